"""bench.py's launch contract: `python bench.py --gpus N` starts N ranks by itself and
fails loudly when it cannot; the N > 1 step is ShardedIndexFlatIP, the class the gloo tests cover."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=600):
    env = dict(os.environ)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_more_gpus_than_visible_is_an_error_not_a_silent_single_gpu_run():
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip("box has 64 GPUs")
    env = {k: "" for k in ()}
    r = _run(["--gpus", "64", "--steps", "1", "--warmup", "0"], env)
    assert r.returncode != 0
    assert "--gpus 64" in r.stderr and "n_gpus" not in r.stdout


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_rehearsal_spawns_two_ranks_and_checks_the_sharded_result():
    """Two ranks on this box's one GPU (gloo exchange staged through the host): the same code path the
    driver's `python bench.py --gpus N` takes with RCCL, minus the backend."""
    env = {"OTTER_BENCH_REHEARSE": "1"}
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    r = _run(["--gpus", "2", "--rows", "70000", "--no-encode", "--no-cpu-baseline", "--steps", "2", "--warmup", "1"], env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["ranks"] == 2 and j["config"]["backend"] == "gloo"
    assert j["self_check"] == "ok" and all(j["self_checks"].values())
    assert "all-gather" in j["config"]["exchange"] or "gather" in j["config"]["exchange"]
