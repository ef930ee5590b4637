#!/usr/bin/env python3
"""bench.py — headline benchmark of the dense-retrieval hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): exact top-100 inner-product search of a 2,214-query
batch (OTT-QA dev size) over a 1,000,000 x 768 fp32 corpus resident in HBM, per GPU.
A "step" is one search of the whole query batch.

N > 1: one process per GPU.  `python bench.py --gpus N` starts the N ranks itself (a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, started before this process
touches the GPU; the reference's multi-GPU modes are single-command too,
scripts/eval_ottqa_retrieval.py:110-112, scripts/encode_corpus.py:109-113); launched under
torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.  Every rank
holds its own 1M-row shard of an N x 1M-row corpus (row-sharded, weak scaling) inside
otter_amd.sharded.ShardedIndexFlatIP -- the class the gloo tests cover -- which searches the
shard for all queries, exchanges the packed candidates (8 bytes each: fp32 score + 32-bit shard-local row)
with ONE RCCL all-gather and merges them on the GPU.  `value` counts the 1M-row query scans all ranks complete per second
(N * 2214 per step), which at N = 1 is plain queries/s on configs[1].  The run fails if the
process group that forms is not N ranks.

One JSON line on stdout (rank 0).  `roofline` is for the scan kernel of the timed step: large
batches run the coarse pass of the exact search on the bf16 matrix cores (bf16 planes of the fp32
operands), so the kernel's work per launch is nprod * 2*N*d*nq bf16 flops against the dense bf16
MFMA peak; the fp32-equivalent rate 2*N*d*nq / t is printed beside it.  Time = HIP events around
the kernel, taken inside the timed region by the library's timing hook on the launch stream.
Untimed-by-`value` legs on rank 0 at N = 1: `small_batches` (nq = 1 ... 256: the reference's
real call pattern, the HBM-bound regime and the sizes just above it: scan ms, wall ms, GB/s against 8 TB/s), `pcie`
(host numpy in / out of the same search), `dim2304` (the dimension the reference script uses,
scripts/eval_ottqa_retrieval.py:74), `encode` (corpus-encode blocks/s, BASELINE.json configs[3]).
`cpu_baseline` is oracle/flat_ip.py (numpy sgemm + k-select port of IndexFlatIP, one batched call)
timed on this box's host cores on a bounded row sample, with the reference-style form beside it
(one scan + heap select per query, the shape of the reference's nq = 1 calls from 24 processes).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: Peak FP32 (matrix), dense
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: Peak BF16 MFMA, dense


def gen_rows(n, d, seed, device, chunk=131072):
    """i.i.d. N(0,1) fp32 rows generated on the device (SURVEY.md §8d synthetic corpus)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        out[lo:hi] = torch.randn((hi - lo, d), generator=g, device=device, dtype=torch.float32)
    return out


def host_cores():
    try:
        from threadpoolctl import threadpool_info
        return int(max([p.get("num_threads", 1) for p in threadpool_info()] or [1]))
    except Exception:
        return int(os.cpu_count() or 1)


def cpu_baseline(xb_sample_fn, q, k, rows_full, budget_s=12.0):
    """Time the CPU port on bounded samples of the same workload (rank 0, N = 1 only).
    (ii) of SURVEY.md §8d: one batched call; (i): the reference's way, one scan per query."""
    from oracle import flat_ip
    cores = host_cores()
    nq = q.shape[0]
    # calibration pass, then one pass sized to the budget
    rows = 20000
    xb = xb_sample_fn(rows)
    t0 = time.perf_counter()
    flat_ip.search_batched(xb, q, k)
    dt = time.perf_counter() - t0
    rows2 = int(min(rows_full, max(rows, rows * budget_s / max(dt, 1e-3))))
    rows2 = min(rows2, 1000000)
    if rows2 > rows * 2:
        xb = xb_sample_fn(rows2)
        t0 = time.perf_counter()
        flat_ip.search_batched(xb, q, k)
        dt = time.perf_counter() - t0
        rows = rows2
    scans_per_s = nq / dt * (rows / rows_full)
    out = {"value": scans_per_s, "unit": "queries/s", "cores": cores, "kind": "port",
           "sample": "%d queries x first %d of %d rows, one batched call: numpy sgemm + k-select "
                     "(oracle/flat_ip.py), %.1f s, scaled by rows" % (nq, rows, rows_full, dt)}
    # the reference's call pattern (scripts/eval_ottqa_retrieval.py:126-129,156): nq = 1 per call from a
    # pool of processes = one sequential scan + heap select per query, queries spread over the cores
    nthreads = flat_ip.num_threads()
    nq1 = min(nq, 4 * nthreads)
    rows1 = min(rows_full, 100000)
    xb1 = xb_sample_fn(rows1)
    t0 = time.perf_counter()
    flat_ip.search_c(xb1, q[:nq1], k, mode=flat_ip.MODE_SIMD8)
    dt1 = time.perf_counter() - t0
    out["per_query_calls"] = {
        "value": nq1 / dt1 * (rows1 / rows_full), "unit": "queries/s", "cores": nthreads, "kind": "port",
        "sample": "%d queries x first %d of %d rows, one scan + heap k-select per query (oracle/flat_ip.c, "
                  "8-lane SIMD-shaped dot product), %.1f s, scaled by rows" % (nq1, rows1, rows_full, dt1)}
    return out


def encode_leg(rank, world, local_rank, dist, steps, warmup, batch, cpu_ref, rehearse):
    """Second half of the metric: corpus-encode blocks/s (BASELINE.json configs[3] shape: roberta-base,
    512-token blocks, bf16, data-parallel = every rank encodes its own blocks with resident weights,
    no data-path collective).  Synthetic ids per SURVEY.md §8d, random weights (seed 99).  The timed loop
    feeds DISTINCT host batches through the encode driver's pipelined predict(): pinned token upload on a
    copy stream, forward, asynchronous download of the embeddings -- blocks/s includes both PCIe legs."""
    from types import SimpleNamespace
    from otter_amd import encoder as E
    from otter_amd import synthetic as S
    from otter_amd.encode_driver import predict
    L, H, layers = 512, 768, 12
    cfg = S.make_config("roberta")
    sd = S.make_state_dict(cfg, 99)
    hf = SimpleNamespace(**{k: cfg[k] for k in ("vocab_size", "hidden_size", "num_hidden_layers",
                                                "num_attention_heads", "intermediate_size",
                                                "max_position_embeddings", "type_vocab_size", "pad_token_id",
                                                "layer_norm_eps")})
    a = SimpleNamespace(shared_encoder=True, no_proj=True, part_pooling="first", encode_table=True)
    model = E.RobertaSingleEncoderThreeCatPool(hf, a, dtype="bf16")
    model.load_state_dict(sd, strict=False)
    model.to("cuda:%d" % local_rank)

    def host_batch(seed):
        rs = np.random.RandomState(seed)
        ids = rs.randint(3, cfg["vocab_size"], size=(batch, L)).astype(np.int64)
        ids[:, 0] = 0
        ids[:, 128] = 2
        ids[:, 129] = 2
        ids[:, L - 1] = 2
        p2 = np.zeros((batch, L), np.int64)
        p2[:, 1:128] = 1
        p3 = np.zeros((batch, L), np.int64)
        p3[:, 130:L - 1] = 1
        return {"input_ids": torch.from_numpy(ids), "input_mask": torch.ones((batch, L), dtype=torch.int64),
                "part2_mask": torch.from_numpy(p2), "part3_mask": torch.from_numpy(p3)}

    nb_distinct = 4
    hbs = [host_batch(777 + 1000 * rank + i) for i in range(nb_distinct)]
    b = {k: v.cuda() for k, v in hbs[0].items()}
    for _ in range(warmup):
        out = model(b)["embed"]
    # pipeline warm-up: pinned staging buffers, copy streams, and one pass of the timed length -- the first predict()
    # of a given length was measured 11 ms (= 0.5 ms per batch of 20) slower than every later one
    # (tools/exp_predict_gap.py: 13.94, 13.46, 13.39 ms per batch on three calls in a row)
    predict(model, [hbs[i % nb_distinct] for i in range(steps)])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb = predict(model, [hbs[i % nb_distinct] for i in range(steps)])      # (a list: predict() sizes its result once)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    assert emb.shape == (steps * batch, 3 * H)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else b["input_ids"].device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # device time of the forward alone (resident batch), for the roofline figure
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        out = model(b)["embed"]
    e1.record()
    torch.cuda.synchronize()
    if rank != 0:
        return None
    dev_ms = e0.elapsed_time(e1) / steps
    flops_block = layers * (24.0 * L * H * H + 4.0 * L * L * H)
    # HBM-side bytes per forward from a separate rocprofv3 --pmc pass over tools/bench_encoder.py (same shape, same
    # batch): a committed constant, labelled as such
    enc_traffic, enc_traffic_source = None, None
    tp = os.path.join(ROOT, "profiles", "r03_encode_traffic.json")
    if os.path.exists(tp) and (batch, L, layers) == (128, 512, 12):
        with open(tp) as f:
            tj = json.load(f)
        enc_traffic = float(tj["encode_hbm_bytes_per_forward"])
        enc_traffic_source = {"file": "profiles/r03_encode_traffic.json", "measured_at_commit": tj.get("commit"),
                              "how": "separate rocprofv3 --pmc passes (tools/pmc.sh + tools/encode_traffic_from_pmc.py), "
                                     "a committed constant, not measured in this run"}
    achieved = flops_block * batch / (dev_ms * 1e-3) / 1e12
    res = {"value": world * batch * steps / elapsed, "unit": "blocks/s", "dtype": "bf16", "batch_per_gpu": batch,
           "seq_len": L, "model": "roberta-base shape, random init", "ms_per_batch": elapsed / steps * 1e3,
           "includes": "pinned H2D of 4 x int64 [B, 512] per batch on a copy stream + D2H of the float32 [B, 2304] "
                       "embeddings (otter_amd.encode_driver.predict), %d distinct host batches" % nb_distinct,
           "finite": bool(torch.isfinite(emb).all().item()),
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_BF16_MFMA_TFLOPS, "traffic": enc_traffic, "traffic_source": enc_traffic_source,
                        "kernel": "whole forward (12 x [qkv gemm, attention, o gemm, ffn1 gemm, ffn2 gemm], LayerNorms folded into the gemm epilogues)",
                        "device_ms_per_batch": dev_ms, "algorithmic_flops_per_block": flops_block}}
    # SURVEY.md §8d ragged variant, reported separately: token counts ~U[64, 512] padded to 512 the way the
    # reference's tokenizer call pads (retrieval/data/tr_dataset.py:139-141); the packed layout skips those rows
    rs2 = np.random.RandomState(778 + rank)
    ids2 = np.full((batch, L), cfg["pad_token_id"], np.int64)
    m2 = np.zeros((batch, L), np.int64)
    q2 = np.zeros((batch, L), np.int64)
    q3 = np.zeros((batch, L), np.int64)
    for i in range(batch):
        n = int(rs2.randint(64, L + 1))
        a_end = max(2, n // 4)
        ids2[i, :n] = rs2.randint(3, cfg["vocab_size"], size=n)
        ids2[i, 0] = 0
        ids2[i, a_end] = 2
        ids2[i, a_end + 1] = 2
        ids2[i, n - 1] = 2
        m2[i, :n] = 1
        q2[i, 1:a_end] = 1
        q3[i, a_end + 2:n - 1] = 1
    rb = {"input_ids": torch.from_numpy(ids2).cuda(), "input_mask": torch.from_numpy(m2).cuda(),
          "part2_mask": torch.from_numpy(q2).cuda(), "part3_mask": torch.from_numpy(q3).cuda()}
    model(rb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rout = model(rb)["embed"]
    torch.cuda.synchronize()
    rdt = (time.perf_counter() - t0) / steps
    res["ragged_variant"] = {"value": batch / rdt, "unit": "blocks/s (this rank)", "mean_tokens": float(m2.sum()) / batch,
                             "padded_to": L, "ms_per_batch": rdt * 1e3,
                             "finite": bool(torch.isfinite(rout).all().item()),
                             "note": "lengths ~U[64,512]; rows past the last masked-in position are not computed; "
                                     "resident batch (no PCIe legs)"}
    if cpu_ref:
        # cosine of the bf16 path against the fp32 CPU restatement + CPU baseline timing (oracle).  The sample
        # is bounded by this file's budget (10-30 s of CPU work: ~9 blocks/s on 16 threads), not the >= 512
        # blocks SURVEY.md §8d suggests, which would take a minute by themselves.
        from oracle import encoder_ref as R
        ref = R.EncoderRef(cfg, sd)
        nb = min(batch, 128)
        small = {k: v[:nb] for k, v in hbs[0].items()}
        ref({k: v[:4] for k, v in small.items()})    # warm-up (thread pool, allocations)
        t0 = time.perf_counter()
        exp = torch.cat([ref({k: v[i:i + 8] for k, v in small.items()})["embed"] for i in range(0, nb, 8)], 0)
        dt = time.perf_counter() - t0
        got = out[:nb].cpu()
        cos = torch.nn.functional.cosine_similarity(got.view(nb * 3, H), exp.view(nb * 3, H), dim=1)
        res["bf16_vs_fp32_oracle_min_cosine"] = float(cos.min())
        res["cpu_baseline"] = {"value": nb / dt, "unit": "blocks/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "%d blocks of 512 tokens, oracle/encoder_ref.py (torch CPU fp32), %.1f s"
                                         % (nb, dt)}
    return res


def scan16_dispatches(n, nq, k, d, nprod):
    """Dispatches of scan16_kernel per search: the sample levels scan16_and_merge (otter_amd/csrc/index.hip) plans
    for this shape, plus the main scan."""
    margin = {3: max(28, k // 8), 2: max(44, math.ceil(k * max(0.6, 0.022 * math.sqrt(d)))),
              1: max(60, math.ceil(k * max(1.0, 0.04 * math.sqrt(d))))}[nprod]
    kk = k + margin
    ncu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    qtiles, total = -(-nq // 256), -(-n // 256)
    wgs = max(1, ncu // qtiles)
    cover = total // (4 if qtiles >= 8 else 8)
    if nprod == 1 and nq > 64 and os.environ.get("OTTER_OPT_THR") != "0":
        # the optimistic start (one small level 0 + the scan of everything else) where the library takes it
        l0 = min(32, int(1.6 * total / kk))
        if wgs >= 24:
            l0 = min(l0, wgs)
        if l0 >= 24 and l0 <= total // 8 and l0 * 256 >= 48 and 12 <= kk:
            return 2
    l0 = max(1, -(-4 * kk // (wgs * 256))) * wgs
    if l0 == wgs:
        l0 = min(wgs, max(-(-4 * kk // 256), 16, min(32, wgs // 2)))
    if l0 > total // 8:
        return 1
    more, c = 0, l0
    while c < cover and more < (3 if qtiles >= 8 else 2) - 1:
        more, c = more + 1, c * 4
    ratio = (max(cover, l0) / l0) ** (1.0 / more) if more else 1.0
    levels, done, target = 1, l0, float(l0)
    for _ in range(more):
        target *= ratio
        nx = -(-(math.ceil(target) - done) // wgs) * wgs
        if nx <= 0:
            continue
        if done + nx > total // 2:
            break
        levels, done = levels + 1, done + nx
    return levels + 1


def timed_search(index, q, k, steps, warmup):
    """(scan ms per search from the library's HIP-event hook, wall ms per search) of index.search(q, k).  Two loops:
    the wall clock runs without the hook (its two event records per search sit between the kernels of a call and
    cost a small batch 10-20 us), the hook's loop follows."""
    for _ in range(warmup):
        index.search(q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        index.search(q, k)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    index.set_timing(True)
    index.timing()
    for _ in range(steps):
        index.search(q, k)
    torch.cuda.synchronize()
    ms, launches = index.timing()
    index.set_timing(False)
    return ms / max(launches, 1), wall


def small_batch_leg(index, q, k, n, d):
    """The reference's real call pattern (nq = 1 per call, scripts/eval_ottqa_retrieval.py:126-129) and the
    rest of the memory-bound regime.  Algorithmic bytes per search: the fp32 corpus, N*d*4.  The
    default path reads HALF of that -- its coarse pass streams the bf16 hi plane -- and re-scores k + margin rows
    exactly; the exact fp32 stream kernels (the allocation-free entry point) read all of it: both are timed."""
    out = []
    for nq in (1, 8, 16, 32, 33, 64, 128, 256):
        if nq > q.shape[0]:
            break
        qq = q[:nq].contiguous()
        scan_ms, wall_ms = timed_search(index, qq, k, 30, 5)
        byts = float(n) * d * 4
        read = byts / 2                         # what the default path's scan really reads: the bf16 hi plane
        pt = {"nq": nq, "scan_ms": scan_ms, "wall_ms": wall_ms, "queries_per_s": nq / (wall_ms * 1e-3),
              # fractions of the 8 TB/s peak, on the bytes REALLY read (never above 1)
              "hbm_frac_scan": read / (scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
              "hbm_frac_wall": read / (wall_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
              # the algorithmic N*d*4 bytes per second the call is worth (may exceed the HBM peak: half of them
              # are never read) -- a rate, not a roofline fraction
              "fp32_equivalent_GBps_scan": byts / (scan_ms * 1e-3) / 1e9,
              "fp32_equivalent_GBps_wall": byts / (wall_ms * 1e-3) / 1e9}
        if k <= 128 and d <= 4096 and nq > 32:
            # above the allocation-free entry point's 32 queries: compared only, in calls of 32
            De = torch.empty((nq, k), dtype=torch.float32, device=qq.device)
            Ie = torch.empty((nq, k), dtype=torch.int64, device=qq.device)
            for lo in range(0, nq, 32):
                index.search_noalloc(qq[lo:lo + 32].contiguous(), k, De[lo:lo + 32], Ie[lo:lo + 32])
            Dd, Id = index.search(qq, k)
            pt["exact_fp32_kernels"] = {"bit_equal_to_default_path": bool(torch.equal(Id, Ie) and torch.equal(
                Dd.view(torch.int32), De.view(torch.int32)))}
        elif k <= 128 and d <= 4096:
            De = torch.empty((nq, k), dtype=torch.float32, device=qq.device)
            Ie = torch.empty((nq, k), dtype=torch.int64, device=qq.device)
            for _ in range(5):
                index.search_noalloc(qq, k, De, Ie)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                index.search_noalloc(qq, k, De, Ie)
            torch.cuda.synchronize()
            ems = (time.perf_counter() - t0) / 30 * 1e3
            Dd, Id = index.search(qq, k)
            pt["exact_fp32_kernels"] = {"wall_ms": ems, "hbm_frac_wall": byts / (ems * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                        "bit_equal_to_default_path": bool(torch.equal(Id, Ie) and torch.equal(
                                            Dd.view(torch.int32), De.view(torch.int32)))}
        out.append(pt)
    return {"bound": "hbm", "peak_GBps": PEAK_HBM_GBS, "algorithmic_bytes_per_search": float(n) * d * 4,
            "bytes_read_per_search_default_path": float(n) * d * 2,
            "kernel": "default path: coarse pass over the bf16 hi plane of the fp32 corpus (N*d*2 bytes read once, "
                      "v_mfma_f32_16x16x32_bf16): stream16_kernel up to 32 queries, stream64_kernel (queries resident in LDS, "
                      "corpus through a register ring) for 33 ... 64, in-kernel threshold exchange in both; scan16_kernel "
                      "(256-column tiles, optimistic threshold from block maxima) from 65 on; + merge + exact fp32 re-scoring "
                      "of the k + margin candidates + per-query proof (up to 64 queries: one launch, finish_small_kernel, "
                      "its proof word polled from pinned memory): results are the exact fp32 kernels', bit for bit.  "
                      "exact_fp32_kernels (otter_index_search_noalloc): stream_kernel (<= 16 queries: lane = row, per-wave "
                      "LDS-DMA rings with non-temporal loads, v_mfma_f32_4x4x1 fma chain) / stream32_kernel (17 ... 32 "
                      "queries: v_mfma_f32_32x32x2 blocks) + one merge_kernel launch, N*d*4 bytes read once",
            "note": "scan_ms = HIP events around the kernel(s) that read corpus rows; wall_ms = host clock per search() call "
                    "incl. every side kernel, the proof word and the launch gaps; hbm_frac_* price the "
                    "bytes the scan REALLY reads (N*d*2 on the default path, N*d*4 for exact_fp32_kernels) against 8 TB/s; "
                    "fp32_equivalent_GBps_* = the algorithmic N*d*4 bytes per second, a rate that may exceed the HBM peak",
            "points": out}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def count_gpus_sysfs():
    """GPUs of this machine from the kfd topology (nodes with SIMDs; CPU nodes have simd_count 0), WITHOUT any HIP or
    torch.cuda call: the process that spawns the ranks must never have touched the GPU.  None if unreadable."""
    import glob
    if not os.path.isdir("/sys/class/kfd"):
        return 0                       # no amdgpu compute driver on this machine: no GPU
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for path in nodes:
        try:
            with open(path) as f:
                for ln in f:
                    if ln.startswith("simd_count") and int(ln.split()[1]) > 0:
                        n += 1
        except (OSError, ValueError, IndexError):
            return None
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    if vis:
        n = min(n, len([v for v in vis.split(",") if v.strip()]))
    return n


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as children of a
    torch.distributed.run child process and pass its exit code on.  This process never touches the GPU: no HIP
    call, no torch.cuda.* call (the device count comes from sysfs; if that cannot be read the check is left to
    the ranks, which fail loudly on their own)."""
    rehearse = os.environ.get("OTTER_BENCH_REHEARSE") == "1"
    if not rehearse:
        have = count_gpus_sysfs()
        if have is not None and have < n:
            print("bench.py: --gpus %d but only %d GPU(s) visible (OTTER_BENCH_REHEARSE=1 runs every rank on GPU 0 "
                  "with a gloo exchange)" % (n, have), file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1000000, help="corpus rows per GPU")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=2214)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encode", action="store_true", help="skip the corpus-encode leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the small-batch / PCIe / d=2304 legs")
    ap.add_argument("--encode-batch", type=int, default=128)
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per scan launch from a separate rocprofv3 --pmc pass (profiles/)")
    args = ap.parse_args()
    args.traffic_source = ({"file": None, "how": "--traffic-bytes on the command line"} if args.traffic_bytes is not None
                           else None)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if (args.traffic_bytes is None and (args.rows, args.dim, args.nq, args.k) == (1000000, 768, 2214, 100)
            and not os.environ.get("OTTER_NO_SPLIT")):
        # HBM-side bytes per scan launch from the separate rocprofv3 --pmc passes (profiles/)
        for name in ("r03_traffic_split.json", "r02_traffic_split.json", "r01_traffic_split.json"):
            tp = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tp):
                with open(tp) as f:
                    tj = json.load(f)
                args.traffic_bytes = float(tj["scan_kernel_hbm_bytes_per_launch"])
                args.traffic_source = {"file": "profiles/" + name, "measured_at_commit": tj.get("commit"),
                                       "how": "separate rocprofv3 --pmc passes (tools/pmc.sh + tools/traffic_from_pmc.py), "
                                              "a committed constant, not measured in this run"}
                break

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU fallback")
    # rehearsal on a one-GPU box (OTTER_BENCH_REHEARSE=1): every rank uses GPU 0 and the exchange goes through
    # gloo; the driver's real runs use one GPU per rank and RCCL
    rehearse = os.environ.get("OTTER_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if rehearse else "nccl"
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))

    from otter_amd.index import IndexFlatIP
    from otter_amd.sharded import ShardedIndexFlatIP

    n, d, nq, k = args.rows, args.dim, args.nq, args.k
    # shard `rank` of the global synthetic corpus; ids are global (id_base = rank * n)
    xb = gen_rows(n, d, 1234 + rank, dev)
    six = ShardedIndexFlatIP(d, device=local_rank, exchange_device="cpu" if (world > 1 and rehearse) else None)
    six.add_shard(xb, rank * n, world * n)
    index = six.local
    xb_head = xb.cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    check_rows = xb[:65536].clone()
    del xb
    torch.cuda.empty_cache()
    q = gen_rows(nq, d, 4321, dev)
    qh = q.cpu().numpy()

    def step():
        return six.search(q, k)

    for _ in range(args.warmup):
        step()
    index.set_timing(True)
    index.timing()  # clear
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        D, I = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    scan_ms, launches = index.timing()
    index.set_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fallback = index.last_fallback
    refined = index.last_refined

    # ---- sanity, outside the timed region.  The timed batch went through the coarse bf16 pass; the check
    # queries are searched again by the EXACT fp32 stream kernels (otter_index_search_noalloc: <= 32 queries per
    # call, no coarse pass on that entry point), so the comparison is between two different paths.
    nchk = min(64, nq)
    De = torch.empty((nchk, k), dtype=torch.float32, device=dev)
    Ie = torch.empty((nchk, k), dtype=torch.int64, device=dev)
    for lo in range(0, nchk, 32):
        hi = min(nchk, lo + 32)
        if k <= 128 and d <= 4096:
            index.search_noalloc(q[lo:hi].contiguous(), k, De[lo:hi], Ie[lo:hi])
        else:
            De[lo:hi], Ie[lo:hi] = index.search(q[lo:hi].contiguous(), k)
    ok = True
    checks = {}
    if world == 1:
        checks["timed_batch_equals_exact_kernel_bitwise_on_%d_queries" % nchk] = bool(
            torch.equal(Ie, I[:nchk]) and torch.equal(De.view(torch.int32), D[:nchk].view(torch.int32)))
    else:
        # every rank's exact local lists for the check queries, gathered and merged by sorting (torch): the
        # merged result of the timed sharded search must equal it bit for bit
        Dl = [torch.empty_like(De.cpu()) for _ in range(world)]
        Il = [torch.empty_like(Ie.cpu()) for _ in range(world)]
        if rehearse:
            dist.all_gather(Dl, De.cpu())
            dist.all_gather(Il, Ie.cpu())
        else:
            Dg = [torch.empty_like(De) for _ in range(world)]
            Ig = [torch.empty_like(Ie) for _ in range(world)]
            dist.all_gather(Dg, De)
            dist.all_gather(Ig, Ie)
            Dl, Il = [t.cpu() for t in Dg], [t.cpu() for t in Ig]
        if rank == 0:
            Dc, Ic = torch.cat(Dl, dim=1), torch.cat(Il, dim=1)          # [nchk, world * k]
            o1 = torch.argsort(Ic, dim=1, stable=True)
            Dc, Ic = torch.gather(Dc, 1, o1), torch.gather(Ic, 1, o1)
            o2 = torch.argsort(-Dc.double(), dim=1, stable=True)[:, :k]
            Dm, Im = torch.gather(Dc, 1, o2), torch.gather(Ic, 1, o2)
            Dt, It = D[:nchk].cpu(), I[:nchk].cpu()
            checks["sharded_result_equals_sorted_merge_of_exact_shard_lists_on_%d_queries" % nchk] = bool(
                torch.equal(Im, It) and torch.equal(Dm.view(torch.int32), Dt.view(torch.int32)))
            srt = torch.sort(I, dim=1).values
            checks["sharded_ids_unique_and_in_range"] = bool(
                (srt[:, 1:] != srt[:, :-1]).all().item() and ((I >= 0) & (I < world * n)).all().item())
    if rank == 0:
        # independent arbiter: float64 scores (torch) of the first 65,536 rows for 8 queries
        small = IndexFlatIP(d, device=local_rank)
        small.add(check_rows)
        Ds, Is = small.search(q[:8], k)
        ref = (q[:8].double() @ check_rows.double().T)
        rs, ri = torch.topk(ref, k, dim=1)
        got = torch.gather(ref, 1, Is)
        checks["fp64_arbiter_8_queries_x_65536_rows"] = bool(
            (got >= rs[:, -1:] - 1e-3).all().item() and ((Ds.double() - got).abs() < 1e-2).all().item())
        small.close()
        ok = all(checks.values())

    extra = {}
    if rank == 0 and world == 1 and not args.no_extra:
        extra["small_batches"] = small_batch_leg(index, q, k, n, d)
        # PCIe-inclusive rate of the timed workload: host numpy queries in, host numpy (D, I) out
        index.search(qh, k)
        t0 = time.perf_counter()
        reps = max(3, args.steps // 2)
        for _ in range(reps):
            index.search(qh, k)
        pdt = (time.perf_counter() - t0) / reps
        # the same regime on a bf16-storage index (BASELINE.json configs[4]'s storage type; reported separately)
        xb16 = gen_rows(n, d, 1234 + rank, dev)
        ix16 = IndexFlatIP(d, device=local_rank, dtype="bf16")
        ix16.add(xb16)
        del xb16
        torch.cuda.empty_cache()
        pts = []
        for nq1 in (1, 16, 32):
            s_ms, w_ms = timed_search(ix16, q[:nq1].contiguous(), k, 30, 5)
            pts.append({"nq": nq1, "scan_ms": s_ms, "wall_ms": w_ms, "GBps_wall": float(n) * d * 2 / (w_ms * 1e-3) / 1e9,
                        "hbm_frac_scan": float(n) * d * 2 / (s_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                        "hbm_frac_wall": float(n) * d * 2 / (w_ms * 1e-3) / 1e9 / PEAK_HBM_GBS})
        extra["small_batches_bf16_index"] = {"bound": "hbm", "algorithmic_bytes_per_search": float(n) * d * 2,
                                             "kernel": "stream16_kernel (v_mfma_f32_16x16x32_bf16; one block of 16 query columns up to 16 queries, two up to 32)", "points": pts}
        ix16.close()
        del ix16
        torch.cuda.empty_cache()
        extra["pcie"] = {"value": nq / pdt, "unit": "queries/s", "ms_per_step": pdt * 1e3,
                         "note": "same search with pageable host numpy buffers: H2D of %d x %d fp32 queries + D2H of "
                                 "(D, I) inside the call; never `value`" % (nq, d)}

    exchange = None if world == 1 else "%s of packed (score, shard-local row) candidates, %d bytes sent per rank and step" % (
        six.last_exchange, six.last_exchange_bytes)
    six.local.close()
    del index, six, check_rows
    torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_extra and (n, d) == (1000000, 768):
        # the dimension the reference's script actually searches (three-cat embeddings, d = 768 * 3,
        # scripts/eval_ottqa_retrieval.py:74), same rows / queries / k
        d3 = 2304
        x3 = gen_rows(n, d3, 2234, dev)
        ix3 = IndexFlatIP(d3, device=local_rank)
        ix3.add(x3)
        del x3
        torch.cuda.empty_cache()
        q3 = gen_rows(nq, d3, 5321, dev)
        s3, w3 = timed_search(ix3, q3, k, max(4, args.steps // 3), 2)
        D3, I3 = ix3.search(q3, k)
        Dx = torch.empty((32, k), dtype=torch.float32, device=dev)
        Ix = torch.empty((32, k), dtype=torch.int64, device=dev)
        ix3.search_noalloc(q3[:32].contiguous(), k, Dx, Ix)          # exact fp32 stream kernel
        ok3 = bool(torch.equal(Ix, I3[:32]) and torch.equal(Dx.view(torch.int32), D3[:32].view(torch.int32)))
        f3 = 2.0 * n * d3 * nq
        extra["dim2304"] = {"value": nq / (w3 * 1e-3), "unit": "queries/s", "ms_per_step": w3, "scan_ms": s3,
                            "rows": n, "dim": d3, "nq": nq, "k": k,
                            "f32_equivalent_tflops": f3 / (w3 * 1e-3) / 1e12,
                            "coarse_pass_bf16_tflops_scan": f3 / (s3 * 1e-3) / 1e12,
                            "bf16_mfma_frac_scan": f3 / (s3 * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                            "timed_batch_equals_exact_kernel_bitwise_on_32_queries": ok3,
                            "exact_fallback_queries": ix3.last_fallback, "refined_queries": ix3.last_refined}
        ok = ok and ok3
        ix3.close()
        del ix3, q3
        torch.cuda.empty_cache()

    enc = None
    if not args.no_encode:
        enc = encode_leg(rank, world, local_rank, dist, max(8, args.steps), min(3, max(1, args.warmup)),
                         args.encode_batch, world == 1 and not args.no_cpu_baseline, rehearse)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * nq * args.steps / elapsed
        kern_ms = scan_ms / max(launches, 1)
        flops = 2.0 * n * d * nq
        split = nq >= 33 and n >= 65536 and not os.environ.get("OTTER_NO_SPLIT")
        nprod = {"1": 3, "2": 2}.get(os.environ.get("OTTER_SPLIT_MODE", ""), 1)
        if split:   # coarse pass on bf16 planes of the fp32 operands: 1 (default), 2 or 3 MFMA products per term
            kernel_flops, peak = float(nprod) * flops, PEAK_BF16_MFMA_TFLOPS
            kname = ("scan16_kernel<%d,*> (256x256 tile, v_mfma_f32_16x16x32_bf16 on bf16 planes of the fp32 "
                     "operands: %s, fused top-k filter; then exact fp32 re-scoring + margin proof; unproven queries "
                     "go to the pass with one more product / the exact kernel). " % (
                         {1: 0, 2: 2, 3: 1}[nprod],
                         {1: "x_hi*q_hi", 2: "x_hi*q_hi + x_lo*q_hi", 3: "x_hi*q_hi + x_hi*q_lo + x_lo*q_hi"}[nprod]) +
                     "kernel_ms is one HIP-event bracket per search on the launch stream = %d dispatches of this kernel "
                     "and what runs between them: with the optimistic start (2 dispatches) a block-maxima launch over 28 ... 64 "
                     "sample tiles, the 5 us threshold kernel and ONE scan of every tile (sample tiles scored twice: "
                     "+0.7%% kernel flops, not counted); otherwise threshold-seeding sample levels, the merges between "
                     "them and the main scan.  rocprof: TotalDurationNs of scan16_kernel / steps + those kernels"
                     % scan16_dispatches(n, nq, k, d, nprod))
        else:
            kernel_flops, peak = flops, PEAK_FP32_MFMA_TFLOPS
            kname = "scan_kernel (fp32 v_mfma_f32_32x32x2_f32 + fused top-k filter)"
        achieved = kernel_flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        out = {
            "metric": "exact top-100 MIPS queries/s over 768-d corpus; corpus encode blocks/s",
            "value": value,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16 hi/lo split of f32 operands on the matrix cores, f32 accumulate, exact f32 re-scoring (results = f32 IndexFlatIP)"
                     if split else "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: exact top-%d IP search, %d-query batch over %dx%d fp32 corpus "
                                   "per GPU (row-sharded at N>1: ShardedIndexFlatIP, one packed all-gather + GPU merge)"
                                   % (k, nq, n, d),
                       "rows_per_gpu": n, "dim": d, "nq": nq, "k": k,
                       "value_counts": "1M-row query scans per second over all ranks (= queries/s at N=1)",
                       "ranks": world, "backend": backend,
                       "exchange": exchange},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak,
                         "traffic": args.traffic_bytes,
                         "traffic_source": args.traffic_source,
                         "kernel": kname,
                         "kernel_ms": kern_ms, "launches": int(launches),
                         "kernel_dispatches_per_launch": scan16_dispatches(n, nq, k, d, nprod) if split else 1,
                         "kernel_flops_per_launch": kernel_flops,
                         "f32_equivalent_tflops": flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0,
                         "f32_equivalent_vs_f32_matrix_peak": (flops / (kern_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS)
                         if kern_ms > 0 else 0.0,
                         "exact_fallback_queries_last_step": int(fallback),
                         "refined_queries_last_step": int(refined),
                         "algorithmic_flops_per_launch": flops,
                         "algorithmic_bytes_per_launch": float(n) * d * 4 + nq * d * 4 + nq * k * 12,
                         "hbm_frac_if_bandwidth_bound": (float(n) * d * 4 / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS)
                         if kern_ms > 0 else 0.0},
            "self_check": "ok" if ok else "FAILED",
            "self_checks": checks,
        }
        out.update(extra)
        if enc is not None:
            out["encode"] = enc
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lambda r: xb_head[:r], qh, k, n)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
