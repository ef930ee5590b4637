#!/usr/bin/env python3
"""tools/exp_predict_overhead.py — resident forward loop vs the pipelined predict() loop, alternating, same process:
is the difference between bench.py's `encode.ms_per_batch` and `device_ms_per_batch` the pipeline or the box?"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import encoder as E  # noqa: E402
from otter_amd import synthetic as S  # noqa: E402
from otter_amd.encode_driver import predict  # noqa: E402

B, L, steps = 128, 512, 20
cfg = S.make_config("roberta")
sd = S.make_state_dict(cfg, 99)
hf = SimpleNamespace(**{k: cfg[k] for k in ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads",
                                            "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                            "pad_token_id", "layer_norm_eps")})
a = SimpleNamespace(shared_encoder=True, no_proj=True, part_pooling="first", encode_table=True)
model = E.RobertaSingleEncoderThreeCatPool(hf, a, dtype="bf16")
model.load_state_dict(sd, strict=False)
model.to("cuda:0")


def host_batch(seed):
    rs = np.random.RandomState(seed)
    ids = rs.randint(3, cfg["vocab_size"], size=(B, L)).astype(np.int64)
    ids[:, 0] = 0
    p2 = np.zeros((B, L), np.int64)
    p2[:, 1:128] = 1
    p3 = np.zeros((B, L), np.int64)
    p3[:, 130:L - 1] = 1
    return {"input_ids": torch.from_numpy(ids), "input_mask": torch.ones((B, L), dtype=torch.int64),
            "part2_mask": torch.from_numpy(p2), "part3_mask": torch.from_numpy(p3)}


hbs = [host_batch(i) for i in range(4)]
b = {k: v.cuda() for k, v in hbs[0].items()}
for _ in range(3):
    model(b)
predict(model, hbs[:2])
torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == 'prefault':
    t0 = time.perf_counter()
    tmp = torch.empty((steps * B, 2304), dtype=torch.float32).zero_()
    print('first touch of a result-sized host tensor: %.1f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
    del tmp
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(steps):
        model(b)
    torch.cuda.synchronize()
    t_res = (time.perf_counter() - t0) / steps * 1e3
    t0 = time.perf_counter()
    predict(model, [hbs[i % 4] for i in range(steps)])
    torch.cuda.synchronize()
    t_pipe = (time.perf_counter() - t0) / steps * 1e3
    t0 = time.perf_counter()
    predict(model, [b for i in range(steps)], to_cpu=False)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / steps * 1e3
    print("rep %d: resident forward %.2f ms, predict (host in, host out) %.2f ms, predict (device in, device out) %.2f ms"
          % (rep, t_res, t_pipe, t_dev), flush=True)

# ---- which host phase stalls?  (wrap the pipeline's steps with timers; report anything above 5 ms)
from otter_amd import encode_driver as ED  # noqa: E402

slow = []


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        dt = (time.perf_counter() - t) * 1e3
        if dt > 5.0:
            slow.append((name, round(dt, 1)))
        return r
    return w


ED._PipeSlot.upload = timed("upload", ED._PipeSlot.upload)
ED._PipeSlot.download = timed("download", ED._PipeSlot.download)
orig_run = model._run
model._run = timed("forward call", orig_run)
orig_empty = torch.empty
for rep in range(4):
    slow.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    predict(model, [hbs[i % 4] for i in range(steps)])
    torch.cuda.synchronize()
    print("instrumented predict: %.2f ms per batch; host phases above 5 ms: %s" % ((time.perf_counter() - t0) / steps * 1e3, slow), flush=True)
