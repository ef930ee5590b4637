#!/usr/bin/env python3
"""tools/hbm_probe.py — what a plain streaming read of the corpus reaches on this box (torch reductions over a
3.07 GB fp32 buffer), to put the scan kernel's HBM rate beside the achievable one rather than the 8 TB/s spec."""
import json
import torch

n = 1000000 * 768
x = torch.randn(n, device="cuda")
res = {}
for name, fn in (("sum", lambda: x.sum()), ("max", lambda: x.max()), ("dot", lambda: torch.dot(x, x))):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    res[name] = {"ms": round(ms, 4), "TBps": round(n * 4 / (ms * 1e-3) / 1e12, 3)}
print(json.dumps(res))
