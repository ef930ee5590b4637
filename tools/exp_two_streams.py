#!/usr/bin/env python3
"""tools/exp_two_streams.py — experiment: does a 2214-query search finish sooner as two concurrent searches of 1280 and
934 queries on two streams (two host threads, two indexes over the same rows)?  The merges and the re-scoring gather of
one half would run beside the matrix-bound scan of the other."""
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd.index import IndexFlatIP  # noqa: E402

N, D, NQ, K = 1000000, 768, 2214, 100
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1234)
xb = torch.empty((N, D), dtype=torch.float32, device=dev)
for lo in range(0, N, 131072):
    hi = min(N, lo + 131072)
    xb[lo:hi] = torch.randn((hi - lo, D), generator=g, device=dev)
g.manual_seed(4321)
q = torch.randn((NQ, D), generator=g, device=dev)
cut = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
ia, ib = IndexFlatIP(D), IndexFlatIP(D)
ia.add(xb)
ib.add(xb)
del xb
qa, qb = q[:cut].contiguous(), q[cut:].contiguous()
steps = 20


def whole():
    for _ in range(3):
        ia.search(q, K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        Dw, Iw = ia.search(q, K)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, Dw, Iw


res = {}


def worker(ix, qq, name, barrier):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        for _ in range(3):
            ix.search(qq, K)
        s.synchronize()
        barrier.wait()
        for _ in range(steps):
            out = ix.search(qq, K)
            barrier.wait()          # both halves of a step finish before the next step starts
        s.synchronize()
    res[name] = out


ms_whole, Dw, Iw = whole()
bar = threading.Barrier(3)
ta = threading.Thread(target=worker, args=(ia, qa, "a", bar))
tb = threading.Thread(target=worker, args=(ib, qb, "b", bar))
ta.start()
tb.start()
bar.wait()
t0 = time.perf_counter()
for _ in range(steps):
    bar.wait()
dt = (time.perf_counter() - t0) / steps * 1e3
ta.join()
tb.join()
torch.cuda.synchronize()
same = bool(torch.equal(torch.cat([res["a"][1], res["b"][1]]), Iw))
print(json.dumps({"whole_ms": ms_whole, "two_streams_ms": dt, "cut": cut, "same_ids": same}))
