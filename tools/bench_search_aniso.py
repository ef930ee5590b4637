#!/usr/bin/env python3
"""tools/bench_search_aniso.py — the large-batch search on a corpus that is NOT i.i.d. Gaussian: a shared mean with a
few large ("rogue") dimensions, a low-rank part that carries most of the variance and a small isotropic part -- the
shape transformer sentence embeddings have.  Reports time, how many queries the one-product proof left to the
two-/three-product passes or the exact kernel, and checks the result bit for bit against <= 32-query calls
(exact kernels)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd.index import IndexFlatIP  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1000000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--nq", type=int, default=2214)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--rank", type=int, default=64)
ap.add_argument("--iso", type=float, default=0.3, help="standard deviation of the isotropic part")
ap.add_argument("--rogue", type=float, default=10.0, help="mean of the two rogue dimensions")
ap.add_argument("--normalize", action="store_true", help="unit-length rows and queries (cosine retrieval)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(7)
mean = torch.randn((a.dim,), generator=g, device=dev) * 0.3
mean[:2] = a.rogue
L = torch.randn((a.rank, a.dim), generator=g, device=dev) * (1.0 / a.rank ** 0.5) * torch.linspace(3.0, 0.3, a.rank, device=dev)[:, None]


def draw(n):
    out = torch.empty((n, a.dim), dtype=torch.float32, device=dev)
    for lo in range(0, n, 131072):
        hi = min(n, lo + 131072)
        z = torch.randn((hi - lo, a.rank), generator=g, device=dev)
        out[lo:hi] = mean + z @ L + a.iso * torch.randn((hi - lo, a.dim), generator=g, device=dev)
    if a.normalize:
        out /= out.norm(dim=1, keepdim=True)
    return out


xb = draw(a.rows)
q = draw(a.nq)
ix = IndexFlatIP(a.dim)
ix.add(xb)
del xb
for _ in range(2):
    D, I = ix.search(q, a.k)
refined, fallback = ix.last_refined, ix.last_fallback
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    D, I = ix.search(q, a.k)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
ok = True
for lo in range(0, min(a.nq, 128), 32):
    De, Ie = ix.search(q[lo:lo + 32].contiguous(), a.k)
    ok = ok and bool(torch.equal(Ie, I[lo:lo + 32]) and torch.equal(De.view(torch.int32), D[lo:lo + 32].view(torch.int32)))
print(json.dumps({"rows": a.rows, "dim": a.dim, "nq": a.nq, "k": a.k, "normalize": a.normalize, "ms_per_batch": round(ms, 3),
                  "queries_per_s": round(a.nq / (ms * 1e-3), 1), "refined_queries": refined, "exact_fallback_queries": fallback,
                  "coarse_state": ix.coarse_state, "bit_equal_to_exact_kernels_on_128_queries": ok,
                  "score_top1_mean": float(D[:, 0].mean()), "score_kth_mean": float(D[:, -1].mean())}))
