#!/usr/bin/env python3
"""tools/dbg_opt.py — how many queries of a 2214-query batch fail the optimistic-threshold check (last_refined / last_fallback)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd.index import IndexFlatIP
n, d, nq, k = 1000000, 768, 2214, 100
g = torch.Generator(device="cuda"); g.manual_seed(1234)
xb = torch.randn((n, d), generator=g, device="cuda")
ix = IndexFlatIP(d); ix.add(xb)
for seed in (1, 2, 3):
    g.manual_seed(4000 + seed)
    q = torch.randn((nq, d), generator=g, device="cuda")
    D, I = ix.search(q, k)
    print("seed", seed, "refined", ix.last_refined, "fallback", ix.last_fallback, "state", ix.coarse_state)
