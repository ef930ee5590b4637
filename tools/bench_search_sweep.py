#!/usr/bin/env python3
"""tools/bench_search_sweep.py — scan-kernel time of the HIP index over a sweep of query-batch sizes.

For each nq: HIP-event time of the scan kernel (library timing hook, launch stream), the
algorithmic HBM rate N*d*4 / t (binding roofline for nq <~ 39) and the algorithmic
fp32 matrix rate 2*N*d*nq / t (binding above).  One JSON line per point.

    python tools/bench_search_sweep.py --rows 1000000 --nq 1,8,32,64,256,2214 --k 100
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd.index import IndexFlatIP  # noqa: E402

PEAK_HBM, PEAK_F32 = 8000.0, 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1000000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", default="1,8,32,64,256,2214")
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="fp32", help="corpus storage: fp32 (IndexFlatIP arithmetic) or bf16")
    ap.add_argument("--recall", action="store_true",
                    help="bf16 only: also build the fp32 index and report recall@k of the bf16 ids against it")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    xb = torch.empty((a.rows, a.dim), dtype=torch.float32, device=dev)
    for lo in range(0, a.rows, 131072):
        hi = min(a.rows, lo + 131072)
        xb[lo:hi] = torch.randn((hi - lo, a.dim), generator=g, device=dev)
    ix = IndexFlatIP(a.dim, device=0, dtype=a.dtype)
    ix.add(xb)
    ref = None
    if a.recall and a.dtype == "bf16":
        ref = IndexFlatIP(a.dim, device=0)
        ref.add(xb)
    del xb
    torch.cuda.empty_cache()
    g.manual_seed(4321)
    qall = torch.randn((max(int(s) for s in a.nq.split(",")), a.dim), generator=g, device=dev)
    for s in a.nq.split(","):
        nq = int(s)
        q = qall[:nq].contiguous()
        for _ in range(a.warmup):
            ix.search(q, a.k)
        ix.set_timing(True)
        ix.timing()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ix.search(q, a.k)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.steps * 1e3
        ms, n = ix.timing()
        ix.set_timing(False)
        kms = ms / max(n, 1)
        gbs = a.rows * a.dim * (2 if a.dtype == 'bf16' else 4) / (kms * 1e-3) / 1e9
        tf = 2.0 * a.rows * a.dim * nq / (kms * 1e-3) / 1e12
        # which matrix pipe serves this point, and what its scan reads: an fp32 index on a corpus of >= 65,536 rows runs
        # a coarse pass on the bf16 hi plane (N*d*2 bytes: the stream kernel up to 32 queries, the 256-query-tile scan
        # above) and re-scores exactly; OTTER_NO_SPLIT=1 / OTTER_SPLIT_MIN_NQ keep batches on the exact fp32 kernels
        smin = int(os.environ.get("OTTER_SPLIT_MIN_NQ", "1"))
        coarse16 = a.dtype == "bf16" or (nq >= smin and a.rows >= 65536 and not os.environ.get("OTTER_NO_SPLIT"))
        bytes_per = 2 if a.dtype == "bf16" else 4
        read_per = 2 if coarse16 else 4
        rec = {"nq": nq, "k": a.k, "rows": a.rows, "scan_ms": round(kms, 4), "wall_ms": round(wall, 4),
               "GBps": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM, 4),
               "hbm_frac_wall": round(a.rows * a.dim * bytes_per / (wall * 1e-3) / 1e9 / PEAK_HBM, 4),
               "scan_bytes_read": a.rows * a.dim * read_per,
               "hbm_frac_scan_of_bytes_read": round(a.rows * a.dim * read_per / (kms * 1e-3) / 1e9 / PEAK_HBM, 4),
               "TFLOPs": round(tf, 2),
               "mfma_pipe": "bf16" if coarse16 else "fp32",
               "mfma_frac": round(tf / (2500.0 if coarse16 else PEAK_F32), 4),
               "f32_equivalent_vs_f32_matrix_peak": round(tf / PEAK_F32, 4),
               "queries_per_s": round(nq / (wall * 1e-3), 1),
               "dtype": a.dtype, "fallback_queries": ix.last_fallback, "refined_queries": ix.last_refined}
        if ref is not None:
            _, I16 = ix.search(q, a.k)
            _, I32 = ref.search(q, a.k)
            I16 = torch.as_tensor(I16).cpu()
            I32 = torch.as_tensor(I32).cpu()
            hit = 0
            for r in range(nq):
                hit += len(set(I16[r].tolist()) & set(I32[r].tolist()))
            rec["recall_at_k_vs_fp32"] = round(hit / float(nq * a.k), 5)
        print(json.dumps(rec), flush=True)
    ix.close()
    if ref is not None:
        ref.close()


if __name__ == "__main__":
    main()
