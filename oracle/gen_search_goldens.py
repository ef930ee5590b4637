#!/usr/bin/env python3
"""oracle/gen_search_goldens.py — committed search fixtures of SURVEY.md §8c (tests/golden/search_*.npz).

TEST INFRASTRUCTURE.  PARITY UNPINNED versus faiss: faiss is absent from /root/reference and from this image
and the reference holds no search fixture, so these vectors cannot come from the reference itself.  What they
pin instead is independent of every kernel's summation order: scores are computed in FLOAT64 (numpy), ranked
under (score descending, id ascending), and stored with the float64 scores of the top k + 8 so that a test can
tell a legitimate float32 near-tie swap from a wrong row.  They hold results across rounds: the HIP path and
the CPU restatements (oracle/flat_ip.c, oracle/flat_ip.py) are both checked against these files.

Corpora are not stored: `make_case` regenerates them from numpy's frozen RandomState stream.

    python oracle/gen_search_goldens.py            # rewrites tests/golden/search_*.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
EXTRA = 8

# name -> (seed, N, d, Q, k, kind)
CASES = {
    "n100_d768_q7_k1": (101, 100, 768, 7, 1, "gauss"),
    "n100_d768_q7_k100": (102, 100, 768, 7, 100, "gauss"),          # k == N
    "n100_d2304_q1_k1000": (103, 100, 2304, 1, 1000, "gauss"),      # k > N: -1 / -FLT_MAX padding
    "n4096_d768_q64_k100": (104, 4096, 768, 64, 100, "gauss"),
    "n4096_d2304_q7_k1000": (105, 4096, 2304, 7, 1000, "gauss"),
    "n65536_d768_q64_k100": (106, 65536, 768, 64, 100, "gauss"),
    "n65536_d2304_q7_k100": (107, 65536, 2304, 7, 100, "gauss"),
    "n65536_d768_q1_k1000": (108, 65536, 768, 1, 1000, "gauss"),
    "dup_n4096_d768_q7_k100": (109, 4096, 768, 7, 100, "dup"),      # every row of the upper half repeats a lower row
    "equal_n300_d768_q3_k100": (110, 300, 768, 3, 100, "equal"),    # all scores of a query are equal: ids 0..k-1
}


def make_case(name):
    """(xb float32 [N, d], q float32 [Q, d], k) of a named case."""
    seed, n, d, nq, k, kind = CASES[name]
    rs = np.random.RandomState(seed)
    xb = rs.standard_normal(size=(n, d)).astype(np.float32)
    q = rs.standard_normal(size=(nq, d)).astype(np.float32)
    if kind == "dup":
        xb[n // 2:] = xb[: n - n // 2]
    elif kind == "equal":
        xb[:] = xb[0]
    return xb, q, k


def golden_of(xb, q, k):
    n = xb.shape[0]
    s = q.astype(np.float64) @ xb.astype(np.float64).T                 # [Q, N] float64
    keep = min(n, k + EXTRA)
    order = np.lexsort((np.broadcast_to(np.arange(n), s.shape), -s), axis=1)[:, :keep]
    s64 = np.take_along_axis(s, order, axis=1)
    ids = np.full((q.shape[0], k + EXTRA), -1, np.int64)
    sc = np.full((q.shape[0], k + EXTRA), -np.inf, np.float64)
    ids[:, :keep] = order
    sc[:, :keep] = s64
    D = np.full((q.shape[0], k), -np.finfo(np.float32).max, np.float32)
    I = np.full((q.shape[0], k), -1, np.int64)
    kk = min(k, n)
    I[:, :kk] = order[:, :kk]
    D[:, :kk] = s64[:, :kk].astype(np.float32)
    return D, I, sc, ids


def main():
    os.makedirs(GOLD, exist_ok=True)
    for name in CASES:
        xb, q, k = make_case(name)
        D, I, s64, ids64 = golden_of(xb, q, k)
        qn = np.sqrt((q.astype(np.float64) ** 2).sum(1))
        xmax = float(np.sqrt((xb.astype(np.float64) ** 2).sum(1)).max())
        np.savez_compressed(os.path.join(GOLD, "search_%s.npz" % name), D=D, I=I.astype(np.int32), s64=s64,
                            ids64=ids64.astype(np.int32), qnorm=qn, xmax=np.float64(xmax),
                            xb_checksum=np.float64(xb.astype(np.float64).sum()), q_checksum=np.float64(q.astype(np.float64).sum()))
        print(name, "D", D.shape, "kth", float(D[0, min(k, xb.shape[0]) - 1]))


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    main()
