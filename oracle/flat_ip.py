"""oracle/flat_ip.py — CPU restatement of faiss.IndexFlatIP (numpy) + loader for flat_ip.c.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by otter_amd/.

PARITY UNPINNED — see the header of oracle/flat_ip.c: faiss is an un-vendored,
version-unpinned dependency (reference README.md:23) that is absent here, and the
reference holds no fixture for this path.  The restatement follows the call sites
scripts/eval_ottqa_retrieval.py:108-109 (IndexFlatIP(d); add(xb)) and :129
(D, I = index.search(q, k)).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_flat_ip.so")
NEG = -np.finfo(np.float32).max  # faiss pads D with -FLT_MAX and I with -1


def build(force=False):
    """Compile oracle/flat_ip.c with gcc (also called from __graft_entry__.build)."""
    src = os.path.join(_HERE, "flat_ip.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-mavx2", "-mfma", "-fopenmp", "-shared", "-fPIC",
                               src, "-o", _SO, "-lm"])
    return _SO


_lib = None


def _c():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = ctypes.CDLL(_SO)
        lib.oracle_flat_ip_search.restype = ctypes.c_int
        lib.oracle_flat_ip_search.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                              ctypes.c_int64, ctypes.c_int]
        lib.oracle_merge_topk.restype = ctypes.c_int
        lib.oracle_merge_topk.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.oracle_num_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


MODE_PLAIN, MODE_GPU_ORDER, MODE_F64, MODE_SIMD8 = 0, 1, 2, 3


def search_c(xb, q, k, mode=MODE_GPU_ORDER, id_base=0, nthreads=0):
    """Per-query scan + heap k-select in C (the shape of the reference's nq=1 calls,
    scripts/eval_ottqa_retrieval.py:126-129)."""
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    n, d = (xb.shape if xb.ndim == 2 else (0, q.shape[1]))
    nq = q.shape[0]
    assert q.shape[1] == d
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    rc = _c().oracle_flat_ip_search(xb.ctypes.data, n, d, q.ctypes.data, nq, k,
                                    D.ctypes.data, I.ctypes.data, mode, id_base, nthreads)
    if rc != 0:
        raise RuntimeError("oracle_flat_ip_search failed: %d" % rc)
    return D, I


def merge_c(Dc, Ic):
    Dc = np.ascontiguousarray(Dc, dtype=np.float32)
    Ic = np.ascontiguousarray(Ic, dtype=np.int64)
    G, nq, k = Dc.shape
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    rc = _c().oracle_merge_topk(Dc.ctypes.data, Ic.ctypes.data, G, nq, k, D.ctypes.data, I.ctypes.data)
    if rc != 0:
        raise RuntimeError("oracle_merge_topk failed: %d" % rc)
    return D, I


def num_threads():
    return _c().oracle_num_threads()


class IndexFlatIP:
    """numpy restatement with the faiss surface the reference uses: d, ntotal, add, search, reset.

    search() is the batched form faiss itself takes for nq >= 20: blocked sgemm
    (numpy -> BLAS) followed by a per-row k-select.  Ties: (score desc, id asc)."""

    def __init__(self, d):
        self.d = int(d)
        self._chunks = []
        self.ntotal = 0

    def add(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError("add: expected [n, %d] float32, got %r" % (self.d, x.shape))
        self._chunks.append(x.copy())
        self.ntotal += x.shape[0]

    def reset(self):
        self._chunks = []
        self.ntotal = 0

    def _xb(self):
        if len(self._chunks) > 1:
            self._chunks = [np.concatenate(self._chunks, axis=0)]
        return self._chunks[0] if self._chunks else np.zeros((0, self.d), np.float32)

    def search(self, q, k, block=65536, float64=False):
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError("search: expected [nq, %d] float32, got %r" % (self.d, q.shape))
        xb = self._xb()
        nq, n = q.shape[0], xb.shape[0]
        D = np.full((nq, k), NEG, dtype=np.float32)
        I = np.full((nq, k), -1, dtype=np.int64)
        if n == 0 or nq == 0:
            return D, I
        qq = q.astype(np.float64) if float64 else q
        best_s = np.empty((nq, 0), dtype=np.float32)
        best_i = np.empty((nq, 0), dtype=np.int64)
        for lo in range(0, n, block):
            hi = min(n, lo + block)
            xx = xb[lo:hi].astype(np.float64) if float64 else xb[lo:hi]
            s = (qq @ xx.T).astype(np.float32)
            ids = np.broadcast_to(np.arange(lo, hi, dtype=np.int64), s.shape)
            cs = np.concatenate([best_s, s], axis=1)
            ci = np.concatenate([best_i, ids], axis=1)
            kk = min(k, cs.shape[1])
            # stable (score desc, id asc): lexsort by id then by -score
            order = np.lexsort((ci, -cs.astype(np.float64)), axis=1)[:, :kk]
            best_s = np.take_along_axis(cs, order, axis=1)
            best_i = np.take_along_axis(ci, order, axis=1)
        kk = best_s.shape[1]
        D[:, :kk] = best_s
        I[:, :kk] = best_i
        return D, I


def search_batched(xb, q, k, block=16384):
    """The batched form faiss takes for large nq: sgemm over corpus blocks (numpy -> BLAS,
    multi-threaded) + per-row k-select (argpartition), exact under (score desc, id asc).
    Used as the timed CPU baseline of bench.py ("port"); same results as IndexFlatIP.search."""
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    nq, n = q.shape[0], xb.shape[0]
    D = np.full((nq, k), NEG, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if n == 0 or nq == 0:
        return D, I
    best_s = np.empty((nq, 0), dtype=np.float32)
    best_i = np.empty((nq, 0), dtype=np.int64)
    rows = np.arange(nq)[:, None]
    for lo in range(0, n, block):
        hi = min(n, lo + block)
        s = q @ xb[lo:hi].T
        nb = hi - lo
        kk = min(k, nb)
        if kk < nb:
            sel = np.argpartition(-s, kk - 1, axis=1)[:, :kk]
            thr = s[rows, sel].min(axis=1)
            # ties at the cut: argpartition keeps an arbitrary subset of the rows equal to thr
            tied = (s >= thr[:, None]).sum(axis=1) > kk
            for r in np.nonzero(tied)[0]:
                sel[r] = np.lexsort((np.arange(nb), -s[r].astype(np.float64)))[:kk]
        else:
            sel = np.broadcast_to(np.arange(nb), (nq, nb))
        cs = np.concatenate([best_s, s[rows, sel]], axis=1)
        ci = np.concatenate([best_i, sel.astype(np.int64) + lo], axis=1)
        order = np.lexsort((ci, -cs.astype(np.float64)), axis=1)[:, :min(k, cs.shape[1])]
        best_s = np.take_along_axis(cs, order, axis=1)
        best_i = np.take_along_axis(ci, order, axis=1)
    kk = best_s.shape[1]
    D[:, :kk] = best_s
    I[:, :kk] = best_i
    return D, I


def scores_f64(xb, q, ids):
    """float64 scores of the given (query, id) pairs; ids [nq, k] with -1 padding."""
    xb = np.asarray(xb, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    safe = np.where(ids < 0, 0, ids)
    out = np.einsum("qkd,qd->qk", xb[safe], q)
    out[ids < 0] = -np.inf
    return out


def check_topk(xb, q, D, I, k, rtol=1e-5, id_base=0):
    """The comparison rule of SURVEY.md §8c for a result (D, I) of search(q, k) over xb.

    Independent of any summation order: with float64 scores as the arbiter,
      * |D - s64(I)| <= tol, tol = rtol * |q| * max|x|   (a bound on fp32 dot error)
      * D is non-increasing along each row and ids are unique (padding aside)
      * every returned row scores >= (true k-th best score - 2 tol): ids may only
        differ from the exact top-k inside runs of scores closer than the tolerance
      * when n < k the tail is I = -1, D = -FLT_MAX
    Returns the number of positions whose id differs from the float64 ranking
    (all of them inside near-tie runs when this function does not raise)."""
    xb = np.asarray(xb, dtype=np.float32)
    q = np.asarray(q, dtype=np.float32)
    n = xb.shape[0]
    nq = q.shape[0]
    assert D.shape == (nq, k) and I.shape == (nq, k), (D.shape, I.shape)
    assert D.dtype == np.float32 and I.dtype == np.int64
    kk = min(k, n)
    if kk < k:
        assert (I[:, kk:] == -1).all(), "tail ids must be -1"
        assert (D[:, kk:] == NEG).all(), "tail scores must be -FLT_MAX"
    if kk == 0 or nq == 0:
        return 0
    Iv = I[:, :kk] - id_base
    assert (Iv >= 0).all() and (Iv < n).all(), "ids out of range"
    srt = np.sort(Iv, axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all(), "duplicate ids in a row"
    assert (D[:, 1:kk] <= D[:, : kk - 1]).all(), "scores not sorted descending"
    xmax = float(np.sqrt((xb.astype(np.float64) ** 2).sum(1)).max())
    qn = np.sqrt((q.astype(np.float64) ** 2).sum(1))
    tol = rtol * qn * xmax + 1e-30
    s64 = scores_f64(xb, q, Iv)
    err = np.abs(D[:, :kk].astype(np.float64) - s64)
    assert (err <= tol[:, None]).all(), "score error %g > tol %g" % (err.max(), tol.min())
    ndiff = 0
    xb64 = xb.astype(np.float64)
    for iq in range(nq):
        full = xb64 @ q[iq].astype(np.float64)
        order = np.lexsort((np.arange(n), -full))[:kk]
        kth = full[order[-1]]
        assert (s64[iq] >= kth - 2 * tol[iq]).all(), "query %d returned a row outside the top-%d" % (iq, k)
        ndiff += int((order != Iv[iq]).sum())
    return ndiff


def check_against_golden(g, D, I, k):
    """The comparison rule of SURVEY.md §8c against a committed fixture of oracle/gen_search_goldens.py
    (g: the loaded npz).  With tol = 1e-5 * |q| * max|x|:
      * position by position, the returned id must be a row of the float64 top-(k+8) whose float64 score is
        within tol of the score the float64 ranking has at that position (ids may differ from the float64
        ranking only inside runs of scores closer than the tolerance);
      * |D - float64 score of the returned row| <= tol; D non-increasing; ids unique;
      * rows the corpus does not have: I = -1, D = -FLT_MAX.
    Returns the number of positions whose id differs from the float64 ranking."""
    Ig, s64, ids64 = g["I"].astype(np.int64), g["s64"], g["ids64"].astype(np.int64)
    nq = Ig.shape[0]
    assert D.shape == (nq, k) and I.shape == (nq, k) and D.dtype == np.float32 and I.dtype == np.int64
    tol = 1e-5 * g["qnorm"] * float(g["xmax"]) + 1e-30
    ndiff = 0
    for iq in range(nq):
        nvalid = int((Ig[iq] >= 0).sum())
        assert (I[iq, nvalid:] == -1).all() and (D[iq, nvalid:] == NEG).all(), "padding of query %d" % iq
        got = I[iq, :nvalid]
        assert len(set(got.tolist())) == nvalid, "duplicate ids in query %d" % iq
        assert (D[iq, 1:nvalid] <= D[iq, :nvalid - 1]).all(), "query %d not sorted" % iq
        score_of = {int(i): float(s) for i, s in zip(ids64[iq], s64[iq]) if i >= 0}
        for p in range(nvalid):
            gid = int(got[p])
            assert gid in score_of, "query %d position %d: row %d is not in the float64 top-(k+8)" % (iq, p, gid)
            sg = score_of[gid]
            assert abs(sg - s64[iq, p]) <= 2 * tol[iq], "query %d position %d: row %d ranks elsewhere (%g vs %g)" % (
                iq, p, gid, sg, s64[iq, p])
            assert abs(float(D[iq, p]) - sg) <= tol[iq], "query %d position %d: score %g, float64 %g" % (iq, p, D[iq, p], sg)
            ndiff += int(gid != Ig[iq, p])
    return ndiff
