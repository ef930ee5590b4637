"""GPU parity of the HIP index (through the C ABI) against the CPU oracle.

Oracle: oracle/flat_ip.c mode 1 reproduces the kernel's fp32 fma order, so D and I
must match BIT FOR BIT; independently, oracle.flat_ip.check_topk applies the
order-free rule of SURVEY.md §8c (float64 arbiter, tolerance 1e-5*|q|*max|x|).
Parity vs faiss itself is unpinned (faiss absent; see oracle/flat_ip.c header).
"""
import numpy as np
import pytest

from oracle import flat_ip

pytestmark = pytest.mark.gpu


def _mk(n, d, nq, seed):
    rng = np.random.RandomState(seed)
    return rng.randn(n, d).astype(np.float32), rng.randn(nq, d).astype(np.float32)


def _search(xb, q, k, chunks=1):
    from otter_amd.index import IndexFlatIP
    ix = IndexFlatIP(xb.shape[1])
    for part in np.array_split(xb, chunks):
        ix.add(part)
    assert ix.ntotal == xb.shape[0]
    D, I = ix.search(q, k)
    _also_exact_stream_kernels(ix, q, k, D, I)
    ix.close()
    return D, I


def _also_exact_stream_kernels(ix, q, k, D, I):
    """search() serves small batches on a large fp32 corpus with a coarse pass over its bf16 hi plane + exact
    re-scoring; the allocation-free entry point keeps them on the exact fp32 stream kernels.  Same bits either way."""
    import torch
    nq = q.shape[0]
    if not (1 <= nq <= 32 and k <= 128 and ix.d <= 4096):
        return
    qd = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).cuda()
    De = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    Ie = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    ix.search_noalloc(qd, k, De, Ie)
    np.testing.assert_array_equal(Ie.cpu().numpy(), np.asarray(I))
    np.testing.assert_array_equal(De.cpu().numpy().view(np.uint32), np.asarray(D).view(np.uint32))


CASES = [
    # n, d, nq, k
    (100, 768, 1, 1),
    (100, 768, 7, 100),
    (100, 768, 7, 1000),     # k > n: -1 padding
    (4096, 768, 1, 100),
    (4096, 768, 64, 100),
    (4096, 2304, 7, 100),
    (5000, 100, 33, 10),     # d not a multiple of 64, 64-query tile
    (65536, 768, 7, 100),
    (65536, 768, 200, 100),  # 128-query tile, ragged query count
    (65536, 768, 64, 1000),
    (20000, 768, 130, 2048),
    (300, 768, 5, 129),
    (70000, 64, 32, 500),
    # >= 32 * 16384 rows and <= 64 queries: the sample pass seeds every strip's threshold
    (600000, 768, 5, 100),
    (540000, 128, 40, 200),
    (530000, 64, 1, 1),
    (530000, 64, 64, 256),
    # stream kernel (<= 16 queries, k <= 128): every query-group width, ragged tiles, threshold exchange on
    # (>= 2 tiles per workgroup) and off, d = 2304 and a d that needs the padded copy of the queries
    (300000, 768, 2, 128),
    (300001, 768, 3, 100),
    (200000, 2304, 9, 100),
    (140000, 100, 16, 37),
    (1000, 768, 13, 128),
    (257, 64, 4, 100),
    # <= 64 queries with k > 256 on a corpus large enough for the sample pass: n >= 12 * 64 k
    (800000, 64, 40, 1000),
    (1600000, 64, 3, 2048),
    (900000, 64, 20, 1000),
]


@pytest.mark.parametrize("n,d,nq,k", CASES)
def test_search_matches_oracle(n, d, nq, k):
    xb, q = _mk(n, d, nq, seed=n + d + nq + k)
    D, I = _search(xb, q, k)
    Dr, Ir = flat_ip.search_c(xb, q, k, mode=flat_ip.MODE_GPU_ORDER)
    assert D.dtype == np.float32 and I.dtype == np.int64
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D.view(np.uint32), Dr.view(np.uint32))
    flat_ip.check_topk(xb, q, D, I, k, rtol=1e-5)


def test_stream_kernel_survives_a_threshold_that_passes_everything():
    """All rows equal: every block maximum ties the threshold, so every row of every tile is accepted and the
    candidate buffers compact after each tile; ids decide (0 .. k-1)."""
    from otter_amd.index import IndexFlatIP
    rng = np.random.RandomState(3)
    row = rng.randn(128).astype(np.float32)
    xb = np.tile(row, (300000, 1))
    q = rng.randn(5, 128).astype(np.float32)
    ix = IndexFlatIP(128)
    ix.add(xb)
    D, I = ix.search(q, 100)
    _also_exact_stream_kernels(ix, q, 100, D, I)
    assert (I == np.arange(100)[None, :]).all()
    assert (D == D[:, :1]).all()
    # the same corpus with one better row far away: it must come first
    xb[299999] = q[2] * 3.0
    ix.reset()
    ix.add(xb)
    D, I = ix.search(q, 100)
    _also_exact_stream_kernels(ix, q, 100, D, I)
    assert I[2, 0] == 299999 and (I[2, 1:] == np.arange(99)).all()
    ix.close()


def test_stream_kernel_repeated_calls_leave_clean_counters():
    """The exchange counters are zeroed by the merge of the previous search: results must not depend on history
    (different batch sizes and k alternate on one index)."""
    from otter_amd.index import IndexFlatIP
    xb, q = _mk(280000, 256, 16, seed=9)
    ix = IndexFlatIP(256)
    ix.add(xb)
    ref = {}
    for nq, k in [(1, 100), (16, 10), (5, 128), (1, 1), (9, 64), (1, 100), (16, 10)]:
        D, I = ix.search(q[:nq], k)
        _also_exact_stream_kernels(ix, q[:nq], k, D, I)
        key = (nq, k)
        if key not in ref:
            Dr, Ir = flat_ip.search_c(xb, q[:nq], k, mode=flat_ip.MODE_GPU_ORDER)
            np.testing.assert_array_equal(I, Ir)
            np.testing.assert_array_equal(D, Dr)
            ref[key] = (D, I)
        else:
            np.testing.assert_array_equal(I, ref[key][1])
            np.testing.assert_array_equal(D, ref[key][0])
    ix.close()


def test_incremental_add_and_reset():
    from otter_amd.index import IndexFlatIP
    xb, q = _mk(3000, 768, 9, seed=5)
    ix = IndexFlatIP(768)
    ix.add(xb[:1000]); ix.add(xb[1000:1001]); ix.add(xb[1001:])
    assert ix.ntotal == 3000 and ix.d == 768
    D, I = ix.search(q, 50)
    Dr, Ir = flat_ip.search_c(xb, q, 50)
    np.testing.assert_array_equal(I, Ir)
    ix.reset()
    assert ix.ntotal == 0
    D, I = ix.search(q, 5)
    assert (I == -1).all() and (D == flat_ip.NEG).all()
    ix.add(xb[:10])
    D, I = ix.search(q, 5)
    Dr, Ir = flat_ip.search_c(xb[:10], q, 5)
    np.testing.assert_array_equal(I, Ir)


def test_ties_duplicate_rows_and_all_equal():
    rng = np.random.RandomState(11)
    base = rng.randn(500, 768).astype(np.float32)
    xb = np.concatenate([base, base, base[:100]], 0)      # every row 2-3 times: exact score ties
    q = rng.randn(6, 768).astype(np.float32)
    D, I = _search(xb, q, 100)
    Dr, Ir = flat_ip.search_c(xb, q, 100)
    np.testing.assert_array_equal(I, Ir)                  # (score desc, id asc)
    np.testing.assert_array_equal(D, Dr)
    xb = np.ones((1000, 768), np.float32)                 # all scores equal: ids 0..k-1
    q = np.ones((3, 768), np.float32)
    D, I = _search(xb, q, 100)
    np.testing.assert_array_equal(I, np.tile(np.arange(100), (3, 1)))
    assert (D == 768.0).all()
    xb = np.zeros((700, 768), np.float32)
    xb[::2] *= -1.0                                       # -0.0 rows tie with +0.0 rows
    D, I = _search(xb, q, 10)
    np.testing.assert_array_equal(I, np.tile(np.arange(10), (3, 1)))


def test_sample_pass_ties_across_the_sample_boundary():
    # duplicates of the best rows sit on both sides of the sample boundary (row 16384) and all
    # scores of a block are equal: the seeded threshold must keep (score desc, id asc) exact
    rng = np.random.RandomState(17)
    n, d = 540000, 64
    xb = (rng.randn(n, d) * 0.1).astype(np.float32)
    q = rng.randn(9, d).astype(np.float32)
    best = (q[:3] * 4.0).astype(np.float32)
    for j, r in enumerate((5, 16383, 16384, 20000, 300000, 539999)):
        xb[r] = best[j % 3]                       # same top rows in the sample and in the main scan
    xb[100000:100300] = xb[7]                     # 300 exact copies of a sample row, far in the main scan
    D, I = _search(xb, q, 100)
    Dr, Ir = flat_ip.search_c(xb, q, 100, mode=flat_ip.MODE_GPU_ORDER)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D.view(np.uint32), Dr.view(np.uint32))
    xb[:] = 1.0                                   # every score equal: ids 0..k-1
    D, I = _search(xb, np.ones((3, d), np.float32), 50)
    np.testing.assert_array_equal(I, np.tile(np.arange(50), (3, 1)))


def test_negative_and_infinite_scores():
    xb, q = _mk(2000, 768, 4, seed=3)
    xb[5] = -np.inf
    xb[7] = np.inf
    xb *= 1.0
    q = np.abs(q)
    D, I = _search(xb, q, 2000)
    assert (I[:, 0] == 7).all() and (I[:, -1] == 5).all()
    assert np.isposinf(D[:, 0]).all() and np.isneginf(D[:, -1]).all()


def test_torch_device_io_and_id_base():
    import torch
    from otter_amd.index import IndexFlatIP
    xb, q = _mk(10000, 768, 40, seed=8)
    ix = IndexFlatIP(768, device=0, id_base=1 << 33)
    ix.add(torch.from_numpy(xb).cuda())
    D, I = ix.search(torch.from_numpy(q).cuda(), 100)
    assert D.is_cuda and I.is_cuda and I.dtype == torch.int64
    Dr, Ir = flat_ip.search_c(xb, q, 100, id_base=1 << 33)
    np.testing.assert_array_equal(I.cpu().numpy(), Ir)
    np.testing.assert_array_equal(D.cpu().numpy(), Dr)


def test_merge_topk_matches_oracle():
    import torch
    from otter_amd.index import merge_topk
    rng = np.random.RandomState(2)
    G, nq, k = 8, 50, 100
    Dc = np.sort(rng.randn(G, nq, k).astype(np.float32), axis=2)[:, :, ::-1].copy()
    Ic = rng.randint(0, 1 << 40, size=(G, nq, k)).astype(np.int64)
    Dc[3, :, 60:] = flat_ip.NEG
    Ic[3, :, 60:] = -1                                    # a short shard
    Dc[5, 0, :] = Dc[4, 0, :]                             # cross-shard score ties
    Dr, Ir = flat_ip.merge_c(Dc, Ic)
    D, I = merge_topk(torch.from_numpy(Dc).cuda(), torch.from_numpy(Ic).cuda())
    np.testing.assert_array_equal(I.cpu().numpy(), Ir)
    np.testing.assert_array_equal(D.cpu().numpy(), Dr)


def test_errors_are_loud():
    from otter_amd.index import IndexFlatIP
    ix = IndexFlatIP(768)
    with pytest.raises(ValueError):
        ix.add(np.zeros((3, 5), np.float32))
    with pytest.raises(ValueError):
        ix.search(np.zeros((3, 768), np.float32), 0)
    with pytest.raises(ValueError):
        ix.search(np.zeros((3, 768), np.float32), 4096)
    with pytest.raises(ValueError):
        IndexFlatIP(0)


@pytest.mark.parametrize("n,d,nq,k,mmax", [(5000, 768, 9, 100, 700), (3000, 2304, 4, 10, 64), (70000, 128, 33, 100, 3000),
                                            (500, 64, 5, 50, 20)])
def test_subset_search_matches_a_per_query_index(n, d, nq, k, mmax):
    # the reference builds IndexFlatIP over each question's filtered blocks and searches it with that one
    # question (eval_ottqa_retrieval_tfidffilter.py:530-546): the oracle does literally that, the library does
    # one batched call; ids exact, scores bit-exact (same fp32 chain as the full search)
    from otter_amd.index import IndexFlatIP
    rng = np.random.RandomState(n + d + nq + k)
    xb, q = _mk(n, d, nq, seed=n + k)
    xb[17] = xb[3]                                               # an exact tie inside some subsets
    subsets = []
    for i in range(nq):
        m = int(rng.randint(1, mmax + 1)) if i else mmax          # ragged; query 0 has the longest list
        s = np.sort(rng.choice(n, size=m, replace=False))
        if i == 1:
            s = np.unique(np.concatenate([s, [3, 17]]))[:mmax]
        subsets.append(s)
    ix = IndexFlatIP(d)
    ix.add(xb)
    D, I = ix.search_subsets(q, k, subsets)
    D2, P = ix.search_subsets(q, k, subsets, return_positions=True)
    Df, If = ix.search(q, min(k, n))
    ix.close()
    assert D.dtype == np.float32 and I.dtype == np.int64 and D.shape == (nq, k)
    for i in range(nq):
        s = subsets[i]
        kk = min(k, len(s))
        Dr, Ir = flat_ip.search_c(xb[s], q[i:i + 1], k, mode=flat_ip.MODE_GPU_ORDER)
        np.testing.assert_array_equal(P[i], Ir[0])              # the reference's convention: positions in the subset
        np.testing.assert_array_equal(I[i, :kk], s[Ir[0, :kk]])
        assert (I[i, kk:] == -1).all()
        np.testing.assert_array_equal(D[i].view(np.uint32), Dr[0].view(np.uint32))
        # a row that is in the subset AND in the full search's top k carries the same score there
        common = np.intersect1d(I[i, :kk], If[i])
        for r in common[:5]:
            assert D[i][list(I[i]).index(r)] == Df[i][list(If[i]).index(r)]
    np.testing.assert_array_equal(D, D2)


def test_subset_search_device_io_id_base_and_errors():
    import torch
    from otter_amd.index import IndexFlatIP
    xb, q = _mk(2000, 96, 6, seed=3)
    ix = IndexFlatIP(96, id_base=1000)
    ix.add(xb)
    ids = np.full((6, 40), -1, np.int64)
    rng = np.random.RandomState(0)
    for i in range(6):
        m = int(rng.randint(5, 41))
        ids[i, :m] = 1000 + np.sort(rng.choice(2000, size=m, replace=False))
    ids[2, [1, 4]] = -1                                           # holes inside a list are skipped
    Dh, Ih = ix.search_subsets(q, 8, ids)
    Dd, Id = ix.search_subsets(torch.from_numpy(q).cuda(), 8, torch.from_numpy(ids).cuda())
    np.testing.assert_array_equal(Ih, Id.cpu().numpy())
    np.testing.assert_array_equal(Dh.view(np.uint32), Dd.cpu().numpy().view(np.uint32))
    for i in range(6):
        s = ids[i][ids[i] >= 0] - 1000
        Dr, Ir = flat_ip.search_c(xb[s], q[i:i + 1], 8, mode=flat_ip.MODE_GPU_ORDER)
        kk = min(8, len(s))
        np.testing.assert_array_equal(Ih[i, :kk], 1000 + s[Ir[0, :kk]])
    bad = ids.copy()
    bad[0, 0] = 5                                                 # below id_base
    with pytest.raises(ValueError):
        ix.search_subsets(q, 8, bad)
    bad[0, 0] = 3000                                              # past the last row
    with pytest.raises(ValueError):
        ix.search_subsets(q, 8, bad)
    with pytest.raises(ValueError):
        ix.search_subsets(q, 8, ids[:3])
    ix.close()
    ib = IndexFlatIP(96, dtype="bf16")
    ib.add(xb)
    with pytest.raises(NotImplementedError):
        ib.search_subsets(q, 8, np.zeros((6, 4), np.int64))
    ib.close()


def test_reconstruct_and_index_file_round_trip(tmp_path):
    from otter_amd import index_io as IO
    from otter_amd.index import IndexFlatIP
    xb, q = _mk(70000, 100, 9, seed=8)
    ix = IndexFlatIP(100)
    ix.add(xb[:30000])
    ix.add(xb[30000:])
    np.testing.assert_array_equal(ix.reconstruct_n(), xb)
    np.testing.assert_array_equal(ix.reconstruct_n(65530, 100), xb[65530:65630])
    with pytest.raises(ValueError):
        ix.reconstruct_n(69999, 2)
    D, I = ix.search(q, 10)
    path = str(tmp_path / "corpus.index")
    IO.write_index(ix, path)
    ix.close()
    ix2 = IO.read_index(path)
    assert ix2.ntotal == 70000 and ix2.d == 100
    D2, I2 = ix2.search(q, 10)
    ix2.close()
    np.testing.assert_array_equal(I, I2)
    np.testing.assert_array_equal(D.view(np.uint32), D2.view(np.uint32))
    # the bf16 index hands back its rounded rows (tile-blocked storage underneath)
    import torch
    ib = IndexFlatIP(100, dtype="bf16")
    ib.add(xb[:1000])
    ib.add(xb[1000:3001])
    exp = torch.from_numpy(xb[:3001]).to(torch.bfloat16).float().numpy()
    np.testing.assert_array_equal(ib.reconstruct_n(), exp)
    ib.close()
