"""GPU parity of the two bf16-matrix-core search paths (otter_amd/csrc/index.hip: scan16_kernel).

* OTTER_F32 index, large query batch ("split" path): coarse pass on hi/lo bf16 planes + exact
  re-scoring + per-query proof, exact kernel for unproven queries.  The result must be what the exact
  kernel returns: D and I BIT-EXACT against oracle mode 1 (oracle/flat_ip.c), ties included.
* OTTER_BF16 index (BASELINE config 5 storage): rows and queries rounded to bfloat16, fp32
  accumulation.  Reported separately from fp32 parity, as north_star says: checked with the order-free
  rule of SURVEY.md §8c against float64 scores of the ROUNDED operands, plus recall@k against the
  fp32 index of the unrounded data.
"""
import numpy as np
import pytest
import torch

from oracle import flat_ip

pytestmark = pytest.mark.gpu


def _mk(n, d, nq, seed):
    rng = np.random.RandomState(seed)
    return rng.randn(n, d).astype(np.float32), rng.randn(nq, d).astype(np.float32)


def _bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()


def _search(xb, q, k, dtype="fp32"):
    from otter_amd.index import IndexFlatIP
    ix = IndexFlatIP(xb.shape[1], dtype=dtype)
    ix.add(xb)
    D, I = ix.search(q, k)
    fb = ix.last_fallback
    ix.close()
    return D, I, fb


@pytest.mark.parametrize("n,d,nq,k", [(70000, 768, 200, 100), (66000, 128, 700, 10), (131072, 768, 256, 1),
                                        (80000, 64, 300, 500), (70001, 2304, 193, 100), (90000, 128, 70, 1000)])
def test_split_path_is_bit_exact(n, d, nq, k):
    xb, q = _mk(n, d, nq, seed=n + d + nq + k)
    D, I, fb = _search(xb, q, k)
    Dr, Ir = flat_ip.search_c(xb, q, k, mode=flat_ip.MODE_GPU_ORDER)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D.view(np.uint32), Dr.view(np.uint32))
    assert fb <= nq // 20, "gaussian data should almost never need the exact fallback (%d of %d)" % (fb, nq)


def test_split_path_falls_back_on_near_ties_and_stays_exact():
    # 70,000 rows that are tiny perturbations of 40 prototypes: hundreds of rows per query sit inside the
    # coarse pass's error bound of the k-th score, so the proof fails and the exact kernel must take over
    rng = np.random.RandomState(3)
    n, d, nq, k = 70000, 256, 200, 100
    proto = rng.randn(40, d).astype(np.float32)
    xb = proto[rng.randint(0, 40, size=n)] * (1.0 + 1e-6 * rng.randn(n, 1).astype(np.float32))
    xb = xb.astype(np.float32)
    xb[1000:1100] = xb[7]                                   # exact duplicates too: (score desc, id asc)
    q = rng.randn(nq, d).astype(np.float32)
    D, I, fb = _search(xb, q, k)
    Dr, Ir = flat_ip.search_c(xb, q, k, mode=flat_ip.MODE_GPU_ORDER)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D.view(np.uint32), Dr.view(np.uint32))
    assert fb > 0, "this corpus is built to defeat the margin proof"


def test_two_product_pass_hands_over_to_the_three_product_pass():
    # clusters whose members differ by ~6 score units: too close for the two-product pass's bound
    # (|q - q_hi| max|x| ~ 1.3) to prove the top 100 with 60 spare candidates, far enough apart for the
    # three-product bound (~0.15).  All 200 queries are searched again by the three-product pass, none by the
    # exact kernel; the corpus then starts with three products until reset().
    from otter_amd.index import IndexFlatIP
    rng = np.random.RandomState(5)
    n, d, nq, k = 70000, 768, 200, 100
    proto = rng.randn(40, d).astype(np.float32)
    xb = (proto[rng.randint(0, 40, size=n)] + 0.22 * rng.randn(n, d)).astype(np.float32)
    q = rng.randn(nq, d).astype(np.float32)
    Dr, Ir = flat_ip.search_c(xb, q, k, mode=flat_ip.MODE_GPU_ORDER)
    ix = IndexFlatIP(d)
    ix.add(xb)
    D, I = ix.search(q, k)
    refined, fb = ix.last_refined, ix.last_fallback
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D.view(np.uint32), Dr.view(np.uint32))
    assert refined > nq // 2, "built to defeat the two-product proof (%d of %d refined)" % (refined, nq)
    assert fb <= nq // 10, "the three-product pass should prove these (%d to the exact kernel)" % fb
    D2, I2 = ix.search(q, k)                                 # sticky: straight to three products
    assert ix.last_refined == 0
    np.testing.assert_array_equal(I2, Ir)
    np.testing.assert_array_equal(D2.view(np.uint32), Dr.view(np.uint32))
    ix.reset()
    ix.add(rng.randn(n, d).astype(np.float32))              # a friendly corpus again: two products suffice
    ix.search(q, k)
    assert ix.last_refined <= 2 and ix.last_fallback <= 2
    ix.close()


def test_split_path_non_finite_rows_and_incremental_add():
    from otter_amd.index import IndexFlatIP
    xb, q = _mk(90000, 128, 220, seed=21)
    q = np.abs(q)
    xb[5] = np.inf
    xb[77777] = -np.inf
    ix = IndexFlatIP(128)
    ix.add(xb[:30000])
    D0, I0 = ix.search(q, 20)                               # n < 65536: exact kernel
    ix.add(xb[30000:])                                      # planes are built / extended on the next large batch
    D, I = ix.search(q, 20)
    assert ix.last_fallback == q.shape[0]                   # an infinite norm bound proves nothing: all exact
    ix.close()
    Dr, Ir = flat_ip.search_c(xb, q, 20, mode=flat_ip.MODE_GPU_ORDER)
    np.testing.assert_array_equal(I, Ir)
    assert (I[:, 0] == 5).all() and np.isposinf(D[:, 0]).all()
    np.testing.assert_array_equal(I0, flat_ip.search_c(xb[:30000], q, 20, mode=flat_ip.MODE_GPU_ORDER)[1])


@pytest.mark.parametrize("n,d,nq,k", [(5000, 768, 7, 100), (70000, 768, 300, 100), (66000, 100, 33, 1000),
                                        (300, 64, 5, 400), (100000, 768, 1, 10)])
def test_bf16_index_against_rounded_operands(n, d, nq, k):
    xb, q = _mk(n, d, nq, seed=n + d + nq + k + 1)
    D, I, _ = _search(xb, q, k, dtype="bf16")
    assert D.dtype == np.float32 and I.dtype == np.int64
    xr, qr = _bf16_round(xb), _bf16_round(q)
    # scores are exact products of bf16 values accumulated in fp32 (any order): 1e-5 |q| max|x| covers it
    flat_ip.check_topk(xr, qr, D, I, k, rtol=1e-5)
    # recall@k against the fp32 ranking of the unrounded data (reported, with a loose floor)
    _, Ir = flat_ip.search_c(xb, q, min(k, n), mode=flat_ip.MODE_GPU_ORDER)
    kk = min(k, n)
    rec = np.mean([len(set(I[i, :kk]) & set(Ir[i])) / kk for i in range(nq)])
    assert rec >= 0.9, rec


@pytest.mark.parametrize("n,d,k", [(300000, 768, 100), (70000, 100, 128), (999, 768, 10)])
def test_bf16_small_batches_share_the_large_batch_arithmetic(n, d, k):
    """<= 32 queries take stream16_kernel (one or two blocks of 16 query columns), larger batches scan16_kernel<0>:
    same instruction, same operand roles, same k order -> the same scores and rows, bit for bit, whatever the
    batch size."""
    from otter_amd.index import IndexFlatIP
    xb, q = _mk(n, d, 40, seed=n + d + k)
    ix = IndexFlatIP(d, dtype="bf16")
    ix.add(xb)
    Dl, Il = ix.search(q, k)                      # 40 queries: scan16 path
    for lo, hi in [(0, 1), (1, 4), (4, 20), (20, 37), (33, 40), (0, 32), (8, 33)]:
        D, I = ix.search(q[lo:hi], k)
        np.testing.assert_array_equal(I, Il[lo:hi])
        np.testing.assert_array_equal(D.view(np.int32), Dl[lo:hi].view(np.int32))
    flat_ip.check_topk(_bf16_round(xb), _bf16_round(q[:7]), *ix.search(q[:7], k), k, rtol=1e-5)
    ix.close()


def test_bf16_index_incremental_device_add_and_limits():
    from otter_amd.index import IndexFlatIP
    xb, q = _mk(20000, 768, 16, seed=2)
    ix = IndexFlatIP(768, dtype="bf16", id_base=1 << 35)
    ix.add(xb[:7000])
    ix.add(torch.from_numpy(xb[7000:]).cuda())
    assert ix.ntotal == 20000
    D, I = ix.search(torch.from_numpy(q).cuda(), 50)
    flat_ip.check_topk(_bf16_round(xb), _bf16_round(q), D.cpu().numpy(), I.cpu().numpy(), 50, rtol=1e-5,
                       id_base=1 << 35)
    with pytest.raises(ValueError):
        ix.search(q, 2000)                                  # the bf16 scan keeps at most 1280 per query
    ix.reset()
    D, I = ix.search(q, 5)
    assert (I == -1).all()
    ix.close()


def test_coarse_pass_switches_itself_off_on_a_corpus_it_cannot_prove():
    """Near-duplicate rows: every coarse pass leaves most queries unproven, the exact kernel re-does them.  After one
    such batch the index stops paying for coarse passes (coarse_state 4) until reset(); results stay the oracle's."""
    from otter_amd.index import IndexFlatIP
    rng = np.random.RandomState(4)
    n, d, nq, k = 70000, 256, 120, 50
    proto = rng.randn(30, d).astype(np.float32)
    xb = (proto[rng.randint(0, 30, size=n)] * (1.0 + 1e-6 * rng.randn(n, 1).astype(np.float32))).astype(np.float32)
    q = rng.randn(nq, d).astype(np.float32)
    Dr, Ir = flat_ip.search_c(xb, q, k, mode=flat_ip.MODE_GPU_ORDER)
    ix = IndexFlatIP(d)
    ix.add(xb)
    D, I = ix.search(q, k)
    assert ix.last_fallback > nq // 4 and ix.coarse_state == 4
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D.view(np.uint32), Dr.view(np.uint32))
    D2, I2 = ix.search(q, k)                          # served by the exact kernel alone
    assert ix.last_fallback == 0
    np.testing.assert_array_equal(I2, Ir)
    np.testing.assert_array_equal(D2.view(np.uint32), Dr.view(np.uint32))
    ix.reset()
    g = np.random.RandomState(5).randn(n, d).astype(np.float32)
    ix.add(g)
    ix.search(q, k)
    assert ix.coarse_state == 1 and ix.last_fallback == 0
    ix.close()


def test_one_query_calls_also_switch_the_coarse_pass_off():
    """The reference's call pattern (one query per search() call) on a near-duplicate corpus large enough for the
    small-batch coarse pass (>= 2e8 values): no single call is a batch worth a verdict, so the verdict is taken over
    a running window of 64 queries -- after that the calls are served by the exact stream kernels alone (nothing
    left to re-search: last_fallback 0); results are the oracle's throughout."""
    from otter_amd.index import IndexFlatIP
    rng = np.random.RandomState(4)
    n, d, k = 270000, 768, 20
    proto = rng.randn(30, d).astype(np.float32)
    xb = (proto[rng.randint(0, 30, size=n)] * (1.0 + 1e-6 * rng.randn(n, 1).astype(np.float32))).astype(np.float32)
    q = rng.randn(80, d).astype(np.float32)
    Dr, Ir = flat_ip.search_c(xb, q, k, mode=flat_ip.MODE_GPU_ORDER)
    ix = IndexFlatIP(d)
    ix.add(xb)
    redone = []
    for i in range(q.shape[0]):
        D, I = ix.search(q[i:i + 1], k)
        np.testing.assert_array_equal(I, Ir[i:i + 1])
        np.testing.assert_array_equal(D.view(np.uint32), Dr[i:i + 1].view(np.uint32))
        redone.append(ix.last_fallback)
    assert sum(redone[:64]) > 16 and sum(redone[64:]) == 0, redone
    ix.close()


def test_optimistic_threshold_failures_are_searched_again(capfd):
    """Large batches start their coarse scan from an optimistic threshold (the 12th best of a small sample) and verify
    afterwards.  Forced to fail here: rank 1 (the best sample score) is far above the true kc-th best, so nearly every
    query misses candidates, is flagged by the check and goes through the next pass -- the result must still be the
    oracle's, bit for bit; with the default rank nothing is refined.  A batch that misses its guesses for more than a
    quarter of its queries says so once and sends the index back to the sample levels -- not to the two-product pass:
    the guess failed, not the one-product bound -- where the same batch proves every query again."""
    import os
    from otter_amd.index import IndexFlatIP
    rng = np.random.RandomState(11)
    n, d, nq, k = 300000, 128, 600, 100
    xb = rng.randn(n, d).astype(np.float32)
    q = rng.randn(nq, d).astype(np.float32)
    Dr, Ir = flat_ip.search_c(xb, q[:64], k, mode=flat_ip.MODE_GPU_ORDER)
    ix = IndexFlatIP(d)
    ix.add(xb)
    D0, I0 = ix.search(q, k)
    assert ix.last_refined == 0 and ix.last_fallback == 0
    np.testing.assert_array_equal(I0[:64], Ir)
    np.testing.assert_array_equal(D0[:64].view(np.uint32), Dr.view(np.uint32))
    os.environ["OTTER_OPT_THR"] = "1"
    os.environ["OTTER_OPT_L0"] = "28"
    try:
        D1, I1 = ix.search(q, k)
        refined = ix.last_refined
    finally:
        del os.environ["OTTER_OPT_THR"], os.environ["OTTER_OPT_L0"]
    assert refined > nq // 2, refined
    np.testing.assert_array_equal(I1, I0)
    np.testing.assert_array_equal(D1.view(np.uint32), D0.view(np.uint32))
    assert "optimistic threshold" in capfd.readouterr().err
    for _ in range(2):                         # sample levels from here on: same bits, nothing refined, no second note
        D2, I2 = ix.search(q, k)
        assert ix.last_refined == 0 and ix.last_fallback == 0
        np.testing.assert_array_equal(I2, I0)
        np.testing.assert_array_equal(D2.view(np.uint32), D0.view(np.uint32))
    assert "optimistic threshold" not in capfd.readouterr().err
    ix.close()


def test_planes_setting_changes_the_footprint_not_the_result():
    """otter_index_set_planes: 'all' (2.5x the corpus in HBM), 'hi' (1.5x: one-product pass only), 'none' (1.0x: exact
    kernels only) -- the same result bits for one query, a small and a large batch; switching back and forth frees
    and rebuilds the planes."""
    from otter_amd.index import IndexFlatIP
    g = torch.Generator(device="cuda").manual_seed(21)
    n, d, k = 300000, 768, 100
    xb = torch.randn((n, d), generator=g, device="cuda")
    q = torch.randn((600, d), generator=g, device="cuda")
    ix = IndexFlatIP(d, planes="none")
    ix.add(xb)
    ref = {nq: ix.search(q[:nq].contiguous(), k) for nq in (1, 40, 600)}
    assert ix.coarse_state == 0                       # no planes were built
    free_none = torch.cuda.mem_get_info()[0]
    used = []
    for planes in ("hi", "all", "hi", "none"):
        ix.set_planes(planes)
        for nq, (Dr, Ir) in ref.items():
            D, I = ix.search(q[:nq].contiguous(), k)
            assert torch.equal(I, Ir) and torch.equal(D.view(torch.int32), Dr.view(torch.int32)), (planes, nq)
        torch.cuda.synchronize()
        used.append(free_none - torch.cuda.mem_get_info()[0])
    # (the searches' own workspace is in every figure; the planes are what differs)
    corpus = n * d * 4
    hi1, all2, hi3, none4 = used
    assert 0.9 * corpus < all2 - hi3 < 1.1 * corpus, (all2 - hi3) / corpus      # the hi/lo planes: the corpus' bytes again
    assert 0.45 * corpus < hi3 - none4 < 0.6 * corpus, (hi3 - none4) / corpus   # the hi plane: half of them
    with pytest.raises(ValueError):
        ix.set_planes("some")
    ix.close()


@pytest.mark.parametrize("n,d,nq,k", [(70000, 768, 33, 100), (70001, 768, 64, 100), (66000, 832, 50, 10),
                                        (131072, 128, 64, 1), (80000, 64, 48, 100), (70000, 960, 37, 20),
                                        (70000, 1024, 40, 100)])
def test_33_to_64_queries_are_bit_exact(n, d, nq, k):
    """33 ... 64 queries of an fp32 index: the coarse pass is stream64_kernel's single read of the hi plane (queries
    resident in LDS; odd and even slab counts, one- and two-slab rows, a row length whose image does not fit and
    stays with the 256-column scan), exact re-scoring and proof behind it: D and I bit-exact against oracle mode 1."""
    xb, q = _mk(n, d, nq, seed=n + d + nq + k)
    D, I, fb = _search(xb, q, k)
    Dr, Ir = flat_ip.search_c(xb, q, k, mode=flat_ip.MODE_GPU_ORDER)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D.view(np.uint32), Dr.view(np.uint32))
    assert fb <= 3


def test_bf16_index_gives_one_result_from_16_to_64_queries():
    """OTTER_BF16 index: the scores ARE the bf16 products, so the three kernels that serve 16, 32 and 64 queries
    (stream16_kernel, stream64_kernel) and the 256-column scan (65) must agree bit for bit, ties and all."""
    from otter_amd.index import IndexFlatIP
    xb, q = _mk(150000, 768, 65, seed=9)
    ix = IndexFlatIP(768, dtype="bf16")
    ix.add(xb)
    D65, I65 = ix.search(q, 100)
    for nq in (16, 32, 33, 48, 64):
        D, I = ix.search(q[:nq], 100)
        np.testing.assert_array_equal(I, I65[:nq])
        np.testing.assert_array_equal(D.view(np.uint32), D65[:nq].view(np.uint32))
    ix.close()


@pytest.mark.parametrize("n,d,k", [(270000, 768, 100), (270000, 832, 20), (100000, 2304, 100), (270000, 768, 1)])
def test_small_batch_finish_kernel_equals_the_four_launches(n, d, k, monkeypatch):
    """Small batches finish in one launch (finish_small_kernel: re-scoring by a plain fmaf chain in the exact kernels'
    k order, rank-count selection, proof, the count of unproven queries polled from pinned memory).  It must give what
    the four launches it replaces give (OTTER_NO_FINISH=1: MFMA re-scoring, merge, margin check, copy), bit for bit,
    for row lengths of one, two and three rounds of K-blocks, and what the exact fp32 kernels give."""
    from otter_amd.index import IndexFlatIP
    xb, q = _mk(n, d, 64, seed=n + d + k)
    ix = IndexFlatIP(d)
    ix.add(xb)
    ref = IndexFlatIP(d, planes="none")
    ref.add(xb)
    for nq in (1, 5, 32, 40, 64):
        D, I = ix.search(q[:nq], k)
        monkeypatch.setenv("OTTER_NO_FINISH", "1")
        D4, I4 = ix.search(q[:nq], k)
        monkeypatch.delenv("OTTER_NO_FINISH")
        np.testing.assert_array_equal(I, I4)
        np.testing.assert_array_equal(D.view(np.uint32), D4.view(np.uint32))
        De, Ie = [], []
        for lo in range(0, nq, 16):
            a, b = ref.search(q[lo:min(nq, lo + 16)], k)
            De.append(a)
            Ie.append(b)
        np.testing.assert_array_equal(I, np.concatenate(Ie))
        np.testing.assert_array_equal(D.view(np.uint32), np.concatenate(De).view(np.uint32))
    assert ix.last_fallback == 0
    ix.close()
    ref.close()


def test_optimistic_sample_with_more_tiles_than_scan_strips():
    """4,500 queries = 18 query tiles leave the scan 14 strips per tile, while the optimistic threshold samples 29 tiles:
    the block-maxima launch must not touch the per-strip candidate counts (it once wrote 29 rows of a 14-row array).
    Result = the exact kernels'."""
    from otter_amd.index import IndexFlatIP
    xb, q = _mk(300000, 128, 4500, seed=5)
    ix = IndexFlatIP(128)
    ix.add(xb)
    D, I = ix.search(q, 10)
    assert ix.last_fallback == 0
    ref = IndexFlatIP(128, planes="none")
    ref.add(xb)
    Dr, Ir = ref.search(q, 10)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D.view(np.uint32), Dr.view(np.uint32))
    ix.close()
    ref.close()
