"""otter_sharded_search / MultiGpuIndexFlatIP: the one-process, every-GPU form of the index
(faiss.index_cpu_to_all_gpus, scripts/eval_ottqa_retrieval.py:110-112).  The box has one GPU, so the shards here
share device 0 (each still has its own host thread, stream and staging); the result must equal the one-index
search bit for bit -- scores are per-row chains that do not depend on which shard holds the row, and ties are
broken by the global id in both."""
import numpy as np
import pytest
import torch

from oracle import flat_ip
from otter_amd.index import IndexFlatIP, MultiGpuIndexFlatIP, index_cpu_to_all_gpus

pytestmark = pytest.mark.gpu


def _data(n, d, nq, seed):
    rng = np.random.RandomState(seed)
    return rng.standard_normal((n, d)).astype(np.float32), rng.standard_normal((nq, d)).astype(np.float32)


@pytest.mark.parametrize("n,d,G,nqs,k", [
    (210_000, 128, 3, (1, 40, 300), 100),      # every shard large enough for the coarse pass (>= 65,536 rows)
    (5_000, 768, 8, (1, 7, 64), 100),          # 8 small shards: the exact kernels
    (30_000, 64, 5, (3, 33), 2048),            # G * k > 8192: hierarchical merge
    (90, 32, 4, (2,), 100),                    # fewer rows than k: -1 padding survives the merge
])
def test_sharded_equals_single_index(n, d, G, nqs, k):
    x, _ = _data(n, d, 1, 5)
    one = IndexFlatIP(d)
    one.add(x)
    multi = MultiGpuIndexFlatIP(d, devices=[0] * G)
    multi.add(x)
    assert multi.ntotal == n and [s.id_base for s in multi.shards] == [min(n, g * -(-n // G)) for g in range(G)]
    for nq in nqs:
        _, q = _data(1, d, nq, 100 + nq)
        D1, I1 = one.search(q, k)
        Dm, Im = multi.search(q, k)                                   # host in, host out
        assert np.array_equal(I1, Im) and np.array_equal(D1, Dm)
        qd = torch.from_numpy(q).cuda()
        Dd, Id = multi.search(qd, k)                                  # device in, device out
        assert Dd.is_cuda and np.array_equal(Id.cpu().numpy(), I1) and np.array_equal(Dd.cpu().numpy(), D1)
    # and against the CPU oracle on a few queries (order-free fp64 check)
    _, q = _data(1, d, 4, 9)
    Dm, Im = multi.search(q, min(k, 100))
    flat_ip.check_topk(x, q, Dm, Im, min(k, 100))
    multi.close()
    one.close()


def test_duplicate_rows_across_shards_keep_id_order():
    """The same row in every shard: equal scores, so the merged order is by global id."""
    d, G = 64, 4
    rng = np.random.RandomState(3)
    base = rng.standard_normal((50, d)).astype(np.float32)
    x = np.concatenate([base] * G)                                    # shard g holds a copy of `base`
    multi = MultiGpuIndexFlatIP(d, devices=[0] * G)
    multi.add(x)
    q = rng.standard_normal((5, d)).astype(np.float32)
    D, I = multi.search(q, 40)
    one = IndexFlatIP(d)
    one.add(x)
    D1, I1 = one.search(q, 40)
    assert np.array_equal(I, I1) and np.array_equal(D, D1)
    for r in range(5):
        for j in range(0, 40, G):                                     # runs of G equal scores, ids ascending
            assert len(set(D[r, j:j + G])) == 1 and list(I[r, j:j + G]) == sorted(I[r, j:j + G])


def test_index_cpu_to_all_gpus_moves_the_rows():
    d, n = 96, 20_000
    x, q = _data(n, d, 9, 11)
    one = IndexFlatIP(d)
    one.add(x)
    D1, I1 = one.search(q, 10)
    multi = index_cpu_to_all_gpus(one, devices=[0, 0, 0])
    assert isinstance(multi, MultiGpuIndexFlatIP) and multi.ntotal == n
    Dm, Im = multi.search(q, 10)
    assert np.array_equal(I1, Im) and np.array_equal(D1, Dm)
    with pytest.raises(RuntimeError):
        multi.add(x)
    multi.close()
    with pytest.raises(RuntimeError):
        multi.search(q, 10)
    same = IndexFlatIP(d)
    assert index_cpu_to_all_gpus(same, devices=[0]) is same


def test_sharded_search_rejects_bad_arguments():
    a, b = IndexFlatIP(32), IndexFlatIP(48)
    multi = MultiGpuIndexFlatIP(32, devices=[0, 0])
    with pytest.raises(ValueError):
        multi.search(np.zeros((2, 31), np.float32), 5)
    import ctypes
    from otter_amd import _lib
    lib = _lib.load()
    hs = (ctypes.c_void_p * 2)(a._h, b._h)                            # different dimensions
    q = np.zeros((1, 32), np.float32)
    D, I = np.zeros((1, 5), np.float32), np.zeros((1, 5), np.int64)
    rc = lib.otter_sharded_search(hs, 2, ctypes.c_void_p(q.ctypes.data), 1, 5, ctypes.c_void_p(D.ctypes.data),
                                  ctypes.c_void_p(I.ctypes.data), 0, None)
    assert rc == -1 and b"d=48" in lib.otter_last_error()
    hs = (ctypes.c_void_p * 2)(a._h, a._h)                            # the same handle twice
    rc = lib.otter_sharded_search(hs, 2, ctypes.c_void_p(q.ctypes.data), 1, 5, ctypes.c_void_p(D.ctypes.data),
                                  ctypes.c_void_p(I.ctypes.data), 0, None)
    assert rc == -1


def test_eval_driver_builds_the_multi_gpu_index(tmp_path):
    """eval_driver.build_index(devices=[...]) = the reference's --gpu branch (index_cpu_to_all_gpus) on a memmapped .npy."""
    from otter_amd import eval_driver as EV
    d, n = 2304, 3_000
    x, q = _data(n, d, 6, 21)
    np.save(tmp_path / "xb.npy", x)
    xb = np.load(tmp_path / "xb.npy", mmap_mode="r")
    one = EV.build_index(xb)
    multi = EV.build_index(xb, devices=[0, 0])
    assert isinstance(multi, MultiGpuIndexFlatIP) and multi.ntotal == n
    D1, I1 = one.search(q, 100)
    Dm, Im = multi.search(q, 100)
    assert np.array_equal(I1, Im) and np.array_equal(D1, Dm)
    one.close()
    multi.close()


def test_one_query_calls_on_four_shards_have_small_host_overhead():
    """The reference's call pattern -- one query per search() call -- through otter_sharded_search: the per-call host
    work (persistent shard threads, peer access asked for once, staging reused, the merge) must stay small beside the
    searches.  1,000 consecutive calls on 4 shards.  This box has ONE device, on which the four shard searches take
    turns (each launch fills the chip), so the yardstick is four searches of a quarter-size index issued one after
    the other from one thread, not one search of the whole index (printed beside it: with one GPU per shard that is
    the comparison that counts)."""
    import time
    n, d, k, G = 1_000_000, 768, 100, 4
    x = torch.randn(n, d, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
    one = IndexFlatIP(d)
    one.add(x)
    quarter = IndexFlatIP(d)
    quarter.add(x[: n // G])
    multi = MultiGpuIndexFlatIP(d, devices=[0] * G)
    multi.add(x)
    q = torch.randn(1, d, device="cuda", generator=torch.Generator(device="cuda").manual_seed(8))
    D1, I1 = one.search(q, k)
    Dm, Im = multi.search(q, k)
    assert torch.equal(I1, Im) and torch.equal(D1, Dm)

    def timed(ix, calls):
        for _ in range(20):
            ix.search(q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            ix.search(q, k)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / calls

    t_one, t_quarter, t_multi = timed(one, 1000), timed(quarter, 1000), timed(multi, 1000)
    print("one index %.3f ms, quarter index %.3f ms, 4 shards on one device %.3f ms per call (%.2f x four quarter "
          "searches in a row, %.2f x the one index)" % (t_one * 1e3, t_quarter * 1e3, t_multi * 1e3,
                                                        t_multi / (G * t_quarter), t_multi / t_one))
    assert t_multi <= 1.3 * G * t_quarter, (t_one, t_quarter, t_multi)
    multi.close()
    quarter.close()
    one.close()
