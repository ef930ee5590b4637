"""BASELINE.json configs[1] at FULL size on the GPU: exact top-100 over 1M x 768 fp32 rows, 2,214 queries.
The CPU oracle cannot finish this in seconds, so parity is checked through size-independent properties, with
torch's own fp32/fp64 arithmetic (rocBLAS) as the independent arbiter for every query, plus the exact oracle
rule (oracle/flat_ip.check_topk) on a sample of queries:

  * rows sorted by score descending, ids unique and in range
  * every returned score equals the float64 dot product of its row within tol = 1e-5 |q| max|x|
  * completeness: no row outside the result scores above the k-th returned score + 2 tol
  * sharding invariance: two 500k-row shards with id_base, merged on the GPU, give the identical (D, I)
    bit for bit (the multi-GPU path's merge, on one device)
"""
import numpy as np
import pytest
import torch

from oracle import flat_ip

pytestmark = pytest.mark.gpu

N, DIM, NQ, K = 1000000, 768, 2214, 100


def _rows(n, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    out = torch.empty((n, DIM), dtype=torch.float32, device="cuda")
    for lo in range(0, n, 131072):
        hi = min(n, lo + 131072)
        out[lo:hi] = torch.randn((hi - lo, DIM), generator=g, device="cuda")
    return out


def test_config1_full_size_properties():
    from otter_amd.index import IndexFlatIP, merge_topk
    xb = _rows(N, 1234)
    q = _rows(NQ, 4321)
    ix = IndexFlatIP(DIM)
    ix.add(xb)
    D, I = ix.search(q, K)
    fallback = ix.last_fallback
    ix.close()
    assert D.shape == (NQ, K) and I.dtype == torch.int64 and D.is_cuda
    assert fallback == 0                                   # gaussian scores: every query proven by the margin test
    # order, range, uniqueness
    assert bool((D[:, 1:] <= D[:, :-1]).all())
    assert bool(((I >= 0) & (I < N)).all())
    srt = torch.sort(I, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())
    # scores against float64, completeness against torch's fp32 matmul, 246 queries at a time
    xmax = float(xb.double().pow(2).sum(1).sqrt().max())
    tol = 1e-5 * q.double().pow(2).sum(1).sqrt() * xmax
    for lo in range(0, NQ, 246):
        hi = min(NQ, lo + 246)
        rows = xb[I[lo:hi].reshape(-1)].view(hi - lo, K, DIM)
        s64 = (rows.double() * q[lo:hi].double()[:, None, :]).sum(-1)
        assert bool(((D[lo:hi].double() - s64).abs() <= tol[lo:hi, None]).all())
        del rows, s64
        S = q[lo:hi] @ xb.t()                              # [246, 1M] fp32, rocBLAS
        above = (S.double() > (D[lo:hi, K - 1].double() + 2 * tol[lo:hi])[:, None]).sum(1)
        assert bool((above < K).all()), "a row outside the result outranks the k-th returned score"
        kth = torch.topk(S, K, dim=1).values[:, K - 1]
        assert bool(((D[lo:hi, K - 1] - kth).abs().double() <= 2 * tol[lo:hi]).all())
        del S
    # the oracle's own rule on a sample (float64 ranking on the host)
    sample = [0, 1, 777, NQ - 1]
    xh = xb.cpu().numpy()
    nd = flat_ip.check_topk(xh, q[sample].cpu().numpy(), D[sample].cpu().numpy(), I[sample].cpu().numpy(), K, rtol=1e-5)
    assert nd <= 2                                         # ids differ from the fp64 ranking only inside near-ties
    del xh
    # sharding invariance
    half = N // 2
    parts = []
    for r in range(2):
        sh = IndexFlatIP(DIM, id_base=r * half)
        sh.add(xb[r * half:(r + 1) * half])
        parts.append(sh.search(q, K))
        sh.close()
    Dm, Im = merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), device=0)
    assert torch.equal(Im, I)
    assert torch.equal(Dm.view(torch.int32), D.view(torch.int32))


@pytest.fixture(scope="module")
def corpus_1m():
    xb = _rows(N, 1234)
    yield xb
    del xb
    torch.cuda.empty_cache()


def test_batch_size_ladder_is_one_result(corpus_1m):
    """configs[2]'s query-batch range on one 1M x 768 fp32 shard: whatever regime serves a batch (coarse pass by the
    bf16 stream kernels up to 64 queries, by the 256-query-tile scan from 65 on, exact re-scoring and proof behind both),
    the rows and scores of a query are the same bits.  Reference = the EXACT fp32 stream kernels (the allocation-free
    entry point, 16 queries per call); independent arbiter = the oracle's float64 rule on a sample."""
    from otter_amd.index import IndexFlatIP
    xb = corpus_1m
    q = _rows(4096, 97)
    ix = IndexFlatIP(DIM)
    ix.add(xb)
    Dr = torch.empty((4096, K), dtype=torch.float32, device="cuda")
    Ir = torch.empty((4096, K), dtype=torch.int64, device="cuda")
    for lo in range(0, 4096, 16):
        ix.search_noalloc(q[lo:lo + 16].contiguous(), K, Dr[lo:lo + 16], Ir[lo:lo + 16])
    for nq in (1, 8, 16, 17, 32, 33, 48, 64, 65, 96, 192, 256, 4096):
        D, I = ix.search(q[:nq].contiguous(), K)
        assert torch.equal(I, Ir[:nq]), "batch of %d: ids differ from the 16-query calls" % nq
        assert torch.equal(D.view(torch.int32), Dr[:nq].view(torch.int32)), "batch of %d: scores differ" % nq
    assert ix.last_fallback == 0
    ix.close()
    sample = [0, 5, 4095]
    nd = flat_ip.check_topk(xb.cpu().numpy(), q[sample].cpu().numpy(), Dr[sample].cpu().numpy(), Ir[sample].cpu().numpy(),
                            K, rtol=1e-5)
    assert nd <= 2


def test_eight_shards_on_one_device_merge_to_the_unsharded_result(corpus_1m):
    """configs[2]'s row sharding (8 shards with id_base) on one device: per-shard top-k + the GPU merge of the
    multi-GPU path = the unsharded index's (D, I), bit for bit, for a small and a large batch."""
    from otter_amd.index import IndexFlatIP, merge_topk
    xb = corpus_1m
    q = _rows(300, 98)
    whole = IndexFlatIP(DIM)
    whole.add(xb)
    per = N // 8
    shards = []
    for r in range(8):
        sh = IndexFlatIP(DIM, id_base=r * per)
        sh.add(xb[r * per:(r + 1) * per])
        shards.append(sh)
    for nq in (1, 300):
        qq = q[:nq].contiguous()
        D, I = whole.search(qq, K)
        parts = [sh.search(qq, K) for sh in shards]
        Dm, Im = merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), device=0)
        assert torch.equal(Im, I) and torch.equal(Dm.view(torch.int32), D.view(torch.int32))
    for sh in shards:
        sh.close()
    whole.close()


def test_config5_shard_bf16_top1000_properties():
    """BASELINE.json configs[4], one GPU's shard at full size: 12.5M x 768 rows stored as bf16 (19.2 GB), 16,384
    queries, k = 1000.  Properties that do not need an oracle run over 12.5M rows: order, uniqueness, range; scores
    = float64 dot products of the ROUNDED operands (the bf16 index's definition) within 1e-5 |q| max|x|;
    completeness on a query sample against torch's matmul over all rows; recall against an fp32 index on a
    1M-row slice (reported separately, as north_star asks for bf16 paths)."""
    from otter_amd.index import IndexFlatIP
    n5, nq5, k5 = 12500000, 16384, 1000
    ix = IndexFlatIP(DIM, dtype="bf16")
    g = torch.Generator(device="cuda")
    g.manual_seed(55)
    slab = 500000
    for lo in range(0, n5, slab):
        ix.add(torch.randn((slab, DIM), generator=g, device="cuda"))
    assert ix.ntotal == n5
    q = _rows(nq5, 56)
    D, I = ix.search(q, k5)
    assert D.shape == (nq5, k5)
    assert bool((D[:, 1:] <= D[:, :-1]).all())
    assert bool(((I >= 0) & (I < n5)).all())
    for lo in range(0, nq5, 2048):                                   # uniqueness, 2048 queries at a time
        srt = torch.sort(I[lo:lo + 2048], dim=1).values
        assert bool((srt[:, 1:] != srt[:, :-1]).all())
        del srt
    # scores of a query sample against float64 products of the rounded operands + completeness over ALL rows
    sample = torch.tensor([0, 1, 4097, 9000, nq5 - 1], device="cuda")
    qs = q[sample].to(torch.bfloat16).double()
    kth = D[sample, k5 - 1].double()
    tol = 1e-5 * qs.pow(2).sum(1).sqrt() * float((DIM ** 0.5) * 1.6)   # |x| <= ~1.6 sqrt(d) for gaussian rows
    above = torch.zeros(len(sample), dtype=torch.int64, device="cuda")
    g.manual_seed(55)
    found = torch.zeros((len(sample), k5), dtype=torch.bool, device="cuda")
    for lo in range(0, n5, slab):
        xs = torch.randn((slab, DIM), generator=g, device="cuda").to(torch.bfloat16).double()
        S = qs @ xs.t()                                                # [5, slab] float64
        above += (S > (kth + 2 * tol)[:, None]).sum(1)
        ids = I[sample] - lo
        inside = (ids >= 0) & (ids < slab)
        got = torch.gather(S, 1, ids.clamp(0, slab - 1))
        ok = (got - D[sample].double()).abs() <= tol[:, None]
        assert bool((ok | ~inside).all()), "a returned score is not the float64 product of the rounded operands"
        found |= inside
        del xs, S
    assert bool(found.all())
    assert bool((above < k5).all()), "a row outside the result outranks the k-th returned score"
    # recall@1000 against the fp32 ranking of the unrounded first 1M rows (both indexes over that slice)
    g.manual_seed(55)
    x1 = torch.cat([torch.randn((slab, DIM), generator=g, device="cuda") for _ in range(2)])
    ix.close()
    a16 = IndexFlatIP(DIM, dtype="bf16")
    a16.add(x1)
    a32 = IndexFlatIP(DIM)
    a32.add(x1)
    _, I16 = a16.search(q[:256].contiguous(), k5)
    _, I32 = a32.search(q[:256].contiguous(), k5)
    hit = 0
    for r in range(256):
        hit += len(set(I16[r].tolist()) & set(I32[r].tolist()))
    recall = hit / (256.0 * k5)
    assert recall >= 0.97, recall
    a16.close()
    a32.close()
