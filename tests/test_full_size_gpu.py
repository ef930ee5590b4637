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
