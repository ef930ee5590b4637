"""otter_index_search_noalloc (include/otter_hip.h): the entry point that allocates nothing.  Its output must equal
otter_index_search's bit for bit, on two calls in a row (the second smaller: the static scratch keeps the first
call's contents), through the C ABI."""
import ctypes

import numpy as np
import pytest
import torch

from otter_amd import _lib
from otter_amd.index import IndexFlatIP

pytestmark = pytest.mark.gpu


def _search_noalloc_raw(ix, q, k):
    """Straight through ctypes (no Python-side helper): device pointers in, device pointers out."""
    lib = _lib.load()
    nq = q.shape[0]
    D = torch.full((nq, k), 7.0, dtype=torch.float32, device=q.device)
    I = torch.full((nq, k), 7, dtype=torch.int64, device=q.device)
    st = ctypes.c_void_p(torch.cuda.current_stream(q.device.index).cuda_stream)
    rc = lib.otter_index_search_noalloc(ix._h, ctypes.c_void_p(q.data_ptr()), nq, k, ctypes.c_void_p(D.data_ptr()),
                                        ctypes.c_void_p(I.data_ptr()), st)
    _lib.check(rc, "otter_index_search_noalloc")
    return D, I


def test_noalloc_equals_search_twice_second_smaller():
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    n, d = 300000, 768                      # large enough for the sample pass + 256 strips + a two-level merge
    xb = torch.randn((n, d), generator=g, device=dev)
    xb[1234] = xb[77]                       # an exact tie (id order decides)
    ix = IndexFlatIP(d, device=0)
    ix.add(xb)
    q1 = torch.randn((32, d), generator=g, device=dev)
    q2 = torch.randn((3, d), generator=g, device=dev)
    q1[5] = xb[77]
    # expected results first (the allocating entry point grows its own buffers here)
    De1, Ie1 = ix.search(q1, 100)
    De2, Ie2 = ix.search(q2, 17)
    _search_noalloc_raw(ix, q2, 17)        # warm-up: torch loads its fill kernels and grows its cache here
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    D1, I1 = _search_noalloc_raw(ix, q1, 100)
    D2, I2 = _search_noalloc_raw(ix, q2, 17)          # right behind it, smaller nq and k
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]             # (before torch loads any further kernel of its own)
    assert torch.equal(I1, Ie1) and torch.equal(D1.view(torch.int32), De1.view(torch.int32))
    assert torch.equal(I2, Ie2) and torch.equal(D2.view(torch.int32), De2.view(torch.int32))
    assert int(I1[5, 0]) == 77 and int(I1[5, 1]) == 1234
    # no device allocation between the two probes: the library's scratch is the static array, and torch's
    # (D, I) tensors come out of its cache (at most one 2 MiB cache block otherwise)
    assert free0 - free1 <= (2 << 20), (free0, free1)
    # a small corpus (no sample pass, fewer strips than CUs) right after the large one
    small = IndexFlatIP(d, device=0)
    small.add(xb[:999])
    Ds, Is = _search_noalloc_raw(small, q2, 128)
    Dx, Ix = small.search(q2, 128)
    assert torch.equal(Is, Ix) and torch.equal(Ds.view(torch.int32), Dx.view(torch.int32))
    # and the first index again: its scratch contents were overwritten in between
    D3, I3 = _search_noalloc_raw(ix, q1[:1].contiguous(), 1)
    De3, Ie3 = ix.search(q1[:1].contiguous(), 1)
    assert torch.equal(I3, Ie3) and torch.equal(D3.view(torch.int32), De3.view(torch.int32))
    small.close()
    ix.close()


def test_noalloc_rejects_what_it_cannot_hold():
    dev = torch.device("cuda", 0)
    ix = IndexFlatIP(64, device=0)
    ix.add(torch.randn((500, 64), device=dev))
    q = torch.randn((33, 64), device=dev)
    with pytest.raises(ValueError, match="nq=33"):
        _search_noalloc_raw(ix, q, 10)
    with pytest.raises(ValueError, match="k=129"):
        _search_noalloc_raw(ix, q[:4].contiguous(), 129)
    b16 = IndexFlatIP(64, device=0, dtype="bf16")
    b16.add(torch.randn((500, 64), device=dev))
    with pytest.raises(NotImplementedError):
        _search_noalloc_raw(b16, q[:4].contiguous(), 10)
    # an empty index answers with padding
    e = IndexFlatIP(64, device=0)
    D, I = _search_noalloc_raw(e, q[:2].contiguous(), 5)
    assert (I == -1).all() and (D == -np.finfo(np.float32).max).all()
    for x in (ix, b16, e):
        x.close()
