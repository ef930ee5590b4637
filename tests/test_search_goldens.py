"""Committed search fixtures (oracle/gen_search_goldens.py): float64-ranked (D, I) plus the
float64 scores of the top k + 8, for seeded corpora N in {100, 4096, 65536}, d in {768, 2304}, Q in {1, 7, 64},
k in {1, 100, 1000}, a duplicate-row fixture and an all-equal fixture.  PARITY UNPINNED versus faiss (absent;
the reference ships no search vectors): these files pin the results across rounds, independently of any
kernel's summation order.  CPU: both restatements of the oracle against the files.  GPU: the HIP path,
through the C ABI, against the same files."""
import os

import numpy as np
import pytest

from oracle import flat_ip
from oracle.gen_search_goldens import CASES, GOLD, make_case


def _load(name):
    return np.load(os.path.join(GOLD, "search_%s.npz" % name))


@pytest.mark.parametrize("name", sorted(CASES))
def test_fixture_inputs_regenerate(name):
    if CASES[name][1] > 4096 and CASES[name][2] > 768:
        pytest.skip("large corpus: covered by the oracle test below")
    xb, q, k = make_case(name)
    g = _load(name)
    assert float(g["xb_checksum"]) == float(xb.astype(np.float64).sum())
    assert float(g["q_checksum"]) == float(q.astype(np.float64).sum())
    assert g["D"].shape == (q.shape[0], k)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_restatements_against_the_fixtures(name):
    xb, q, k = make_case(name)
    g = _load(name)
    for mode in (flat_ip.MODE_GPU_ORDER, flat_ip.MODE_F64, flat_ip.MODE_PLAIN):
        D, I = flat_ip.search_c(xb, q, k, mode=mode)
        nd = flat_ip.check_against_golden(g, D, I, k)
        if mode == flat_ip.MODE_F64:
            assert nd == 0, "float64 accumulation must reproduce the float64 ranking"
    if xb.shape[0] <= 4096:
        ix = flat_ip.IndexFlatIP(xb.shape[1])
        ix.add(xb)
        D, I = ix.search(q, k)
        flat_ip.check_against_golden(g, D, I, k)
    if name.startswith("equal"):
        D, I = flat_ip.search_c(xb, q, k, mode=flat_ip.MODE_GPU_ORDER)
        assert (I == np.arange(k)[None, :]).all()          # all scores equal: id order decides
    if name.startswith("dup"):
        D, I = flat_ip.search_c(xb, q, k, mode=flat_ip.MODE_GPU_ORDER)
        half = xb.shape[0] // 2
        lo = I[I < half]
        assert np.isin(lo + half, I).mean() > 0.9           # a row and its copy both rank (ties by id)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_search_against_the_fixtures(name):
    from otter_amd.index import IndexFlatIP
    xb, q, k = make_case(name)
    g = _load(name)
    ix = IndexFlatIP(xb.shape[1], device=0)
    ix.add(xb)
    D, I = ix.search(q, k)                                   # numpy in / out: host path of the C ABI
    flat_ip.check_against_golden(g, D, I, k)
    if name.startswith("equal"):
        assert (I == np.arange(k)[None, :]).all()
    # the same queries one by one (the reference's call pattern) give the same rows
    for i in range(min(3, q.shape[0])):
        D1, I1 = ix.search(q[i:i + 1], k)
        assert (I1[0] == I[i]).all() and (D1[0] == D[i]).all()
    ix.close()


@pytest.mark.gpu
def test_hip_bf16_coarse_path_against_a_fixture():
    """64 queries over 65,536 rows take the coarse bf16 pass + exact re-scoring: same files, same rule."""
    from otter_amd.index import IndexFlatIP
    name = "n65536_d768_q64_k100"
    xb, q, k = make_case(name)
    ix = IndexFlatIP(xb.shape[1], device=0)
    ix.add(xb)
    D, I = ix.search(q, k)
    assert ix.coarse_state == 1
    flat_ip.check_against_golden(_load(name), D, I, k)
    ix.close()
