"""In-batch score matrix (reference scripts/train_1hop_tb_retrieval.py:304-322,
retrieval/criterions.py:15-26).  CPU: the oracle restatement against the golden the reference produced
(oracle/gen_inbatch_golden.py).  GPU: the HIP kernel through the C-ABI against the oracle and the golden.

Tolerance of the float32 scores: 2e-6 * d^0.5 * |q| |c| — the products are float32 sums of d terms in an
order that differs from torch.mm's (which is itself unspecified); ranks must agree wherever the float64
scores of the target and every other column differ by more than that tolerance."""
import os

import numpy as np
import pytest
import torch

from oracle import inbatch_ref as REF

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inbatch.npz")


def _tol(q, c):
    d = q.shape[1]
    return 2e-6 * np.sqrt(d) * np.linalg.norm(q, axis=1).max() * np.linalg.norm(c, axis=1).max()


def _ranks_agree(product64, got, tol):
    """got[i] must lie between the ranks obtained by resolving every near-tie (within tol) against / for the target."""
    B = product64.shape[0]
    t = product64[np.arange(B), np.arange(B)][:, None]
    lo = (product64 > t + tol).sum(1)
    hi = (product64 >= t - tol).sum(1) - 1
    return np.all((got >= lo) & (got <= hi))


def test_oracle_matches_the_reference_golden():
    g = np.load(GOLD)
    for n in ("a", "b"):
        q, c, neg = g[n + "_q"], g[n + "_c"], g[n + "_neg"]
        p = REF.product_matrix(q, c, neg)
        assert p.shape == g[n + "_product"].shape and p.dtype == np.float32
        np.testing.assert_allclose(p, g[n + "_product"], rtol=0, atol=_tol(q, c))
        assert np.array_equal(REF.target_ranks(g[n + "_product"]), g[n + "_ranks"])
        assert np.array_equal(REF.target_ranks(p), g[n + "_ranks"])        # no near-ties in the golden
        mrr, acc = REF.mrr_accuracy(g[n + "_ranks"])
        assert abs(mrr - float(g[n + "_mrr"])) < 1e-12 and abs(acc - float(g[n + "_acc"])) < 1e-12
        # the number the reference's loss_single returned for these inputs
        assert abs(REF.cross_entropy(p) - float(g[n + "_loss"])) < 1e-4 * max(1.0, float(g[n + "_loss"]))
    # equal scores: the lower column sorts first
    p = np.array([[1.0, 1.0, 0.5], [2.0, 1.0, 2.0], [0.0, 0.0, 0.0]], np.float32)
    assert REF.target_ranks(p).tolist() == [0, 2, 2]


@pytest.mark.gpu
def test_hip_inbatch_scores_match_golden_and_oracle():
    from otter_amd import inbatch as IB
    g = np.load(GOLD)
    dev = torch.device("cuda:0")
    for n in ("a", "b"):
        q, c, neg = g[n + "_q"], g[n + "_c"], g[n + "_neg"]
        P, R = IB.inbatch_scores(torch.from_numpy(q).to(dev), torch.from_numpy(c).to(dev), torch.from_numpy(neg).to(dev))
        assert P.dtype == torch.float32 and tuple(P.shape) == g[n + "_product"].shape and R.dtype == torch.int32
        np.testing.assert_allclose(P.cpu().numpy(), g[n + "_product"], rtol=0, atol=_tol(q, c))
        assert np.array_equal(R.cpu().numpy(), g[n + "_ranks"])
        mrr, acc = IB.mrr_accuracy(R)
        assert abs(mrr - float(g[n + "_mrr"])) < 1e-12 and abs(acc - float(g[n + "_acc"])) < 1e-12
        assert abs(REF.cross_entropy(P.cpu().numpy()) - float(g[n + "_loss"])) < 1e-4 * float(g[n + "_loss"])


@pytest.mark.gpu
@pytest.mark.parametrize("B,d,with_neg", [(1, 768, True), (7, 768, False), (64, 2304, True), (300, 768, True),
                                          (513, 100, True)])
def test_hip_inbatch_scores_random(B, d, with_neg):
    from otter_amd import inbatch as IB
    rs = np.random.RandomState(B * 7 + d)
    c = rs.randn(B, d).astype(np.float32)
    q = (1.5 / np.sqrt(d) * c + rs.randn(B, d)).astype(np.float32)
    neg = (1.2 / np.sqrt(d) * c + rs.randn(B, d)).astype(np.float32) if with_neg else None
    dev = torch.device("cuda:0")
    P, R = IB.inbatch_scores(torch.from_numpy(q).to(dev), torch.from_numpy(c).to(dev),
                             torch.from_numpy(neg).to(dev) if with_neg else None)
    p64 = q.astype(np.float64) @ c.astype(np.float64).T
    if with_neg:
        p64 = np.concatenate([p64, (q.astype(np.float64) * neg).sum(-1)[:, None]], 1)
    tol = _tol(q, c)
    assert tuple(P.shape) == p64.shape
    np.testing.assert_allclose(P.cpu().numpy(), p64, rtol=0, atol=tol)
    np.testing.assert_allclose(P.cpu().numpy(), REF.product_matrix(q, c, neg), rtol=0, atol=tol)
    got = R.cpu().numpy()
    assert _ranks_agree(p64, got, tol)
    # and exactly the oracle's ranks of the kernel's own float32 matrix (the tie rule included)
    assert np.array_equal(got, REF.target_ranks(P.cpu().numpy()))


@pytest.mark.gpu
def test_hip_inbatch_ties_empty_and_errors():
    from otter_amd import inbatch as IB
    dev = torch.device("cuda:0")
    # duplicate blocks: rows 0 and 2 of c are equal, so product[i][0] == product[i][2] bit for bit
    rs = np.random.RandomState(3)
    c = rs.randn(4, 64).astype(np.float32)
    c[2] = c[0]
    q = c.copy()
    P, R = IB.inbatch_scores(torch.from_numpy(q).to(dev), torch.from_numpy(c).to(dev))
    p = P.cpu().numpy()
    assert p.shape == (4, 4) and np.array_equal(p[:, 0], p[:, 2])
    assert R.cpu().numpy().tolist() == [0, 0, 1, 0]          # row 2's target ties with column 0, which sorts first
    assert np.array_equal(R.cpu().numpy(), REF.target_ranks(p))
    P, R = IB.inbatch_scores(torch.zeros((0, 8), device=dev), torch.zeros((0, 8), device=dev))
    assert tuple(P.shape) == (0, 0) and R.numel() == 0
    assert all(np.isnan(v) for v in IB.mrr_accuracy(R))
    with pytest.raises(ValueError):
        IB.inbatch_scores(torch.zeros((3, 8), device=dev), torch.zeros((4, 8), device=dev))
    with pytest.raises(ValueError):
        IB.inbatch_scores(torch.zeros((3, 8)), torch.zeros((3, 8)))
    with pytest.raises(ValueError, match="exceeds 8192"):
        IB.inbatch_scores(torch.zeros((8193, 2), device=dev), torch.zeros((8193, 2), device=dev))


@pytest.mark.gpu
def test_predict_runs_a_hip_encoder_over_a_dev_loader():
    """predict() end to end: three encoder passes per batch (q / c / neg_c) + the score kernel; compared with
    the same three embeddings scored by the oracle."""
    from otter_amd import inbatch as IB
    from otter_amd.encoder import RobertaSingleEncoderThreeCatPool
    from oracle import encoder_ref as ER
    from tests import _enc_util as U
    cfg = ER.make_config("roberta", layers=2, vocab_size=1000)
    model = RobertaSingleEncoderThreeCatPool(U.hf_like_config(cfg), U.ref_args(cfg), dtype="fp32")
    model.load_state_dict(U.state_dict_for(cfg, 3), strict=False)
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)

    def part(B, L):
        ids = torch.from_numpy(rs.randint(3, cfg["vocab_size"], size=(B, L))).long()
        ids[:, 0] = 0
        m = torch.ones((B, L), dtype=torch.long)
        p2 = torch.zeros((B, L), dtype=torch.long)
        p2[:, 1:L // 2] = 1
        p3 = torch.zeros((B, L), dtype=torch.long)
        p3[:, L // 2 + 1:L - 1] = 1
        return ids.to(dev), m.to(dev), p2.to(dev), p3.to(dev)

    batches = []
    for B in (5, 3):
        b = {}
        for p, L in (("q", 16), ("c", 32), ("neg", 32)):
            b[p + "_input_ids"], b[p + "_mask"], b[p + "_part2_mask"], b[p + "_part3_mask"] = part(B, L)
        batches.append(b)
    mrr = IB.predict(model, batches)
    ranks = []
    for b in batches:
        o = IB.retriever_outputs(model, b)
        p = REF.product_matrix(o["q"].float().cpu().numpy(), o["c"].float().cpu().numpy(), o["neg_c"].float().cpu().numpy())
        ranks.append(REF.target_ranks(p))
    assert abs(mrr - REF.mrr_accuracy(np.concatenate(ranks))[0]) < 1e-12
