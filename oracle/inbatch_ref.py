"""oracle/inbatch_ref.py — CPU restatement of the in-batch score matrix and the metrics the reference derives
from it (scripts/train_1hop_tb_retrieval.py:304-322; retrieval/criterions.py:15-26).  TEST INFRASTRUCTURE ONLY:
imported by tests/ (and nothing else); the product path is otter_amd/inbatch.py -> libotter_hip.so.

Pinned by tests/golden/inbatch.npz: the cross-entropy the REFERENCE's loss_single returns for the golden
inputs must equal the cross-entropy of this restatement's product, and product / ranks equal the values the
reference's own torch statements produced (oracle/gen_inbatch_golden.py)."""
import numpy as np


def product_matrix(q, c, neg_c=None):
    """:304-306 — float32 torch.mm(q, c.t()) and (q * neg_c).sum(-1), concatenated."""
    q = np.asarray(q, np.float32)
    p = q @ np.asarray(c, np.float32).T
    if neg_c is not None:
        p = np.concatenate([p, (q * np.asarray(neg_c, np.float32)).sum(-1, dtype=np.float32)[:, None]], axis=1)
    return p.astype(np.float32)


def target_ranks(product):
    """:308-315 — rank of column i in row i under a descending sort (stable: equal scores keep column order)."""
    product = np.asarray(product)
    ranked = np.argsort(-product.astype(np.float64), axis=1, kind="stable")
    idx2rank = np.argsort(ranked, axis=1, kind="stable")
    return idx2rank[np.arange(product.shape[0]), np.arange(product.shape[0])].astype(np.int32)


def mrr_accuracy(ranks):
    """:312-322."""
    r = np.asarray(ranks, np.int64)
    return float(np.mean(1.0 / (r + 1))), float(np.mean(r == 0))


def cross_entropy(product):
    """criterions.py:25-26 — F.cross_entropy(product, arange(B)), in float64."""
    p = np.asarray(product, np.float64)
    m = p.max(axis=1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(p - m).sum(axis=1))
    return float(np.mean(lse - p[np.arange(p.shape[0]), np.arange(p.shape[0])]))
