"""PCA whitening of the embeddings before indexing: host-side numpy, the transform the
reference applies under --whitening (retrieval/utils/utils.py:301-324, used from
retrieval/eval_ottqa_retrieval_tfidffilter.py).  No kernels: one d x d SVD and two [N, d] x [d, d] products;
the whitened matrices then go through IndexFlatIP.add / search like any other embeddings.

    kernel = U diag(1 / sqrt(S)),  cov(vecs) = U S V^T        utils.py:301-306
    vecs'  = (vecs - mean) kernel                             utils.py:309-313 (no re-normalisation: commented out)
    whitening(e1, e2): statistics over concat([e1, e2]), results cast back to the input dtypes   utils.py:316-324
"""
import numpy as np


def compute_kernel_bias(vecs):
    """(W, -mu) of retrieval/utils/utils.py:301-306."""
    vecs = np.asarray(vecs)
    mu = vecs.mean(axis=0, keepdims=True)
    cov = np.cov(vecs.T)
    u, s, _ = np.linalg.svd(cov)
    return np.dot(u, np.diag(1 / np.sqrt(s))), -mu


def transform_and_normalize(vecs, kernel=None, bias=None):
    """retrieval/utils/utils.py:309-313 (the normalisation step is disabled in the reference)."""
    if not (kernel is None or bias is None):
        vecs = (vecs + bias).dot(kernel)
    return vecs


def whitening(embedding1, embedding2):
    """retrieval/utils/utils.py:316-324."""
    t1, t2 = embedding1.dtype, embedding2.dtype
    kernel, bias = compute_kernel_bias(np.concatenate([embedding1, embedding2], axis=0))
    e1 = transform_and_normalize(embedding1, kernel, bias).astype(t1)
    e2 = transform_and_normalize(embedding2, kernel, bias).astype(t2)
    return e1, e2
