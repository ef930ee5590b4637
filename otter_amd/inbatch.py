"""In-batch score matrix and the dev-set evaluation built on it.

The reference's fine-tuning script scores every question of a batch against every block of the batch plus
its own hard negative, and reports accuracy and MRR of the diagonal (scripts/train_1hop_tb_retrieval.py:290-327;
the same matrix is the input of loss_single, retrieval/criterions.py:10-27).  Here the matrix and the rank of
each row's target come from one HIP kernel (otter_inbatch_scores, include/otter_hip.h); the three encoder
passes that produce q / c / neg_c are the HIP encoder's (RobertaSingleRetrieverThreeCatPool.forward,
retrieval/models/tb_retriever.py:235-239, restated in `retriever_outputs`).

Ties: torch's argsort leaves the order of equal scores unspecified; here an equal score in a lower column
sorts first (argmax's first-index rule), so `rank == 0` is exactly `argmax == target`.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def inbatch_scores(q, c, neg_c=None):
    """q, c, neg_c: float32 [B, d] tensors on one GPU.  Returns (product float32 [B, B + (neg_c is not None)],
    rank int32 [B]) on that GPU — train_1hop_tb_retrieval.py:304-315 without the host round trip."""
    lib = _lib.load()
    if not (torch.is_tensor(q) and q.is_cuda):
        raise ValueError("inbatch_scores: q must be a tensor on a GPU (the HIP path has no CPU fallback)")
    if q.dim() != 2 or c.shape != q.shape or (neg_c is not None and neg_c.shape != q.shape):
        raise ValueError("inbatch_scores: q, c and neg_c must all be [B, d]; got %s %s %s" % (
            tuple(q.shape), tuple(c.shape), None if neg_c is None else tuple(neg_c.shape)))
    dev = q.device
    q = q.to(torch.float32).contiguous()
    c = c.to(device=dev, dtype=torch.float32).contiguous()
    if neg_c is not None:
        neg_c = neg_c.to(device=dev, dtype=torch.float32).contiguous()
    B, d = q.shape
    product = torch.empty((B, B + (neg_c is not None)), dtype=torch.float32, device=dev)
    rank = torch.empty((B,), dtype=torch.int32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.otter_inbatch_scores(
        dev.index or 0, ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(c.data_ptr()),
        ctypes.c_void_p(neg_c.data_ptr()) if neg_c is not None else None, B, d,
        ctypes.c_void_p(product.data_ptr()), ctypes.c_void_p(rank.data_ptr()), stream), "otter_inbatch_scores")
    return product, rank


def mrr_accuracy(ranks):
    """(MRR, accuracy) of 0-based target ranks — train_1hop_tb_retrieval.py:312-322."""
    r = np.asarray(ranks.cpu() if torch.is_tensor(ranks) else ranks, dtype=np.int64)
    if r.size == 0:
        return float("nan"), float("nan")
    return float(np.mean(1.0 / (r + 1))), float(np.mean(r == 0))


def retriever_outputs(model, batch):
    """{'q', 'c', 'neg_c'} of the retriever's forward (tb_retriever.py:74-78 BERT family, :235-239 RoBERTa;
    retriever.py has the same shape without part masks) from a HIP encoder of otter_amd.encoder."""
    def enc(fn, p):
        types = None if model.roberta else batch.get(p + "_type_ids")
        if model.three_cat:
            return fn(batch[p + "_input_ids"], batch[p + "_mask"], types, part2_mask=batch[p + "_part2_mask"],
                      part3_mask=batch[p + "_part3_mask"])
        return fn(batch[p + "_input_ids"], batch[p + "_mask"], types)
    return {"q": enc(model.encode_q, "q"), "c": enc(model.encode_seq, "c"), "neg_c": enc(model.encode_seq, "neg")}


def predict(model, eval_dataloader, logger=None):
    """train_1hop_tb_retrieval.py:290-327: MRR (returned) and accuracy (logged) over a dev loader.  `model` is
    either a callable returning {'q', 'c', 'neg_c'} or a HIP encoder (then retriever_outputs runs it)."""
    ranks = []
    for batch in eval_dataloader:
        out = retriever_outputs(model, batch) if hasattr(model, "encode_seq") else model(batch)
        _, rank = inbatch_scores(out["q"], out["c"], out["neg_c"])
        ranks.append(rank.cpu().numpy())
    ranks = np.concatenate(ranks) if ranks else np.zeros((0,), np.int32)
    mrr, acc = mrr_accuracy(ranks)
    if logger is not None:
        logger.info("evaluated %s examples...", float(len(ranks)))
        logger.info("avg. Acc: %s", acc)
        logger.info("MRR: %s", mrr)
    return mrr
