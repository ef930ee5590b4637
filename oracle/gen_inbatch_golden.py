#!/usr/bin/env python3
"""oracle/gen_inbatch_golden.py — golden vectors for the in-batch score matrix.  Build
container only (needs /root/reference).  TEST INFRASTRUCTURE ONLY.

What runs from the reference: retrieval/criterions.py:loss_single, imported as is and fed a stand-in model
that returns fixed q / c / neg_c (its result, the cross-entropy of the product matrix, pins the matrix).
scripts/train_1hop_tb_retrieval.py itself cannot be imported here (apex is absent), so the statements of its
predict() that turn the matrix into ranks (:304-315) are executed below as the same torch calls on the same
inputs; their outputs are the expected product / ranks.

    python oracle/gen_inbatch_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
from retrieval.criterions import loss_single  # noqa: E402  (the reference)


def main():
    out = {}
    for name, B, d, seed in (("a", 12, 96, 5), ("b", 33, 384, 6)):
        g = torch.Generator().manual_seed(seed)
        c = torch.randn(B, d, generator=g)
        a = 1.5 / d ** 0.5                                    # q_i . c_i leads by about 1.5 sigma: likely winner, not certain
        q = a * c + torch.randn(B, d, generator=g)
        neg = 0.8 * a * c + torch.randn(B, d, generator=g)
        loss = float(loss_single(lambda batch: {"q": q, "c": c, "neg_c": neg}, None))
        product = torch.cat([torch.mm(q, c.t()), (q * neg).sum(-1).unsqueeze(1)], dim=-1)
        ranked = product.argsort(dim=1, descending=True)
        idx2rank = ranked.argsort(dim=1)
        ranks = idx2rank[torch.arange(B), torch.arange(B)]
        pred = product.argmax(-1)
        out.update({name + "_q": q.numpy(), name + "_c": c.numpy(), name + "_neg": neg.numpy(),
                    name + "_loss": np.float64(loss), name + "_product": product.numpy(),
                    name + "_ranks": ranks.numpy().astype(np.int32),
                    name + "_acc": np.float64((pred == torch.arange(B)).double().mean()),
                    name + "_mrr": np.float64(np.mean([1 / (int(r) + 1) for r in ranks]))})
        print(name, "loss", loss, "acc", out[name + "_acc"], "mrr", out[name + "_mrr"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "inbatch.npz"), **out)
    print("wrote tests/golden/inbatch.npz")


if __name__ == "__main__":
    main()
