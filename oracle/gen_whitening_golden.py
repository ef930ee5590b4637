#!/usr/bin/env python3
"""oracle/gen_whitening_golden.py — run the REFERENCE's whitening (retrieval/utils/utils.py:301-324) on a small
seeded input and commit input + output as tests/golden/whitening.npz.  Build container only (needs
/root/reference).  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_whitening_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
from retrieval.utils import utils as ref_utils  # noqa: E402  (the reference)


def main():
    rs = np.random.RandomState(12)
    mix = rs.randn(24, 24) * 0.5 + np.eye(24)
    e1 = (rs.randn(300, 24) @ mix + rs.randn(24)).astype(np.float32)        # "corpus"
    e2 = (rs.randn(40, 24) @ mix + rs.randn(24)).astype(np.float32)         # "queries"
    w1, w2 = ref_utils.whitening(e1, e2)
    kernel, bias = ref_utils.compute_kernel_bias(np.concatenate([e1, e2], axis=0))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "whitening.npz"), e1=e1, e2=e2, w1=w1, w2=w2,
                        kernel=kernel, bias=bias)
    print("wrote tests/golden/whitening.npz", w1.shape, w2.shape, w1.dtype)


if __name__ == "__main__":
    main()
