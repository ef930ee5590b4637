#!/usr/bin/env python3
"""tools/bench_gemm.py — TFLOP/s of the encoder's bf16 GEMM (otter_gemm_bf16) on the four projection
shapes of a roberta-base layer at B*L tokens, random operands, HIP-event timed."""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--zeros", action="store_true",
                    help="all-zero operands: the same instruction stream at the clock the chip sustains without data toggling")
    a = ap.parse_args()
    lib = _lib.load()
    M = a.tokens
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    for name, N, K, epi in (("qkv", 2304, 768, 0), ("attn_out", 768, 768, 2), ("ffn_up", 3072, 768, 1),
                            ("ffn_down", 768, 3072, 2), ("plain_768", 768, 768, 0), ("plain_3072x768", 3072, 768, 0)):
        A = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
        W = (torch.randn((N, K), generator=g, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn((N,), generator=g, device="cuda")
        R = torch.randn((M, N), generator=g, device="cuda").to(torch.bfloat16)
        C = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        if a.zeros:
            A.zero_(); W.zero_(); bias.zero_(); R.zero_()
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

        def run():
            _lib.check(lib.otter_gemm_bf16(0, epi, ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(W.data_ptr()),
                                           ctypes.c_void_p(bias.data_ptr()), ctypes.c_void_p(R.data_ptr()),
                                           ctypes.c_void_p(C.data_ptr()), M, N, K, st), "gemm")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(json.dumps({"gemm": name, "M": M, "N": N, "K": K, "epi": epi, "operands": "zeros" if a.zeros else "random", "us": round(ms * 1e3, 1),
                          "TFLOPs": round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1)}), flush=True)


if __name__ == "__main__":
    main()
