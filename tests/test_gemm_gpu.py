"""GPU numerics of the encoder's bf16 GEMM (otter_gemm_bf16, otter_amd/csrc/gemm_bf16.hip) against a
plain PyTorch fp32 matmul of the same bf16 operands.

Tolerance: the kernel accumulates in fp32 and rounds once to bf16, so |C - ref| <= 2^-8 |ref| + 1e-3
(one bf16 ulp of the result plus accumulation-order noise on sums of K <= 3072 products).
Shapes are the encoder's four projections (K, N in {768, 2304, 3072}) at several token counts,
including more tiles than CUs (persistent loop over several tiles per workgroup) and fewer.
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _gemm(epi, A, W, bias, R):
    from otter_amd import _lib
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    C = torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.otter_gemm_bf16(0, epi, ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(W.data_ptr()),
                                   ctypes.c_void_p(bias.data_ptr()) if bias is not None else None,
                                   ctypes.c_void_p(R.data_ptr()) if R is not None else None,
                                   ctypes.c_void_p(C.data_ptr()), M, N, K, st), "otter_gemm_bf16")
    torch.cuda.synchronize()
    return C


def _ref(epi, A, W, bias, R):
    out = A.float() @ W.float().t()
    if bias is not None:
        out = out + bias
    if epi == 1:
        out = torch.nn.functional.gelu(out)     # exact erf form
    if epi == 2:
        out = out + R.float()
    return out


SHAPES = [
    # M, N, K
    (256, 256, 64),          # one tile, one K-step
    (256, 256, 128),
    (512, 768, 768),         # attention output projection
    (1024, 2304, 768),       # QKV
    (768, 3072, 768),        # FFN up
    (768, 768, 3072),        # FFN down
    (65536, 768, 768),       # 768 tiles on 256 CUs: 3 tiles per workgroup
    (16384, 2304, 768),      # 576 tiles: ragged tile count per workgroup
]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_matches_fp32_matmul(M, N, K, epi):
    if M >= 16384 and epi == 1:
        pytest.skip("large shapes cover the loop structure; GELU is covered at small M")
    g = torch.Generator(device="cuda")
    g.manual_seed(M + N + K + epi)
    A = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    W = (torch.randn((N, K), generator=g, device="cuda") * (K ** -0.5)).to(torch.bfloat16)
    bias = torch.randn((N,), generator=g, device="cuda")
    R = torch.randn((M, N), generator=g, device="cuda").to(torch.bfloat16) if epi == 2 else None
    C = _gemm(epi, A, W, bias, R).float()
    ref = _ref(epi, A, W, bias, R)
    err = (C - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 1e-3
    bad = err > tol
    assert not bool(bad.any()), "max err %g (ref %g) at %r, %d bad" % (
        float(err.max()), float(ref.flatten()[err.argmax()]), divmod(int(err.argmax()), N), int(bad.sum()))


def test_gemm_repeatable_and_no_bias():
    # a race in the LDS ring shows as run-to-run differences: 20 launches must agree bit for bit
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    A = torch.randn((8192, 768), generator=g, device="cuda").to(torch.bfloat16)
    W = (torch.randn((3072, 768), generator=g, device="cuda") * 0.04).to(torch.bfloat16)
    first = _gemm(0, A, W, None, None)
    ref = A.float() @ W.float().t()
    assert float((first.float() - ref).abs().max()) < 2e-2
    for _ in range(20):
        again = _gemm(0, A, W, None, None)
        assert torch.equal(first.view(torch.int16), again.view(torch.int16))


def test_gemm_rejects_bad_shapes():
    A = torch.zeros((256, 64), dtype=torch.bfloat16, device="cuda")
    W = torch.zeros((256, 64), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError):
        _gemm(0, A[:100], W, None, None)
    with pytest.raises(ValueError):
        _gemm(2, A, W, None, None)          # residual epilogue without R
