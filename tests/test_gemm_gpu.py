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


# ---------------------------------------------------------------- LayerNorm folded around the GEMM
def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _gemm_ln(epi, mode, A, W, bias, R, stat, v0, v1, want_part):
    from otter_amd import _lib
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    C = torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
    part = torch.full((4 * (N // 256), M, 2), float("nan"), device=A.device) if want_part else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.otter_gemm_bf16_ln(0, epi, mode, _p(A), _p(W), _p(bias), _p(R), _p(C), _p(stat), _p(v0), _p(v1),
                                      _p(part), M, N, K, st), "otter_gemm_bf16_ln")
    torch.cuda.synchronize()
    return C, part


def _stats(part, N, eps):
    from otter_amd import _lib
    lib = _lib.load()
    P, M, _ = part.shape
    stat = torch.empty((M, 2), device=part.device)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.otter_ln_stats(0, _p(part), P, M, N, ctypes.c_float(eps), _p(stat), st), "otter_ln_stats")
    torch.cuda.synchronize()
    return stat


def _fold(W32, gamma, beta, bias):
    from otter_amd import _lib
    lib = _lib.load()
    N, K = W32.shape
    Wout = torch.empty((N, K), dtype=torch.bfloat16, device=W32.device)
    s = torch.empty((N,), device=W32.device)
    c = torch.empty((N,), device=W32.device)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.otter_ln_fold(0, _p(W32), _p(gamma), _p(beta), _p(bias), _p(Wout), _p(s), _p(c), N, K, st), "otter_ln_fold")
    torch.cuda.synchronize()
    return Wout, s, c


@pytest.mark.parametrize("M,H,I", [(256, 256, 256), (768, 768, 3072), (33024, 768, 768)])
def test_layernorm_folded_chain_matches_torch(M, H, I):
    """One post-LN transformer half-layer the way the fused encoder path runs it:
         T  = x + ctx Wo^T + bo            (epi 2, mode 1: statistics out)
         U  = gelu(LN(T) W1^T + b1)        (epi 1, mode 1: LayerNorm folded into W1)
         T2 = LN(T) + U W2^T + b2          (epi 2, mode 2: residual normalised on the fly)
       against torch on the bf16-rounded T and the bf16-rounded gamma*W1 (the fold rounds gamma*W once,
       the unfused path rounds LN(T) and W separately; the encoder tests bound that difference end to end).
       Tolerances: statistics 1e-5 relative; outputs one bf16 ulp of the result + 2e-3 absolute."""
    eps = 1e-5
    g = torch.Generator(device="cuda")
    g.manual_seed(M + H + I)
    x = torch.randn((M, H), generator=g, device="cuda").to(torch.bfloat16)
    ctx = torch.randn((M, H), generator=g, device="cuda").to(torch.bfloat16)
    Wo = (torch.randn((H, H), generator=g, device="cuda") * H ** -0.5).to(torch.bfloat16)
    bo = torch.randn((H,), generator=g, device="cuda") * 0.1 + 0.3       # non-zero row mean
    W1 = torch.randn((I, H), generator=g, device="cuda") * H ** -0.5
    b1 = torch.randn((I,), generator=g, device="cuda") * 0.1
    W2 = (torch.randn((H, I), generator=g, device="cuda") * I ** -0.5).to(torch.bfloat16)
    b2 = torch.randn((H,), generator=g, device="cuda") * 0.1
    gamma = 1.0 + 0.2 * torch.randn((H,), generator=g, device="cuda")
    beta = 0.2 * torch.randn((H,), generator=g, device="cuda")

    T, part = _gemm_ln(2, 1, ctx, Wo, bo, x, None, None, None, True)
    T_plain = _gemm(2, ctx, Wo, bo, x)
    assert torch.equal(T.view(torch.int16), T_plain.view(torch.int16))       # statistics do not change the product
    assert bool(torch.isfinite(part).all())
    Tf = T.float()
    assert torch.allclose(part[..., 0].sum(0), Tf.sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(part[..., 1].sum(0), (Tf * Tf).sum(1), rtol=1e-5, atol=1e-3)
    stat = _stats(part, H, eps)
    mean = Tf.mean(1)
    rstd = torch.rsqrt(Tf.var(1, unbiased=False) + eps)
    assert torch.allclose(stat[:, 0], rstd, rtol=1e-4, atol=1e-6)
    assert torch.allclose(stat[:, 1], -mean * rstd, rtol=1e-4, atol=1e-5)

    W1f, s, c = _fold(W1, gamma, beta, b1)
    assert torch.equal(W1f.view(torch.int16), (W1 * gamma).to(torch.bfloat16).view(torch.int16))
    assert torch.allclose(s, W1f.float().sum(1), rtol=1e-5, atol=1e-4)
    assert torch.allclose(c, b1 + W1 @ beta, rtol=1e-5, atol=1e-4)

    ln = torch.nn.functional.layer_norm(Tf, (H,), gamma, beta, eps)
    U, _ = _gemm_ln(1, 1, T, W1f, None, None, stat, s, c, False)
    ln_plain = torch.nn.functional.layer_norm(Tf, (H,), None, None, eps)
    U_ref = torch.nn.functional.gelu(ln_plain @ W1f.float().t() + (b1 + W1 @ beta))
    err = (U.float() - U_ref).abs()
    assert not bool((err > 2.0 ** -8 * U_ref.abs() + 2e-3).any()), float(err.max())
    U_true = torch.nn.functional.gelu(ln @ W1.t() + b1)                      # unrounded weights: looser
    assert float((U.float() - U_true).abs().max()) < 0.05

    T2, part2 = _gemm_ln(2, 2, U, W2, b2, T, stat, gamma, beta, True)
    T2_ref = ln + U.float() @ W2.float().t() + b2
    err = (T2.float() - T2_ref).abs()
    assert not bool((err > 2.0 ** -8 * T2_ref.abs() + 2e-3).any()), float(err.max())
    assert torch.allclose(part2[..., 0].sum(0), T2.float().sum(1), rtol=1e-5, atol=1e-3)


def test_layernorm_fold_rejects_bad_modes():
    A = torch.zeros((256, 64), dtype=torch.bfloat16, device="cuda")
    W = torch.zeros((256, 64), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError):
        _gemm_ln(0, 2, A, W, None, None, None, None, None, False)        # mode 2 is a residual-epilogue mode
    with pytest.raises(ValueError):
        _gemm_ln(0, 1, A, W, None, None, None, None, None, False)        # folding without stat / s / c
    with pytest.raises(ValueError):
        _gemm_ln(2, 1, A, W, None, A.repeat(1, 4), None, None, None, False)   # statistics mode without an output
