// otter_gemm.h — internal interface of the bf16 encoder GEMM (gemm_bf16.hip).
#pragma once
#include "otter_common.h"

namespace otter {

enum { GEMM_EPI_BIAS = 0, GEMM_EPI_BIAS_GELU = 1, GEMM_EPI_BIAS_RESID = 2 };

// C[M_pad, N] = A[M_pad, K] * W[N, K]^T + bias[N] (+ gelu | + R[M_pad, N]); bf16 in/out, fp32 accumulate.
// M_pad % 256 == 0, N % 256 == 0, K % 64 == 0; every pointer addresses whole 256-row tiles.
// `ncu` = compute units of the device (persistent grid size).
int launch_gemm_bf16(int epi, const void* A, const void* W, const float* bias, const void* R, void* C,
                     long long M_pad, int N, int K, int ncu, hipStream_t st);

// LayerNorm folded around the GEMM (bf16 encoder path).
//   epilogue 0 / 1, mode 1: the rows of A are un-normalised; W must be the gamma-scaled matrix of
//       launch_fold_ln, v0 = s_n, v1 = c_n, stat = (rstd, -mean rstd) of the rows of A; `bias` is unused.
//   epilogue 2, mode 1: also write (sum, sum of squares) partials of the stored rows to
//       part[4 * N / 256][M_pad] (float2), for launch_ln_stats.
//   epilogue 2, mode 2: mode 1, and the rows of R are un-normalised: R' = (R rstd - mean rstd) gamma + beta
//       with stat of the rows of R, v0 = gamma, v1 = beta.
struct GemmLN {
    int mode = 0;
    const void* stat = nullptr;   // float2 [M_pad]
    const float* v0 = nullptr;    // [N]
    const float* v1 = nullptr;    // [N]
    void* part = nullptr;         // float2 [4 * N / 256][M_pad]
    // independent of `mode`: device int, the number of 256-row tiles in use (<= M_pad / 256); the
    // packed variable-length encoder path knows its row count on the device only
    const int* tiles_m_dev = nullptr;
};
int launch_gemm_bf16_ln(int epi, const void* A, const void* W, const float* bias, const void* R, void* C,
                        long long M_pad, int N, int K, const GemmLN& ln, int ncu, hipStream_t st);
// stat[r] = (rstd, -mean * rstd) from P partial (sum, sum of squares) pairs per row over N features
int launch_ln_stats(const void* part, int P, long long M_pad, int N, float eps, void* stat, hipStream_t st);
// Wout[n,k] = bf16(gamma_k W[n,k]); s_out[n] = sum_k Wout[n,k]; c_out[n] = bias_n + sum_k beta_k W[n,k]
int launch_fold_ln(const float* W, const float* gamma, const float* beta, const float* bias, void* Wout, float* s_out,
                   float* c_out, int N, int K, hipStream_t st);

}  // namespace otter
