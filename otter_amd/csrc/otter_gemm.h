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

}  // namespace otter
