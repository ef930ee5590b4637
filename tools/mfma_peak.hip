// tools/mfma_peak.hip — microbenchmark: what rate does v_mfma_f32_32x32x2_f32 sustain on this box,
// alone and with the scan kernel's ds_read_b128 mix?  (measurement aid, not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float4 a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 2, seed, 1.f, 2.f};
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) ((float*)lds)[i] = seed + i;
    __syncthreads();
    const char* p = lds + (lane & 31) * 256 + (((lane >> 5)) ^ (lane & 15)) * 16;
    float4 na = a, nb = b, na2 = a, nb2 = b;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {   // prefetch next iteration's fragments (register double buffer), as the scan kernel does
            a = na; b = nb;
            if (MODE == 2) { a.x += na2.y; b.x += nb2.y; }
            na = *(const float4*)(p + (((it + 1) & 7) * 32));
            nb = *(const float4*)(p + 8192 + (((it + 1) & 7) * 32));
            if (MODE == 2) {
                na2 = *(const float4*)(p + 16384 + (((it + 1) & 7) * 32));
                nb2 = *(const float4*)(p + 24576 + (((it + 1) & 7) * 32));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(const char* name, int threads) {
    float* out; (void)hipMalloc(&out, 4096 * 512 * 4);
    const int iters = 20000, grid = 256 * (256 / threads == 0 ? 1 : 1);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NACC>), dim3(grid), dim3(threads), 0, 0, out, iters, 1.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)grid * (threads / 64) * iters * 4.0 * NACC * 4096.0;
        if (rep == 2) printf("%-40s threads=%d  %.3f ms  %.1f TFLOP/s\n", name, threads, ms, flops / ms / 1e9);
    }
    (void)hipFree(out);
}

int main() {
    run<0, 4>("pure mfma, 4 acc", 256);
    run<0, 2>("pure mfma, 2 acc", 256);
    run<0, 1>("pure mfma, 1 acc (dependent chain)", 256);
    run<0, 2>("pure mfma, 2 acc, 2 waves/SIMD", 512);
    run<1, 4>("mfma 4 acc + 2 prefetched ds_read_b128 / 16 mfma", 256);
    run<2, 4>("mfma 4 acc + 4 prefetched ds_read_b128 / 16 mfma", 256);
    run<1, 2>("mfma 2 acc + 2 ds_read_b128 / 8 mfma, 2w/SIMD", 512);
    run<2, 2>("mfma 2 acc + 4 ds_read_b128 / 8 mfma, 2w/SIMD", 512);
    return 0;
}
