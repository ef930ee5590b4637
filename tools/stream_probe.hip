// tools/stream_probe.hip — what a plain streaming read of a 3.07 GB buffer reaches on this GPU (gfx950):
// the ceiling the HBM-bound small-batch scan is measured against, beside the 8 TB/s spec.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o variants/stream_probe && variants/stream_probe
// Variants: (a) 16-byte global loads to registers, 8 in flight per lane, default / non-temporal policy;
//           (b) LDS-DMA (global_load_lds_dwordx4) of 1 KiB per wave-instruction into a ring, nothing consumed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int NT_POLICY>
__global__ __launch_bounds__(256) void read_regs(const f32x4v* __restrict__ x, size_t n4, float* out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        f32x4v v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = NT_POLICY ? __builtin_nontemporal_load(&x[i + u * stride]) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) acc += x[i].x;
    if (acc == 123.456f) out[0] = acc;     // (keeps the loads alive)
}

__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst, bool) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// one workgroup per CU streams a contiguous strip through a ring of SLOTS x 32 KiB; every wave issues 8 KiB per slot
template <int SLOTS>
__global__ __launch_bounds__(256) void read_dma(const char* __restrict__ x, size_t bytes_per_wg, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = x + (size_t)blockIdx.x * bytes_per_wg;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int nslots = (int)(bytes_per_wg / 32768);
    const unsigned voff = (unsigned)lane * 16u;
    auto issue = [&](int s) {
        const char* g = base + (size_t)s * 32768 + wave * 8192;
        const unsigned l = lds0 + (unsigned)(s % SLOTS) * 32768u + wave * 8192u;
#pragma unroll
        for (int p = 0; p < 8; ++p) dma16(g + p * 1024, voff, l + p * 1024, false);
    };
    for (int s = 0; s < SLOTS - 1 && s < nslots; ++s) issue(s);
    for (int s = 0; s < nslots; ++s) {
        if (s + SLOTS - 1 < nslots) {
            issue(s + SLOTS - 1);
            if (SLOTS == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (SLOTS == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
    }
    if (smem[threadIdx.x] == 77 && out) out[0] = 1.f;
}

int main() {
    const size_t bytes = (size_t)1000000 * 768 * 4;
    char* x;
    float* out;
    CK(hipMalloc(&x, bytes));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(x, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 20;
        printf("{\"probe\": \"%s\", \"ms\": %.4f, \"TBps\": %.3f}\n", name, ms, bytes / (ms * 1e-3) / 1e12);
    };
    for (int wgs : {2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, sizeof nm, "regs16B_x8_default_%dwg", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL(read_regs<0>, dim3(wgs), dim3(256), 0, 0, (const f32x4v*)x, bytes / 16, out); });
        snprintf(nm, sizeof nm, "regs16B_x8_nt_%dwg", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL(read_regs<1>, dim3(wgs), dim3(256), 0, 0, (const f32x4v*)x, bytes / 16, out); });
    }
    const size_t per_wg = bytes / 256 / 32768 * 32768;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(read_dma<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(read_dma<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(read_dma<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768));
    timeit("lds_dma_ring2x32K_256wg", [&] { hipLaunchKernelGGL(read_dma<2>, dim3(256), dim3(256), 2 * 32768, 0, x, per_wg, out); });
    timeit("lds_dma_ring3x32K_256wg", [&] { hipLaunchKernelGGL(read_dma<3>, dim3(256), dim3(256), 3 * 32768, 0, x, per_wg, out); });
    timeit("lds_dma_ring4x32K_256wg", [&] { hipLaunchKernelGGL(read_dma<4>, dim3(256), dim3(256), 4 * 32768, 0, x, per_wg, out); });
    return 0;
}
