// otter_device.h — device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace otter {

typedef uint32_t u32;
typedef unsigned long long u64;
typedef unsigned short bf16_t;  // storage type of a bfloat16
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define OTTER_AS1 __attribute__((address_space(1)))
#define OTTER_AS3 __attribute__((address_space(3)))

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS-DMA: 64 lanes x 16 B, wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset,
// to LDS [lds_dst, lds_dst + 1 KiB) in lane order.  Inline asm on purpose: hipcc then neither
// counts it nor guards later LDS accesses with vmcnt(0); callers order it by hand
// (counted s_waitcnt vmcnt + workgroup barrier before the first ds_read of the data).
__device__ __forceinline__ void dma16s(const void* sbase, u32 voff, u32 lds_dst) {
    u32 keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

// workgroup barrier that does NOT drain the vector-memory queue (LDS-DMA stays in flight)
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float((u32)x << 16); }
__device__ __forceinline__ bf16_t f2bf(float x) { return __builtin_bit_cast(bf16_t, (__bf16)x); }  // RNE, NaN kept

template <typename T> __device__ __forceinline__ float to_f(T x);
template <> __device__ __forceinline__ float to_f<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t x) { return bf2f(x); }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return f2bf(x); }

// XCD-aware, bijective remap of a 1-D grid: workgroups b, b+8, ... share an XCD; give each XCD
// a contiguous range of the logical index space.  Speed only; every mapping is correct.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int per = nwg >> 3, rem = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + idx;
}

}  // namespace otter
