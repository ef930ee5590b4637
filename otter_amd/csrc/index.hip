// index.hip — exact inner-product top-k over a row-major fp32 corpus in HBM (gfx950).
//
// Replaces faiss.IndexFlatIP add()/search() as the reference calls them
// (scripts/eval_ottqa_retrieval.py:108-109,129).  The N x Q score matrix is never
// written: one kernel streams corpus rows through LDS (LDS-DMA ring), multiplies
// them with a resident tile of queries on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32, exact fp32 fma chain) and filters every score against a
// per-query running threshold; survivors go to a small per-(strip, query)
// candidate buffer that is compacted by a wavefront radix select whenever it fills.
// A second, tiny kernel bitonic-sorts the per-strip survivors of each query.
//
// Data layout
//   corpus   X  [N][ld]   fp32, ld = d rounded up to 64 (zero padded), row i = id i
//   queries  Qp [nq_pad][ld] fp32, zero padded to a multiple of the query tile
//   cand     [S][nq_pad][CAP] u64 composite = key32(score) << 32 | (0xFFFFFFFF - row)
//            (larger composite == ranks earlier: score desc, id asc; composites are unique)
//   cnt      [S][nq_pad] u32 number of valid composites
#include <float.h>
#include <stdlib.h>

#include <algorithm>
#include <new>
#include <utility>
#include <vector>

#include "otter_common.h"

namespace {

using namespace otter;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;
typedef uint32_t u32;

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

#ifndef SCAN_G0
#define SCAN_G0 4                 // k-groups (of 8) of a stage over which its LDS-DMA pieces are issued
#endif
constexpr int LD_ALIGN = 64;       // corpus / query row length is padded to a multiple of this
constexpr int NTHREADS = 256;      // 4 waves, one per SIMD

// monotone map float -> u32 (larger float <=> larger key); -0 is folded into +0
__device__ __forceinline__ u32 f2key(float s) {
    s = s + 0.0f;
    u32 u = __float_as_uint(s);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key2f(u32 k) {
    u32 u = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(u);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS [lds_dst, lds_dst + 1 KiB).
// Issued from inline asm on purpose: hipcc then neither counts it nor guards later LDS
// accesses with vmcnt(0); the kernel orders it by hand (counted vmcnt + barrier).
__device__ __forceinline__ void dma16(const void* gsrc, u32 lds_dst) {
    u32 keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
// Same with a wave-uniform 64-bit base (SGPR pair) and a per-lane 32-bit byte offset:
// the per-lane part is computed once per kernel, the base with scalar ALU per stage.
__device__ __forceinline__ void dma16s(const void* sbase, u32 voff, u32 lds_dst) {
    u32 keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}
// workgroup barrier that does NOT drain the vector-memory queue (LDS-DMA stays in flight)
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct ScanArgs {
    const float* X;     // [n][ld]
    const float* Q;     // [nq_pad][ld]
    u64* cand;          // [S][nq_pad][CAP]
    u32* cnt;           // [S][nq_pad]
    long long n;        // corpus rows
    int ld;             // padded row length (floats), multiple of 64
    int nq;             // real queries
    int nq_pad;         // multiple of BQ
    int k;
    int qtiles;         // nq_pad / BQ
    int strips;         // S
    int total_tiles;    // tiles scanned by this launch
    int tile_begin;     // first tile (rows below tile_begin * BM belong to the sample pass)
    const u32* thr0;    // [nq_pad] initial score-key threshold per query (null: none)
    int order;          // 0: strip fastest within an XCD's range, 1: query tile fastest (speed only)
    u64* dbg;           // SCAN_DIAG builds only: per-wave cycle sums {total, stage wait, epilogue, stages}
};

#ifdef SCAN_DIAG
__device__ __forceinline__ u64 stamp() {
    u64 t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#endif

// One wavefront selects the k largest of the n (<= 64*EPL) composites at `base`,
// moves them to base[0..k) and returns the k-th largest.  Radix select, one bit
// per round, counted with ballots; composites are unique so exactly k survive.
template <int EPL>
__device__ __forceinline__ u64 wave_compact(u64* base, int n, int k, int lane) {
    u64 x[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        int idx = e * 64 + lane;
        // agent-scope load: bypass this CU's vector L1 (entries were stored by other waves)
        x[e] = idx < n ? __hip_atomic_load(base + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
    u64 T = 0;
    for (int b = 63; b >= 0; --b) {
        const u64 c = T | (1ull << b);
        int cntv = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) cntv += __popcll(__ballot(x[e] >= c));
        if (cntv >= k) T = c;
        if (cntv == k) break;   // exactly k composites are >= T: T selects the same set as the k-th largest
    }
    int off = 0;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const bool keep = x[e] >= T;
        const u64 m = __ballot(keep);
        const int pos = off + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0));
        if (keep) base[pos] = x[e];
        off += __popcll(m);
    }
    return T;
}

// WM x WQ waves; each wave owns (MB*32) corpus rows x (QB*32) queries of the tile.
// BK floats of k per LDS stage (64: 256-B rows, 32: 128-B rows); NSTAGE ring slots.
template <int WM, int WQ, int MB, int QB, int BK, int NSTAGE, int EPL>
__global__ __launch_bounds__(WM * WQ * 64, WM * WQ / 4) void scan_kernel(ScanArgs a) {
    static_assert(BK == 64 || BK == 32, "stage depth");
    constexpr int ROW_BYTES = BK * 4;
    constexpr int CH = BK / 4;           // 16-B chunks per row
    constexpr int RPI = 64 / CH;         // rows per LDS-DMA instruction (1 KiB)
    constexpr int NW = WM * WQ;          // waves per workgroup (4: one per SIMD, 8: two)
    constexpr int NT = NW * 64;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    constexpr int BM = WM * MB * 32;
    constexpr int BQ = WQ * QB * 32;
    constexpr int CAP = EPL * 64;
    constexpr int A_BYTES = BM * ROW_BYTES;
    constexpr int B_BYTES = BQ * ROW_BYTES;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_INSTR = BM / RPI / NW;  // LDS-DMA instructions per wave per stage (1 KiB each)
    constexpr int B_INSTR = BQ / RPI / NW;
    static_assert(A_INSTR * RPI * NW == BM && B_INSTR * RPI * NW == BQ, "tile rows must split over the waves");
    constexpr int IPS = A_INSTR + B_INSTR;
    constexpr int G0 = SCAN_G0 < BK / 8 ? SCAN_G0 : BK / 8;  // k-groups over which a stage's DMA is issued
    static_assert(CAP - BM >= 128, "candidate buffer too small");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32* sCnt = reinterpret_cast<u32*>(smem + NSTAGE * STAGE_BYTES);
    u32* sThr = sCnt + BQ;
    u32* sFlag = sThr + BQ;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WQ;
    const int wq = wave % WQ;

    // XCD-aware placement: workgroups b, b+8, b+16, ... share an XCD (and its 4 MiB L2).
    // Give each XCD a contiguous range of the (query tile, strip) list with the strip
    // varying fastest, so that one XCD works on few query tiles (their 393 KB stay in
    // its L2) while the corpus rows stream through.  Pure speed: any placement is correct.
    int wg;
    {
        const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wg = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + idx;
    }
    int qt, strip;
    if (a.order) {   // workgroups of an XCD walk the same corpus strips for different query tiles: rows hit in L2
        strip = wg / a.qtiles;
        qt = wg - strip * a.qtiles;
    } else {
        qt = wg / a.strips;
        strip = wg - qt * a.strips;
    }
    const int q0 = qt * BQ;
    const int t0 = a.tile_begin + (int)(((long long)strip * a.total_tiles) / a.strips);
    const int t1 = a.tile_begin + (int)(((long long)(strip + 1) * a.total_tiles) / a.strips);
    const int KS = a.ld / BK;

    for (int i = tid; i < BQ; i += NT) {
        sCnt[i] = 0;
        // padded queries never accept; a sample pass may have left the score key of each query's
        // k-th best among rows with smaller ids: only strictly larger keys can still enter its top k
        sThr[i] = (q0 + i < a.nq) ? (a.thr0 ? a.thr0[q0 + i] : 0u) : 0xFFFFFFFFu;
    }
    if (tid == 0) *sFlag = 0;

    u64* candBase = a.cand + ((size_t)strip * a.nq_pad + q0) * CAP;

    // ---- LDS-DMA producer: one k-stage of one tile into a ring slot ----
    // LDS image of a stage: [row][CH chunks of 16 B]; chunk p of row r holds the global
    // chunk p ^ swz(r) (swizzle on the SOURCE address; the DMA writes linearly), swz chosen
    // so that the 16 rows one ds_read_b128 lane group touches fall on 16 distinct 16-B slots
    // of the 256-B bank row: r & 15 for 256-B rows, (r >> 1) & 7 for 128-B rows.
    // A stage is IPS wave-instructions of 1 KiB (4 rows) per wave.  Per-lane byte offsets
    // are fixed for the whole kernel; the corpus allocation is padded to whole tiles, so
    // no row clamp is needed (rows >= n are masked in the epilogue).
    const int dma_r = lane / CH;   // row within the row group of one instruction
    const int dma_p = lane % CH;   // LDS chunk position
    auto swz = [](int r) { return CH == 16 ? (r & 15) : ((r >> 1) & 7); };
    const u32 lds0 = (u32)(uintptr_t)(AS3 char*)smem;
    u32 voff[IPS];
#pragma unroll
    for (int t = 0; t < IPS; ++t) {
        const int i = (t < A_INSTR ? t : t - A_INSTR) * NW + wave;
        const int r = RPI * i + dma_r;
        voff[t] = ((u32)r * (u32)a.ld + (u32)((dma_p ^ swz(r)) << 2)) * 4u;
    }
    const float* gQ = a.Q + (size_t)q0 * a.ld;

    long long f = 0;   // flat index of the stage being consumed
    // the stage being produced: scalar bases + ring slot; pieces are issued one by one
    int it_tile = t0, it_ks = 0;
    bool pr_have = false;
    const float* pr_gA = a.X;
    const float* pr_gB = gQ;
    u32 pr_lds = lds0;
    auto produce_begin = [&](int slot) {
        pr_have = it_tile < t1;
        if (pr_have) {
            pr_gA = a.X + ((size_t)it_tile * BM * a.ld + (size_t)it_ks * BK);
            pr_gB = gQ + (size_t)it_ks * BK;
            pr_lds = lds0 + slot * STAGE_BYTES;
            if (++it_ks == KS) { it_ks = 0; ++it_tile; }
        }
    };
    auto produce_piece = [&](int t) {   // t in [0, IPS), compile-time after unrolling
#ifdef SCAN_ABL_NODMA
        if (pr_have && f < 0) {   // timing-only ablation: never issue (results are garbage)
#else
        if (pr_have) {
#endif
            if (t < A_INSTR) dma16s(pr_gA, voff[t], pr_lds + (t * NW + wave) * 1024);
            else dma16s(pr_gB, voff[t], pr_lds + A_BYTES + ((t - A_INSTR) * NW + wave) * 1024);
        }
    };

    f32x16 acc[MB][QB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int q = 0; q < QB; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.0f;

    // rare path: compact every over-full candidate buffer of this workgroup
    auto compact_pass = [&](bool final_pass) {
        wait_vmcnt<0>();  // own candidate stores have reached L2
        wg_barrier();
        for (int q = wave; q < BQ; q += NW) {
            const int n = (int)sCnt[q];
            const bool need = final_pass ? (n > a.k) : (n > CAP - BM);
            if (need) {
                const u64 T = wave_compact<EPL>(candBase + (size_t)q * CAP, n, a.k, lane);
                if (lane == 0) {
                    sCnt[q] = (u32)a.k;
                    sThr[q] = (u32)(T >> 32);
                }
            }
        }
        if (tid == 0) *sFlag = 0;
        wait_vmcnt<0>();
        wg_barrier();
    };

    const long long total_stages = (long long)(t1 - t0) * KS;
#ifdef SCAN_DIAG
    u64 dg_wait = 0, dg_epi = 0;
    const u64 dg_t0 = stamp();
#endif

    // prologue: fill NSTAGE-1 ring slots
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) {
        produce_begin(s);
#pragma unroll
        for (int t = 0; t < IPS; ++t) produce_piece(t);
    }

    int slot = 0;
    const int h = lane >> 5;
    const int l31 = lane & 31;
    const int lswz = swz(l31);     // block row offsets are multiples of 32: swz(row) == swz(lane & 31)

    for (int tile = t0; tile < t1; ++tile) {
        for (int ks = 0; ks < KS; ++ks, ++f) {
#ifdef SCAN_DIAG
            const u64 dg_a = stamp();
#endif
            // stage f has landed once at most (NSTAGE-2) younger stages are outstanding
            if (total_stages - 1 - f >= NSTAGE - 2) wait_vmcnt<(NSTAGE - 2) * IPS>();
            else wait_vmcnt<0>();
            wg_barrier();
#ifdef SCAN_DIAG
            dg_wait += stamp() - dg_a;
#endif
            if (ks == 0 && tile > t0) {
                // every wave has finished the previous tile's epilogue
                if (__builtin_amdgcn_readfirstlane(*sFlag)) compact_pass(false);  // LDS read, reloaded after the barrier
            }
            const char* sA = smem + slot * STAGE_BYTES;
            const char* sB = sA + A_BYTES;
            // fragments of k-group j: 16 B per lane = 4 MFMA k-steps (k = 8j + 4h + t)
            float4 af[2][MB], bf[2][QB];
            auto load_frags = [&](int j, int buf) {
                const int coff = ((2 * j + h) ^ lswz) << 4;
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    af[buf][m] =
                        *reinterpret_cast<const float4*>(sA + ((wm * MB + m) * 32 + l31) * ROW_BYTES + coff);
#pragma unroll
                for (int q = 0; q < QB; ++q)
                    bf[buf][q] =
                        *reinterpret_cast<const float4*>(sB + ((wq * QB + q) * 32 + l31) * ROW_BYTES + coff);
            };
            load_frags(0, 0);
            // slot (slot-1) was consumed by every wave before the barrier above: refill it
            // with stage f+NSTAGE-1, a few pieces per k-group so the issue hides under MFMAs
            produce_begin((slot + NSTAGE - 1) % NSTAGE);
#pragma unroll
            for (int j = 0; j < BK / 8; ++j) {
                const int cb = j & 1;
                if (j + 1 < BK / 8) load_frags(j + 1, cb ^ 1);  // register double buffer
#pragma unroll
                for (int t = j; t < IPS; t += G0)
                    if (j < G0) produce_piece(t);
                // keep the reads of group j+1 ahead of group j's MFMAs (hipcc sinks them otherwise)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int q = 0; q < QB; ++q) {
                        acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][m].x, bf[cb][q].x, acc[m][q], 0, 0, 0);
                        acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][m].y, bf[cb][q].y, acc[m][q], 0, 0, 0);
                        acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][m].z, bf[cb][q].z, acc[m][q], 0, 0, 0);
                        acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][m].w, bf[cb][q].w, acc[m][q], 0, 0, 0);
                    }
            }
            slot = (slot + 1) % NSTAGE;
        }

        // ---- epilogue of one tile: threshold filter, append survivors ----
#ifdef SCAN_DIAG
        const u64 dg_e = stamp();
#endif
        const long long row0 = (long long)tile * BM;
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const int ql = (wq * QB + q) * 32 + l31;
            const u32 thr = sThr[ql];
            // float image of the threshold for the cheap pre-test (thr == 0: accept everything)
            const float thrf = key2f(thr);
            u64* qbuf = candBase + (size_t)ql * CAP;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                // after warm-up almost no score beats the running k-th best: test the block's
                // maximum first (8 v_max3) and only then look at the 16 scores one by one
                float mx = fmaxf(acc[m][q][0], acc[m][q][1]);
#pragma unroll
                for (int r = 2; r < 16; ++r) mx = fmaxf(mx, acc[m][q][r]);
                if (thr == 0u || mx > thrf || mx != mx) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rt = (wm * MB + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const long long grow = row0 + rt;
                        const u32 key = f2key(acc[m][q][r]);
                        if (key > thr && grow < a.n) {
                            const u32 pos = atomicAdd(&sCnt[ql], 1u);
                            if (pos < (u32)CAP) qbuf[pos] = ((u64)key << 32) | (u64)(0xFFFFFFFFu - (u32)grow);
                            if (pos + 1 > (u32)(CAP - BM)) *sFlag = 1;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.0f;
            }
        }
#ifdef SCAN_DIAG
        dg_epi += stamp() - dg_e;
#endif
    }

#ifdef SCAN_DIAG
    if (a.dbg && lane == 0) {
        u64* o = a.dbg + ((size_t)blockIdx.x * NW + wave) * 4;
        o[0] = stamp() - dg_t0;
        o[1] = dg_wait;
        o[2] = dg_epi;
        o[3] = (u64)total_stages;
    }
#endif
    // ---- end of strip: leave at most k composites per query, publish counts ----
    compact_pass(true);
    for (int i = tid; i < BQ; i += NT) a.cnt[(size_t)strip * a.nq_pad + q0 + i] = sCnt[i];
}

// queries -> zero-padded [nq_pad][ld]
__global__ void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long long rows,
                                long long rows_pad, int d, int ld) {
    const long long total = rows_pad * ld;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / ld;
        const int c = (int)(i - r * ld);
        dst[i] = (r < rows && c < d) ? src[r * d + c] : 0.0f;
    }
}

// ---- merge: bitonic sort (descending) of the composites of one query ----
struct MergeArgs {
    const u64* in;       // [S_in][nq_pad][cap_in]
    const u32* in_cnt;   // [S_in][nq_pad]
    u64* out;            // [S_out][nq_pad][k]       (intermediate levels)
    u32* out_cnt;        // [S_out][nq_pad]
    float* D;            // [nq][k]                   (final level)
    long long* I;        // [nq][k]
    long long id_base;
    int cap_in;
    int S_in;
    int fan;             // strips merged by one workgroup
    int nq_pad;
    int k;
    int npad;            // power of two >= fan * min(cap_in, k) (+ k with an extra list)
    int final_level;
    const u64* ex;       // final level only: one more sorted list [nq_pad][k] (the sample pass), or null
    const u32* ex_cnt;   // [nq_pad]
    u32* thr_out;        // non-final level with one group: [nq_pad] key of the k-th best, 0 if fewer than k
};

constexpr int MAX_SELECT = 8192;       // composites one merge workgroup selects from in LDS
constexpr int MERGE_THREADS = 1024;
constexpr int MERGE_MAX_STRIPS = 1024;   // strips one merge workgroup gathers (host keeps fan below it)

__device__ __forceinline__ void bitonic_desc_u64(u64* l, int n, int tid, int nthreads) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = tid; i < (n >> 1); i += nthreads) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const u64 x = l[lo], y = l[hi];
                if ((x < y) == desc) { l[lo] = y; l[hi] = x; }
            }
        }
    }
    __syncthreads();
}

// One workgroup merges the candidate lists of up to `fan` strips of one query: counts are read
// in parallel and prefix-summed in LDS (no chain of dependent global loads), one wave copies one
// strip at a time, and only the live power-of-two prefix is sorted.
__global__ __launch_bounds__(MERGE_THREADS) void merge_kernel(MergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* l = reinterpret_cast<u64*>(smem);
    int* sOff = reinterpret_cast<int*>(smem + (size_t)a.npad * sizeof(u64));   // [ns + 2]
    const int q = blockIdx.x;
    const int g = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s0 = g * a.fan;
    const int ns = min(a.S_in, s0 + a.fan) - s0;
    const bool extra = a.final_level && a.ex;
    for (int t = tid; t < ns + (extra ? 1 : 0); t += MERGE_THREADS) {
        const int c = t < ns ? (int)a.in_cnt[(size_t)(s0 + t) * a.nq_pad + q] : (int)a.ex_cnt[q];
        sOff[t + 1] = min(c, a.k);
    }
    __syncthreads();
    if (wave == 0) {   // inclusive scan of <= MERGE_MAX_STRIPS + 1 counts by one wave
        int run = 0;
        const int n = ns + (extra ? 1 : 0);
        for (int c0 = 0; c0 < n; c0 += 64) {
            int v = c0 + lane < n ? sOff[c0 + lane + 1] : 0;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(v, o);
                if (lane >= o) v += u;
            }
            if (c0 + lane < n) sOff[c0 + lane + 1] = run + v;
            run += __shfl(v, 63);
        }
        if (lane == 0) sOff[0] = 0;
    }
    __syncthreads();
    const int nlist = ns + (extra ? 1 : 0);
    const int off = sOff[nlist];
    for (int t = wave; t < nlist; t += MERGE_THREADS / 64) {
        const int o0 = sOff[t], c = sOff[t + 1] - o0;
        const u64* src = t < ns ? a.in + ((size_t)(s0 + t) * a.nq_pad + q) * a.cap_in : a.ex + (size_t)q * a.k;
        for (int i = lane; i < c; i += 64) l[o0 + i] = src[i];
    }
    __syncthreads();
    // Block-wide radix select of the k largest (composites are unique and non-zero): every thread
    // keeps its <= MAX_SELECT / MERGE_THREADS elements in registers, one ballot count per bit,
    // stop as soon as exactly k remain; only the survivors are sorted.
    constexpr int EPT = MAX_SELECT / MERGE_THREADS;
    u64 x[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int idx = e * MERGE_THREADS + tid;
        x[e] = idx < off ? l[idx] : 0ull;
    }
    u64 T = 1ull;                               // off <= k: keep every gathered composite
    int* sW = sOff + MERGE_MAX_STRIPS + 4;      // [2][16] per-wave counts, double buffered by round
    int* sPos = sW + 32;
    if (off > a.k) {
        T = 0ull;
        int par = 0;
        for (int b = 63; b >= 0; --b, par ^= 1) {
            const u64 c = T | (1ull << b);
            int cw = 0;
#pragma unroll
            for (int e = 0; e < EPT; ++e) cw += __popcll(__ballot(x[e] >= c));
            if (lane == 0) sW[par * 16 + wave] = cw;
            __syncthreads();
            int cntv = 0;
#pragma unroll
            for (int w = 0; w < MERGE_THREADS / 64; ++w) cntv += sW[par * 16 + w];
            if (cntv >= a.k) T = c;
            if (cntv == a.k) break;             // uniform: every thread sees the same counts
        }
    }
    if (tid == 0) *sPos = 0;
    __syncthreads();                            // every thread holds its elements: l may be overwritten
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const bool keep = x[e] >= T && x[e] != 0ull;
        const u64 m = __ballot(keep);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(sPos, __popcll(m));
        base = __shfl(base, 0);
        if (keep) l[base + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0))] = x[e];
    }
    __syncthreads();
    const int kept = *sPos;                     // == min(off, k)
    int npad = 2;
    while (npad < kept) npad <<= 1;             // <= next_pow2(k) <= a.npad
    for (int i = kept + tid; i < npad; i += MERGE_THREADS) l[i] = 0ull;
    bitonic_desc_u64(l, npad, tid, MERGE_THREADS);
    const int total = min(off, a.k);
    if (a.final_level) {
        for (int t = tid; t < a.k; t += MERGE_THREADS) {
            const u64 e = t < total ? l[t] : 0ull;
            const bool valid = t < total;
            a.D[(size_t)q * a.k + t] = valid ? key2f((u32)(e >> 32)) : -FLT_MAX;
            a.I[(size_t)q * a.k + t] = valid ? a.id_base + (long long)(0xFFFFFFFFu - (u32)e) : -1ll;
        }
    } else {
        u64* dst = a.out + ((size_t)g * a.nq_pad + q) * a.k;
        for (int t = tid; t < total; t += MERGE_THREADS) dst[t] = l[t];
        if (tid == 0) {
            a.out_cnt[(size_t)g * a.nq_pad + q] = (u32)total;
            if (a.thr_out) a.thr_out[q] = total >= a.k ? (u32)(l[a.k - 1] >> 32) : 0u;
        }
    }
}

// ---- merge of G sorted (D, I) lists per query: the sharded search's last step ----
struct PairMergeArgs {
    const float* Dc;       // [G][nq][k]
    const long long* Ic;   // [G][nq][k]
    float* D;              // [nq][k]
    long long* I;
    long long nq;
    int G;
    int k;
    int npad;
};

__device__ __forceinline__ bool pair_before(u32 ka, long long ia, u32 kb, long long ib) {
    return ka > kb || (ka == kb && ia < ib);
}

__global__ __launch_bounds__(NTHREADS) void pair_merge_kernel(PairMergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long* lid = reinterpret_cast<long long*>(smem);
    u32* lkey = reinterpret_cast<u32*>(smem + (size_t)a.npad * 8);
    const long long q = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = a.G * a.k;
    for (int i = tid; i < a.npad; i += NTHREADS) {
        u32 key = 0;
        long long id = 0x7FFFFFFFFFFFFFFFll;  // padding ranks last
        if (i < n) {
            const int g = i / a.k, t = i - g * a.k;
            const size_t src = ((size_t)g * a.nq + q) * a.k + t;
            const long long v = a.Ic[src];
            if (v >= 0) { id = v; key = f2key(a.Dc[src]); }
        }
        lkey[i] = key;
        lid[i] = id;
    }
    for (int size = 2; size <= a.npad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = tid; i < (a.npad >> 1); i += NTHREADS) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const u32 kx = lkey[lo], ky = lkey[hi];
                const long long ix = lid[lo], iy = lid[hi];
                // descending run: the element that ranks first goes to `lo`
                const bool swap = desc ? pair_before(ky, iy, kx, ix) : pair_before(kx, ix, ky, iy);
                if (swap) { lkey[lo] = ky; lkey[hi] = kx; lid[lo] = iy; lid[hi] = ix; }
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < a.k; t += NTHREADS) {
        const bool valid = lid[t] != 0x7FFFFFFFFFFFFFFFll;
        a.D[(size_t)q * a.k + t] = valid ? key2f(lkey[t]) : -FLT_MAX;
        a.I[(size_t)q * a.k + t] = valid ? lid[t] : -1ll;
    }
}

// ------------------------------------------------------------------ host side

constexpr long long QCHUNK = 16384;    // queries per scan launch (bounds scratch)

struct ScanCfg {
    int bm, bq, cap;
    size_t lds;
    void (*launch)(const ScanArgs&, int grid, size_t lds, hipStream_t);
};

template <int WM, int WQ, int MB, int QB, int BK, int NSTAGE, int EPL>
void launch_scan(const ScanArgs& a, int grid, size_t lds, hipStream_t st) {
    auto kern = scan_kernel<WM, WQ, MB, QB, BK, NSTAGE, EPL>;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WQ * 64), lds, st, a);
}

template <int WM, int WQ, int MB, int QB, int BK, int NSTAGE, int EPL>
ScanCfg make_cfg() {
    constexpr int BM = WM * MB * 32, BQ = WQ * QB * 32;
    ScanCfg c;
    c.bm = BM;
    c.bq = BQ;
    c.cap = EPL * 64;
    c.lds = (size_t)NSTAGE * (BM + BQ) * BK * 4 + (size_t)(2 * BQ + 4) * sizeof(u32);
    c.launch = &launch_scan<WM, WQ, MB, QB, BK, NSTAGE, EPL>;
    return c;
}

// query-tile shapes: S (<=32 queries, HBM-bound), M (<=64), L (MFMA-bound)
// EPL = candidate slots per lane of the compaction (buffer capacity 64*EPL): the k class's
// base (2 * kmax + 128, i.e. room for kmax new survivors between compactions at 128-row
// tiles) plus 2 per extra 128 rows of the tile.
template <int EPL>
ScanCfg pick_shape(long long nq) {
    if (nq <= 32) return make_cfg<4, 1, 1, 1, 64, 3, EPL>();
    if (nq <= 64) return make_cfg<4, 1, 1, 2, 64, 3, EPL>();
    static const char* v = getenv("OTTER_SCAN_VARIANT");   // A/B switch for tuning runs
    if (v && v[0] == 'a') return make_cfg<2, 2, 2, 2, 64, 2, EPL>();      // 128x128, 4 waves, 2 x 64 KB
    if (v && v[0] == 'b') return make_cfg<2, 4, 2, 1, 64, 2, EPL>();      // 128x128, 8 waves, 2 x 64 KB
    if (v && v[0] == 'c') return make_cfg<2, 4, 2, 1, 32, 4, EPL>();      // 128x128, 8 waves, 4 x 32 KB
    if (EPL >= 64) return make_cfg<2, 4, 2, 1, 32, 4, EPL>();             // k > 1024: keep registers for the select
    return make_cfg<4, 2, 2, 2, 32, 3, (EPL >= 64 ? 6 : EPL + 2)>();      // 256x128, 8 waves, 3 x 48 KB
}

int pick_cfg(long long nq, int k, ScanCfg* out) {
    if (k <= 128) *out = pick_shape<6>(nq);
    else if (k <= 512) *out = pick_shape<18>(nq);
    else if (k <= 1024) *out = pick_shape<34>(nq);
    else if (k <= 2048) *out = pick_shape<64>(nq);
    else return set_err(OTTER_EINVAL, "search: k=%d exceeds OTTER_MAX_K=%d", k, OTTER_MAX_K);
    return OTTER_OK;
}

int next_pow2(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace

struct otter_index {
    int d = 0, ld = 0, dtype = OTTER_F32, device = 0, ncu = 256;
    long long n = 0, cap_rows = 0, id_base = 0;
    float* x = nullptr;
    otter::DevBuf qpad, cand, cnt, lvl[2], lvlcnt[2], smp, smpcnt, thr0, io_q, io_D, io_I;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending, ev_free;
};

namespace {

int index_reserve_rows(otter_index* ix, long long need_rows, hipStream_t st) {
    if (need_rows <= ix->cap_rows) return OTTER_OK;
    // capacity is kept a multiple of the largest scan tile (256 rows): the kernel reads whole tiles
    // and masks rows >= n, so the tail of the last tile must be addressable
    need_rows = round_up64(need_rows, 256);
    long long want = ix->cap_rows == 0 ? need_rows : std::max(need_rows, ix->cap_rows + ix->cap_rows / 2);
    want = round_up64(want, 256);
    float* nx = nullptr;
    hipError_t e = hipMalloc((void**)&nx, (size_t)want * ix->ld * sizeof(float));
    if (e != hipSuccess && want > need_rows) {
        (void)hipGetLastError();
        want = need_rows;
        e = hipMalloc((void**)&nx, (size_t)want * ix->ld * sizeof(float));
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return set_err(OTTER_ENOMEM, "add: hipMalloc of %lld rows x %d floats failed: %s", want, ix->ld,
                       hipGetErrorString(e));
    }
    if (ix->n > 0) {
        OTTER_HIP_TRY(hipMemcpyAsync(nx, ix->x, (size_t)ix->n * ix->ld * sizeof(float), hipMemcpyDeviceToDevice, st));
        OTTER_HIP_TRY(hipStreamSynchronize(st));
    }
    if (ix->x) OTTER_HIP_TRY(hipFree(ix->x));
    ix->x = nx;
    ix->cap_rows = want;
    return OTTER_OK;
}

// Merge the per-strip candidate lists of one scan, level by level (one workgroup sorts at most
// MAX_SELECT composites).  final: write (D, I), optionally merging one more sorted list (ex);
// otherwise leave ONE sorted list of <= k composites per query in list_out and the score key of
// each query's k-th best in thr_out.
int run_merge(otter_index* ix, const u64* in, const u32* in_cnt, int cap_in, int S_in, long long nq, long long nq_pad,
              int k, bool final, float* D, long long* I, const u64* ex, const u32* ex_cnt, u64* list_out,
              u32* list_cnt, u32* thr_out, hipStream_t st) {
    const int fan = std::min(MERGE_MAX_STRIPS, std::max(2, MAX_SELECT / k - (ex ? 1 : 0)));
    int pp = 0, rc;
    for (;;) {
        const int S_out = (S_in + fan - 1) / fan;
        const bool last = S_out == 1;
        MergeArgs ma;
        ma.in = in;
        ma.in_cnt = in_cnt;
        ma.out = nullptr;
        ma.out_cnt = nullptr;
        ma.D = D;
        ma.I = I;
        ma.id_base = ix->id_base;
        ma.cap_in = cap_in;
        ma.S_in = S_in;
        ma.fan = fan;
        ma.nq_pad = (int)nq_pad;
        ma.k = k;
        ma.final_level = (last && final) ? 1 : 0;
        ma.ex = ma.final_level ? ex : nullptr;
        ma.ex_cnt = ma.final_level ? ex_cnt : nullptr;
        ma.thr_out = nullptr;
        ma.npad = next_pow2(std::max(2, (std::min(S_in, fan) + (ma.ex ? 1 : 0)) * k));
        ma.npad = std::max(ma.npad, next_pow2(k));
        if (!ma.final_level) {
            if (last) {
                ma.out = list_out;
                ma.out_cnt = list_cnt;
                ma.thr_out = thr_out;
            } else {
                if ((rc = ix->lvl[pp].reserve((size_t)S_out * nq_pad * k * sizeof(u64)))) return rc;
                if ((rc = ix->lvlcnt[pp].reserve((size_t)S_out * nq_pad * sizeof(u32)))) return rc;
                ma.out = (u64*)ix->lvl[pp].p;
                ma.out_cnt = (u32*)ix->lvlcnt[pp].p;
            }
        }
        const size_t lds = (size_t)ma.npad * sizeof(u64) + (size_t)(MERGE_MAX_STRIPS + 4 + 32 + 4) * sizeof(int);
        static bool attr_set[64] = {};
        if (ix->device < 64 && !attr_set[ix->device]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(merge_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      MAX_SELECT * 2 * (int)sizeof(u64) + (MERGE_MAX_STRIPS + 40) * (int)sizeof(int));
            attr_set[ix->device] = true;
        }
        hipLaunchKernelGGL(merge_kernel, dim3((unsigned)nq, (unsigned)S_out), dim3(MERGE_THREADS), lds, st, ma);
        OTTER_HIP_TRY(hipGetLastError());
        if (last) break;
        in = ma.out;
        in_cnt = ma.out_cnt;
        cap_in = k;
        S_in = S_out;
        pp ^= 1;
    }
    return OTTER_OK;
}

// one scan + merge over queries [qd, qd + nq) (device pointers), results to device D/I
int search_chunk(otter_index* ix, const float* qd, long long nq, int k, float* D, long long* I, hipStream_t st) {
    ScanCfg cfg;
    int rc = pick_cfg(nq, k, &cfg);
    if (rc) return rc;
    const long long nq_pad = round_up64(nq, cfg.bq);
    const int qtiles = (int)(nq_pad / cfg.bq);
    const int all_tiles = (int)((ix->n + cfg.bm - 1) / cfg.bm);

    // Sample pass (small query batches on large corpora): every strip otherwise re-learns its own
    // top-k threshold, appending ~k ln(rows/k) candidates per query and strip.  Scanning the first
    // n_s rows first and merging them gives each query the key of its k-th best among rows
    // [0, n_s); the main scan over [n_s, n) starts every strip from that threshold.  Exact: the
    // sample rows have the smallest ids, so a later row that only ties the k-th key ranks behind it.
    static const char* no_sample = getenv("OTTER_NO_SAMPLE");
    const long long n_s = round_up64(std::max<long long>(16384, 64ll * k), 256);
    const bool sample = !no_sample && nq <= 64 && ix->n >= 12 * n_s;
    const int sample_tiles = sample ? (int)(n_s / cfg.bm) : 0;

    const int main_tiles = all_tiles - sample_tiles;
    int strips = std::max(1, ix->ncu / qtiles);
    strips = std::min(strips, main_tiles);
    // few enough sample strips for a one-level merge (k <= 128 on this path's sizes: fan >= 64)
    int strips_s = sample ? std::min(std::min(std::max(1, ix->ncu / qtiles), sample_tiles), 64) : 0;

    if ((rc = ix->qpad.reserve((size_t)nq_pad * ix->ld * sizeof(float)))) return rc;
    if ((rc = ix->cand.reserve((size_t)std::max(strips, strips_s) * nq_pad * cfg.cap * sizeof(u64)))) return rc;
    if ((rc = ix->cnt.reserve((size_t)std::max(strips, strips_s) * nq_pad * sizeof(u32)))) return rc;
    if (sample) {
        if ((rc = ix->smp.reserve((size_t)nq_pad * k * sizeof(u64)))) return rc;
        if ((rc = ix->smpcnt.reserve((size_t)nq_pad * sizeof(u32)))) return rc;
        if ((rc = ix->thr0.reserve((size_t)nq_pad * sizeof(u32)))) return rc;
    }

    {
        const long long total = nq_pad * ix->ld;
        const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(pad_rows_kernel, dim3(grid), dim3(256), 0, st, qd, (float*)ix->qpad.p, nq, nq_pad, ix->d,
                           ix->ld);
    }

    ScanArgs sa;
    sa.X = ix->x;
    sa.Q = (const float*)ix->qpad.p;
    sa.cand = (u64*)ix->cand.p;
    sa.cnt = (u32*)ix->cnt.p;
    sa.n = ix->n;
    sa.ld = ix->ld;
    sa.nq = (int)nq;
    sa.nq_pad = (int)nq_pad;
    sa.k = k;
    sa.qtiles = qtiles;
    sa.dbg = nullptr;
    static const char* ord = getenv("OTTER_SCAN_ORDER");
    sa.order = ord ? (ord[0] == '1') : 1;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    const bool timed = ix->timing && ix->ev_pending.size() < 4096;
    if (timed) {
        if (!ix->ev_free.empty()) {
            ev = ix->ev_free.back();
            ix->ev_free.pop_back();
        } else {
            OTTER_HIP_TRY(hipEventCreate(&ev.first));
            OTTER_HIP_TRY(hipEventCreate(&ev.second));
        }
        OTTER_HIP_TRY(hipEventRecord(ev.first, st));
    }
    if (sample) {
        sa.strips = strips_s;
        sa.total_tiles = sample_tiles;
        sa.tile_begin = 0;
        sa.thr0 = nullptr;
        cfg.launch(sa, qtiles * strips_s, cfg.lds, st);
        OTTER_HIP_TRY(hipGetLastError());
        if ((rc = run_merge(ix, (const u64*)ix->cand.p, (const u32*)ix->cnt.p, cfg.cap, strips_s, nq, nq_pad, k, false,
                            nullptr, nullptr, nullptr, nullptr, (u64*)ix->smp.p, (u32*)ix->smpcnt.p, (u32*)ix->thr0.p,
                            st)))
            return rc;
    }
    sa.strips = strips;
    sa.total_tiles = main_tiles;
    sa.tile_begin = sample_tiles;
    sa.thr0 = sample ? (const u32*)ix->thr0.p : nullptr;
#ifdef SCAN_DIAG
    static otter::DevBuf dbgbuf;
    const size_t dbg_n = (size_t)qtiles * strips * 8 * 4;
    if ((rc = dbgbuf.reserve(dbg_n * sizeof(u64)))) return rc;
    OTTER_HIP_TRY(hipMemsetAsync(dbgbuf.p, 0, dbg_n * sizeof(u64), st));
    sa.dbg = (u64*)dbgbuf.p;
#endif
    cfg.launch(sa, qtiles * strips, cfg.lds, st);
    OTTER_HIP_TRY(hipGetLastError());
    if (timed) {
        OTTER_HIP_TRY(hipEventRecord(ev.second, st));
        ix->ev_pending.push_back(ev);
    }
#ifdef SCAN_DIAG
    {
        std::vector<u64> hbuf(dbg_n);
        OTTER_HIP_TRY(hipStreamSynchronize(st));
        OTTER_HIP_TRY(hipMemcpy(hbuf.data(), dbgbuf.p, dbg_n * sizeof(u64), hipMemcpyDeviceToHost));
        double tot = 0, wait = 0, epi = 0, stages = 0, nwv = 0, mx = 0;
        for (size_t i = 0; i < dbg_n; i += 4) {
            if (hbuf[i] == 0) continue;
            tot += hbuf[i]; wait += hbuf[i + 1]; epi += hbuf[i + 2]; stages += hbuf[i + 3]; nwv += 1;
            mx = std::max(mx, (double)hbuf[i]);
        }
        if (nwv > 0)
            fprintf(stderr, "[scan diag] waves=%.0f mean cycles/wave: total %.0f (max %.0f) stage-wait %.0f epilogue %.0f; "
                            "per stage: total %.0f wait %.0f\n",
                    nwv, tot / nwv, mx, wait / nwv, epi / nwv, tot / stages, wait / stages);
    }
#endif
    return run_merge(ix, (const u64*)ix->cand.p, (const u32*)ix->cnt.p, cfg.cap, strips, nq, nq_pad, k, true, D, I,
                     sample ? (const u64*)ix->smp.p : nullptr, sample ? (const u32*)ix->smpcnt.p : nullptr, nullptr,
                     nullptr, nullptr, st);
}

// n == 0: every result is padding
__global__ void fill_empty_kernel(float* D, long long* I, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        D[i] = -FLT_MAX;
        I[i] = -1;
    }
}

}  // namespace

extern "C" {

int otter_index_create(int d, int dtype, int device, otter_index** out) {
    if (!out) return set_err(OTTER_EINVAL, "index_create: out is NULL");
    *out = nullptr;
    if (d <= 0 || d > (1 << 20)) return set_err(OTTER_EINVAL, "index_create: bad dimension d=%d", d);
    if (dtype != OTTER_F32)
        return set_err(OTTER_ENOTIMPL, "index_create: only OTTER_F32 corpus storage is built (dtype=%d)", dtype);
    int ndev = 0;
    OTTER_HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return set_err(OTTER_EINVAL, "index_create: device %d out of range (%d visible)", device, ndev);
    hipDeviceProp_t prop;
    OTTER_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(OTTER_ESTATE, "index_create: device %d is %s; this library is built for gfx950 only", device,
                       prop.gcnArchName);
    otter_index* ix = new (std::nothrow) otter_index();
    if (!ix) return set_err(OTTER_ENOMEM, "index_create: host allocation failed");
    ix->d = d;
    ix->ld = round_up(d, LD_ALIGN);
    ix->dtype = dtype;
    ix->device = device;
    ix->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = ix;
    return OTTER_OK;
}

int otter_index_add(otter_index* ix, const void* x, int64_t n, int x_is_device, void* stream) {
    if (!ix) return set_err(OTTER_EINVAL, "add: index is NULL");
    if (n < 0) return set_err(OTTER_EINVAL, "add: n=%lld < 0", (long long)n);
    if (n == 0) return OTTER_OK;
    if (!x) return set_err(OTTER_EINVAL, "add: x is NULL");
    if (ix->n + n > 0xFFFFFFF0ll)
        return set_err(OTTER_EINVAL, "add: %lld rows exceed the 32-bit local id space of one index; shard the corpus",
                       (long long)(ix->n + n));
    DeviceGuard g(ix->device);
    if (!g.ok) return set_err(OTTER_EHIP, "add: hipSetDevice(%d) failed", ix->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = index_reserve_rows(ix, ix->n + n, st);
    if (rc) return rc;
    float* dst = ix->x + (size_t)ix->n * ix->ld;
    const hipMemcpyKind kind = x_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (ix->ld == ix->d) {
        OTTER_HIP_TRY(hipMemcpyAsync(dst, x, (size_t)n * ix->d * sizeof(float), kind, st));
    } else {
        OTTER_HIP_TRY(hipMemsetAsync(dst, 0, (size_t)n * ix->ld * sizeof(float), st));
        OTTER_HIP_TRY(hipMemcpy2DAsync(dst, (size_t)ix->ld * sizeof(float), x, (size_t)ix->d * sizeof(float),
                                       (size_t)ix->d * sizeof(float), (size_t)n, kind, st));
    }
    if (!x_is_device) OTTER_HIP_TRY(hipStreamSynchronize(st));
    ix->n += n;
    return OTTER_OK;
}

int otter_index_search(otter_index* ix, const void* q, int64_t nq, int k, float* D, int64_t* I, int io_is_device,
                       void* stream) {
    if (!ix) return set_err(OTTER_EINVAL, "search: index is NULL");
    if (nq < 0) return set_err(OTTER_EINVAL, "search: nq=%lld < 0", (long long)nq);
    if (k < 1 || k > OTTER_MAX_K) return set_err(OTTER_EINVAL, "search: k=%d outside [1, %d]", k, OTTER_MAX_K);
    if (nq == 0) return OTTER_OK;
    if (!q || !D || !I) return set_err(OTTER_EINVAL, "search: q, D and I must be non-NULL");
    DeviceGuard g(ix->device);
    if (!g.ok) return set_err(OTTER_EHIP, "search: hipSetDevice(%d) failed", ix->device);
    hipStream_t st = (hipStream_t)stream;
    int rc;

    const float* qd = (const float*)q;
    float* Dd = D;
    long long* Id = (long long*)I;
    if (!io_is_device) {
        if ((rc = ix->io_q.reserve((size_t)nq * ix->d * sizeof(float)))) return rc;
        if ((rc = ix->io_D.reserve((size_t)nq * k * sizeof(float)))) return rc;
        if ((rc = ix->io_I.reserve((size_t)nq * k * sizeof(long long)))) return rc;
        OTTER_HIP_TRY(hipMemcpyAsync(ix->io_q.p, q, (size_t)nq * ix->d * sizeof(float), hipMemcpyHostToDevice, st));
        qd = (const float*)ix->io_q.p;
        Dd = (float*)ix->io_D.p;
        Id = (long long*)ix->io_I.p;
    }

    if (ix->n == 0) {
        const long long total = (long long)nq * k;
        hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)),
                           dim3(256), 0, st, Dd, Id, total);
        OTTER_HIP_TRY(hipGetLastError());
    } else {
        for (long long q0 = 0; q0 < nq; q0 += QCHUNK) {
            const long long c = std::min<long long>(QCHUNK, nq - q0);
            rc = search_chunk(ix, qd + (size_t)q0 * ix->d, c, k, Dd + (size_t)q0 * k, Id + (size_t)q0 * k, st);
            if (rc) return rc;
        }
    }

    if (!io_is_device) {
        OTTER_HIP_TRY(hipMemcpyAsync(D, Dd, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
        OTTER_HIP_TRY(hipMemcpyAsync(I, Id, (size_t)nq * k * sizeof(long long), hipMemcpyDeviceToHost, st));
        OTTER_HIP_TRY(hipStreamSynchronize(st));
    }
    return OTTER_OK;
}

int64_t otter_index_ntotal(const otter_index* ix) { return ix ? ix->n : -1; }
int otter_index_d(const otter_index* ix) { return ix ? ix->d : -1; }

int otter_index_reset(otter_index* ix) {
    if (!ix) return set_err(OTTER_EINVAL, "reset: index is NULL");
    ix->n = 0;
    return OTTER_OK;
}

int otter_index_set_id_base(otter_index* ix, int64_t id_base) {
    if (!ix) return set_err(OTTER_EINVAL, "set_id_base: index is NULL");
    if (id_base < 0) return set_err(OTTER_EINVAL, "set_id_base: id_base=%lld < 0", (long long)id_base);
    ix->id_base = id_base;
    return OTTER_OK;
}

int otter_index_set_timing(otter_index* ix, int enable) {
    if (!ix) return set_err(OTTER_EINVAL, "set_timing: index is NULL");
    ix->timing = enable != 0;
    return OTTER_OK;
}

int otter_index_timing(otter_index* ix, double* scan_ms, int64_t* scan_launches) {
    if (!ix || !scan_ms || !scan_launches) return set_err(OTTER_EINVAL, "timing: NULL argument");
    DeviceGuard g(ix->device);
    double total = 0.0;
    for (auto& ev : ix->ev_pending) {
        OTTER_HIP_TRY(hipEventSynchronize(ev.second));
        float ms = 0.f;
        OTTER_HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
        total += ms;
        ix->ev_free.push_back(ev);
    }
    *scan_ms = total;
    *scan_launches = (int64_t)ix->ev_pending.size();
    ix->ev_pending.clear();
    return OTTER_OK;
}

void otter_index_destroy(otter_index* ix) {
    if (!ix) return;
    DeviceGuard g(ix->device);
    for (auto& ev : ix->ev_pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : ix->ev_free) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (ix->x) (void)hipFree(ix->x);
    ix->qpad.release();
    ix->cand.release();
    ix->cnt.release();
    for (int i = 0; i < 2; ++i) { ix->lvl[i].release(); ix->lvlcnt[i].release(); }
    ix->smp.release();
    ix->smpcnt.release();
    ix->thr0.release();
    ix->io_q.release();
    ix->io_D.release();
    ix->io_I.release();
    delete ix;
}

int otter_merge_topk(int device, const float* Dc, const int64_t* Ic, int G, int64_t nq, int k, float* D, int64_t* I,
                     void* stream) {
    if (G < 1 || k < 1 || nq < 0) return set_err(OTTER_EINVAL, "merge_topk: bad G=%d k=%d nq=%lld", G, k, (long long)nq);
    if ((long long)G * k > MAX_SELECT)
        return set_err(OTTER_EINVAL, "merge_topk: G*k=%lld exceeds %d; merge hierarchically", (long long)G * k,
                       MAX_SELECT);
    if (nq == 0) return OTTER_OK;
    if (!Dc || !Ic || !D || !I) return set_err(OTTER_EINVAL, "merge_topk: NULL buffer");
    DeviceGuard g(device);
    if (!g.ok) return set_err(OTTER_EHIP, "merge_topk: hipSetDevice(%d) failed", device);
    PairMergeArgs a;
    a.Dc = Dc;
    a.Ic = (const long long*)Ic;
    a.D = D;
    a.I = (long long*)I;
    a.nq = nq;
    a.G = G;
    a.k = k;
    a.npad = next_pow2(std::max(2, G * k));
    const size_t lds = (size_t)a.npad * 12;
    static bool attr_set[64] = {};
    if (device < 64 && !attr_set[device]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pair_merge_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, MAX_SELECT * 12);
        attr_set[device] = true;
    }
    hipLaunchKernelGGL(pair_merge_kernel, dim3((unsigned)nq), dim3(NTHREADS), lds, (hipStream_t)stream, a);
    OTTER_HIP_TRY(hipGetLastError());
    return OTTER_OK;
}

}  // extern "C"
