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
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "otter_common.h"

namespace {

using namespace otter;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
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
// the corpus stream of the exact kernels: every byte is read once per search, so the loads carry the non-temporal
// policy (measured on the stream kernel, 1 query x 1M x 768: 0.536 -> 0.482 ms; SCAN_NO_NT builds without it)
__device__ __forceinline__ void dma16s_stream(const void* sbase, u32 voff, u32 lds_dst) {
#ifndef SCAN_NO_NT
    u32 keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
#else
    dma16s(sbase, voff, lds_dst);
#endif
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
    int q_rows;         // rows that exist behind Q (nq_pad for a padded copy; nq when the caller's rows are read in place)
    int skip_final;     // 1: leave the strip's candidates uncompacted (<= CAP per query: one-tile strips); the merge selects
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
// The score keys (high words) are selected first with 32-bit compares -- they are almost always
// distinct, so the rounds end there; only entries that tie the k-th key go on to the row words.
template <int EPL>
__device__ __forceinline__ void wave_compact_load(u64 (&x)[EPL], const u64* base, int n, int lane) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        int idx = e * 64 + lane;
        // agent-scope load: bypass this CU's vector L1 (entries were stored by other waves)
        x[e] = idx < n ? __hip_atomic_load(base + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
}

template <int EPL>
__device__ __forceinline__ u64 wave_compact_select(u64 (&x)[EPL], u64* base, int k, int lane) {
    u32 Th = 0;
    bool exact = false;          // exactly k entries have a key >= Th
    for (int b = 31; b >= 0; --b) {
        const u32 c = Th | (1u << b);
        int cntv = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) cntv += __popcll(__ballot((u32)(x[e] >> 32) >= c));
        if (cntv >= k) Th = c;
        if (cntv == k) { exact = true; break; }
    }
    u32 Tl = 0;
    if (!exact) {
        // more than k entries reach key Th: all with a larger key stay, the ties compete on the row word
        int above = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) above += __popcll(__ballot((u32)(x[e] >> 32) > Th));
        const int need = k - above;
        for (int b = 31; b >= 0; --b) {
            const u32 c = Tl | (1u << b);
            int cntv = 0;
#pragma unroll
            for (int e = 0; e < EPL; ++e) cntv += __popcll(__ballot((u32)(x[e] >> 32) == Th && (u32)x[e] >= c));
            if (cntv >= need) Tl = c;
            if (cntv == need) break;
        }
    }
    const u64 T = ((u64)Th << 32) | Tl;
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

template <int EPL>
__device__ __forceinline__ u64 wave_compact(u64* base, int n, int k, int lane) {
    u64 x[EPL];
    wave_compact_load<EPL>(x, base, n, lane);
    return wave_compact_select<EPL>(x, base, k, lane);
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
        // query side: a row past the batch reads the batch's last row instead (its scores are never accepted)
        const int rs = t < A_INSTR ? r : min(r, max(a.q_rows - 1 - q0, 0));
        voff[t] = ((u32)rs * (u32)a.ld + (u32)((dma_p ^ swz(r)) << 2)) * 4u;
#ifdef SCAN_ABL_BLOCKED
        // timing-only ablation: read every stage of the corpus as ONE contiguous block (results are garbage)
        if (t < A_INSTR) voff[t] = ((u32)r * (u32)BK + (u32)((dma_p ^ swz(r)) << 2)) * 4u;
#endif
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
#ifdef SCAN_ABL_BLOCKED
            pr_gA = a.X + ((size_t)it_tile * BM * a.ld + (size_t)it_ks * BK * BM);
#else
            pr_gA = a.X + ((size_t)it_tile * BM * a.ld + (size_t)it_ks * BK);
#endif
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
            if (t < A_INSTR) dma16s_stream(pr_gA, voff[t], pr_lds + (t * NW + wave) * 1024);
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
#ifdef SCAN_ABL_NOMFMA
                        // timing-only ablation: the fragments are read, one product instead of four (results are garbage)
                        asm volatile("" ::"v"(af[cb][m].x), "v"(af[cb][m].y), "v"(af[cb][m].z), "v"(af[cb][m].w), "v"(bf[cb][q].x),
                                     "v"(bf[cb][q].y), "v"(bf[cb][q].z), "v"(bf[cb][q].w));
                        if (j == 0) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][m].x, bf[cb][q].x, acc[m][q], 0, 0, 0);
                        continue;
#endif
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
    if (a.skip_final) {
        wait_vmcnt<0>();
        wg_barrier();
    } else {
        compact_pass(true);
    }
    for (int i = tid; i < BQ; i += NT) a.cnt[(size_t)strip * a.nq_pad + q0 + i] = min(sCnt[i], (u32)CAP);
}

// ------------------------------------------------------------------ stream kernel (at most 32 queries)
// The reference's real call pattern is ONE query per search() call (scripts/eval_ottqa_retrieval.py:126-129):
// N*d*4 bytes per call, HBM-bound up to about 32 queries.  A 32x32x2 MFMA computes 32 query columns whatever nq
// is (55 % of a stage's time at 1 query), so this regime has a kernel of its own:
//   lane = corpus row.  A wave owns 64 consecutive rows of a 256-row tile and streams ITS OWN rows through
//   its own LDS ring (slots of [8 KiB rows | the queries' 32 floats], LDS-DMA with the non-temporal policy --
//   every corpus byte is read once per search --, counted vmcnt): no workgroup barrier in the main loop.
//   Scores: v_mfma_f32_4x4x1_16b_f32 -- 16 blocks of (4 rows) x (4 queries) x (ONE k) per instruction, i.e. a
//   plain fma per output and instruction, issued in scan_kernel's k order (0,4,1,5,2,6,3,7 per 8): bit-identical
//   to the 32x32x2 kernels and to the oracle's fmaf chain.  G = ceil(nq / 4) instructions per k: the matrix-core
//   time scales with nq (1/8 of the 32x32x2 tile's at <= 4 queries).
//   No sample pass: after its FIRST tile every wave publishes, per query, the best of its 64 rows (a composite);
//   the workgroup that arrives last selects each query's k-th best of these <= 1024 block maxima -- a valid lower
//   bound of the k-th best score, since it is the k-th best of a subset of rows -- and publishes it; the other
//   workgroups hold the scores of their first tiles in registers until the thresholds are there (polled at tile
//   boundaries, never waited for: after ST_HOLD held tiles a workgroup falls back to learning its own threshold).
//   Hand-offs follow MI355X_MICROARCH.md "Valid forms": sc1 stores, every storing wave's vmcnt(0), workgroup
//   barrier, one agent-scope atomic add per workgroup; consumers poll with an sc1 load and read the payload with
//   sc1 loads.  Acceptance is key >= threshold (rows of other strips may tie it).
struct StreamArgs {
    const float* X;      // [n][ld]
    const float* Q;      // [q_rows][ld]
    u64* cand;           // [S][4 G][CAP]
    u32* cnt;            // [S][4 G]
    long long n;
    int ld, nq, k;
    int strips;          // workgroups
    int total_tiles;     // 256-row tiles
    int q_rows;          // rows that exist behind Q
    int seed;            // 1: threshold exchange on
    u64* bmax;           // [4 G][4 * strips] block maxima of every wave's first tile
    u32* sync;           // [0] arrivals, [1] ready (stream16 / stream64: number of queries selected), [2..33] thresholds (keys;
                         // stream64_kernel: [2..65]), [35] next query to select (stream16_kernel; stream64_kernel: [67]);
                         // ST_SYNC_WORDS words, zero at launch, zeroed again by the merge
    float* hold = nullptr;   // stream64_kernel: [strips][S64_HOLD][16 accumulators][256 lanes] float4, tiles awaiting thresholds
};

constexpr int ST_BM = 256, ST_SYNC_WORDS = 128;   // (see StreamArgs::sync)
// stream64_kernel: thresholds of up to 64 queries at [2..65], claim counter at [67]; held tiles; LDS image of the queries
#ifndef S64_HOLD_TILES
#define S64_HOLD_TILES 3
#endif
constexpr int S64_CLAIM = 67, S64_HOLD = S64_HOLD_TILES, S64_NQP = 64;
__host__ __device__ constexpr u32 s64_pitch(int ld) { return (u32)ld * 2u + (((ld / 64) & 1) ? 0u : 128u); }   // odd multiple of 128 B
constexpr int st_qinstr(int G, int BK) { return (4 * G * BK * 4 + 1023) / 1024; }              // LDS-DMA instructions for the queries
constexpr int st_slot(int G, int BK) { return 64 * BK * 4 + st_qinstr(G, BK) * 1024; }        // per wave and stage
constexpr int st_ips(int G, int BK) { return BK / 4 + st_qinstr(G, BK); }                     // LDS-DMA instructions per wave and stage

// k-th largest of the n <= 64 * EPL composites in x (unique, zero = empty): radix select by ballots
// key of the k-th largest composite (high words only; equal keys may straddle the cut, which a ">=" filter
// tolerates): 32 ballot rounds
template <int EPL>
__device__ __forceinline__ u32 wave_kth_key(const u64 (&x)[EPL], int k) {
    u32 T = 0u;
    for (int b = 31; b >= 0; --b) {
        const u32 c = T | (1u << b);
        int cntv = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) cntv += __popcll(__ballot((u32)(x[e] >> 32) >= c));
        if (cntv >= k) T = c;
        if (cntv == k) break;        // (no further bit can be set without dropping below k)
    }
    return T;
}

template <int EPL>
__device__ __forceinline__ u64 wave_kth(const u64 (&x)[EPL], int k) {
    u64 T = 0ull;
    for (int b = 63; b >= 0; --b) {
        const u64 c = T | (1ull << b);
        int cntv = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) cntv += __popcll(__ballot(x[e] >= c));
        if (cntv >= k) T = c;
        if (cntv == k) {
            // exactly k composites are >= c: the k-th largest is their minimum
            u64 mn = ~0ull;
#pragma unroll
            for (int e = 0; e < EPL; ++e) mn = (x[e] >= c && x[e] < mn) ? x[e] : mn;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const u64 v = __shfl_xor(mn, o);
                mn = v < mn ? v : mn;
            }
            return mn;
        }
    }
    return T;
}

template <int G, int EPL, int ST_BK, int ST_NSTAGE>
__global__ __launch_bounds__(256) void stream_kernel(StreamArgs a) {
    constexpr int NQP = 4 * G;                 // query columns computed (padded to whole 4x4 blocks)
    constexpr int ST_HOLD = G <= 2 ? 6 : 4;    // tiles whose scores wait in registers for the thresholds (4 G per tile)
    constexpr int CAP = EPL * 64;
    constexpr int QI = st_qinstr(G, ST_BK), ST_SLOT = st_slot(G, ST_BK), ST_IPS = st_ips(G, ST_BK);
    constexpr int CH = ST_BK / 4;              // 16-byte chunks per row of a stage
    constexpr int RPI = 64 / CH;               // rows per LDS-DMA instruction
    constexpr int ROWB = ST_BK * 4;            // bytes per row of a stage
    static_assert(ST_BK == 32 || ST_BK == 64, "stage depth");
    static_assert(CAP - ST_BM >= 128, "candidate buffer too small");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32* sCnt = reinterpret_cast<u32*>(smem + 4 * ST_NSTAGE * ST_SLOT);
    u32* sThr = sCnt + 32;
    u32* sFlag = sThr + 32;       // [0] compaction wanted, [1] thresholds ready, [2] this workgroup selects
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = blockIdx.x;
    const int t0 = (int)(((long long)strip * a.total_tiles) / a.strips);
    const int t1 = (int)(((long long)(strip + 1) * a.total_tiles) / a.strips);
    const int KS = a.ld / ST_BK;
    if (tid < 32) {
        sCnt[tid] = 0;
        sThr[tid] = 0u;            // accept everything (key >= 0) until a threshold is known
    }
    if (tid < 3) sFlag[tid] = 0;
    u64* candBase = a.cand + (size_t)strip * NQP * CAP;

    // ---- producer: this wave's 64 rows x BK floats (1 KiB per instruction) + the queries' BK floats
    // LDS image: [row][CH chunks of 16 B], chunk position p of row r holds global chunk p ^ swz(r), swz chosen so
    // that the rows one ds_read_b128 lane group touches fall on distinct 16-byte slots of the 256-byte bank row
    auto swz = [](int r) { return CH == 16 ? (r & 15) : ((r >> 1) & 7); };
    const u32 lds0 = (u32)(uintptr_t)(AS3 char*)smem + (u32)wave * (ST_NSTAGE * ST_SLOT);
    u32 voffA[CH], voffQ[QI];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int r = RPI * i + lane / CH;
        voffA[i] = ((u32)(wave * 64 + r) * (u32)a.ld + (u32)(((lane % CH) ^ swz(r)) << 2)) * 4u;
    }
#pragma unroll
    for (int i = 0; i < QI; ++i) {
        const int r = RPI * i + lane / CH;       // a row past the batch reads its last row (never accepted)
        voffQ[i] = ((u32)min(r, max(a.q_rows - 1, 0)) * (u32)a.ld + (u32)((lane % CH) << 2)) * 4u;
    }
    const long long total_stages = (long long)(t1 - t0) * KS;
    int it_tile = t0, it_ks = 0;
    long long it_f = 0;
    auto issue_stage = [&]() {
        if (it_f < total_stages) {
            const float* gA = a.X + ((size_t)it_tile * ST_BM * a.ld + (size_t)it_ks * ST_BK);
            const float* gQ = a.Q + (size_t)it_ks * ST_BK;
            const u32 l = lds0 + (u32)(it_f % ST_NSTAGE) * ST_SLOT;
#pragma unroll
            for (int i = 0; i < CH; ++i) dma16s_stream(gA, voffA[i], l + i * 1024);
#pragma unroll
            for (int i = 0; i < QI; ++i) dma16s(gQ, voffQ[i], l + CH * 1024 + i * 1024);
            ++it_f;
            if (++it_ks == KS) { it_ks = 0; ++it_tile; }
        }
    };

    // ---- candidate path.  Lane l holds, per query group g, the scores of rows 4 (l >> 2) + i (i = 0..3) of this
    // wave's 64 for query 4 g + (l & 3)
    const int rsub = wave * 64 + 4 * (lane >> 2);
    auto append_tile = [&](const f32x4 (&sc)[G], int tile) {
        const long long row0 = (long long)tile * ST_BM + rsub;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int q = 4 * g + (lane & 3);
            const u32 thr = sThr[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32 key = f2key(sc[g][i]);
                if (q < a.nq && key >= thr && row0 + i < a.n) {
                    const u32 pos = atomicAdd(&sCnt[q], 1u);
                    if (pos < (u32)CAP)
                        candBase[(u32)q * (u32)CAP + pos] = ((u64)key << 32) | (u64)(0xFFFFFFFFu - (u32)(row0 + i));
                    if (pos + 1 > (u32)(CAP - ST_BM)) sFlag[0] = 1;
                }
            }
        }
    };
    auto compact_pass = [&](bool final_pass) {      // workgroup-wide; every wave calls it at the same point
        wait_vmcnt<0>();
        wg_barrier();
        for (int q = wave; q < NQP; q += 4) {
            const int n = (int)min(sCnt[q], (u32)CAP);
            const bool need = final_pass ? (n > a.k) : (n > CAP - ST_BM);
            if (need) {
                const u64 T = wave_compact<EPL>(candBase + (size_t)q * CAP, n, a.k, lane);
                if (lane == 0) {
                    sCnt[q] = (u32)a.k;
                    sThr[q] = max(sThr[q], (u32)(T >> 32));
                }
            }
        }
        if (tid == 0) sFlag[0] = 0;
        wait_vmcnt<0>();
        wg_barrier();
    };

    f32x4 acc[G];
    f32x4 held[ST_HOLD][G];
    int nheld = 0, held_tile0 = t0;
    bool have_thr = a.seed == 0;     // no exchange: thresholds are learned locally from the start
    bool got_global = a.seed == 0;   // the exchanged thresholds have been taken over (or there are none to wait for)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s2 = 0; s2 < ST_NSTAGE - 1; ++s2) issue_stage();

    const u32 rowoff = (u32)lane * (u32)ROWB;
    const u32 rswz = (u32)swz(lane);
    const u32 qoff = (u32)(lane & 3) * (u32)ROWB;
    long long f = 0;
    for (int tile = t0; tile < t1; ++tile) {
        for (int ks = 0; ks < KS; ++ks, ++f) {
            // stage f (this wave's own loads) has landed once at most NSTAGE-2 younger stages are outstanding
            if (total_stages - 1 - f >= ST_NSTAGE - 2) wait_vmcnt<(ST_NSTAGE - 2) * ST_IPS>();
            else wait_vmcnt<0>();
            // the slot of stage f-1 was read by this wave only, and those reads have returned (their values fed
            // the products above): refill it with stage f + NSTAGE - 1
            wait_lgkm0();
            issue_stage();
            const char* sA = smem + wave * (ST_NSTAGE * ST_SLOT) + (int)(f % ST_NSTAGE) * ST_SLOT;
            const char* sQ = sA + CH * 1024;
#pragma unroll
            for (int j = 0; j < ST_BK / 8; ++j) {
                const float4 x0 = *reinterpret_cast<const float4*>(sA + rowoff + (((u32)(2 * j) ^ rswz) << 4));
                const float4 x1 = *reinterpret_cast<const float4*>(sA + rowoff + (((u32)(2 * j + 1) ^ rswz) << 4));
                // lanes l, l + 4, ... read the same query row: four addresses per instruction
                float4 q0[G], q1[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    q0[g] = *reinterpret_cast<const float4*>(sQ + g * 4 * ROWB + qoff + (2 * j) * 16);
                    q1[g] = *reinterpret_cast<const float4*>(sQ + g * 4 * ROWB + qoff + (2 * j + 1) * 16);
                }
                // one k per instruction, the G query groups interleaved (independent accumulators back to back)
#define ST_STEP(XA, QA)                                                                              \
    _Pragma("unroll") for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(XA, QA, acc[g], 0, 0, 0);
                ST_STEP(x0.x, q0[g].x)
                ST_STEP(x1.x, q1[g].x)
                ST_STEP(x0.y, q0[g].y)
                ST_STEP(x1.y, q1[g].y)
                ST_STEP(x0.z, q0[g].z)
                ST_STEP(x1.z, q1[g].z)
                ST_STEP(x0.w, q0[g].w)
                ST_STEP(x1.w, q1[g].w)
#undef ST_STEP
            }
        }
        // ---- end of a tile
        if (have_thr) {
            append_tile(acc, tile);
        } else {
            if (nheld == 0) held_tile0 = tile;
#pragma unroll
            for (int h = 0; h < ST_HOLD; ++h)
                if (h == nheld)
#pragma unroll
                    for (int g = 0; g < G; ++g) held[h][g] = acc[g];
            ++nheld;
        }
        if (a.seed && tile == t0) {
            // publish the best row of this wave's first tile, per query (composites; 0 = no valid row)
            const long long row0 = (long long)tile * ST_BM + rsub;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int q = 4 * g + (lane & 3);
                u64 c = 0ull;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u64 v = (q < a.nq && row0 + i < a.n)
                                      ? (((u64)f2key(acc[g][i]) << 32) | (u64)(0xFFFFFFFFu - (u32)(row0 + i))) : 0ull;
                    c = v > c ? v : c;
                }
#pragma unroll
                for (int o = 32; o >= 4; o >>= 1) {      // over the lanes that hold this query
                    const u64 v = __shfl_xor(c, o);
                    c = v > c ? v : c;
                }
                if (lane < 4 && q < a.nq)
                    __hip_atomic_store(a.bmax + (size_t)q * (4 * a.strips) + 4 * strip + wave, c, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            wait_vmcnt<0>();            // this wave's stores have left (once per launch: the ring drains with them)
        }
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        wg_barrier();                   // tile boundary: every wave has appended / published
        if (a.seed && tile == t0) {
            if (tid == 0) {
                const u32 old = __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sFlag[2] = (old == (u32)a.strips - 1u) ? 1u : 0u;
            }
            wg_barrier();
            if (sFlag[2]) {
                // last arrival: every block maximum is published.  Wave w selects for queries w, w + 4, ...
                const int M = 4 * a.strips;          // <= 1024 (host)
                for (int q = wave; q < a.nq; q += 4) {
                    u64 x[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int idx = e * 64 + lane;
                        x[e] = idx < M ? __hip_atomic_load(a.bmax + (size_t)q * M + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : 0ull;
                    }
                    int valid = 0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) valid += __popcll(__ballot(x[e] != 0ull));
                    const u32 T = valid >= a.k ? wave_kth_key<16>(x, a.k) : 0u;
                    if (lane == 0)
                        __hip_atomic_store(a.sync + 2 + q, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                wait_vmcnt<0>();
                wg_barrier();
                if (tid == 0) __hip_atomic_store(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!got_global) {
            // thresholds there yet?  (one sc1 poll per tile boundary; nobody waits)
            if (wave == 0) {
                const u32 r = __hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (r && lane < a.nq) {
                    const u32 t = __hip_atomic_load(a.sync + 2 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sThr[lane] = max(sThr[lane], t);
                }
                if (lane == 0) sFlag[1] = r;
            }
            wg_barrier();
            const bool ready = sFlag[1] != 0u;
            got_global = ready;
            if (!have_thr && (ready || nheld == ST_HOLD || tile + 1 == t1)) {
                // flush the held tiles through the filter (without thresholds: one tile at a time, compacting)
#pragma unroll
                for (int h = 0; h < ST_HOLD; ++h) {
                    if (h < nheld) {
                        append_tile(held[h], held_tile0 + h);
                        wg_barrier();
                        if (__builtin_amdgcn_readfirstlane(sFlag[0])) compact_pass(false);
                    }
                }
                nheld = 0;
                have_thr = true;
            } else if (have_thr && __builtin_amdgcn_readfirstlane(sFlag[0])) {
                compact_pass(false);
            }
        } else if (__builtin_amdgcn_readfirstlane(sFlag[0])) {
            compact_pass(false);
        }
    }
    compact_pass(true);
    if (tid < NQP) a.cnt[(size_t)strip * NQP + tid] = min(sCnt[tid], (u32)CAP);
}

// ------------------------------------------------------------------ bf16-MFMA scan
// Same product, on the bf16 matrix cores (16x the fp32 matrix rate).  Two uses:
//   MODE 0  corpus stored as bf16 (dtype OTTER_BF16; BASELINE config 5): rows are [ld] bf16,
//           queries rounded to bf16, score = sum over k of x_k q_k accumulated in fp32.
//   MODE 1  coarse pass of the exact fp32 search: every fp32 value v is split into
//           hi = bf16(v), lo = bf16(v - hi) (|v - hi - lo| <= 2^-16 |v|); rows are stored
//           interleaved per 32 values of k as [hi(32) | lo(32)] (128 B), and the score is
//           x_hi q_hi + x_hi q_lo + x_lo q_hi (the dropped terms are <= 3 * 2^-16 |x_k q_k| each).
//           The caller keeps k + margin candidates by this coarse score, re-scores them exactly
//           (rescore_lds_kernel: the fp32 fma chain of scan_kernel) and proves the margin sufficient
//           (margin_check_kernel) or falls back to scan_kernel for that query.
// Structure: the encoder GEMM's persistent ping-pong main loop (gemm_bf16.hip: 8 waves = 2 groups
// one barrier apart, 256 x 256 tile, K-step of 128 B rows as four 16 KiB half-tiles, LDS-DMA six
// half-tiles ahead with a counted vmcnt, source-side XOR swizzle, v_mfma_f32_16x16x32_bf16) with
// the queries on the GEMM's token side (lanes) and the corpus rows on its feature side
// (accumulator rows), streamed over a strip of corpus tiles, and scan_kernel's epilogue
// (per-query threshold in LDS, candidate buffers, wave radix compaction).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------ stream kernel, 17 ... 32 queries
// stream_kernel's structure (per-wave rings, non-temporal corpus loads, threshold exchange, no sample pass) with
// scan_kernel's v_mfma_f32_32x32x2_f32 fragments: a wave's 64 rows are two 32-row blocks against all 32 query
// columns; from 17 queries on that tile is cheaper than 4x4x1 products (same fma chain either way).
template <int EPL, int ST_NSTAGE>
__global__ __launch_bounds__(256) void stream32_kernel(StreamArgs a) {
    constexpr int NQP = 32;                    // query columns of one 32x32x2 block
    constexpr int G = 8;                       // (query rows of a stage = 4 G, as in stream_kernel)
    constexpr int RB = 2;                      // 32-row blocks of a wave's 64 rows
    constexpr int ST_BK = 32;
    constexpr int ST_HOLD = 3;                 // tiles whose scores wait in registers for the thresholds (32 per tile)
    constexpr int CAP = EPL * 64;
    constexpr int QI = st_qinstr(G, ST_BK), ST_SLOT = st_slot(G, ST_BK), ST_IPS = st_ips(G, ST_BK);
    constexpr int CH = ST_BK / 4;              // 16-byte chunks per row of a stage
    constexpr int RPI = 64 / CH;               // rows per LDS-DMA instruction
    constexpr int ROWB = ST_BK * 4;            // bytes per row of a stage
    static_assert(ST_BK == 32 || ST_BK == 64, "stage depth");
    static_assert(CAP - ST_BM >= 128, "candidate buffer too small");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32* sCnt = reinterpret_cast<u32*>(smem + 4 * ST_NSTAGE * ST_SLOT);
    u32* sThr = sCnt + 32;
    u32* sFlag = sThr + 32;       // [0] compaction wanted, [1] thresholds ready, [2] this workgroup selects
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = blockIdx.x;
    const int t0 = (int)(((long long)strip * a.total_tiles) / a.strips);
    const int t1 = (int)(((long long)(strip + 1) * a.total_tiles) / a.strips);
    const int KS = a.ld / ST_BK;
    if (tid < 32) {
        sCnt[tid] = 0;
        sThr[tid] = 0u;            // accept everything (key >= 0) until a threshold is known
    }
    if (tid < 3) sFlag[tid] = 0;
    u64* candBase = a.cand + (size_t)strip * NQP * CAP;

    // ---- producer: this wave's 64 rows x BK floats (1 KiB per instruction) + the queries' BK floats
    // LDS image: [row][CH chunks of 16 B], chunk position p of row r holds global chunk p ^ swz(r), swz chosen so
    // that the rows one ds_read_b128 lane group touches fall on distinct 16-byte slots of the 256-byte bank row
    auto swz = [](int r) { return CH == 16 ? (r & 15) : ((r >> 1) & 7); };
    const u32 lds0 = (u32)(uintptr_t)(AS3 char*)smem + (u32)wave * (ST_NSTAGE * ST_SLOT);
    u32 voffA[CH], voffQ[QI];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int r = RPI * i + lane / CH;
        voffA[i] = ((u32)(wave * 64 + r) * (u32)a.ld + (u32)(((lane % CH) ^ swz(r)) << 2)) * 4u;
    }
#pragma unroll
    for (int i = 0; i < QI; ++i) {
        const int r = RPI * i + lane / CH;       // a row past the batch reads its last row (never accepted)
        voffQ[i] = ((u32)min(r, max(a.q_rows - 1, 0)) * (u32)a.ld + (u32)(((lane % CH) ^ swz(r)) << 2)) * 4u;   // (swizzled like the rows)
    }
    const long long total_stages = (long long)(t1 - t0) * KS;
    int it_tile = t0, it_ks = 0;
    long long it_f = 0;
    auto issue_stage = [&]() {
        if (it_f < total_stages) {
            const float* gA = a.X + ((size_t)it_tile * ST_BM * a.ld + (size_t)it_ks * ST_BK);
            const float* gQ = a.Q + (size_t)it_ks * ST_BK;
            const u32 l = lds0 + (u32)(it_f % ST_NSTAGE) * ST_SLOT;
#pragma unroll
            for (int i = 0; i < CH; ++i) dma16s_stream(gA, voffA[i], l + i * 1024);
#pragma unroll
            for (int i = 0; i < QI; ++i) dma16s(gQ, voffQ[i], l + CH * 1024 + i * 1024);
            ++it_f;
            if (++it_ks == KS) { it_ks = 0; ++it_tile; }
        }
    };

    // ---- candidate path.  Lane l (query l & 31, half h = l >> 5) holds, per 32-row block b, the scores of rows
    // 32 b + (r & 3) + 8 (r >> 2) + 4 h (r = 0..15) of this wave's 64
    const int l31 = lane & 31, hh = lane >> 5;
    auto append_tile = [&](const f32x16 (&sc)[RB], int tile) {
        const int q = l31;
        const u32 thr = sThr[q];
        const float thrf = key2f(thr);
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            // after warm-up almost nothing passes: test the block's maximum first
            float mx = fmaxf(sc[b][0], sc[b][1]);
#pragma unroll
            for (int r = 2; r < 16; ++r) mx = fmaxf(mx, sc[b][r]);
            if (!(thr == 0u || mx >= thrf || mx != mx)) continue;
            const long long row0 = (long long)tile * ST_BM + wave * 64 + 32 * b + 4 * hh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = row0 + (r & 3) + 8 * (r >> 2);
                const u32 key = f2key(sc[b][r]);
                if (q < a.nq && key >= thr && row < a.n) {
                    const u32 pos = atomicAdd(&sCnt[q], 1u);
                    if (pos < (u32)CAP) candBase[(u32)q * (u32)CAP + pos] = ((u64)key << 32) | (u64)(0xFFFFFFFFu - (u32)row);
                    if (pos + 1 > (u32)(CAP - ST_BM)) sFlag[0] = 1;
                }
            }
        }
    };
    auto compact_pass = [&](bool final_pass) {      // workgroup-wide; every wave calls it at the same point
        wait_vmcnt<0>();
        wg_barrier();
        for (int q = wave; q < NQP; q += 4) {
            const int n = (int)min(sCnt[q], (u32)CAP);
            const bool need = final_pass ? (n > a.k) : (n > CAP - ST_BM);
            if (need) {
                const u64 T = wave_compact<EPL>(candBase + (size_t)q * CAP, n, a.k, lane);
                if (lane == 0) {
                    sCnt[q] = (u32)a.k;
                    sThr[q] = max(sThr[q], (u32)(T >> 32));
                }
            }
        }
        if (tid == 0) sFlag[0] = 0;
        wait_vmcnt<0>();
        wg_barrier();
    };

    f32x16 acc[RB];
    f32x16 held[ST_HOLD][RB];
    int nheld = 0, held_tile0 = t0;
    bool have_thr = a.seed == 0;     // no exchange: thresholds are learned locally from the start
    bool got_global = a.seed == 0;   // the exchanged thresholds have been taken over (or there are none to wait for)
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;

#pragma unroll
    for (int s2 = 0; s2 < ST_NSTAGE - 1; ++s2) issue_stage();

    const u32 rowoff = (u32)l31 * (u32)ROWB;
    const u32 rswz = (u32)swz(l31);              // (swz(32 b + l31) == swz(l31))
    long long f = 0;
    for (int tile = t0; tile < t1; ++tile) {
        for (int ks = 0; ks < KS; ++ks, ++f) {
            // stage f (this wave's own loads) has landed once at most NSTAGE-2 younger stages are outstanding
            if (total_stages - 1 - f >= ST_NSTAGE - 2) wait_vmcnt<(ST_NSTAGE - 2) * ST_IPS>();
            else wait_vmcnt<0>();
            // the slot of stage f-1 was read by this wave only, and those reads have returned (their values fed
            // the products above): refill it with stage f + NSTAGE - 1
            wait_lgkm0();
            issue_stage();
            const char* sA = smem + wave * (ST_NSTAGE * ST_SLOT) + (int)(f % ST_NSTAGE) * ST_SLOT;
            const char* sQ = sA + CH * 1024;
            // scan_kernel's fragments: 16 bytes per lane = four k-steps (k = 8 j + 4 h + t), same fma chain
#pragma unroll
            for (int j = 0; j < ST_BK / 8; ++j) {
                const u32 co = ((u32)(2 * j + hh) ^ rswz) << 4;
                const float4 qf = *reinterpret_cast<const float4*>(sQ + rowoff + co);
#pragma unroll
                for (int b = 0; b < RB; ++b) {
                    const float4 xf = *reinterpret_cast<const float4*>(sA + b * 32 * ROWB + rowoff + co);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf.x, qf.x, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf.y, qf.y, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf.z, qf.z, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf.w, qf.w, acc[b], 0, 0, 0);
                }
            }
        }
        // ---- end of a tile
        if (have_thr) {
            append_tile(acc, tile);
        } else {
            if (nheld == 0) held_tile0 = tile;
#pragma unroll
            for (int h = 0; h < ST_HOLD; ++h)
                if (h == nheld)
#pragma unroll
                    for (int b = 0; b < RB; ++b) held[h][b] = acc[b];
            ++nheld;
        }
        if (a.seed && tile == t0) {
            // publish the best row of this wave's first tile, per query (composites; 0 = no valid row)
            {
                const int q = l31;
                u64 c = 0ull;
#pragma unroll
                for (int b = 0; b < RB; ++b) {
                    const long long row0 = (long long)tile * ST_BM + wave * 64 + 32 * b + 4 * hh;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long long row = row0 + (r & 3) + 8 * (r >> 2);
                        const u64 v = (q < a.nq && row < a.n)
                                          ? (((u64)f2key(acc[b][r]) << 32) | (u64)(0xFFFFFFFFu - (u32)row)) : 0ull;
                        c = v > c ? v : c;
                    }
                }
                {
                    const u64 v = __shfl_xor(c, 32);     // the other half of this query's rows
                    c = v > c ? v : c;
                }
                if (lane < 32 && q < a.nq)
                    __hip_atomic_store(a.bmax + (size_t)q * (4 * a.strips) + 4 * strip + wave, c, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            wait_vmcnt<0>();            // this wave's stores have left (once per launch: the ring drains with them)
        }
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
        wg_barrier();                   // tile boundary: every wave has appended / published
        if (a.seed && tile == t0) {
            if (tid == 0) {
                const u32 old = __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sFlag[2] = (old == (u32)a.strips - 1u) ? 1u : 0u;
            }
            wg_barrier();
            if (sFlag[2]) {
                // last arrival: every block maximum is published.  Wave w selects for queries w, w + 4, ...
                const int M = 4 * a.strips;          // <= 1024 (host)
                for (int q = wave; q < a.nq; q += 4) {
                    u64 x[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int idx = e * 64 + lane;
                        x[e] = idx < M ? __hip_atomic_load(a.bmax + (size_t)q * M + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : 0ull;
                    }
                    int valid = 0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) valid += __popcll(__ballot(x[e] != 0ull));
                    const u32 T = valid >= a.k ? wave_kth_key<16>(x, a.k) : 0u;
                    if (lane == 0)
                        __hip_atomic_store(a.sync + 2 + q, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                wait_vmcnt<0>();
                wg_barrier();
                if (tid == 0) __hip_atomic_store(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!got_global) {
            // thresholds there yet?  (one sc1 poll per tile boundary; nobody waits)
            if (wave == 0) {
                const u32 r = __hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (r && lane < a.nq) {
                    const u32 t = __hip_atomic_load(a.sync + 2 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sThr[lane] = max(sThr[lane], t);
                }
                if (lane == 0) sFlag[1] = r;
            }
            wg_barrier();
            const bool ready = sFlag[1] != 0u;
            got_global = ready;
            if (!have_thr && (ready || nheld == ST_HOLD || tile + 1 == t1)) {
                // flush the held tiles through the filter (without thresholds: one tile at a time, compacting)
#pragma unroll
                for (int h = 0; h < ST_HOLD; ++h) {
                    if (h < nheld) {
                        append_tile(held[h], held_tile0 + h);
                        wg_barrier();
                        if (__builtin_amdgcn_readfirstlane(sFlag[0])) compact_pass(false);
                    }
                }
                nheld = 0;
                have_thr = true;
            } else if (have_thr && __builtin_amdgcn_readfirstlane(sFlag[0])) {
                compact_pass(false);
            }
        } else if (__builtin_amdgcn_readfirstlane(sFlag[0])) {
            compact_pass(false);
        }
    }
    compact_pass(true);
    if (tid < NQP) a.cnt[(size_t)strip * NQP + tid] = min(sCnt[tid], (u32)CAP);
}

// ------------------------------------------------------------------ bf16-MFMA scan
// Same product, on the bf16 matrix cores (16x the fp32 matrix rate).  Two uses:
//   MODE 0  corpus stored as bf16 (dtype OTTER_BF16; BASELINE config 5): rows are [ld] bf16,
//           queries rounded to bf16, score = sum over k of x_k q_k accumulated in fp32.
//   MODE 1  coarse pass of the exact fp32 search: every fp32 value v is split into
//           hi = bf16(v), lo = bf16(v - hi) (|v - hi - lo| <= 2^-16 |v|); rows are stored
//           interleaved per 32 values of k as [hi(32) | lo(32)] (128 B), and the score is
//           x_hi q_hi + x_hi q_lo + x_lo q_hi (the dropped terms are <= 3 * 2^-16 |x_k q_k| each).
//           The caller keeps k + margin candidates by this coarse score, re-scores them exactly
//           (rescore_lds_kernel: the fp32 fma chain of scan_kernel) and proves the margin sufficient
//           (margin_check_kernel) or falls back to scan_kernel for that query.
// Structure: the encoder GEMM's persistent ping-pong main loop (gemm_bf16.hip: 8 waves = 2 groups
// one barrier apart, 256 x 256 tile, K-step of 128 B rows as four 16 KiB half-tiles, LDS-DMA six
// half-tiles ahead with a counted vmcnt, source-side XOR swizzle, v_mfma_f32_16x16x32_bf16) with
// the queries on the GEMM's token side (lanes) and the corpus rows on its feature side
// (accumulator rows), streamed over a strip of corpus tiles, and scan_kernel's epilogue
// (per-query threshold in LDS, candidate buffers, wave radix compaction).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------ stream kernel, bf16 storage (at most 32 queries)
// stream_kernel's structure on an OTTER_BF16 index: a wave streams its own 64 rows of every 256-row tile, one
// 64-element K-slab (128 B per row; the planes are tile-blocked, so a wave's rows of a slab are 8 KiB of
// consecutive bytes) per stage, v_mfma_f32_16x16x32_bf16 with scan16_kernel<0>'s operand roles and k order
// (identical scores), the same threshold exchange, candidate path and merge.  X and Q are the bf16 planes;
// a.ld is the row length in elements.
template <int QG, int EPL, int ST_NSTAGE>
__global__ __launch_bounds__(256) void stream16_kernel(StreamArgs a) {
    constexpr int NQP = 16 * QG;               // query columns: QG blocks of 16 (one 16x16x32 block each)
    constexpr int G = 4;                       // 16-row blocks of a wave's 64 rows
    constexpr int ST_HOLD = QG == 1 ? 6 : 3;   // tiles whose scores wait in registers for the thresholds (16 QG per tile)
    constexpr int ST_BK = 32;                  // (a stage = one 64-element K-slab = 128 bytes per row, as 32 floats are)
    constexpr int CAP = EPL * 64;
    constexpr int QI = 2 * QG, ST_SLOT = 64 * 128 + QI * 1024, ST_IPS = 8 + QI;   // 16 QG query rows x 128 B
    constexpr int CH = ST_BK / 4;              // 16-byte chunks per row of a stage
    constexpr int RPI = 64 / CH;               // rows per LDS-DMA instruction
    static_assert(CAP - ST_BM >= 128, "candidate buffer too small");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32* sCnt = reinterpret_cast<u32*>(smem + 4 * ST_NSTAGE * ST_SLOT);
    u32* sThr = sCnt + 32;
    u32* sFlag = sThr + 32;       // [0] compaction wanted, [1] thresholds ready, [2] this workgroup selects
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = blockIdx.x;
    const int t0 = (int)(((long long)strip * a.total_tiles) / a.strips);
    const int t1 = (int)(((long long)(strip + 1) * a.total_tiles) / a.strips);
    const int KS = a.ld / 64;                  // a.ld = row length in bf16 elements (multiple of 64)
    if (tid < 32) {
        sCnt[tid] = 0;
        sThr[tid] = 0u;            // accept everything (key >= 0) until a threshold is known
    }
    if (tid < 3) sFlag[tid] = 0;
    u64* candBase = a.cand + (size_t)strip * NQP * CAP;

    // ---- producer: this wave's 64 rows x BK floats (1 KiB per instruction) + the queries' BK floats
    // LDS image: [row][CH chunks of 16 B], chunk position p of row r holds global chunk p ^ swz(r), swz chosen so
    // that the rows one ds_read_b128 lane group touches fall on distinct 16-byte slots of the 256-byte bank row
    auto swz = [](int r) { return CH == 16 ? (r & 15) : ((r >> 1) & 7); };
    const u32 lds0 = (u32)(uintptr_t)(AS3 char*)smem + (u32)wave * (ST_NSTAGE * ST_SLOT);
    u32 voffA[CH], voffQ[QI];
    // (tile-blocked planes: the K-slab of a 256-row tile is one 32 KiB block of 256 rows x 128 B, plane_index)
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int r = RPI * i + lane / CH;
        voffA[i] = (u32)(wave * 64 + r) * 128u + (u32)(((lane % CH) ^ swz(r)) << 4);
    }
#pragma unroll
    for (int i = 0; i < QI; ++i) {
        const int r = RPI * i + lane / CH;       // rows past the batch are zero rows of the query plane
        voffQ[i] = (u32)r * 128u + (u32)(((lane % CH) ^ swz(r)) << 4);
    }
    const long long total_stages = (long long)(t1 - t0) * KS;
    int it_tile = t0, it_ks = 0;
    long long it_f = 0;
    auto issue_stage = [&]() {
        if (it_f < total_stages) {
            const char* gA = reinterpret_cast<const char*>(a.X) + ((size_t)((u32)it_tile * (u32)KS + (u32)it_ks) << 15);
            const char* gQ = reinterpret_cast<const char*>(a.Q) + ((size_t)(u32)it_ks << 15);
            const u32 l = lds0 + (u32)(it_f % ST_NSTAGE) * ST_SLOT;
#pragma unroll
            for (int i = 0; i < CH; ++i) dma16s_stream(gA, voffA[i], l + i * 1024);
#pragma unroll
            for (int i = 0; i < QI; ++i) dma16s(gQ, voffQ[i], l + CH * 1024 + i * 1024);
            ++it_f;
            if (++it_ks == KS) { it_ks = 0; ++it_tile; }
        }
    };

    // ---- candidate path.  Lane l holds, per 16-row block g, the scores of rows 16 g + 4 (l >> 4) + i (i = 0..3)
    // of this wave's 64 for query l & 15
    const int rsub = wave * 64 + 4 * (lane >> 4);
    auto append_tile = [&](const f32x4 (&sc)[QG][G], int tile) {
#pragma unroll
        for (int c = 0; c < QG; ++c) {
            const int q = 16 * c + (lane & 15);
            const u32 thr = sThr[q];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const long long row0 = (long long)tile * ST_BM + rsub + 16 * g;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32 key = f2key(sc[c][g][i]);
                    if (q < a.nq && key >= thr && row0 + i < a.n) {
                        const u32 pos = atomicAdd(&sCnt[q], 1u);
                        if (pos < (u32)CAP)
                            candBase[(u32)q * (u32)CAP + pos] = ((u64)key << 32) | (u64)(0xFFFFFFFFu - (u32)(row0 + i));
                        if (pos + 1 > (u32)(CAP - ST_BM)) sFlag[0] = 1;
                    }
                }
            }
        }
    };
    auto compact_pass = [&](bool final_pass) {      // workgroup-wide; every wave calls it at the same point
        wait_vmcnt<0>();
        wg_barrier();
        for (int q = wave; q < NQP; q += 4) {
            const int n = (int)min(sCnt[q], (u32)CAP);
            const bool need = final_pass ? (n > a.k) : (n > CAP - ST_BM);
            if (need) {
                const u64 T = wave_compact<EPL>(candBase + (size_t)q * CAP, n, a.k, lane);
                if (lane == 0) {
                    sCnt[q] = (u32)a.k;
                    sThr[q] = max(sThr[q], (u32)(T >> 32));
                }
            }
        }
        if (tid == 0) sFlag[0] = 0;
        wait_vmcnt<0>();
        wg_barrier();
    };

    f32x4 acc[QG][G];
    f32x4 held[ST_HOLD][QG][G];
    int nheld = 0, held_tile0 = t0;
    bool have_thr = a.seed == 0;     // no exchange: thresholds are learned locally from the start
    bool got_global = a.seed == 0;   // the exchanged thresholds have been taken over (or there are none to wait for)
#pragma unroll
    for (int c = 0; c < QG; ++c)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[c][g] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s2 = 0; s2 < ST_NSTAGE - 1; ++s2) issue_stage();

    const u32 fr = (u32)(lane & 15), fq = (u32)(lane >> 4);
    const u32 rowoff = fr * 128u;
    const u32 rswz = (u32)swz((int)fr);          // (swz(16 g + fr) == swz(fr))
    long long f = 0;
    for (int tile = t0; tile < t1; ++tile) {
        for (int ks = 0; ks < KS; ++ks, ++f) {
            // stage f (this wave's own loads) has landed once at most NSTAGE-2 younger stages are outstanding
            if (total_stages - 1 - f >= ST_NSTAGE - 2) wait_vmcnt<(ST_NSTAGE - 2) * ST_IPS>();
            else wait_vmcnt<0>();
            // the slot of stage f-1 was read by this wave only, and those reads have returned (their values fed
            // the products above): refill it with stage f + NSTAGE - 1
            wait_lgkm0();
            issue_stage();
            const char* sA = smem + wave * (ST_NSTAGE * ST_SLOT) + (int)(f % ST_NSTAGE) * ST_SLOT;
            const char* sQ = sA + CH * 1024;
            // two k-steps of 32 per slab; operand roles as in scan16_kernel<0> (corpus rows = A, queries = B), so the
            // scores are that kernel's, bit for bit
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                const u32 co = ((4u * ks2 + fq) ^ rswz) << 4;
                float4 qf[QG];
#pragma unroll
                for (int c = 0; c < QG; ++c) qf[c] = *reinterpret_cast<const float4*>(sQ + c * 16 * 128 + rowoff + co);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float4 xf = *reinterpret_cast<const float4*>(sA + g * 16 * 128 + rowoff + co);
#pragma unroll
                    for (int c = 0; c < QG; ++c)
                        acc[c][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xf),
                                                                            __builtin_bit_cast(bf16x8, qf[c]), acc[c][g], 0, 0, 0);
                }
            }
        }
        // ---- end of a tile
        if (have_thr) {
            append_tile(acc, tile);
        } else {
            if (nheld == 0) held_tile0 = tile;
#pragma unroll
            for (int h = 0; h < ST_HOLD; ++h)
                if (h == nheld)
#pragma unroll
                    for (int c = 0; c < QG; ++c)
#pragma unroll
                        for (int g = 0; g < G; ++g) held[h][c][g] = acc[c][g];
            ++nheld;
        }
        if (a.seed && tile == t0) {
            // publish the best row of this wave's first tile, per query (composites; 0 = no valid row)
#pragma unroll
            for (int cg = 0; cg < QG; ++cg) {
                const int q = 16 * cg + (lane & 15);
                u64 c = 0ull;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const long long row0 = (long long)tile * ST_BM + rsub + 16 * g;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u64 v = (q < a.nq && row0 + i < a.n)
                                          ? (((u64)f2key(acc[cg][g][i]) << 32) | (u64)(0xFFFFFFFFu - (u32)(row0 + i))) : 0ull;
                        c = v > c ? v : c;
                    }
                }
#pragma unroll
                for (int o = 32; o >= 16; o >>= 1) {     // over the lanes that hold this query
                    const u64 v = __shfl_xor(c, o);
                    c = v > c ? v : c;
                }
                if (lane < 16 && q < a.nq)
                    __hip_atomic_store(a.bmax + (size_t)q * (4 * a.strips) + 4 * strip + wave, c, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            wait_vmcnt<0>();            // this wave's stores have left (once per launch: the ring drains with them)
        }
#pragma unroll
        for (int c = 0; c < QG; ++c)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[c][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        wg_barrier();                   // tile boundary: every wave has appended / published
        // Threshold exchange, selection shared out: once every workgroup has published its block maxima
        // (arrivals == strips), any workgroup that reaches a tile boundary claims four queries (one per wave) and
        // selects their k-th best block maximum; sync[1] counts the queries done.  Nobody waits for anybody: a
        // workgroup only ever looks at the counters when it is at a tile boundary anyway.  (One workgroup selecting
        // all 32 queries takes about 50 us, more than the three tiles this kernel can hold at bf16 speed.)
        if (!got_global) {
            if (tid == 0) {
                u32 arr = 0u;
                if (tile == t0) arr = __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                else arr = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u32 done = __hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                u32 claim = 0xFFFFFFFFu;
                if (arr == (u32)a.strips && done < (u32)a.nq) {
                    const u32 c = __hip_atomic_fetch_add(a.sync + 35, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (c < (u32)a.nq) claim = c;
                }
                sFlag[2] = claim;
                sFlag[1] = done >= (u32)a.nq ? 1u : 0u;
            }
            wg_barrier();
            const u32 claim = sFlag[2];
            if (claim != 0xFFFFFFFFu) {
                const int M = 4 * a.strips;          // <= 1024 (host)
                const int q = (int)claim + wave;
                if (q < a.nq) {
                    u64 x[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int idx = e * 64 + lane;
                        x[e] = idx < M ? __hip_atomic_load(a.bmax + (size_t)q * M + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : 0ull;
                    }
                    int valid = 0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) valid += __popcll(__ballot(x[e] != 0ull));
                    const u32 T = valid >= a.k ? wave_kth_key<16>(x, a.k) : 0u;
                    if (lane == 0)
                        __hip_atomic_store(a.sync + 2 + q, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                wait_vmcnt<0>();
                wg_barrier();
                if (tid == 0)
                    __hip_atomic_fetch_add(a.sync + 1, (u32)min(4, a.nq - (int)claim), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (sFlag[1] != 0u && tid < a.nq) {
                const u32 t = __hip_atomic_load(a.sync + 2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sThr[tid] = max(sThr[tid], t);
            }
            wg_barrier();
            const bool ready = sFlag[1] != 0u;
            got_global = ready;
            if (!have_thr && (ready || nheld == ST_HOLD || tile + 1 == t1)) {
                // flush the held tiles through the filter (without thresholds: one tile at a time, compacting)
#pragma unroll
                for (int h = 0; h < ST_HOLD; ++h) {
                    if (h < nheld) {
                        append_tile(held[h], held_tile0 + h);
                        wg_barrier();
                        if (__builtin_amdgcn_readfirstlane(sFlag[0])) compact_pass(false);
                    }
                }
                nheld = 0;
                have_thr = true;
            } else if (have_thr && __builtin_amdgcn_readfirstlane(sFlag[0])) {
                compact_pass(false);
            }
        } else if (__builtin_amdgcn_readfirstlane(sFlag[0])) {
            compact_pass(false);
        }
    }
    compact_pass(true);
    if (tid < NQP) a.cnt[(size_t)strip * NQP + tid] = min(sCnt[tid], (u32)CAP);
}

// ------------------------------------------------------------------ stream kernel, bf16 storage, 33 ... 64 queries
// Between 33 and 256 queries the 256-column scan16 tile computes mostly padding (0.41 ms at 33 queries against
// 0.27 ms at 32).  Up to 64 queries the pass is still one read of the plane: this kernel keeps stream16_kernel's
// wave-private streaming, k order and candidate path, and changes where the operands live:
//   queries   ALL slabs of the (at most 64) query rows stay in LDS for the whole launch (64 x (2 ld + 128) bytes:
//             104 KiB at ld = 768), filled once by LDS-DMA from the query plane with the source-side swizzle; rows
//             are pitched an odd multiple of 128 bytes so that the 16 rows of one ds_read_b128 group fall on 16
//             distinct 16-byte bank slots.  Nothing of Q is loaded again, so a stage costs the corpus bytes only.
//   corpus    from HBM into REGISTERS (eight fully coalesced 1 KiB loads per 8 KiB stage, non-temporal) through a
//             ring of three register stages -- 16 KiB per wave in flight without any LDS behind it --, and only the
//             stage being consumed passes through 8 KiB of wave-private LDS to reach MFMA operand layout.  (Loading
//             operand layout straight from memory -- 16 bytes of 16 different rows per 16 lanes -- ran at 4.2 TB/s.)
//             The loads are ordinary compiler-visible loads: with no LDS-DMA in the main loop the compiler's own
//             vmcnt accounting is exact (it waits for the oldest stage with the two younger ones outstanding), and
//             no register is written behind its back.
//   held      the tiles scored before the exchanged thresholds arrive wait in a global scratch (64 KiB per tile and
//             workgroup, written and read back by the same lane), not in registers: 64 accumulators per lane
//             leave no room for stream16_kernel's register copies.
// Scores are scan16_kernel<0>'s bit for bit (same operand roles, same k order).
template <int QG, int EPL>
__global__ __launch_bounds__(256) void stream64_kernel(StreamArgs a) {
    constexpr int NQP = 16 * QG;
    constexpr int G = 4, HOLD = S64_HOLD, CAP = EPL * 64;
    static_assert(CAP - ST_BM >= 128, "candidate buffer too small");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KS = a.ld / 64;                  // a.ld = row length in bf16 elements (multiple of 64)
    const u32 pitch = s64_pitch(a.ld);
    const u32 qbytes = (u32)NQP * pitch;       // (a multiple of 2048)
    u32* sCnt = reinterpret_cast<u32*>(smem + qbytes + 4 * 8192);      // (behind the waves' staging buffers)
    u32* sThr = sCnt + NQP;
    u32* sFlag = sThr + NQP;      // [0] compaction wanted, [1] thresholds ready, [2] this workgroup selects
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = blockIdx.x;
    const int t0 = (int)(((long long)strip * a.total_tiles) / a.strips);
    const int t1 = (int)(((long long)(strip + 1) * a.total_tiles) / a.strips);
    if (tid < NQP) {
        sCnt[tid] = 0;
        sThr[tid] = 0u;            // accept everything (key >= 0) until a threshold is known
    }
    if (tid < 3) sFlag[tid] = 0;
    u64* candBase = a.cand + (size_t)strip * NQP * CAP;

    // ---- the queries: image row r = query r, [KS slabs of 128 B | pad]; chunk position p of a slab holds global
    // chunk p ^ ((r >> 1) & 7) (consecutive rows alternate between the halves of the 256-byte bank row)
    {
        const u32 ppr = pitch >> 4;                 // 16-byte chunks per image row
        const u32 pieces = qbytes >> 10;
        const u32 ldsq = (u32)(uintptr_t)(AS3 char*)smem;
        for (u32 i = (u32)wave; i < pieces; i += 4) {
            const u32 p = i * 64u + (u32)lane;
            const u32 row = p / ppr, w = p - row * ppr;
            u32 ks = w >> 3;
            const u32 pos = w & 7u;
            if (ks >= (u32)KS) ks = 0;               // the pad chunks of a row: never read
            const u32 chunk = pos ^ ((row >> 1) & 7u);
            dma16s(a.Q, ks * 32768u + row * 128u + chunk * 16u, ldsq + i * 1024u);
        }
        wait_vmcnt<0>();
        wg_barrier();
    }

    // ---- candidate path.  Lane l holds, per 16-row block g, the scores of rows 16 g + 4 (l >> 4) + i (i = 0..3)
    // of this wave's 64 for query l & 15 (of column block c)
    const int rsub = wave * 64 + 4 * (lane >> 4);
    auto append_tile = [&](const f32x4 (&sc)[QG][G], int tile) {
#pragma unroll
        for (int c = 0; c < QG; ++c) {
            const int q = 16 * c + (lane & 15);
            const u32 thr = sThr[q];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const long long row0 = (long long)tile * ST_BM + rsub + 16 * g;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32 key = f2key(sc[c][g][i]);
                    if (q < a.nq && key >= thr && row0 + i < a.n) {
                        const u32 pos = atomicAdd(&sCnt[q], 1u);
                        if (pos < (u32)CAP)
                            candBase[(u32)q * (u32)CAP + pos] = ((u64)key << 32) | (u64)(0xFFFFFFFFu - (u32)(row0 + i));
                        if (pos + 1 > (u32)(CAP - ST_BM)) sFlag[0] = 1;
                    }
                }
            }
        }
    };
    auto compact_pass = [&](bool final_pass) {      // workgroup-wide; every wave calls it at the same point
        wait_vmcnt<0>();
        wg_barrier();
        for (int q = wave; q < NQP; q += 4) {
            const int n = (int)min(sCnt[q], (u32)CAP);
            const bool need = final_pass ? (n > a.k) : (n > CAP - ST_BM);
            if (need) {
                const u64 T = wave_compact<EPL>(candBase + (size_t)q * CAP, n, a.k, lane);
                if (lane == 0) {
                    sCnt[q] = (u32)a.k;
                    sThr[q] = max(sThr[q], (u32)(T >> 32));
                }
            }
        }
        if (tid == 0) sFlag[0] = 0;
        wait_vmcnt<0>();
        wg_barrier();
    };

    f32x4 acc[QG][G];
    f32x4* hold = reinterpret_cast<f32x4*>(a.hold) + (size_t)strip * HOLD * (QG * G * 256) + tid;
    int nheld = 0, held_tile0 = t0;
    bool have_thr = a.seed == 0;     // no exchange: thresholds are learned locally from the start
    bool got_global = a.seed == 0;   // the exchanged thresholds have been taken over (or there are none to wait for)
#pragma unroll
    for (int c = 0; c < QG; ++c)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[c][g] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- end of a tile (workgroup-wide: every wave passes here once per tile)
    auto tile_end = [&](int tile) {
        if (have_thr) {
            append_tile(acc, tile);
        } else {
            if (nheld == 0) held_tile0 = tile;
#pragma unroll
            for (int c = 0; c < QG; ++c)
#pragma unroll
                for (int g = 0; g < G; ++g) hold[(size_t)((nheld * QG + c) * G + g) * 256] = acc[c][g];
            ++nheld;
        }
        if (a.seed && tile == t0) {
            // publish the best row of this wave's first tile, per query (composites; 0 = no valid row)
#pragma unroll
            for (int cg = 0; cg < QG; ++cg) {
                const int q = 16 * cg + (lane & 15);
                u64 c = 0ull;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const long long row0 = (long long)tile * ST_BM + rsub + 16 * g;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u64 v = (q < a.nq && row0 + i < a.n)
                                          ? (((u64)f2key(acc[cg][g][i]) << 32) | (u64)(0xFFFFFFFFu - (u32)(row0 + i))) : 0ull;
                        c = v > c ? v : c;
                    }
                }
#pragma unroll
                for (int o = 32; o >= 16; o >>= 1) {     // over the lanes that hold this query
                    const u64 v = __shfl_xor(c, o);
                    c = v > c ? v : c;
                }
                if (lane < 16 && q < a.nq)
                    __hip_atomic_store(a.bmax + (size_t)q * (4 * a.strips) + 4 * strip + wave, c, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            wait_vmcnt<0>();            // this wave's stores have left (once per launch: the ring drains with them)
        }
#pragma unroll
        for (int c = 0; c < QG; ++c)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[c][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        wg_barrier();                   // tile boundary: every wave has appended / published
        // threshold exchange with the selection shared out, as in stream16_kernel (sync words: [0] arrivals,
        // [1] queries selected, [2 + q] thresholds, [S64_CLAIM] next query to select)
        if (!got_global) {
            if (tid == 0) {
                u32 arr = 0u;
                if (tile == t0) arr = __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                else arr = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u32 done = __hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                u32 claim = 0xFFFFFFFFu;
                if (arr == (u32)a.strips && done < (u32)a.nq) {
                    const u32 c = __hip_atomic_fetch_add(a.sync + S64_CLAIM, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (c < (u32)a.nq) claim = c;
                }
                sFlag[2] = claim;
                sFlag[1] = done >= (u32)a.nq ? 1u : 0u;
            }
            wg_barrier();
            const u32 claim = sFlag[2];
            if (claim != 0xFFFFFFFFu) {
                const int M = 4 * a.strips;          // <= 1024 (host)
                const int q = (int)claim + wave;
                if (q < a.nq) {
                    u64 x[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int idx = e * 64 + lane;
                        x[e] = idx < M ? __hip_atomic_load(a.bmax + (size_t)q * M + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : 0ull;
                    }
                    int valid = 0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) valid += __popcll(__ballot(x[e] != 0ull));
                    const u32 T = valid >= a.k ? wave_kth_key<16>(x, a.k) : 0u;
                    if (lane == 0)
                        __hip_atomic_store(a.sync + 2 + q, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                wait_vmcnt<0>();
                wg_barrier();
                if (tid == 0)
                    __hip_atomic_fetch_add(a.sync + 1, (u32)min(4, a.nq - (int)claim), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (sFlag[1] != 0u && tid < a.nq) {
                const u32 t = __hip_atomic_load(a.sync + 2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sThr[tid] = max(sThr[tid], t);
            }
            wg_barrier();
            const bool ready = sFlag[1] != 0u;
            got_global = ready;
            if (!have_thr && (ready || nheld == HOLD || tile + 1 == t1)) {
                // flush the held tiles through the filter (without thresholds: one tile at a time, compacting)
                for (int h = 0; h < nheld; ++h) {
                    f32x4 sc[QG][G];
#pragma unroll
                    for (int c = 0; c < QG; ++c)
#pragma unroll
                        for (int g = 0; g < G; ++g) sc[c][g] = hold[(size_t)((h * QG + c) * G + g) * 256];
                    append_tile(sc, held_tile0 + h);
                    wg_barrier();
                    if (__builtin_amdgcn_readfirstlane(sFlag[0])) compact_pass(false);
                }
                nheld = 0;
                have_thr = true;
            } else if (have_thr && __builtin_amdgcn_readfirstlane(sFlag[0])) {
                compact_pass(false);
            }
        } else if (__builtin_amdgcn_readfirstlane(sFlag[0])) {
            compact_pass(false);
        }
    };

    // ---- main loop.  K-slab ks of tile t is the 32 KiB block t KS + ks of the tile-blocked plane; this wave's 64
    // rows of it are 8 KiB of consecutive bytes.  The ring is three register stages and the loop body is exactly
    // three steps, so every stage keeps its registers across the back edge (no rotation copies, which would have
    // to wait for loads in flight); a tile is walked in KSP = 3 ceil(KS / 3) steps, the steps past KS load the
    // last slab again and skip the products.
    const u32 fr = (u32)(lane & 15), fq = (u32)(lane >> 4);
    const char* Xw = reinterpret_cast<const char*>(a.X) + ((u32)wave * 8192u + (u32)lane * 16u);
    const int KSP = (KS + 2) / 3 * 3;
    // loads: instruction i reads the 1 KiB of rows 8 i .. 8 i + 7 whole (lane l: row 8 i + (l >> 3), chunk l & 7) --
    // full 128-byte lines per 8 lanes; MFMA operand layout straight from memory would be 16 lines per 16 lanes
    auto load_stage = [&](f32x4 (&x)[2 * G], int tile, int ks) {       // (ks may run up to KSP + 1: into the next tile)
        if (ks >= KSP) { ks -= KSP; ++tile; }
        tile = min(tile, t1 - 1);
        ks = min(ks, KS - 1);
        const char* p = Xw + ((size_t)((u32)tile * (u32)KS + (u32)ks) << 15);
#pragma unroll
        for (int i = 0; i < 2 * G; ++i) x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i * 1024));
    };
    // staging: the stage being consumed goes through this wave's own 8 KiB of LDS to reach operand layout
    // (row r, chunk c at r * 128 + ((c ^ ((r >> 1) & 7)) << 4): conflict-free both ways); same wave, LDS executes
    // a wave's instructions in order: no barrier
    char* sX = smem + qbytes + wave * 8192;
    const u32 wr0 = (u32)(lane >> 3) * 128u + ((((u32)lane & 7u) ^ ((u32)(lane >> 4) & 7u)) << 4);          // rows 8 i, i even
    const u32 wr1 = (u32)(lane >> 3) * 128u + ((((u32)lane & 7u) ^ ((4u + (u32)(lane >> 4)) & 7u)) << 4);   // i odd
    const u32 rswz = (fr >> 1) & 7u;
    const char* sQl = smem + fr * pitch;
    const char* sXl = sX + fr * 128u;
    auto step = [&](const f32x4 (&cur)[2 * G], f32x4 (&nxt)[2 * G], int tile, int ks) {
        load_stage(nxt, tile, ks + 2);
        if (ks < KS) {
#pragma unroll
            for (int i = 0; i < 2 * G; ++i)
                *reinterpret_cast<f32x4*>(sX + i * 1024 + ((i & 1) ? wr1 : wr0)) = cur[i];
            const char* sQ = sQl + ks * 128;
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                const u32 co = ((4u * ks2 + fq) ^ rswz) << 4;
                f32x4 qf[QG], xf[G];
#pragma unroll
                for (int c = 0; c < QG; ++c) qf[c] = *reinterpret_cast<const f32x4*>(sQ + (u32)c * 16u * pitch + co);
#pragma unroll
                for (int g = 0; g < G; ++g) xf[g] = *reinterpret_cast<const f32x4*>(sXl + g * 2048 + co);
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int c = 0; c < QG; ++c)
                        acc[c][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xf[g]),
                                                                            __builtin_bit_cast(bf16x8, qf[c]), acc[c][g], 0, 0, 0);
            }
        }
    };
    if (t1 > t0) {
        f32x4 x0[2 * G], x1[2 * G], x2[2 * G];
        load_stage(x0, t0, 0);
        load_stage(x1, t0, 1);
        for (int tile = t0; tile < t1; ++tile) {
            for (int ks = 0; ks < KSP; ks += 3) {
                step(x0, x2, tile, ks);
                step(x1, x0, tile, ks + 1);
                step(x2, x1, tile, ks + 2);
            }
            tile_end(tile);
        }
    }
    compact_pass(true);
    if (tid < NQP) a.cnt[(size_t)strip * NQP + tid] = min(sCnt[tid], (u32)CAP);
}

struct Scan16Args {
    const char* X;      // corpus, bf16, tile-blocked plane of cap_rows x ldk elements (plane_index)
    const char* Q;      // queries, bf16, tile-blocked plane of nq_pad x ldk elements
    u64* cand;          // [S][nq_pad][CAP]
    u32* cnt;           // [S][nq_pad]
    long long n;
    int ldk;            // row length in bf16 elements: ld (MODE 0) or 2 * ld (MODE 1), multiple of 64
    int nq, nq_pad, k;
    int qtiles, strips, total_tiles;
    int tile_begin;     // first tile of this launch (tiles below belong to the sample pass)
    const u32* thr0;    // [nq_pad] initial score-key threshold per query (null: none)
    int skip_final;     // 1: leave the strip's candidates as they are (<= CAP per query); the merge selects
    u32* bmax = nullptr;   // BMAX instance: [nq_pad][4 * strips] score keys, the best row of every wave's 64 rows per query
    int tile_stride = 1;   // BMAX instance: strip s scans the one tile s * tile_stride (a sample spread over the corpus)
};

constexpr int S16_HT = 16384, S16_BUF = 4 * S16_HT;
constexpr int S16_AF = 0, S16_BF = S16_HT, S16_BS = 2 * S16_HT, S16_AS = 3 * S16_HT;
constexpr int S16_BM = 256, S16_BQ = 256;

template <int MODE, int EPL, bool BMAX = false>
__global__ __launch_bounds__(512, 2) void scan16_kernel(Scan16Args a) {
    constexpr int CAP = EPL * 64;
    static_assert(CAP - S16_BM >= 128, "candidate buffer too small");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32* sCnt = reinterpret_cast<u32*>(smem + 2 * S16_BUF);
    u32* sThr = sCnt + S16_BQ;
    u32* sFlag = sThr + S16_BQ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;

    int wg;
    {
        const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wg = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + idx;
    }
    // (the division runs on the vector ALU: readfirstlane brings the uniform results back to scalar registers,
    // which the DMA bases derived from them must be in)
    const int strip = __builtin_amdgcn_readfirstlane(wg / a.qtiles);           // an XCD's workgroups share corpus strips
    const int qt = wg - strip * a.qtiles;
    const int q0 = qt * S16_BQ;
    const int t0 = BMAX ? strip * a.tile_stride : a.tile_begin + (int)(((long long)strip * a.total_tiles) / a.strips);
    const int t1 = BMAX ? t0 + 1 : a.tile_begin + (int)(((long long)(strip + 1) * a.total_tiles) / a.strips);
    const int KT = a.ldk >> 6;
    const int total_stages = (t1 - t0) * KT;

    for (int i = tid; i < S16_BQ; i += 512) {
        sCnt[i] = 0;
        sThr[i] = (q0 + i < a.nq) ? (a.thr0 ? a.thr0[q0 + i] : 0u) : 0xFFFFFFFFu;
    }
    if (tid == 0) *sFlag = 0;
    u64* candBase = a.cand + ((size_t)strip * a.nq_pad + q0) * CAP;

    // ---- producer (see gemm_bf16.hip): "A" side = queries, "B" side = corpus rows
    u32 voffA[2], voffB[2];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
        const int r = 8 * (wave * 2 + i2) + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int arow = (r & 63) + 128 * (r >> 6);
        const int wrow = 64 * (r >> 5) + (r & 31);
        voffA[i2] = (u32)arow * 128u + (u32)c * 16u;          // inside one 32 KiB K-slab block (plane_index)
        voffB[i2] = (u32)wrow * 128u + (u32)c * 16u;
    }
    const u32 lds0 = (u32)(uintptr_t)(AS3 char*)smem;
    const u32 dma_dst = (u32)(wave * 2) * 1024u;
    const char* gQ = a.Q + ((size_t)(u32)(qt * KT) << 15);
    int is_stage = 0, is_t = 0, is_tile = t0;
    bool is_valid = false;
    const char* is_gA = gQ;
    const char* is_gW = a.X;
    u32 is_lds = lds0;
    auto issue = [&](int kind) -> bool {
        if (kind == 0) {
            is_valid = is_stage < total_stages;
            if (is_valid) {
                is_gA = gQ + ((size_t)(u32)is_t << 15);
                is_gW = a.X + ((size_t)(u32)(is_tile * KT + is_t) << 15);     // block index < 2^31 (host check)
                is_lds = lds0 + (u32)(is_stage & 1) * S16_BUF + dma_dst;
                ++is_stage;
                if (++is_t == KT) { is_t = 0; ++is_tile; }
            }
        }
        if (is_valid) {
            if (kind == 0) {
                dma16s(is_gA, voffA[0], is_lds + S16_AF);
                dma16s(is_gA, voffA[1], is_lds + S16_AF + 1024);
            } else if (kind == 1) {
                dma16s(is_gW, voffB[0], is_lds + S16_BF);
                dma16s(is_gW, voffB[1], is_lds + S16_BF + 1024);
            } else if (kind == 2) {
                const char* b = is_gW + 32 * 128;
                dma16s(b, voffB[0], is_lds + S16_BS);
                dma16s(b, voffB[1], is_lds + S16_BS + 1024);
            } else {
                const char* b = is_gA + 64 * 128;
                dma16s(b, voffA[0], is_lds + S16_AS);
                dma16s(b, voffA[1], is_lds + S16_AS + 1024);
            }
        }
        return is_valid;
    };
    auto stage_and_wait = [&](int kind) {
        if (issue(kind)) wait_vmcnt<8>();
        else wait_vmcnt<0>();
    };
    auto bar = []() { asm volatile("s_barrier" ::: "memory"); };

    // ---- consumer
    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[m][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32 xrow = (u32)(64 * g + fr) * 128u;
    const u32 wrow = (u32)(32 * wn + fr) * 128u;
    const u32 coff0 = (u32)((fq) ^ (fr >> 1)) << 4;
    const u32 coff1 = (u32)((4 + fq) ^ (fr >> 1)) << 4;
    float4 X0[4][2], X1[4][2], Wf[2][2][2];
    auto ldX = [&](float4 (&X)[4][2], const char* base) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            X[mb][0] = *reinterpret_cast<const float4*>(base + xrow + mb * 2048 + coff0);
            X[mb][1] = *reinterpret_cast<const float4*>(base + xrow + mb * 2048 + coff1);
        }
    };
    auto ldW = [&](int nq, const char* base) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            Wf[nq][nb][0] = *reinterpret_cast<const float4*>(base + wrow + nb * 2048 + coff0);
            Wf[nq][nb][1] = *reinterpret_cast<const float4*>(base + wrow + nb * 2048 + coff1);
        }
    };
    auto mfma = [&](f32x4& c, const float4& w, const float4& x) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
    };
    // a batch that ends inside the first 128 queries of this tile leaves group 1 (queries 128 ... 255) with padding
    // only: its products are skipped -- the barriers are not, the groups stay in step -- and group 0 has the matrix
    // cores to itself (65 ... 128 queries: the scan goes from 256 columns' time to about 128 columns')
    const bool idle_group = g == 1 && a.nq - q0 <= 128;
    auto mma = [&](float4 (&X)[4][2], int mq, int nq) {
        if (idle_group) return;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (MODE == 0) {
            // chunk 0 / 1 of the 128-B row = k 0..31 / 32..63 of this K-step
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) mfma(acc[4 * mq + mb][2 * nq + nb], Wf[nq][nb][ks], X[mb][ks]);
        } else {
            // chunk 0 / 1 = hi / lo planes of the same 32 values of k; W = corpus rows, X = queries:
            // x_hi q_hi + x_hi q_lo + x_lo q_hi (MODE 1); MODE 2 leaves out x_hi q_lo (error |q - q_hi| |x|)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) mfma(acc[4 * mq + mb][2 * nq + nb], Wf[nq][nb][0], X[mb][0]);
            if (MODE == 1) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) mfma(acc[4 * mq + mb][2 * nq + nb], Wf[nq][nb][0], X[mb][1]);
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) mfma(acc[4 * mq + mb][2 * nq + nb], Wf[nq][nb][1], X[mb][0]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // threshold filter of one finished tile: lane = query (8 of them), 16 corpus rows per query
    auto epilogue = [&](int tile) {
        const long long rbase = (long long)tile * S16_BM + 64 * wn + 4 * fq;
        const int lim = (int)std::min<long long>(a.n - rbase, 1 << 20);     // this lane's rows rbase + j are real for j < lim
        // The lane's first query, made opaque here: every address below is derived from it INSIDE the epilogue.
        // Left to itself the compiler computes the eight LDS addresses and eight 64-bit row pointers once,
        // before the main loop, spills them, and reloads them at every use behind a wait that also drains the
        // tile loads in flight.
        int qb = 128 * g + fr;
        asm volatile("" : "+v"(qb));
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int ql = qb + 16 * m;
            const u32 thr = sThr[ql];
            const float thrf = key2f(thr);
            const u32 qslot = (u32)ql * (u32)CAP;      // 32-bit slot index against the uniform base
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const f32x4 v = acc[m][nb];
                const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                if (thr == 0u || mx > thrf || mx != mx) {
                    // the lane's survivors among its 4 scores take their slots with ONE LDS atomic
                    u32 key[4];
                    bool hit[4];
                    u32 c = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        key[i] = f2key(v[i]);
                        hit[i] = key[i] > thr && 16 * nb + i < lim;
                        c += hit[i] ? 1u : 0u;
                    }
                    if (c) {
                        u32 pos = atomicAdd(&sCnt[ql], c);
                        const u32 low0 = 0xFFFFFFFFu - (u32)(rbase + 16 * nb);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (hit[i]) {
                                if (pos < (u32)CAP) candBase[qslot + pos] = ((u64)key[i] << 32) | (u64)(low0 - (u32)i);
                                ++pos;
                            }
                        if (pos > (u32)(CAP - S16_BM)) *sFlag = 1;
                    }
                }
            }
        }
    };
    // BMAX instance (level 0 of the optimistic start): no candidates at all -- per query only the best score among
    // this wave's 64 rows of the tile goes out, one key per (query, strip, wave column): the sample the optimistic
    // threshold is guessed from (bmax_thr_kernel).  127 MB of dumped candidates and a 0.14 ms merge per 2214-query
    // search become 1 MB and a 5 us kernel; the main scan then covers the sample tiles too.
    auto bmax_epilogue = [&](int tile) {
        const long long rbase = (long long)tile * S16_BM + 64 * wn + 4 * fq;
        const int lim = (int)std::min<long long>(a.n - rbase, 1 << 20);
        const int M = 4 * a.strips;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int ql = 128 * g + fr + 16 * m;
            float mx = -INFINITY;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (16 * nb + i < lim) mx = fmaxf(mx, acc[m][nb][i]);
            u32 key = f2key(mx);
            key = max(key, (u32)__shfl_xor((int)key, 16));
            key = max(key, (u32)__shfl_xor((int)key, 32));
            if (fq == 0 && q0 + ql < a.nq) a.bmax[(size_t)(q0 + ql) * M + 4 * strip + wn] = key;
        }
    };
    // (the accumulators are cleared after a possible compaction, not in the epilogue: 128 registers that
    // are dead while the compaction runs)
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[m][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto compact_pass = [&](bool final_pass) {
        wait_vmcnt<0>();
        wg_barrier();
        // a compaction is one dependent round trip to L2 (the entries) followed by the select: the next
        // query's entries are requested before the current query's select runs
        u64 xc[EPL];
        int qc = -1;
        for (int q = wave; q < S16_BQ; q += 8) {
            const int n = (int)sCnt[q];
            const bool need = final_pass ? (n > a.k) : (n > CAP - S16_BM);
            if (!need) continue;
            u64 xn[EPL];
            wave_compact_load<EPL>(xn, candBase + (size_t)q * CAP, n, lane);
            if (qc >= 0) {
                const u64 T = wave_compact_select<EPL>(xc, candBase + (size_t)qc * CAP, a.k, lane);
                if (lane == 0) {
                    sCnt[qc] = (u32)a.k;
                    sThr[qc] = (u32)(T >> 32);
                }
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) xc[e] = xn[e];
            qc = q;
        }
        if (qc >= 0) {
            const u64 T = wave_compact_select<EPL>(xc, candBase + (size_t)qc * CAP, a.k, lane);
            if (lane == 0) {
                sCnt[qc] = (u32)a.k;
                sThr[qc] = (u32)(T >> 32);
            }
        }
        if (tid == 0) *sFlag = 0;
        wait_vmcnt<0>();
        wg_barrier();
    };

    if (total_stages > 0) {
        issue(0); issue(1); issue(2); issue(3);
        issue(0);
        if (issue(1)) wait_vmcnt<8>();
        else wait_vmcnt<0>();
    }
    __syncthreads();
    if (total_stages > 0) {
        ldX(X0, smem + S16_AF);
        if (g == 1) bar();               // group 1 runs one barrier behind
        int c_t = 0, c_tile = t0;
        for (int cs = 0; cs < total_stages; ++cs) {
            const char* lb = smem + (cs & 1) * S16_BUF;
            ldW(0, lb + S16_BF);
            stage_and_wait(2);
            bar();
            mma(X0, 0, 0);
            bar();
            ldW(1, lb + S16_BS);
            stage_and_wait(3);
            bar();
            mma(X0, 0, 1);
            bar();
            ldX(X1, lb + S16_AS);
            stage_and_wait(0);
            bar();
            mma(X1, 1, 1);
            bar();
            if (cs + 1 < total_stages) ldX(X0, smem + ((cs + 1) & 1) * S16_BUF + S16_AF);
            stage_and_wait(1);
            bar();
            mma(X1, 1, 0);
            if (++c_t == KT) {
                c_t = 0;
                if (g == 1) { if (BMAX) bmax_epilogue(c_tile); else epilogue(c_tile); }
                wg_barrier();
                if (g == 0) {
                    if (BMAX) bmax_epilogue(c_tile); else epilogue(c_tile);
                    wg_barrier();        // group 0 waits for group 1: every wave is past its epilogue
                }
                if (__builtin_amdgcn_readfirstlane(*sFlag)) compact_pass(false);   // same barriers for all waves
                zero_acc();
                ++c_tile;
                if (g == 1 && cs + 1 < total_stages) bar();   // fall one barrier behind again
            } else {
                bar();
            }
        }
        // after the last tile group 1 did not fall behind again: the groups are level here
    }
    // (a one-tile strip of the first sample level has nothing to gain from a compaction of 32 queries per
    // wave, one after the other: the merge kernel takes all its <= 256 rows per query)
    if (a.skip_final) {
        wait_vmcnt<0>();
        wg_barrier();
    } else {
        compact_pass(true);
    }
    // (the block-maxima instance has no candidates: its strips are sample tiles, not rows of cand / cnt -- there may be
    // more of them than the scan has strips)
    if (!BMAX)
        for (int i = tid; i < S16_BQ; i += 512) a.cnt[(size_t)strip * a.nq_pad + q0 + i] = sCnt[i];
}

// fp32 rows [rows][d] -> bf16 rows [rows_pad][ld] (MODE 0), zero padded
// Plane layout of everything scan16_kernel streams (corpus and query planes alike): tile-blocked.
// Element e (0 .. ldk-1) of row R lives at
//     (((R >> 8) * (ldk / 64) + (e >> 6)) * 256 + (R & 255)) * 64 + (e & 63)
// i.e. the 64-element K-slab of a 256-row tile is ONE contiguous 32 KiB block (256 rows x 128 B): the
// LDS-DMA of a half-tile reads 16 KiB of consecutive addresses instead of 128 rows x 128 B at a 3 KiB
// stride (HBM pages, L2 channel spread).  Buffers hold whole 256-row tiles.
__device__ __forceinline__ size_t plane_index(long long R, int e, int ldk) {
    return ((size_t)((R >> 8) * (ldk >> 6) + (e >> 6)) * 256 + (size_t)(R & 255)) * 64 + (size_t)(e & 63);
}

// `dst` is the plane base, `row0` the first destination row (incremental adds)
__global__ void to_bf16_rows_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long long rows,
                                    long long rows_pad, int d, int ld, long long row0) {
    const long long total = rows_pad * ld;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / ld;
        const int c = (int)(i - r * ld);
        const float v = (r < rows && c < d) ? src[r * d + c] : 0.0f;
        dst[plane_index(row0 + r, c, ld)] = __builtin_bit_cast(unsigned short, (__bf16)v);
    }
}

// fp32 rows [rows][sld] (first d valid) -> split rows of 2 ld elements: per 32 values of k, 32 hi then 32 lo
// (MODE 1), in the tile-blocked plane layout from row `row0` of the plane `dst` on; optionally the squared
// norm of every row (norm2[rows_pad], indexed from 0) for the error bound.
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ src, long long sld,
                                                         unsigned short* __restrict__ dst, long long rows,
                                                         long long rows_pad, int d, int ld, float* norm2,
                                                         long long row0, float* res2 = nullptr,
                                                         unsigned short* __restrict__ hi_only = nullptr) {
    const int lane = threadIdx.x & 63;
    const long long w0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = w0; r < rows_pad; r += nw) {      // one wave per row
        float ss = 0.f, rr = 0.f;
        for (int c = lane; c < ld; c += 64) {
            const float v = (r < rows && c < d) ? src[r * sld + c] : 0.0f;
            const __bf16 hi = (__bf16)v;
            const __bf16 lo = (__bf16)(v - (float)hi);
            rr += (v - (float)hi) * (v - (float)hi);
            unsigned short* o = dst + plane_index(row0 + r, (c >> 5) * 64 + (c & 31), 2 * ld);   // lo: +32, same slab
            o[0] = __builtin_bit_cast(unsigned short, hi);
            o[32] = __builtin_bit_cast(unsigned short, lo);
            if (hi_only) hi_only[plane_index(row0 + r, c, ld)] = __builtin_bit_cast(unsigned short, hi);   // (to_bf16_rows_kernel's plane)
            ss += v * v;
        }
        if (norm2) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
            if (lane == 0) norm2[r] = ss;
        }
        if (res2) {      // |v - hi|^2: what a product without the lo plane of this row leaves out
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) rr += __shfl_xor(rr, o);
            if (lane == 0) res2[r] = rr;
        }
    }
}

// |row - bf16(row)|^2 of fp32 rows [rows][ld] (one wave per row): what a product on rounded rows leaves out
__global__ __launch_bounds__(256) void residual_norm_kernel(const float* __restrict__ src, long long rows, int ld,
                                                            float* __restrict__ res2, float* __restrict__ nrm2) {
    // nrm2 (optional): |x|^2 as well -- an index that keeps only the hi plane has no split_rows_kernel pass to take it from
    const int lane = threadIdx.x & 63;
    const long long w0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = w0; r < rows; r += nw) {
        float rr = 0.f, nn = 0.f;
        for (int c = lane; c < ld; c += 64) {
            const float v = src[r * ld + c];
            const float e = v - (float)(__bf16)v;
            rr += e * e;
            nn += v * v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            rr += __shfl_xor(rr, o);
            nn += __shfl_xor(nn, o);
        }
        if (lane == 0) {
            res2[r] = rr;
            if (nrm2) nrm2[r] = nn;
        }
    }
}

// max over rows of norm2 (non-negative floats order like their bit patterns; NaN patterns rank above +inf)
__global__ void max_norm_kernel(const float* norm2, long long rows, u32* out) {
    u32 m = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < rows; i += (long long)gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(norm2[i]) & 0x7FFFFFFFu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (u32)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// one wave per query: valid ids of the list to the front (order kept), -1 behind them; ids that are neither
// -1 nor rows of this index are counted in *bad
__global__ void subset_ids_kernel(const long long* __restrict__ in, long long* __restrict__ out, int m,
                                  long long id_base, long long n, u32* bad) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const long long* src = in + (size_t)q * m;
    long long* dst = out + (size_t)q * m;
    int kept = 0, nbad = 0;
    for (int c0 = 0; c0 < m; c0 += 64) {
        const int i = c0 + lane;
        const long long id = i < m ? src[i] : -1ll;
        const bool ok = id >= id_base && id - id_base < n;
        if (i < m && !ok && id != -1ll) ++nbad;
        const unsigned long long bal = __ballot(ok);
        if (ok) dst[kept + __popcll(bal & ((1ull << lane) - 1ull))] = id;
        kept += __popcll(bal);
    }
    for (int i = kept + lane; i < m; i += 64) dst[i] = -1ll;
    if (nbad) atomicAdd(bad, (u32)nbad);
}

// Exact scores of the coarse candidates: one wave per (query, 32 candidates), the fp32 matrix
// cores fed exactly like scan_kernel (16-byte fragments, k order 0,4,1,5,2,6,3,7 per 8), the
// query in every column, so every output is bit for bit the score scan_kernel computes.
// Writes composites (exact key, row) into out[s][q][slot] with s = candidate / k.
struct RescoreArgs {
    const float* X;        // [n][ld]
    const float* Q;        // [nq_pad][ld] (zero padded)
    const long long* Ic;   // [nq][kc] coarse ids (local row + id_base), -1 = none
    u64* out;              // [S2][nq_pad][k]
    u32* out_cnt;          // [S2][nq_pad]
    long long id_base;
    int ld, kc, k, nq_pad;
    u32* zero_word;        // a counter to leave at zero for the kernel after this one (null: none)
    int wg_per_q;          // workgroups that share a query's candidates (few queries: spread them over more CUs)
};

// The candidate rows are gathered through LDS.  Loading the 32x32x2 fragments straight from global memory touches 32
// rows x 32 bytes per instruction (a lane group owns a row's k-pair); here a load instruction takes whole 128-byte
// lines -- 8 lanes per row, 8 rows per instruction -- into registers, the wave writes them to its own LDS buffer in
// the swizzled row image the stream kernels use, and reads the fragments from there (round 1's direct form: 0.37 ms
// for 2214 x 211 rows of 768 floats, this one 0.30; 1.79 -> 1.60 ms at d = 2304, 3.03 -> 2.73 ms at k = 1000).
// No barrier: a wave only ever touches its own two 4 KiB buffers.
template <bool QLDS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 3))) void rescore_lds_kernel(RescoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) char rsm[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = (int)(blockDim.x >> 6);
    const int q = blockIdx.x / a.wg_per_q, part = blockIdx.x - q * a.wg_per_q;
    const int h = lane >> 5, l31 = lane & 31;
    const float* qrow = a.Q + (size_t)q * a.ld;
    char* wbuf = rsm + wave * 8192;                       // [2][32 rows][128 B]
    float* sQ = reinterpret_cast<float*>(rsm + nw * 8192);
    if (QLDS) {
        for (int i = threadIdx.x; i < a.ld / 4; i += (int)blockDim.x)
            reinterpret_cast<float4*>(sQ)[i] = reinterpret_cast<const float4*>(qrow)[i];
        __syncthreads();
    }
    const int S = a.ld / 32;                              // steps of 32 floats (ld is a multiple of 64: S is even)
    const int lr = lane >> 3, lc = lane & 7;              // this lane's row within an instruction's 8, its 16-byte chunk
    for (int c0 = (part * nw + wave) * 32; c0 < a.kc; c0 += a.wg_per_q * nw * 32) {
        const int ci = c0 + l31;
        const long long id = ci < a.kc ? a.Ic[(size_t)q * a.kc + ci] : -1ll;
        const long long row = id >= 0 ? id - a.id_base : 0;
        // rows 8 i + lr (i = 0..3) are the ones this lane fetches: their ids sit in lanes 8 i + lr
        const f32x4* src[4];
        u32 wofs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + lr;
            const long long rr = __shfl(row, r);
            src[i] = reinterpret_cast<const f32x4*>(a.X + (size_t)rr * a.ld) + lc;
            wofs[i] = (u32)r * 128u + (u32)((lc ^ ((r >> 1) & 7)) << 4);
        }
        const u32 rofs = (u32)l31 * 128u;
        const u32 rswz = (u32)((l31 >> 1) & 7);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        f32x4 preA[4], preB[4];      // (native vectors: arrays of HIP's float4 class stay in private memory)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            preA[i] = src[i][0];
            preB[i] = src[i][8];                          // (step s: + 8 s float4 = 128 bytes per step)
        }
        auto step = [&](int s2, f32x4 (&pre)[4], char* buf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(buf + wofs[i]) = pre[i];
            if (s2 + 2 < S) {
#pragma unroll
                for (int i = 0; i < 4; ++i) pre[i] = src[i][8 * (s2 + 2)];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 x = *reinterpret_cast<const float4*>(buf + rofs + ((((u32)(2 * j + h)) ^ rswz) << 4));
                const float4 qf = QLDS ? *reinterpret_cast<const float4*>(sQ + 32 * s2 + 8 * j + 4 * h)
                                       : *reinterpret_cast<const float4*>(qrow + 32 * s2 + 8 * j + 4 * h);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, qf.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, qf.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, qf.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, qf.w, acc, 0, 0, 0);
            }
        };
        for (int s2 = 0; s2 < S; s2 += 2) {
            step(s2, preA, wbuf);
            step(s2 + 1, preB, wbuf + 4096);
        }
        // lane (column l31, half h) holds rows (r & 3) + 8 (r >> 2) + 4 h; all columns are equal:
        // column 0's two lanes publish the 32 scores
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rt = (r & 3) + 8 * (r >> 2) + 4 * h;
            const long long idr = __shfl(id, rt);                 // candidate rt is held by lane rt (h = 0 half)
            const int cr = c0 + rt;
            if (l31 == 0 && cr < a.kc) {
                const int s = cr / a.k, slot = cr - s * a.k;
                const u64 comp = idr >= 0 ? (((u64)f2key(acc[r]) << 32) | (u64)(0xFFFFFFFFu - (u32)(idr - a.id_base))) : 0ull;
                a.out[((size_t)s * a.nq_pad + q) * a.k + slot] = comp;
            }
        }
    }
    if (a.zero_word && blockIdx.x == 0 && threadIdx.x == 0) *a.zero_word = 0u;
    // counts: invalid candidates (id < 0) are the tail of the list (one wave counts with ballots)
    if (wave == 0 && part == 0) {
        int valid = 0;
        for (int i0 = 0; i0 < a.kc; i0 += 64) {
            const int i = i0 + lane;
            valid += __popcll(__ballot(i < a.kc && a.Ic[(size_t)q * a.kc + i] >= 0));
        }
        const int S2 = (a.kc + a.k - 1) / a.k;
        for (int s = lane; s < S2; s += 64) a.out_cnt[(size_t)s * a.nq_pad + q] = (u32)max(0, min(a.k, valid - s * a.k));
    }
}

// The same gather with 64 candidates per wave and v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 rows x 4 columns x ONE k: a
// plain fma per output, lane l supplies row l, 8 cycles per k for 64 rows) instead of 32x32x2 with the query in all 32
// columns (64 cycles per two k for 32 rows: per 2214 x 211 candidates the chain of dependent MFMAs alone was 0.17 ms of
// matrix-pipe time in a 0.30 ms kernel).  Same k order (0,4,1,5,2,6,3,7 per 8), same bits (stream_kernel's instruction).
template <bool QLDS>
__global__ __launch_bounds__(512) void rescore64_kernel(RescoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) char rsm[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = (int)(blockDim.x >> 6);
    const int q = blockIdx.x / a.wg_per_q, part = blockIdx.x - q * a.wg_per_q;
    const float* qrow = a.Q + (size_t)q * a.ld;
    char* wbuf = rsm + wave * 16384;                      // [2][64 rows][128 B]
    float* sQ = reinterpret_cast<float*>(rsm + nw * 16384);
    if (QLDS) {
        for (int i = threadIdx.x; i < a.ld / 4; i += (int)blockDim.x)
            reinterpret_cast<float4*>(sQ)[i] = reinterpret_cast<const float4*>(qrow)[i];
        __syncthreads();
    }
    const int S = a.ld / 32;                              // steps of 32 floats (ld is a multiple of 64: S is even)
    const int lr = lane >> 3, lc = lane & 7;              // this lane's row within an instruction's 8, its 16-byte chunk
    for (int c0 = (part * nw + wave) * 64; c0 < a.kc; c0 += a.wg_per_q * nw * 64) {
        const int ci = c0 + lane;
        const long long id = ci < a.kc ? a.Ic[(size_t)q * a.kc + ci] : -1ll;
        const long long row = id >= 0 ? id - a.id_base : 0;
        // rows 8 i + lr (i = 0..7) are the ones this lane fetches: their ids sit in lanes 8 i + lr
        const f32x4* src[8];
        u32 wofs[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 8 * i + lr;
            const long long rr = __shfl(row, r);
            src[i] = reinterpret_cast<const f32x4*>(a.X + (size_t)rr * a.ld) + lc;
            wofs[i] = (u32)r * 128u + (u32)((lc ^ ((r >> 1) & 7)) << 4);
        }
        const u32 rofs = (u32)lane * 128u;
        const u32 rswz = (u32)((lane >> 1) & 7);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 preA[8], preB[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            preA[i] = src[i][0];
            preB[i] = src[i][8];                          // (step s: + 8 s float4 = 128 bytes per step)
        }
        auto step = [&](int s2, f32x4 (&pre)[8], char* buf) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(buf + wofs[i]) = pre[i];
            if (s2 + 2 < S) {
#pragma unroll
                for (int i = 0; i < 8; ++i) pre[i] = src[i][8 * (s2 + 2)];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 x0 = *reinterpret_cast<const float4*>(buf + rofs + ((((u32)(2 * j)) ^ rswz) << 4));
                const float4 x1 = *reinterpret_cast<const float4*>(buf + rofs + ((((u32)(2 * j + 1)) ^ rswz) << 4));
                const float4 q0 = QLDS ? *reinterpret_cast<const float4*>(sQ + 32 * s2 + 8 * j)
                                       : *reinterpret_cast<const float4*>(qrow + 32 * s2 + 8 * j);
                const float4 q1 = QLDS ? *reinterpret_cast<const float4*>(sQ + 32 * s2 + 8 * j + 4)
                                       : *reinterpret_cast<const float4*>(qrow + 32 * s2 + 8 * j + 4);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.x, q0.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(x1.x, q1.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.y, q0.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(x1.y, q1.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.z, q0.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(x1.z, q1.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.w, q0.w, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(x1.w, q1.w, acc, 0, 0, 0);
            }
        };
        for (int s2 = 0; s2 < S; s2 += 2) {
            step(s2, preA, wbuf);
            step(s2 + 1, preB, wbuf + 8192);
        }
        // lane l holds rows 4 (l >> 2) + i (i = 0..3) for column l & 3; all columns are equal: column 0 publishes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rt = 4 * (lane >> 2) + i;
            const long long idr = __shfl(id, rt);                 // candidate rt is held by lane rt
            const int cr = c0 + rt;
            if ((lane & 3) == 0 && cr < a.kc) {
                const int s = cr / a.k, slot = cr - s * a.k;
                const u64 comp = idr >= 0 ? (((u64)f2key(acc[i]) << 32) | (u64)(0xFFFFFFFFu - (u32)(idr - a.id_base))) : 0ull;
                a.out[((size_t)s * a.nq_pad + q) * a.k + slot] = comp;
            }
        }
    }
    if (a.zero_word && blockIdx.x == 0 && threadIdx.x == 0) *a.zero_word = 0u;
    // counts: invalid candidates (id < 0) are the tail of the list (one wave counts with ballots)
    if (wave == 0 && part == 0) {
        int valid = 0;
        for (int i0 = 0; i0 < a.kc; i0 += 64) {
            const int i = i0 + lane;
            valid += __popcll(__ballot(i < a.kc && a.Ic[(size_t)q * a.kc + i] >= 0));
        }
        const int S2 = (a.kc + a.k - 1) / a.k;
        for (int s = lane; s < S2; s += 64) a.out_cnt[(size_t)s * a.nq_pad + q] = (u32)max(0, min(a.k, valid - s * a.k));
    }
}

// 64-candidate groups over as few rounds as four waves allow (2 x 8 KiB of LDS per wave)
void launch_rescore64(const RescoreArgs& ra, unsigned grid, hipStream_t st) {
    const size_t qbytes = (size_t)ra.ld * 4;
    const bool qlds = qbytes <= 32768;
    const int max_w = 4;
    const int groups = std::max(1, (ra.kc + 63) / 64);
    RescoreArgs a2 = ra;
    int nw;
    if ((long long)grid * groups <= 1024) {
        a2.wg_per_q = groups;        // few queries: a query's groups spread over as many CUs
        nw = 1;
    } else {
        const int rounds = (groups + max_w - 1) / max_w;
        a2.wg_per_q = 1;
        nw = (groups + rounds - 1) / rounds;
    }
    const size_t lds = (size_t)nw * 16384 + (qlds ? qbytes : 0);
    static bool attr_set[64][2] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev][qlds ? 1 : 0]) {
        if (qlds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rescore64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384 + 32768);
        else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rescore64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
        attr_set[dev][qlds ? 1 : 0] = true;
    }
    if (qlds) hipLaunchKernelGGL(rescore64_kernel<true>, dim3(grid * a2.wg_per_q), dim3(64 * nw), lds, st, a2);
    else hipLaunchKernelGGL(rescore64_kernel<false>, dim3(grid * a2.wg_per_q), dim3(64 * nw), lds, st, a2);
}

// up to 8 waves per query, one 32-candidate group each; the query row staged in LDS when it fits beside the buffers
void launch_rescore(const RescoreArgs& ra, unsigned grid, hipStream_t st) {
    {
        const char* ev = getenv("OTTER_RESCORE32");       // A/B runs: the 32x32x2 form
        if (!(ev && ev[0] == '1')) return launch_rescore64(ra, grid, st);
    }
    // waves = 32-candidate groups spread evenly over as few rounds as the waves allow (10 groups: 2 rounds of 5);
    // at most 8 waves, and no more than leave room for the query row beside their buffers in 64 KiB of LDS
    const size_t qbytes = (size_t)ra.ld * 4;
    const bool qlds = qbytes <= 32768;
    const int max_w = qlds ? std::min(8, (int)((65536 - qbytes) / 8192)) : 8;
    const int groups = std::max(1, (ra.kc + 31) / 32);
    RescoreArgs a2 = ra;
    int nw;
    if ((long long)grid * groups <= 1024) {
        // few queries: one wave per workgroup, a query's groups spread over as many CUs (1 query x 211 candidates:
        // one workgroup took 28 us, latency-bound on its one CU)
        a2.wg_per_q = groups;
        nw = 1;
    } else {
        const int rounds = (groups + max_w - 1) / max_w;
        a2.wg_per_q = 1;
        nw = (groups + rounds - 1) / rounds;
    }
    if (qlds)
        hipLaunchKernelGGL(rescore_lds_kernel<true>, dim3(grid * a2.wg_per_q), dim3(64 * nw), (size_t)nw * 8192 + qbytes, st, a2);
    else
        hipLaunchKernelGGL(rescore_lds_kernel<false>, dim3(grid * a2.wg_per_q), dim3(64 * nw), (size_t)nw * 8192, st, a2);
}

// A query's exact top k is proven complete when its k-th exact score exceeds every score a row
// OUTSIDE the kc coarse candidates can have: (smallest kept coarse score) + eps,
// eps = rel * |q| * max |x|.  Otherwise (or on any non-finite value) the query is flagged.
// qres2 (two-product coarse pass): |q - q_hi|^2; that pass leaves out x . (q - q_hi), at most |q - q_hi| |x|.
__global__ void margin_check_kernel(const float* Dexact, const float* Dcoarse, const long long* Icoarse,
                                    const float* qnorm2, const u32* xnorm2_max, float rel, long long nq, int k, int kc,
                                    long long n, u32* flags, u32* nflag, const float* qres2,
                                    const u32* xres2_max, const u32* opt_thr, u32* ticket, u32* host_word) {
    const long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    bool ok = true;
    if (q >= nq) {
    } else if (n <= kc) {
        ok = true;                                              // every row is a candidate
    } else {
        const float tc = Dcoarse[q * kc + kc - 1];             // n > kc: the coarse list is full
        const float ek = Dexact[q * k + k - 1];
        const float xmax = sqrtf(__uint_as_float(*xnorm2_max));
        float eps = rel * sqrtf(qnorm2[q]) * xmax;
        if (qres2) eps += 1.001f * sqrtf(qres2[q]) * xmax;
        // one product: q.x - q_hi.x_hi = (q - q_hi).x + q_hi.(x - x_hi), |q_hi| <= (1 + 2^-8) |q|
        if (xres2_max) eps += 1.005f * sqrtf(qnorm2[q]) * sqrtf(__uint_as_float(*xres2_max));
        ok = Icoarse[q * kc + kc - 1] >= 0 && ek > tc + eps;    // false on NaN / inf bounds
        // the coarse scan started from an OPTIMISTIC threshold (a guess at the kc-th best score from a small sample):
        // rows whose key lies below it were never looked at, so the coarse list is the true top kc only if its last
        // entry reaches the threshold
        if (opt_thr && f2key(tc) < opt_thr[q]) {
            ok = false;
            if (ticket) atomicAdd(ticket - 1, 1u);       // counted apart: the guess failed, not the coarse score's bound
        }
    }
    if (q < nq) flags[q] = ok ? 0u : 1u;
    if (!ok) atomicAdd(nflag, 1u);
    if (!host_word) return;
    // the last block to finish hands the count of unproven queries to the host, which polls this pinned word: no copy,
    // no stream synchronisation (the ticket is left at zero)
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const u32 t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            u32 nf = __hip_atomic_load(nflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // bit 31: more than a quarter of the batch missed its optimistic threshold (the word before the ticket counts
            // them; both are left at zero)
            const u32 nopt = __hip_atomic_load(ticket - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned long long)nopt * 4ull > (unsigned long long)nq) nf |= 0x80000000u;
            __hip_atomic_store(ticket - 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(host_word, nf, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- small batches (at most 64 queries): re-scoring, final selection and proof in ONE launch ----
// The reference's call pattern is one query per call: behind the 0.24 ms scan, re-scoring (18 us: a chain of 384
// dependent 64-cycle fp32 MFMAs per candidate group), the merge of its output (9 us), the margin check (5 us), the
// copy of the proof word (4 us) and a stream synchronisation were four latency-bound launches.  Here:
//   * grid = queries x parts, a workgroup re-scores 32 candidates: its four waves bring the rows in as 1 KiB K-blocks
//     (a wave-load = one row's 256 floats, fully coalesced, up to three blocks = 24 loads per lane in flight -- a
//     whole 768-float row), lay them into LDS rows pitched 1040 bytes, and 32 lanes of wave 0 run the plain fmaf
//     chain, lane = candidate, in scan_kernel's k order (0,4,1,5,2,6,3,7 per 8: the oracle's chain, bit for bit);
//   * composites go to global scratch with agent-scope stores; the LAST workgroup of a query (an agent-scope ticket)
//     reads all kc of them back, orders them by rank counting, writes (D, I), and decides the query's proof
//     (margin_check_kernel's rule);
//   * the last query to finish writes the number of unproven queries straight into pinned host memory (system-scope
//     store), which the host polls: no copy, no stream synchronisation.  All counters are left at zero.
struct FinishArgs {
    const float* X;        // [n][ld]
    const float* Q;        // [nq][ld] (zero padded rows)
    const long long* Ic;   // [nq][kc] coarse ids (local row + id_base), -1 = none (the tail)
    const float* Dc;       // [nq][kc] coarse scores
    float* D;              // [nq][k]
    long long* I;          // [nq][k]
    u64* comp;             // [nq][kc] scratch
    u32* tick;             // [nq] arrivals per query, [nq] queries finished | unproven queries << 16; zero at launch and after
    u32* flags;            // [nq] 1 = unproven
    u32* host_word;        // pinned, host-coherent
    const float* qnorm2;
    const u32* xnorm2_max;
    const float* qres2;    // or null
    const u32* xres2_max;  // or null
    float rel;
    long long id_base, n;
    int ld, kc, k, nq, parts;
};

constexpr int FIN_ROWS = 32, FIN_KB = 256, FIN_PITCH = FIN_KB * 4 + 16, FIN_STAGES = 3, FIN_MAX_KC = 320;

__global__ __launch_bounds__(256) void finish_small_kernel(FinishArgs a) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    char* stage = fsm;                                                        // [FIN_STAGES][32 rows][1040 B]
    float* sQ = reinterpret_cast<float*>(fsm + FIN_STAGES * FIN_ROWS * FIN_PITCH);   // [ld]
    __shared__ long long sRow[FIN_ROWS];
    __shared__ u64 sComp[FIN_MAX_KC];
    __shared__ u32 sLast, sValid;
    __shared__ float sEk;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.x / a.parts, part = blockIdx.x - q * a.parts;
    const int c0 = part * FIN_ROWS;
    if (tid < FIN_ROWS) {
        const int ci = c0 + tid;
        const long long id = ci < a.kc ? a.Ic[(size_t)q * a.kc + ci] : -1ll;
        sRow[tid] = id >= 0 ? id - a.id_base : -1ll;
    }
    for (int i = tid; i < a.ld / 4; i += 256)
        reinterpret_cast<f32x4*>(sQ)[i] = reinterpret_cast<const f32x4*>(a.Q + (size_t)q * a.ld)[i];
    __syncthreads();
    const int S = a.ld / FIN_KB + ((a.ld % FIN_KB) ? 1 : 0);     // K-blocks (ld is a multiple of 64: the last may be short)
    // pass i of a block: wave w brings row 4 i + w (lane l its 16 bytes l); rows without a candidate read row 0
    const float* src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long long r = sRow[4 * i + wave];
        src[i] = a.X + (size_t)(r >= 0 ? r : 0) * a.ld + lane * 4;
    }
    f32x4 buf[FIN_STAGES][8];
    auto load_block = [&](f32x4 (&x)[8], int b) {
        const bool in = b * FIN_KB + lane * 4 < a.ld;             // (a short last block: its tail is never read)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            x[i] = in ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src[i] + b * FIN_KB)) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    float acc = 0.0f;
#pragma unroll
    for (int s = 0; s < FIN_STAGES; ++s)
        if (s < S) load_block(buf[s], s);
    for (int g0 = 0; g0 < S; g0 += FIN_STAGES) {
#pragma unroll
        for (int s = 0; s < FIN_STAGES; ++s)
            if (g0 + s < S) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    *reinterpret_cast<f32x4*>(stage + (s * FIN_ROWS + 4 * i + wave) * FIN_PITCH + lane * 16) = buf[s][i];
            }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < FIN_STAGES; ++s)
            if (g0 + FIN_STAGES + s < S) load_block(buf[s], g0 + FIN_STAGES + s);      // in flight during the chain
        if (wave == 0 && lane < FIN_ROWS) {
            for (int s = 0; s < FIN_STAGES && g0 + s < S; ++s) {
                const char* st = stage + (s * FIN_ROWS + lane) * FIN_PITCH;
                const float* qb = sQ + (g0 + s) * FIN_KB;
                const int n8 = min(FIN_KB, a.ld - (g0 + s) * FIN_KB) / 8;
#pragma unroll 4
                for (int j = 0; j < n8; ++j) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(st + j * 32);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(st + j * 32 + 16);
                    const f32x4 q0 = *reinterpret_cast<const f32x4*>(qb + j * 8);
                    const f32x4 q1 = *reinterpret_cast<const f32x4*>(qb + j * 8 + 4);
                    acc = __builtin_fmaf(x0[0], q0[0], acc);
                    acc = __builtin_fmaf(x1[0], q1[0], acc);
                    acc = __builtin_fmaf(x0[1], q0[1], acc);
                    acc = __builtin_fmaf(x1[1], q1[1], acc);
                    acc = __builtin_fmaf(x0[2], q0[2], acc);
                    acc = __builtin_fmaf(x1[2], q1[2], acc);
                    acc = __builtin_fmaf(x0[3], q0[3], acc);
                    acc = __builtin_fmaf(x1[3], q1[3], acc);
                }
            }
        }
        __syncthreads();
    }
    // ---- publish this part's composites; the last part of the query goes on
    if (wave == 0) {
        if (lane < FIN_ROWS && c0 + lane < a.kc) {
            const long long r = sRow[lane];
            const u64 c = r >= 0 ? (((u64)f2key(acc) << 32) | (u64)(0xFFFFFFFFu - (u32)r)) : 0ull;
            __hip_atomic_store(a.comp + (size_t)q * a.kc + c0 + lane, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        wait_vmcnt<0>();                  // this wave's stores have left
        if (lane == 0) {
            const u32 t = __hip_atomic_fetch_add(a.tick + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sLast = (t == (u32)a.parts - 1u) ? 1u : 0u;
        }
    }
    __syncthreads();
    if (sLast == 0u) return;
    for (int t = tid; t < a.kc; t += 256)
        sComp[t] = __hip_atomic_load(a.comp + (size_t)q * a.kc + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
        sValid = 0u;
        sEk = -FLT_MAX;
        __hip_atomic_store(a.tick + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // left at zero for the next call
    }
    __syncthreads();
    const int k = a.k;
    for (int t = tid; t < a.kc; t += 256) {
        const u64 v = sComp[t];
        if (v != 0ull) {
            int r = 0;
            for (int j = 0; j < a.kc; ++j) r += sComp[j] > v;       // (composites of distinct rows are distinct)
            atomicAdd(&sValid, 1u);
            if (r < k) {
                const float sc = key2f((u32)(v >> 32));
                a.D[(size_t)q * k + r] = sc;
                a.I[(size_t)q * k + r] = a.id_base + (long long)(0xFFFFFFFFu - (u32)v);
                if (r == k - 1) sEk = sc;
            }
        }
    }
    __syncthreads();
    const int nv = (int)sValid;
    for (int t = nv + tid; t < k; t += 256) {        // fewer candidates than k: padding
        a.D[(size_t)q * k + t] = -FLT_MAX;
        a.I[(size_t)q * k + t] = -1ll;
    }
    if (tid == 0) {
        bool ok;
        if (a.n <= a.kc) {
            ok = true;
        } else {
            const float tc = a.Dc[(size_t)q * a.kc + a.kc - 1];
            const float ek = sEk;
            const float xmax = sqrtf(__uint_as_float(*a.xnorm2_max));
            float eps = a.rel * sqrtf(a.qnorm2[q]) * xmax;
            if (a.qres2) eps += 1.001f * sqrtf(a.qres2[q]) * xmax;
            if (a.xres2_max) eps += 1.005f * sqrtf(a.qnorm2[q]) * sqrtf(__uint_as_float(*a.xres2_max));
            ok = a.Ic[(size_t)q * a.kc + a.kc - 1] >= 0 && ek > tc + eps;
        }
        a.flags[q] = ok ? 0u : 1u;
        // one word counts both: queries finished (low half) and queries left unproven (high half)
        const u32 old = __hip_atomic_fetch_add(a.tick + a.nq, 1u + (ok ? 0u : 0x10000u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((old & 0xFFFFu) == (u32)a.nq - 1u) {
            const u32 nf = (old >> 16) + (ok ? 0u : 1u);
            __hip_atomic_store(a.tick + a.nq, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.host_word, nf, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Unproven queries of a coarse pass: `which` lists them (ascending).  gather: rows of q -> a dense batch;
// scatter: the dense batch's result rows back to their places in (D, I).
__global__ void compact_flags_kernel(const u32* __restrict__ flags, long long nq, int* __restrict__ which) {
    // one workgroup of 1024 threads: order-preserving compaction of the flagged query numbers
    __shared__ int s_base, s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (long long q0 = 0; q0 < nq; q0 += 1024) {
        const long long q = q0 + tid;
        const bool f = q < nq && flags[q] != 0u;
        const u64 m = __ballot(f);
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        if (f) which[off + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0))] = (int)q;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += s_w[w];
            s_base += t;
        }
        __syncthreads();
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ which, float* __restrict__ dst,
                                   int d) {
    const float* s = src + (size_t)which[blockIdx.x] * d;
    float* o = dst + (size_t)blockIdx.x * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) o[c] = s[c];
}

__global__ void scatter_results_kernel(const float* __restrict__ Ds, const long long* __restrict__ Is,
                                       const int* __restrict__ which, float* __restrict__ D, long long* __restrict__ I,
                                       int k) {
    const size_t so = (size_t)blockIdx.x * k, to = (size_t)which[blockIdx.x] * k;
    for (int c = threadIdx.x; c < k; c += blockDim.x) {
        D[to + c] = Ds[so + c];
        I[to + c] = Is[so + c];
    }
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
    u64* out;            // [nq_pad][k]  unsorted list of the survivors (non-final)
    u32* out_cnt;        // [nq_pad]
    float* D;            // [nq][k]                   (final level)
    long long* I;        // [nq][k]
    long long id_base;
    int cap_in;
    int S_in;
    int nq_pad;
    int k;
    int final_level;
    int take;            // entries gathered from each strip list (k; more only for uncompacted one-tile strips)
    const u64* ex;       // last level only: one more sorted list [nq_pad][k] (the sample passes so far), or null;
                         // may alias `out` (a workgroup has gathered its query's lists before it writes)
    const u32* ex_cnt;   // [nq_pad]
    u32* thr_out;        // non-final: [nq_pad] key of the k-th best, 0 if fewer than k
    int thr_rank;        // 0: as above; r > 0: thr_out = key of the r-th best instead (0 if fewer than r) -- an OPTIMISTIC
                         // threshold for a scan of many more rows, whose result the caller has to verify
    u32* zero_words;     // the stream kernel's arrival counter, ready flag and thresholds (ST_SYNC_WORDS), left zero; or null
};

constexpr int MAX_SELECT = 8192;       // composites one merge workgroup holds in LDS
constexpr int MERGE_MAX_STRIPS = 256;    // strip counts one merge workgroup scans at a time

// ---- merge: one workgroup per query, ONE launch for any number of strips ----
// The strips' candidate lists are gathered into an LDS pool of CAPL composites; whenever the next strips
// would not fit, the pool is cut back to its k largest (select) and the gather goes on.  With seeded
// thresholds the lists are short and one round takes everything.
//   select    distribution select on the 64-bit composites (unique, non-zero): min / max of the live range,
//             256 bins over it (LDS atomics), the bin that holds the k-th largest becomes the next range
//             (256x narrower each pass); a range that holds <= 256 candidates is finished by rank counting.
//   output    final level: the <= k survivors are ordered by rank counting (every survivor counts the
//             survivors above it: k LDS broadcast reads, no barrier) and written as (D, I);
//             otherwise the unsorted list, its count and the key of the k-th best (thr_out) are left for a
//             later merge -- nothing downstream needs an intermediate list ordered.
template <int NT, int CAPL>
__global__ __launch_bounds__(NT) void merge_kernel(MergeArgs a) {
    constexpr int EPT = CAPL / NT;
    constexpr int NWV = NT / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* l = reinterpret_cast<u64*>(smem);                                   // [CAPL]
    u64* tmp = l + CAPL;                                                     // [256] candidates of the last range
    int* sOff = reinterpret_cast<int*>(tmp + 256);                           // [MERGE_MAX_STRIPS + 2]
    u32* hist = reinterpret_cast<u32*>(sOff + MERGE_MAX_STRIPS + 2);         // [256]
    u64* sRed = reinterpret_cast<u64*>(hist + 256);                          // [2 * NWV] per-wave min / max
    u64* sT = sRed + 2 * NWV;                                                // [1] selected threshold
    int* sI = reinterpret_cast<int*>(sT + 1);                                // [8] bin, above, m, position counters
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.k;

    // cut the pool l[0, n) back to its k largest (n > k), compacted into l[0, k); returns their minimum
    auto select = [&](int n) -> u64 {
        u64 x[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int idx = e * NT + tid;
            x[e] = idx < n ? l[idx] : 0ull;
        }
        // live range [lo, hi] of the composites that still compete; `need` of them are to be kept
        u64 mn = ~0ull, mx = 0ull;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (x[e]) { mn = x[e] < mn ? x[e] : mn; mx = x[e] > mx ? x[e] : mx; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u64 a2 = __shfl_xor(mn, o), b2 = __shfl_xor(mx, o);
            mn = a2 < mn ? a2 : mn;
            mx = b2 > mx ? b2 : mx;
        }
        if (lane == 0) { sRed[wave] = mn; sRed[NWV + wave] = mx; }
        __syncthreads();
        u64 lo = sRed[0], hi = sRed[NWV];
#pragma unroll
        for (int w = 1; w < NWV; ++w) {
            lo = sRed[w] < lo ? sRed[w] : lo;
            hi = sRed[NWV + w] > hi ? sRed[NWV + w] : hi;
        }
        int need = k;
        u64 T = lo;
        for (int pass = 0; pass < 16; ++pass) {         // (64 bits / 8 bits per pass = 8 passes at most)
            const u64 range = hi - lo;
            const int bits = range ? 64 - __builtin_clzll(range) : 0;
            const int shift = bits > 8 ? bits - 8 : 0;
            if (tid < 256) hist[tid] = 0u;
            if (tid == 0) sI[3] = 0;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (x[e] >= lo && x[e] <= hi) atomicAdd(&hist[(u32)((x[e] - lo) >> shift)], 1u);
            __syncthreads();
            if (wave == 0) {
                // suffix sums over 256 bins, 4 per lane: the bin B with  sum(bins > B) < need <= sum(bins >= B)
                const u32 h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                u32 own = h0 + h1 + h2 + h3, suf = own;       // inclusive suffix over lanes
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const u32 u = __shfl_down(suf, o);
                    if (lane + o < 64) suf += u;
                }
                const u32 after = suf - own;                   // candidates in the bins of higher lanes
                if (after < (u32)need && suf >= (u32)need) {   // exactly one lane
                    u32 ab = after;
                    int B = 4 * lane + 3;
                    u32 m = h3;
                    if (ab + h3 >= (u32)need) { B = 4 * lane + 3; m = h3; }
                    else if (ab + h3 + h2 >= (u32)need) { ab += h3; B = 4 * lane + 2; m = h2; }
                    else if (ab + h3 + h2 + h1 >= (u32)need) { ab += h3 + h2; B = 4 * lane + 1; m = h1; }
                    else { ab += h3 + h2 + h1; B = 4 * lane; m = h0; }
                    sI[0] = B;
                    sI[1] = (int)ab;
                    sI[2] = (int)m;
                }
            }
            __syncthreads();
            const int B = sI[0], above = sI[1], m = sI[2];
            const u64 nlo = lo + ((u64)B << shift);
            const u64 nhi = shift ? nlo + ((1ull << shift) - 1ull) : nlo;
            lo = nlo;
            hi = nhi < hi ? nhi : hi;
            need -= above;
            if (m == need) { T = lo; break; }                  // the whole bin is kept
            if (m <= 256) {
                // finish by rank counting among the bin's candidates
#pragma unroll
                for (int e = 0; e < EPT; ++e)
                    if (x[e] >= lo && x[e] <= hi) tmp[atomicAdd(&sI[3], 1)] = x[e];
                __syncthreads();
                if (tid < m) {
                    const u64 v = tmp[tid];
                    int r = 0;        // (equal composites -- a caller's repeated ids -- rank by position)
                    for (int j = 0; j < m; ++j) r += (tmp[j] > v) || (tmp[j] == v && j < tid);
                    if (r == need - 1) *sT = v;
                }
                __syncthreads();
                T = *sT;
                break;
            }
        }
        // keep the k largest, compacted to the front of the pool: everything above T, then composites equal
        // to T up to k in all (composites are unique except for a caller's repeated ids)
        if (tid == 0) sI[4] = 0;
        __syncthreads();                                       // (every thread holds its elements: l may be overwritten)
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const bool keep = x[e] > T;
            const u64 mk = __ballot(keep);
            int base = 0;
            if (lane == 0 && mk) base = atomicAdd(&sI[4], __popcll(mk));
            base = __shfl(base, 0);
            if (keep) l[base + __builtin_amdgcn_mbcnt_hi((u32)(mk >> 32), __builtin_amdgcn_mbcnt_lo((u32)mk, 0))] = x[e];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const bool keep = x[e] == T;
            const u64 mk = __ballot(keep);
            int base = 0;
            if (lane == 0 && mk) base = atomicAdd(&sI[4], __popcll(mk));
            base = __shfl(base, 0);
            const int pos = base + __builtin_amdgcn_mbcnt_hi((u32)(mk >> 32), __builtin_amdgcn_mbcnt_lo((u32)mk, 0));
            if (keep && pos < k) l[pos] = x[e];
        }
        __syncthreads();
        return T;
    };

    int n = 0;
    u64 Tmin = 0ull;             // minimum of the pool when it was last cut to k (0: never cut)
    if (a.ex) {
        n = min((int)a.ex_cnt[q], k);
        const u64* src = a.ex + (size_t)q * k;
        for (int i = tid; i < n; i += NT) l[i] = src[i];
    }
    for (int s0 = 0; s0 < a.S_in; s0 += MERGE_MAX_STRIPS) {
        const int ns = min(MERGE_MAX_STRIPS, a.S_in - s0);
        __syncthreads();
        for (int t = tid; t < ns; t += NT) sOff[t + 1] = min((int)a.in_cnt[(size_t)(s0 + t) * a.nq_pad + q], a.take);
        __syncthreads();
        if (wave == 0) {   // inclusive scan of <= MERGE_MAX_STRIPS counts by one wave
            int run = 0;
            for (int c0 = 0; c0 < ns; c0 += 64) {
                int v = c0 + lane < ns ? sOff[c0 + lane + 1] : 0;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int u = __shfl_up(v, o);
                    if (lane >= o) v += u;
                }
                if (c0 + lane < ns) sOff[c0 + lane + 1] = run + v;
                run += __shfl(v, 63);
            }
            if (lane == 0) sOff[0] = 0;
        }
        __syncthreads();
        int sb = 0;
        while (sb < ns) {
            // the strips [sb, se) that fit beside the n composites already in the pool (a strip holds <= take
            // <= CAPL - k entries and n <= k after a cut: at least one strip fits)
            const int room = CAPL - n + sOff[sb];
            int se = sb + 1;
            {
                int lo2 = sb + 1, hi2 = ns;                    // largest se with sOff[se] <= room
                while (lo2 < hi2) {
                    const int mid = (lo2 + hi2 + 1) >> 1;
                    if (sOff[mid] <= room) lo2 = mid; else hi2 = mid - 1;
                }
                se = lo2;
            }
            {
                // flat gather: entry idx of the round belongs to the strip t with sOff[t] <= base + idx < sOff[t + 1];
                // every thread's loads are in flight together (a strip-by-strip copy is a chain of dependent loads)
                const int base = sOff[sb], tot = sOff[se] - base;
                for (int idx = tid; idx < tot; idx += NT) {
                    int lo2 = sb, hi2 = se - 1;
                    while (lo2 < hi2) {
                        const int mid = (lo2 + hi2 + 1) >> 1;
                        if (sOff[mid] <= base + idx) lo2 = mid; else hi2 = mid - 1;
                    }
                    l[n + idx] = a.in[((size_t)(s0 + lo2) * a.nq_pad + q) * a.cap_in + (base + idx - sOff[lo2])];
                }
            }
            n += sOff[se] - sOff[sb];
            sb = se;
            __syncthreads();
            const bool more = sb < ns || s0 + MERGE_MAX_STRIPS < a.S_in;
            if (more && n > k) { Tmin = select(n); n = k; }
        }
    }
    __syncthreads();
    if (n > k) { Tmin = select(n); n = k; }

    if (a.zero_words && q == 0 && tid < ST_SYNC_WORDS) a.zero_words[tid] = 0u;
    if (a.final_level) {
        // order the survivors by rank counting and write (D, I); positions past n are padding
        for (int t = tid; t < k; t += NT) {
            if (t < n) {
                const u64 v = l[t];
                int r = 0;
                for (int j = 0; j < n; ++j) r += (l[j] > v) || (l[j] == v && j < t);
                a.D[(size_t)q * k + r] = key2f((u32)(v >> 32));
                a.I[(size_t)q * k + r] = a.id_base + (long long)(0xFFFFFFFFu - (u32)v);
            } else {
                a.D[(size_t)q * k + t] = -FLT_MAX;
                a.I[(size_t)q * k + t] = -1ll;
            }
        }
    } else {
        u64* dst = a.out + (size_t)q * k;
        for (int t = tid; t < n; t += NT) dst[t] = l[t];
        if (a.thr_out && n >= k && Tmin == 0ull) {
            // exactly k gathered and never cut: their minimum by a block reduction
            u64 mn = ~0ull;
            for (int t = tid; t < n; t += NT) mn = l[t] < mn ? l[t] : mn;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const u64 a2 = __shfl_xor(mn, o);
                mn = a2 < mn ? a2 : mn;
            }
            if (lane == 0) sRed[wave] = mn;
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < NWV; ++w) mn = sRed[w] < mn ? sRed[w] : mn;
                Tmin = mn;
            }
        }
        if (tid == 0) {
            a.out_cnt[q] = (u32)n;
            if (a.thr_out && a.thr_rank <= 0) a.thr_out[q] = n >= k ? (u32)(Tmin >> 32) : 0u;
        }
        if (a.thr_out && a.thr_rank > 0) {
            // the r-th best of the <= k survivors by rank counting (composites are unique: exactly one has r - 1 above it)
            if (n < a.thr_rank) {
                if (tid == 0) a.thr_out[q] = 0u;
            } else {
                for (int t = tid; t < n; t += NT) {
                    const u64 v = l[t];
                    int r = 0;
                    for (int j = 0; j < n; ++j) r += l[j] > v;
                    if (r == a.thr_rank - 1) a.thr_out[q] = (u32)(v >> 32);
                }
            }
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

// ------------------------------------------------------------------ in-batch score matrix
// The dev-set evaluation of the fine-tuning script (reference scripts/train_1hop_tb_retrieval.py:304-319;
// the same matrix feeds loss_single, retrieval/criterions.py:15-17): row i of `product` holds q_i . c_j for
// every block j of the batch and, last, q_i . neg_c_i; the target of row i is column i.  rank[i] is the
// number of columns that sort before the target in a descending order (an equal score in a lower column
// sorts first: argmax's first-index rule), so accuracy = mean(rank == 0) and MRR = mean(1 / (rank + 1)).
// One workgroup per row; one wave per score, lanes strided over d, fp32 fma partial sums and a butterfly.
struct InbatchArgs {
    const float* q;      // [B][d]
    const float* c;      // [B][d]
    const float* neg;    // [B][d] or null
    float* product;      // [B][ncol]
    int* rank;           // [B]
    int B, d, ncol;
};

__global__ __launch_bounds__(256) void inbatch_kernel(InbatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* srow = reinterpret_cast<float*>(smem);         // [ncol]
    int* scount = reinterpret_cast<int*>(srow + a.ncol);
    const int i = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* qi = a.q + (size_t)i * a.d;
    if (threadIdx.x == 0) *scount = 0;
    for (int j = wave; j < a.ncol; j += 4) {
        const float* cj = j < a.B ? a.c + (size_t)j * a.d : a.neg + (size_t)i * a.d;
        float s = 0.0f;
        for (int e = lane; e < a.d; e += 64) s = fmaf(qi[e], cj[e], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
            srow[j] = s;
            a.product[(size_t)i * a.ncol + j] = s;
        }
    }
    __syncthreads();
    const float t = srow[i];
    int before = 0;
    for (int j = threadIdx.x; j < a.ncol; j += 256) {
        const float v = srow[j];
        before += (v > t) || (v == t && j < i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    if (lane == 0 && before) atomicAdd(scount, before);
    __syncthreads();
    if (threadIdx.x == 0) a.rank[i] = *scount;
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
    static const char* sv = getenv("OTTER_SCAN_SMALL");      // A/B switch for tuning runs (nq <= 32)
    if (nq <= 32 && sv && sv[0] == 'a') return make_cfg<4, 1, 1, 1, 32, 7, EPL>();   // 128 rows, 7 x 20 KB
    if (nq <= 32 && sv && sv[0] == 'b') return make_cfg<4, 1, 2, 1, 32, 4, EPL>();   // 256 rows, 4 x 36 KB
    if (nq <= 32 && sv && sv[0] == 'c') return make_cfg<4, 1, 1, 1, 32, 5, EPL>();   // 128 rows, 5 x 20 KB
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

template <int G, int EPL, int BK, int NSTAGE>
void launch_stream(const StreamArgs& a, int grid, hipStream_t st) {
    auto kern = stream_kernel<G, EPL, BK, NSTAGE>;
    const int lds = 4 * NSTAGE * st_slot(G, BK) + 67 * (int)sizeof(u32);
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
}

template <int EPL, int NSTAGE>
void launch_stream32(const StreamArgs& a, int grid, hipStream_t st) {
    auto kern = stream32_kernel<EPL, NSTAGE>;
    const int lds = 4 * NSTAGE * st_slot(8, 32) + 67 * (int)sizeof(u32);
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
}

template <int QG, int EPL, int NSTAGE>
void launch_stream16(const StreamArgs& a, int grid, hipStream_t st) {
    auto kern = stream16_kernel<QG, EPL, NSTAGE>;
    const int lds = 4 * NSTAGE * (64 * 128 + 2 * QG * 1024) + 67 * (int)sizeof(u32);
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
}

template <int QG, int EPL>
void launch_stream64(const StreamArgs& a, int grid, hipStream_t st) {
    auto kern = stream64_kernel<QG, EPL>;
    const int lds = 16 * QG * (int)s64_pitch(a.ld) + (2 * 16 * QG + 4) * (int)sizeof(u32) + 4 * 8192;
    static int attr_lds[64] = {};        // (the image grows with ld: raise the limit when a longer row comes)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && attr_lds[dev] < lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_lds[dev] = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
}

struct Scan16Cfg {
    int cap;
    void (*launch)(const Scan16Args&, int grid, hipStream_t);
};

template <int MODE, int EPL>
void launch_scan16(const Scan16Args& a, int grid, hipStream_t st) {
    auto kern = scan16_kernel<MODE, EPL>;
    const int lds = 2 * S16_BUF + (2 * S16_BQ + 4) * (int)sizeof(u32);
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
}

// level 0 of the optimistic start: block maxima only (see bmax_epilogue)
void launch_scan16_bmax(const Scan16Args& a, int grid, hipStream_t st) {
    auto kern = scan16_kernel<0, 8, true>;
    const int lds = 2 * S16_BUF + (2 * S16_BQ + 4) * (int)sizeof(u32);
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
}

// one wave per query: the key of the rank-th best of its M <= 256 block maxima (0 = fewer than rank blocks)
constexpr int BMAX_THR_MAX = 256;
__global__ __launch_bounds__(64) void bmax_thr_kernel(const u32* __restrict__ bmax, int M, int rank, u32* __restrict__ thr) {
    const int q = blockIdx.x, lane = threadIdx.x;
    u64 x[BMAX_THR_MAX / 64];
#pragma unroll
    for (int e = 0; e < BMAX_THR_MAX / 64; ++e) {
        const int i = e * 64 + lane;
        x[e] = i < M ? ((u64)bmax[(size_t)q * M + i] << 32) | 1ull : 0ull;
    }
    const u32 T = M >= rank ? wave_kth_key<BMAX_THR_MAX / 64>(x, rank) : 0u;
    if (lane == 0) thr[q] = T;
}

// smallest rank r with P(Poisson(mu) >= r) <= p: the sample of an optimistic start holds Poisson(mu) rows above the true
// kk-th best score of the corpus (mu = kk n0 / N); a threshold at the sample's r-th best is too high only if it holds r
int poisson_rank(double mu, double p) {
    for (int r = 1; r <= 96; ++r) {
        double term = std::exp(-mu);
        for (int j = 1; j <= r; ++j) term *= mu / j;            // P(X = r)
        double tail = 0.0;
        for (int j = r; j < r + 100; ++j) {
            tail += term;
            term *= mu / (j + 1);
        }
        if (tail <= p) return r;
    }
    return 97;
}

// largest k the bf16-MFMA scan keeps per query (its compaction holds CAP / 64 composites per lane)
constexpr int S16_MAX_K = 1280;   // (1280: top-1000 of an fp32 index plus its coarse-pass margin of k/8)

template <int MODE>
int pick_cfg16(int k, Scan16Cfg* out) {
    if (k <= 160) *out = Scan16Cfg{8 * 64, &launch_scan16<MODE, 8>};   // 512 slots: kk + a 256-row tile fit
    else if (k <= 320) *out = Scan16Cfg{12 * 64, &launch_scan16<MODE, 12>};
    else if (k <= 512) *out = Scan16Cfg{20 * 64, &launch_scan16<MODE, 20>};
    else if (k <= S16_MAX_K) *out = Scan16Cfg{36 * 64, &launch_scan16<MODE, 36>};
    else return set_err(OTTER_EINVAL, "search: k=%d exceeds %d, the bound of the bf16 scan", k, S16_MAX_K);
    return OTTER_OK;
}

int next_pow2(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace

// One persistent host thread per shard of otter_sharded_search (created at the shard's first sharded search, joined by
// otter_index_destroy): a search may wait for its GPU (proof read-back), and the GPUs must not wait for each other;
// creating and joining G - 1 threads on EVERY call cost as much as a one-query search.
struct ShardWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> task;
    bool has_task = false, done = true, quit = false;
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return has_task || quit; });
            if (quit) return;
            std::function<void()> t = std::move(task);
            has_task = false;
            lk.unlock();
            t();
            lk.lock();
            done = true;
            cv.notify_all();
        }
    }
    void submit(std::function<void()> t) {
        std::lock_guard<std::mutex> lk(m);
        task = std::move(t);
        has_task = true;
        done = false;
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

struct otter_index {
    int d = 0, ld = 0, dtype = OTTER_F32, device = 0, ncu = 256;
    long long n = 0, cap_rows = 0, id_base = 0;
    float* x = nullptr;             // OTTER_F32: [cap_rows][ld] fp32
    unsigned short* x16 = nullptr;  // OTTER_BF16: [cap_rows][ld] bf16
    // OTTER_F32, built on first use by a large query batch: hi/lo bf16 planes [cap_rows][2 ld],
    // squared row norms and their maximum (bit pattern) for the coarse pass's error bound
    otter::DevBuf xs, xnorm2, xnmax;
    long long xs_rows = 0, xs_cap = 0;
    bool xs_failed = false, no_split = false;
    long long last_fallback = 0;    // queries of the last split search that went to the exact kernel
    long long last_refined = 0;     // ... that the two-product pass could not prove (three-product pass or exact kernel)
    bool two_prod_off = false;      // the two-product pass failed for > 1/4 of a batch: this corpus starts with three
    otter::DevBuf qres2, fb2_q, fb2_D, fb2_I, fb_w, fb2_w, fb3_w;
    // one-product pass (stage 0): a third copy of the corpus, bf16 hi values only (half the fp32 bytes), the
    // largest |x - x_hi| over its rows for the error bound, and the rounded queries
    otter::DevBuf xh, xres2, xrmax, qh16, fb3_q, fb3_D, fb3_I;
    long long xh_rows = 0, xh_cap = 0;
    bool xh_failed = false, one_prod_off = false;
    bool opt_off = false;           // the optimistic start guessed too high for more than a quarter of a batch: sample levels until reset()
    int planes = 2;                 // otter_index_set_planes: 0 none, 1 the bf16 hi plane only, 2 all (hi + hi/lo)
    long long win_q[4] = {0, 0, 0, 0}, win_f[4] = {0, 0, 0, 0};   // small calls: queries seen / left unproven, per pass
    bool coarse_off = false;        // even the three-product pass left > 1/4 of a batch unproven (scores within rounding of each
                                    // other, e.g. near-duplicate rows): this corpus is searched by the exact kernel from then on
    long long last_stage0_unproven = 0;
    otter::DevBuf qs, qnorm2, cD, cI, rs, rscnt, flags, fb_q, fb_D, fb_I;
    otter::DevBuf qpad, cand, cnt, lvl[2], lvlcnt[2], smp, smpcnt, thr0, io_q, io_D, io_I, bmax, sync, hold, bmax16;
    bool sync_dirty = true;         // the stream kernel's counters are not known to be zero
    u32* h_poll = nullptr;          // pinned, host-coherent word the last kernel of a coarse pass stores its count of unproven queries into
    otter::DevBuf fin;              // that kernel's counters (zero between launches)
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending, ev_free;
    // otter_sharded_search: this shard's own stream and staging (queries in, candidates out); on shards[0] also
    // the gathered candidates of every shard and the levels of a hierarchical merge
    hipStream_t sh_stream = nullptr;
    ShardWorker* sh_worker = nullptr;
    otter::DevBuf sh_q, sh_D, sh_I, sh_gD, sh_gI, sh_tD, sh_tI;
};

// Scratch of otter_index_search_noalloc (include/otter_hip.h): one fixed region per GPU, part of the code
// object, so the entry point never allocates.  Layout = 512 bytes for the stream kernel's counters, then the bump
// allocation order of search_chunk_stream() for the largest shape that entry point accepts (32 queries, k <= 128,
// d <= 4096, 256 strips):  qpad 32 x 4096 x 4 | cand 256 x 32 x 512 x 8 | cnt | block maxima 32 x 1024 x 8
constexpr int NA_MAX_NQ = 32, NA_MAX_K = 128, NA_MAX_LD = 4096, NA_MAX_STRIPS = 256;
constexpr size_t NA_SCRATCH_BYTES = 34ull << 20;      // 34 MiB (32 queries: 256 strips x 32 x 512 composites = 32 MiB + the rest)
static __device__ __align__(256) unsigned char scratch[NA_SCRATCH_BYTES];

namespace {

// device address of `scratch` on each GPU (the array's name is a host-side stand-in, not a device address)
unsigned char* g_scratch_ptr[64] = {};

int resolve_scratch(int device) {
    if (device < 0 || device >= 64) return set_err(OTTER_EINVAL, "device %d outside [0, 64)", device);
    if (g_scratch_ptr[device]) return OTTER_OK;
    void* p = nullptr;
    OTTER_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));     // the device is current (DeviceGuard of the caller)
    if (!p) return set_err(OTTER_EHIP, "hipGetSymbolAddress(scratch) returned NULL on device %d", device);
    g_scratch_ptr[device] = (unsigned char*)p;
    return OTTER_OK;
}

// Where a search's device scratch comes from: the index's grow-only buffers (may call hipMalloc), or a fixed
// region handed out by a bump pointer (never allocates; running out is an error).
struct Scratch {
    unsigned char* fixed = nullptr;
    size_t fixed_bytes = 0, used = 0;
    int get(otter::DevBuf& b, size_t bytes, void** out) {
        if (!fixed) {
            const int rc = b.reserve(bytes);
            *out = b.p;
            return rc;
        }
        const size_t at = (used + 255) & ~(size_t)255;
        if (at + bytes > fixed_bytes)
            return set_err(OTTER_EINVAL, "search_noalloc: this shape needs more than the %zu-byte static scratch", fixed_bytes);
        *out = fixed + at;
        used = at + bytes;
        return OTTER_OK;
    }
};

int index_reserve_rows(otter_index* ix, long long need_rows, hipStream_t st) {
    if (need_rows <= ix->cap_rows) return OTTER_OK;
    // capacity is kept a multiple of the largest scan tile (256 rows): the kernels read whole tiles
    // and mask rows >= n, so the tail of the last tile must be addressable
    const size_t esz = ix->dtype == OTTER_BF16 ? 2 : 4;
    void* old = ix->dtype == OTTER_BF16 ? (void*)ix->x16 : (void*)ix->x;
    need_rows = round_up64(need_rows, 256);
    long long want = ix->cap_rows == 0 ? need_rows : std::max(need_rows, ix->cap_rows + ix->cap_rows / 2);
    want = round_up64(want, 256);
    void* nx = nullptr;
    hipError_t e = hipMalloc(&nx, (size_t)want * ix->ld * esz);
    if (e != hipSuccess && want > need_rows) {
        (void)hipGetLastError();
        want = need_rows;
        e = hipMalloc(&nx, (size_t)want * ix->ld * esz);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return set_err(OTTER_ENOMEM, "add: hipMalloc of %lld rows x %d x %zu bytes failed: %s", want, ix->ld, esz,
                       hipGetErrorString(e));
    }
    // rows past n are read by whole-tile loads (and multiplied by real queries): keep them finite
    OTTER_HIP_TRY(hipMemsetAsync(nx, 0, (size_t)want * ix->ld * esz, st));
    if (ix->n > 0)
        // (whole tiles: the bf16 plane is tile-blocked, the tail rows of the last tile are zero)
        OTTER_HIP_TRY(hipMemcpyAsync(nx, old, (size_t)round_up64(ix->n, 256) * ix->ld * esz, hipMemcpyDeviceToDevice, st));
    OTTER_HIP_TRY(hipStreamSynchronize(st));
    if (old) OTTER_HIP_TRY(hipFree(old));
    if (ix->dtype == OTTER_BF16) ix->x16 = (unsigned short*)nx;
    else ix->x = (float*)nx;
    ix->cap_rows = want;
    return OTTER_OK;
}

// Merge the per-strip candidate lists of one scan (one launch, one workgroup per query).  An extra unsorted
// list (ex) may join.  final: write sorted (D, I); otherwise leave ONE unsorted list of <= k composites per
// query in list_out (which may be ex itself: a workgroup has read its query's ex list before it writes) and
// the score key of each query's k-th best in thr_out.
template <int NT, int CAPL>
int launch_merge(int device, const MergeArgs& ma, long long nq, hipStream_t st) {
    const size_t lds = (size_t)(CAPL + 256) * sizeof(u64) + (size_t)(MERGE_MAX_STRIPS + 2 + 256) * sizeof(int) +
                       (size_t)(2 * (NT / 64) + 1) * sizeof(u64) + 8 * sizeof(int);
    auto kern = merge_kernel<NT, CAPL>;
    static bool attr_set[64] = {};
    if (device >= 0 && device < 64 && !attr_set[device]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[device] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nq), dim3(NT), lds, st, ma);
    OTTER_HIP_TRY(hipGetLastError());
    return OTTER_OK;
}

int run_merge(otter_index* ix, const u64* in, const u32* in_cnt, int cap_in, int S_in, long long nq, long long nq_pad,
              int k, bool final, float* D, long long* I, const u64* ex, const u32* ex_cnt, u64* list_out,
              u32* list_cnt, u32* thr_out, hipStream_t st, int take_first = 0, Scratch* sc = nullptr,
              u32* zero_words = nullptr, int thr_rank = 0) {
    (void)sc;
    const int take = take_first > 0 ? take_first : std::min(k, cap_in);
    MergeArgs ma;
    ma.in = in;
    ma.in_cnt = in_cnt;
    ma.out = final ? nullptr : list_out;
    ma.out_cnt = final ? nullptr : list_cnt;
    ma.D = D;
    ma.I = I;
    ma.id_base = ix->id_base;
    ma.cap_in = cap_in;
    ma.S_in = S_in;
    ma.take = take;
    ma.nq_pad = (int)nq_pad;
    ma.k = k;
    ma.final_level = final ? 1 : 0;
    ma.ex = ex;
    ma.ex_cnt = ex_cnt;
    ma.thr_out = final ? nullptr : thr_out;
    ma.thr_rank = thr_rank;
    ma.zero_words = zero_words;
    // Four waves per query (cheap barriers; the pool lives in registers during a select), pool sized to the
    // job: a smaller pool lets more workgroups share a CU's LDS.
    static const char* mv = getenv("OTTER_MERGE_VARIANT");      // A/B switch for tuning runs
    if (mv && mv[0] == 'a') return launch_merge<1024, MAX_SELECT>(ix->device, ma, nq, st);
    if (mv && mv[0] == 'b') return launch_merge<512, MAX_SELECT>(ix->device, ma, nq, st);
    const long long all_in = (long long)S_in * take + k;
    if (all_in <= 1024) return launch_merge<256, 1024>(ix->device, ma, nq, st);
    if (mv && mv[0] == 'c') return launch_merge<512, 4096>(ix->device, ma, nq, st);
    // few queries: latency counts, 16 waves gather the strips in parallel; many queries: throughput counts, small
    // workgroups share a CU (measured: 1 ... 256 queries 1024 threads, 2214 queries 256 threads); a pool that
    // takes every strip at once needs one select round
    if (nq < 1024 && k + take <= MAX_SELECT) return launch_merge<1024, MAX_SELECT>(ix->device, ma, nq, st);
    // (a pool that takes every strip at once needs one select round, but its 68 KiB of LDS leave two workgroups per CU:
    // measured at 2214 queries x 28 strips, round 3: 0.096 -> 0.060 ms with the 4096-entry pool; kept behind 'e' for A/B runs)
    if (all_in > 4096 && all_in <= MAX_SELECT && mv && mv[0] == 'e') return launch_merge<512, MAX_SELECT>(ix->device, ma, nq, st);
    if (k + take <= 4096) return launch_merge<256, 4096>(ix->device, ma, nq, st);
    if (k + take > MAX_SELECT)
        return set_err(OTTER_EINVAL, "merge: k=%d + %d entries per strip exceed the %d-entry pool", k, take, MAX_SELECT);
    return launch_merge<512, MAX_SELECT>(ix->device, ma, nq, st);
}

struct Scratch;
int search_chunk(otter_index* ix, const float* qd, long long nq, int k, float* D, long long* I, hipStream_t st, Scratch* sc);
int search_chunk_bf16(otter_index* ix, const float* qd, long long nq, int k, float* D, long long* I, hipStream_t st);
int search_chunk_split(otter_index* ix, const float* qd, long long nq, int k, float* D, long long* I, hipStream_t st,
                       int mode, bool first);
int split_kc(int k, int mode, int d);
int ensure_split_planes(otter_index* ix, hipStream_t st);
int ensure_hi_plane(otter_index* ix, hipStream_t st);

// device-side dirty flags of the static scratch's stream-kernel counters (shared by every index of a GPU)
bool g_scratch_sync_dirty[64] = {};

// 33 ... 64 queries on a bf16 plane: the LDS image of the queries (stream64_kernel) has to fit beside the counters
bool stream64_fits(const otter_index* ix, long long nq) {
    static const char* off = getenv("OTTER_NO_STREAM64");
    return !off && nq > 32 && nq <= S64_NQP && (size_t)S64_NQP * s64_pitch(ix->ld) + 1024 + 4 * 8192 <= 160 * 1024;
}

// at most 32 queries (64 on a bf16 plane), k <= 128: the stream kernel + one merge (no sample pass, no pad kernel for aligned rows)
int search_chunk_stream(otter_index* ix, const float* qd, long long nq, int k, float* D, long long* I, hipStream_t st,
                        Scratch* sc, const void* q16_plane = nullptr, const void* x16_plane = nullptr) {
    // x16_plane: a bf16 plane of an OTTER_F32 index (its hi values: the one-product coarse pass of a small batch);
    // k may then be a coarse k + margin of up to 320, which takes the 768-slot candidate lists
    const bool b16 = ix->dtype == OTTER_BF16 || x16_plane != nullptr;   // stream16_kernel: q16_plane = the queries' bf16 plane
    const int EPL = k <= 160 ? 8 : 12, CAP = EPL * 64;
    const bool wide = nq > 32;                  // 33 ... 64 queries: stream64_kernel (bf16 planes only; see stream64_fits)
    if (wide && (!b16 || sc->fixed || !stream64_fits(ix, nq)))
        return set_err(OTTER_EINVAL, "stream search: %lld queries do not fit the stream kernels", nq);
    const int G = nq <= 16 ? (b16 ? 4 : nq <= 4 ? 1 : nq <= 8 ? 2 : 4) : 8;     // query columns / 4
    const int nqp = wide ? S64_NQP : 4 * G;
    const int total_tiles = (int)((ix->n + ST_BM - 1) / ST_BM);
    int strips = std::min(std::min(ix->ncu, NA_MAX_STRIPS), total_tiles);
    strips = std::max(strips, 1);
    int rc;
    void *w_qpad = nullptr, *w_cand, *w_cnt, *w_bmax, *w_sync;
    const bool q_in_place = b16 || (ix->ld == ix->d && ((uintptr_t)qd & 15) == 0);
    if (!q_in_place && (rc = sc->get(ix->qpad, (size_t)nqp * ix->ld * sizeof(float), &w_qpad))) return rc;
    if ((rc = sc->get(ix->cand, (size_t)strips * nqp * CAP * sizeof(u64), &w_cand))) return rc;
    if ((rc = sc->get(ix->cnt, (size_t)strips * nqp * sizeof(u32), &w_cnt))) return rc;
    if ((rc = sc->get(ix->bmax, (size_t)(wide ? S64_NQP : 32) * 4 * NA_MAX_STRIPS * sizeof(u64), &w_bmax))) return rc;
    void* w_hold = nullptr;
    if (wide && (rc = sc->get(ix->hold, (size_t)strips * S64_HOLD * (S64_NQP / 16 * 4) * 256 * 16, &w_hold))) return rc;
    const void* sync_before = ix->sync.p;
    if (sc->fixed) w_sync = sc->fixed;          // reserved head of the static scratch (no other layout touches it)
    else if ((rc = sc->get(ix->sync, ST_SYNC_WORDS * sizeof(u32), &w_sync))) return rc;
    bool& dirty = sc->fixed ? g_scratch_sync_dirty[ix->device] : ix->sync_dirty;
    if (!sc->fixed && ix->sync.p != sync_before) dirty = true;       // a fresh allocation is not zero
    if (dirty) OTTER_HIP_TRY(hipMemsetAsync(w_sync, 0, ST_SYNC_WORDS * sizeof(u32), st));
    dirty = true;                                                     // until this search's merge has been enqueued
    if (!q_in_place) {
        const long long total = (long long)nqp * ix->ld;
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, qd, (float*)w_qpad, nq,
                           (long long)nqp, ix->d, ix->ld);
    }
    StreamArgs sa;
    sa.X = x16_plane ? (const float*)x16_plane : b16 ? (const float*)ix->x16 : ix->x;
    sa.Q = b16 ? (const float*)q16_plane : q_in_place ? qd : (const float*)w_qpad;
    sa.q_rows = q_in_place ? (int)nq : nqp;
    sa.cand = (u64*)w_cand;
    sa.cnt = (u32*)w_cnt;
    sa.n = ix->n;
    sa.ld = ix->ld;
    sa.nq = (int)nq;
    sa.k = k;
    sa.strips = strips;
    sa.total_tiles = total_tiles;
    // the exchange needs k block maxima at least (4 per workgroup); a strip of one tile gains nothing from it
    static const char* no_seed = getenv("OTTER_NO_SAMPLE");
    sa.seed = (!no_seed && 4 * strips >= 2 * k && total_tiles >= 2 * strips) ? 1 : 0;
    sa.bmax = (u64*)w_bmax;
    sa.sync = (u32*)w_sync;
    sa.hold = (float*)w_hold;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    const bool timed = ix->timing && ix->ev_pending.size() < 4096 && !sc->fixed;
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
    // (ring depth by LDS: 4 waves x slots x (8 KiB + queries))
    if (wide && EPL == 12) launch_stream64<4, 12>(sa, strips, st);
    else if (wide) launch_stream64<4, 8>(sa, strips, st);
    else if (b16 && EPL == 12 && G <= 4) launch_stream16<1, 12, 3>(sa, strips, st);
    else if (b16 && EPL == 12) launch_stream16<2, 12, 3>(sa, strips, st);
    else if (b16 && G <= 4) launch_stream16<1, 8, 3>(sa, strips, st);
    else if (b16) launch_stream16<2, 8, 3>(sa, strips, st);
    else if (G == 1) launch_stream<1, 8, 32, 4>(sa, strips, st);      // (the fp32 kernels serve k <= 128: 512-slot lists)
    else if (G == 2) launch_stream<2, 8, 32, 4>(sa, strips, st);
    else if (G == 4) launch_stream<4, 8, 32, 3>(sa, strips, st);
    else launch_stream32<8, 3>(sa, strips, st);
    OTTER_HIP_TRY(hipGetLastError());
    if (timed) {
        OTTER_HIP_TRY(hipEventRecord(ev.second, st));
        ix->ev_pending.push_back(ev);
    }
    rc = run_merge(ix, (const u64*)w_cand, (const u32*)w_cnt, CAP, strips, nq, nqp, k, true, D, I, nullptr, nullptr, nullptr,
                   nullptr, nullptr, st, 0, sc, (u32*)w_sync);
    if (rc == OTTER_OK) dirty = false;                                // the merge leaves the counters zero
    return rc;
}

// one scan + merge over queries [qd, qd + nq) (device pointers), results to device D/I
int search_chunk(otter_index* ix, const float* qd, long long nq, int k, float* D, long long* I, hipStream_t st,
                 Scratch* sc = nullptr) {
    Scratch own;
    if (!sc) sc = &own;
    if (ix->dtype == OTTER_BF16) return search_chunk_bf16(ix, qd, nq, k, D, I, st);
    if (!sc->fixed) {
        // MFMA-bound batches on a large corpus: coarse pass on the bf16 matrix cores + exact re-scoring
        static const char* no_split_env = getenv("OTTER_NO_SPLIT");
        static const char* split_min_env = getenv("OTTER_SPLIT_MIN_NQ");   // tuning runs only
        // Every batch size: from 33 queries on the coarse pass is the 256-query-tile scan on the matrix cores; up to 32
        // it is ONE read of the bf16 hi plane by the stream kernel -- half the bytes of the fp32 rows the exact stream
        // kernels read (1M x 768: 1 query 0.50 -> 0.35 ms per call, 32 queries 0.63 -> 0.37).  OTTER_SPLIT_MIN_NQ=33
        // keeps small batches on the exact fp32 stream kernels (no extra planes in HBM, no read-back per call).
        static const int split_min = split_min_env ? atoi(split_min_env) : 1;
        static const char* mode_env = getenv("OTTER_SPLIT_MODE");          // "1": three-product coarse pass first
        // (a small batch pays about 60 us for the re-scoring, the proof and its read-back: worth it once the half
        // plane it saves is a few hundred megabytes -- 1M x 768 saves 1.5 GB = 0.15 ms; below, the exact stream kernels)
        const bool small_ok = nq > 32 || (!split_min_env && (long long)ix->n * ix->ld >= 200000000ll) || (split_min_env != nullptr);
        if (!no_split_env && !ix->no_split && !ix->coarse_off && nq >= split_min && small_ok && ix->n >= 65536 &&
            split_kc(k, 1, ix->d) <= S16_MAX_K && ix->planes > 0) {
            // planes == 1: only the hi plane exists (half the corpus' bytes instead of 1.5x): the one-product pass or
            // nothing; what it cannot prove goes to the exact kernels
            const int rs = ix->planes == 2 ? ensure_split_planes(ix, st) : 0;
            if (rs < 0) return rs;
            if (rs == 0) {
                const int forced = mode_env ? atoi(mode_env) : 0;      // 1 / 2: start with that many products
                int mode = ix->planes == 2 ? 1 : 0;
                if (ix->planes == 2 && forced != 1 && !ix->two_prod_off && split_kc(k, 2, ix->d) <= S16_MAX_K) mode = 2;
                if (((forced != 1 && forced != 2) || ix->planes == 1) && !ix->one_prod_off && split_kc(k, 3, ix->d) <= S16_MAX_K) {
                    const int rh = ensure_hi_plane(ix, st);
                    if (rh < 0) return rh;
                    if (rh == 0) mode = 3;
                }
                if (mode == 0) goto exact_kernels;
                // (a small batch gains only from the one-product pass on the hi plane, which the bf16 stream kernel
                // reads once; without it -- no room for that plane, or a corpus that needs more products -- the exact
                // stream kernels below are at least as fast as a 256-query-tile scan with 1 ... 32 real columns)
                if (nq > 32 || k > 128 || (mode == 3 && split_kc(k, 3, ix->d) <= 320))
                    return search_chunk_split(ix, qd, nq, k, D, I, st, mode, true);
            }
        }
    }
exact_kernels:
    static const char* no_stream = getenv("OTTER_NO_STREAM");
    // (1 ... 16 queries: 4x4x1 products, 17 ... 32: 32x32x2 blocks, both in the stream structure)
    if (nq <= 32 && k <= 128 && !no_stream) return search_chunk_stream(ix, qd, nq, k, D, I, st, sc);
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
    // one tile per workgroup of a full launch, at least 64 k rows
    const int wgs_all = std::max(1, ix->ncu / qtiles);
    const long long n_s = round_up64(std::max<long long>(16384, 64ll * k), 256);
    const bool sample = !no_sample && nq <= 64 && ix->n >= 12 * n_s;
    const int sample_tiles = sample ? (int)(n_s / cfg.bm) : 0;

    const int main_tiles = all_tiles - sample_tiles;
    int strips = std::max(1, ix->ncu / qtiles);
    strips = std::min(strips, main_tiles);
    if (sc->fixed) strips = std::min(strips, NA_MAX_STRIPS);
    int strips_s = sample ? std::min(std::min(wgs_all, sample_tiles), 64) : 0;
    // one-tile sample strips may keep all their rows (<= CAP): no compaction in the scan, the merge selects
    const bool sample_whole = sample && sample_tiles == strips_s && cfg.bm <= cfg.cap && k + cfg.bm <= MAX_SELECT &&
                              (long long)strips_s * cfg.bm <= 4096;

    void *w_qpad, *w_cand, *w_cnt, *w_smp = nullptr, *w_smpcnt = nullptr, *w_thr0 = nullptr;
    if ((rc = sc->get(ix->qpad, (size_t)nq_pad * ix->ld * sizeof(float), &w_qpad))) return rc;
    if ((rc = sc->get(ix->cand, (size_t)std::max(strips, strips_s) * nq_pad * cfg.cap * sizeof(u64), &w_cand))) return rc;
    if ((rc = sc->get(ix->cnt, (size_t)std::max(strips, strips_s) * nq_pad * sizeof(u32), &w_cnt))) return rc;
    if (sample) {
        if ((rc = sc->get(ix->smp, (size_t)nq_pad * k * sizeof(u64), &w_smp))) return rc;
        if ((rc = sc->get(ix->smpcnt, (size_t)nq_pad * sizeof(u32), &w_smpcnt))) return rc;
        if ((rc = sc->get(ix->thr0, (size_t)nq_pad * sizeof(u32), &w_thr0))) return rc;
    }

    // queries whose rows already have the index's row length are read in place (16-byte aligned for the LDS-DMA)
    const bool q_in_place = ix->ld == ix->d && ((uintptr_t)qd & 15) == 0;
    if (!q_in_place) {
        const long long total = nq_pad * ix->ld;
        const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(pad_rows_kernel, dim3(grid), dim3(256), 0, st, qd, (float*)w_qpad, nq, nq_pad, ix->d,
                           ix->ld);
    }

    ScanArgs sa;
    sa.X = ix->x;
    sa.Q = q_in_place ? qd : (const float*)w_qpad;
    sa.q_rows = q_in_place ? (int)nq : (int)nq_pad;
    sa.skip_final = 0;
    sa.cand = (u64*)w_cand;
    sa.cnt = (u32*)w_cnt;
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
    const bool timed = ix->timing && ix->ev_pending.size() < 4096 && !sc->fixed;   // (events are created on demand)
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
        sa.skip_final = sample_whole ? 1 : 0;
        cfg.launch(sa, qtiles * strips_s, cfg.lds, st);
        OTTER_HIP_TRY(hipGetLastError());
        sa.skip_final = 0;
        if ((rc = run_merge(ix, (const u64*)w_cand, (const u32*)w_cnt, cfg.cap, strips_s, nq, nq_pad, k, false,
                            nullptr, nullptr, nullptr, nullptr, (u64*)w_smp, (u32*)w_smpcnt, (u32*)w_thr0,
                            st, sample_whole ? cfg.bm : 0, sc)))
            return rc;
    }
    sa.strips = strips;
    sa.total_tiles = main_tiles;
    sa.tile_begin = sample_tiles;
    sa.thr0 = sample ? (const u32*)w_thr0 : nullptr;
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
    return run_merge(ix, (const u64*)w_cand, (const u32*)w_cnt, cfg.cap, strips, nq, nq_pad, k, true, D, I,
                     sample ? (const u64*)w_smp : nullptr, sample ? (const u32*)w_smpcnt : nullptr, nullptr,
                     nullptr, nullptr, st, 0, sc);
}

// common part of the two bf16-MFMA searches: scan `Xk` (bf16 planes of ldk elements per row) for the
// queries in ix->qs and merge to (Dc, Ic) [nq, kk].
// Sample pass (as in search_chunk): on a large corpus every strip would otherwise spend its first tiles
// learning a threshold, appending whole tiles of candidates and compacting after each.  The first
// sample_tiles tiles are scanned by all workgroups first and merged; the key of each query's kk-th
// best among them seeds every strip of the main scan.  Exact: the sample rows have the smallest ids,
// so a later row that only ties that key ranks behind kk sample rows.
// Optimistic start (*optimistic in/out; large batches of the one-product pass): instead of sample LEVELS that end at a
// quarter of the corpus -- scanned at 1.5 ... 2x the main scan's cost per tile, because an epilogue is expensive while
// more than ~1 % of the scores pass, with a merge after each -- ONE small level 0, whose merge hands out the key of
// each query's r-th best sample row (r << kk) as the threshold of the scan of ALL other rows: a guess at the kk-th
// best of the whole corpus, loose enough that it is almost never too high (the number of rows above it is about
// r N / n0 with a relative spread of 1 / sqrt(r); strips that collect more than they can hold tighten their own
// threshold exactly, as always).  The caller verifies: the list is the true top kk iff its kk-th entry reaches the
// threshold (margin_check_kernel), and sends the others through the next pass like any unproven query.
template <int MODE>
int scan16_and_merge(otter_index* ix, const void* Xk, int ldk, long long nq, int kk, float* Dc, long long* Ic,
                     hipStream_t st, const void* Qk = nullptr, bool* optimistic = nullptr) {
    Scan16Cfg cfg;
    int rc = pick_cfg16<MODE>(kk, &cfg);
    if (rc) return rc;
    const long long nq_pad = round_up64(nq, S16_BQ);
    const int qtiles = (int)(nq_pad / S16_BQ);
    const int total_tiles = (int)((ix->n + S16_BM - 1) / S16_BM);
    const int wgs = std::max(1, ix->ncu / qtiles);
    static const char* no_sample = getenv("OTTER_NO_SAMPLE");
    // Sample levels.  A strip that starts without a threshold appends whole tiles and compacts after every
    // second one (about 3x the time of a tile in steady state), so the first level is as small as the launch
    // allows: `per0` tiles per workgroup, at least 4 kk rows in all.  Each further level scans three times what
    // has been scanned so far with the threshold the levels before it produced (a tile then passes a few
    // rows per query), in geometric steps that end at 1/DIV of the corpus.
    static const char* per_env = getenv("OTTER_SAMPLE_PER_WG");     // tuning runs only: tiles per workgroup of level 0
    static const char* lvl_env = getenv("OTTER_SAMPLE_LEVELS");     // ... most levels (1: a single level of 1/16)
    static const char* frac_env = getenv("OTTER_SAMPLE_DIV");       // ... cover 1/DIV of the corpus
    // measured (1M x 768; a level costs a launch and a merge, a looser threshold costs appends): three levels
    // ending at 1/4 from 8 query tiles on (few, long strips), two levels ending at 1/8 below
    const int max_levels = lvl_env ? std::max(1, atoi(lvl_env)) : (qtiles >= 8 ? 3 : 2);
    const int cover = total_tiles / (frac_env ? std::max(2, atoi(frac_env)) : (qtiles >= 8 ? 4 : 8));
    int lv_tiles[8], nlv = 0, sample_tiles = 0;
    const char* opt_env = getenv("OTTER_OPT_THR");                  // 0: off; r: rank of the optimistic threshold (read per call:
    const char* opt_l0_env = getenv("OTTER_OPT_L0");                // tests force failures with r = 1); tiles of its level 0
    int opt_rank = 0, opt_l0 = 0;
    if (optimistic && *optimistic && !no_sample && !(opt_env && atoi(opt_env) == 0)) {
        // rows above the r-th best of n0 sample rows: about r N / n0 in the whole corpus, i.e. Gamma(r) N / n0 with the
        // sample's fluctuation; asked for: >= kk, with a failure probability per query far below 1 / nq.  r = 12 and
        // kk n0 / N <= 1.6: P(Gamma(12) < 1.6) ~ 1.4e-7.  Level 0 stays ONE tile per strip and <= 32 strips -- the
        // uncompacted dump the first merge takes whole (more tiles would be scanned without any threshold, at three
        // times a tile's steady-state cost: measured slower) -- so n0 <= 8192 rows, and the optimistic start applies
        // while kk <= 1.6 N / n0 (k = 100 on a million rows: kk = 211, n0 = 7168 at 2214 queries; not k = 1000).
        opt_rank = opt_env ? std::max(1, atoi(opt_env)) : 12;
        int l0 = 32;            // (block maxima need no workgroup per strip of the main scan: any number of tiles)
        if (opt_l0_env) l0 = std::min(BMAX_THR_MAX / 4, std::max(1, atoi(opt_l0_env)));   // (bmax_thr_kernel's block maxima per query)
        const int l0_safe = (int)(1.6 * total_tiles / kk);
        if (!opt_l0_env) l0 = std::min(l0, l0_safe);
        // one round of workgroups where that still leaves 24 tiles (2214 queries: 9 x 28 = 252 <= 256; 9 x 29 needs a second
        // round for 5 workgroups: 29 -> 51 us)
        if (!opt_l0_env && wgs >= 24) l0 = std::min(l0, wgs);
        // (measured, 1M x 768, k = 100, same box, against the sample levels: 512 queries 1.02 -> 0.91 ms, 1024: 1.80 -> 1.62,
        // 2214: 3.79 -> 3.60, 4096: 6.59 -> 6.27; below 24 sample tiles -- small corpora, large k -- the levels stay)
        if ((l0 >= 24 || opt_l0_env) && l0 <= total_tiles / 8 && (long long)l0 * S16_BM >= 4ll * opt_rank && opt_rank <= kk) {
            opt_l0 = l0;        // (no level: block maxima of l0 tiles spread over the corpus, then ONE scan of every tile)
            // A row the scan passes costs 35 us per kk of them and query (2214 queries; the rank-r threshold passes about
            // r / mu x kk rows, mu = kk n0 / N sample rows expected above the true bound): where the start applies, the
            // sample grows while it is cheap -- every tile one round of workgroups holds (few query tiles), or a second
            // round (2214 queries: 56 tiles) -- up to 64 tiles (256 block maxima) and mu <= 4, and the rank follows from
            // mu at the same failure probability as rank 12 at mu = 1.6 (1.5e-7 per query): 56 tiles -> rank 16, 5.3 kk
            // rows passed instead of 7.9.
            static const char* grow_env = getenv("OTTER_OPT_GROW");       // "0": the one-round sample and rank 12
            if (!opt_l0_env && !opt_env && !(grow_env && grow_env[0] == '0')) {
                int want = wgs >= 48 ? wgs : wgs >= 24 ? 2 * wgs : 3 * wgs;      // (whole rounds of workgroups)
                want = std::min(std::min(want, BMAX_THR_MAX / 4), std::min((int)(4.0 * total_tiles / kk), total_tiles / 8));
                if (want > l0) {
                    const double mu = (double)kk * want * S16_BM / (double)ix->n;
                    const int r = poisson_rank(mu, 1.5e-7);
                    if (r <= kk && (long long)want * S16_BM >= 4ll * r && r / mu < 12.0 / 1.6) {
                        opt_l0 = want;
                        opt_rank = r;
                    }
                }
            }
        } else {
            opt_rank = 0;
        }
    }
    if (optimistic) *optimistic = opt_rank > 0;
    if (!no_sample && opt_rank == 0) {
        int l0 = std::max(1, (int)((4ll * kk + (long long)wgs * S16_BM - 1) / ((long long)wgs * S16_BM))) * wgs;
        if (per_env) l0 = std::max(1, atoi(per_env)) * wgs;
        static const char* l0s_env = getenv("OTTER_SAMPLE_L0_STRIPS");   // tuning runs only: one-tile strips of level 0
        // One tile per workgroup is far more than the 4 kk rows level 0 needs when few query tiles leave many workgroups
        // per tile: 16 ... 32 one-tile strips (half the workgroups in between) give a looser first threshold -- level 1
        // appends more -- but a dump the first merge takes whole (<= 32 strips x 256 rows fit its pool) and selects from
        // quickly.  Measured, 1M x 768: 64 queries 0.49 -> 0.43 ms, 256: 0.66 -> 0.53, 1024: 1.75 -> 1.63, 2214: -1 %.
        const int l0_min = (int)((4ll * kk + S16_BM - 1) / S16_BM);
        if (!per_env && l0 == wgs) l0 = std::min(wgs, std::max(std::max(l0_min, 16), std::min(32, wgs / 2)));
        if (l0s_env) l0 = std::min(wgs, std::max(l0_min, atoi(l0s_env)));
        if (max_levels == 1) l0 = std::max(l0, (std::max(cover, (int)((64ll * kk + S16_BM - 1) / S16_BM)) + wgs - 1) / wgs * wgs);
        if (l0 <= total_tiles / 8) {
            lv_tiles[nlv++] = l0;
            sample_tiles = l0;
            // geometric growth of the rows scanned so far, about 4x per level, ending at `cover`
            int more = 0;
            for (long long c = l0; c < cover && more < max_levels - 1 && more < 7; c *= 4) ++more;
            const double ratio = more ? std::pow((double)std::max(cover, l0) / l0, 1.0 / more) : 1.0;
            double target = l0;
            for (int i = 0; i < more; ++i) {
                target *= ratio;
                const int nx = ((int)std::ceil(target) - sample_tiles + wgs - 1) / wgs * wgs;
                if (nx <= 0) continue;
                if (sample_tiles + nx > total_tiles / 2) break;
                lv_tiles[nlv++] = nx;
                sample_tiles += nx;
            }
        }
    }
    const bool sample = nlv > 0;
    const int main_tiles = total_tiles - sample_tiles;
    const int strips = std::min(wgs, main_tiles);
    if ((rc = ix->cand.reserve((size_t)wgs * nq_pad * cfg.cap * sizeof(u64)))) return rc;
    if ((rc = ix->cnt.reserve((size_t)wgs * nq_pad * sizeof(u32)))) return rc;
    if (sample) {
        if ((rc = ix->smp.reserve((size_t)nq_pad * kk * sizeof(u64)))) return rc;
        if ((rc = ix->smpcnt.reserve((size_t)nq_pad * sizeof(u32)))) return rc;
        if ((rc = ix->thr0.reserve((size_t)nq_pad * sizeof(u32)))) return rc;
    }
    if (opt_l0) {
        if ((rc = ix->bmax16.reserve((size_t)nq_pad * 4 * opt_l0 * sizeof(u32)))) return rc;
        if ((rc = ix->thr0.reserve((size_t)nq_pad * sizeof(u32)))) return rc;
    }
    Scan16Args sa;
    sa.X = (const char*)Xk;
    sa.Q = Qk ? (const char*)Qk : (const char*)ix->qs.p;
    sa.cand = (u64*)ix->cand.p;
    sa.cnt = (u32*)ix->cnt.p;
    sa.n = ix->n;
    sa.ldk = ldk;
    sa.nq = (int)nq;
    sa.nq_pad = (int)nq_pad;
    sa.k = kk;
    sa.qtiles = qtiles;
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
    if (opt_l0) {
        // the optimistic threshold: rank-th best of the block maxima of opt_l0 tiles (4 blocks of 64 rows each).  At
        // least `rank` sample rows reach it, so the Gamma(rank) bound above holds for it as for the rank-th best row.
        sa.strips = opt_l0;
        sa.total_tiles = opt_l0;
        sa.tile_begin = 0;
        sa.tile_stride = total_tiles / opt_l0;
        sa.thr0 = nullptr;
        sa.skip_final = 1;
        sa.bmax = (u32*)ix->bmax16.p;
        launch_scan16_bmax(sa, qtiles * opt_l0, st);
        hipLaunchKernelGGL(bmax_thr_kernel, dim3((unsigned)nq), dim3(64), 0, st, (const u32*)ix->bmax16.p, 4 * opt_l0, opt_rank,
                           (u32*)ix->thr0.p);
        OTTER_HIP_TRY(hipGetLastError());
        sa.bmax = nullptr;
        sa.tile_stride = 1;
    }
    int done_tiles = 0;
    for (int lv = 0; lv < nlv; ++lv) {
        sa.strips = std::min(wgs, lv_tiles[lv]);
        sa.total_tiles = lv_tiles[lv];
        sa.tile_begin = done_tiles;
        sa.thr0 = lv ? (const u32*)ix->thr0.p : nullptr;
        // one-tile strips of level 0 that one merge workgroup can take whole: no compaction in the scan
        const bool whole = lv == 0 && lv_tiles[0] == sa.strips && sa.strips * S16_BM <= MAX_SELECT && S16_BM <= cfg.cap;
        sa.skip_final = whole ? 1 : 0;
        cfg.launch(sa, qtiles * sa.strips, st);
        OTTER_HIP_TRY(hipGetLastError());
        // the list of the levels so far is merged in place with this level's strips
        if ((rc = run_merge(ix, (const u64*)ix->cand.p, (const u32*)ix->cnt.p, cfg.cap, sa.strips, nq, nq_pad, kk, false,
                            nullptr, nullptr, lv ? (const u64*)ix->smp.p : nullptr, lv ? (const u32*)ix->smpcnt.p : nullptr,
                            (u64*)ix->smp.p, (u32*)ix->smpcnt.p, (u32*)ix->thr0.p, st, whole ? S16_BM : 0, nullptr, nullptr,
                            opt_rank)))
            return rc;
        done_tiles += lv_tiles[lv];
    }
#ifdef SCAN_DIAG
    // ablation: an unreachable threshold (no appends in the main scan; results are garbage)
    if (getenv("OTTER_DIAG_NOHIT") && (sample || opt_l0)) (void)hipMemsetAsync(ix->thr0.p, 0xFF, (size_t)nq_pad * 4, st);
#endif
    sa.skip_final = 0;
    sa.strips = strips;
    sa.total_tiles = main_tiles;
    sa.tile_begin = sample_tiles;
    sa.thr0 = (sample || opt_l0) ? (const u32*)ix->thr0.p : nullptr;
    cfg.launch(sa, qtiles * strips, st);
    OTTER_HIP_TRY(hipGetLastError());
    if (timed) {
        OTTER_HIP_TRY(hipEventRecord(ev.second, st));
        ix->ev_pending.push_back(ev);
    }
    return run_merge(ix, (const u64*)ix->cand.p, (const u32*)ix->cnt.p, cfg.cap, strips, nq, nq_pad, kk, true, Dc, Ic,
                     sample ? (const u64*)ix->smp.p : nullptr, sample ? (const u32*)ix->smpcnt.p : nullptr, nullptr,
                     nullptr, nullptr, st);
}

// OTTER_BF16 index: queries rounded to bf16, scores accumulated in fp32
int search_chunk_bf16(otter_index* ix, const float* qd, long long nq, int k, float* D, long long* I, hipStream_t st) {
    const long long nq_pad = round_up64(nq, S16_BQ);
    int rc;
    if ((rc = ix->qs.reserve((size_t)nq_pad * ix->ld * 2))) return rc;
    {
        const long long total = nq_pad * ix->ld;
        hipLaunchKernelGGL(to_bf16_rows_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256),
                           0, st, qd, (unsigned short*)ix->qs.p, nq, nq_pad, ix->d, ix->ld, 0ll);
        OTTER_HIP_TRY(hipGetLastError());
    }
    static const char* no_stream = getenv("OTTER_NO_STREAM");
    if ((nq <= 32 || stream64_fits(ix, nq)) && k <= 128 && !no_stream) {
        Scratch own;
        return search_chunk_stream(ix, qd, nq, k, D, I, st, &own, ix->qs.p);
    }
    // (the optimistic start was tried here too, verified by "the k-th result reaches the threshold" with a re-search of the
    // others: 256 queries 0.541 -> 0.520 ms, 2214: 3.35 -> 3.33, the 12.5M x 768 top-1000 shard 307 -> 309 ms -- this
    // index's levels are cheap (kk = k, no margin) and the check costs it the enqueue-only contract: not kept)
    return scan16_and_merge<0>(ix, ix->x16, ix->ld, nq, k, D, I, st);
}

// relative error bound of the coarse score against scan_kernel's fp32 chain, as a multiple of
// |q| |x|: dropped split terms 3 * 2^-16 = 4.6e-5, fp32 accumulation of 3 d / 32 MFMA partial sums
// and of the d-term fma chain it is compared with (each <= d * 2^-24 = 4.6e-5 at d = 768, growing
// with d): 2e-4 covers d <= 1024 with margin; larger d scales it
float split_rel_eps(int d) { return 2e-4f * std::max(1.0f, (float)d / 1024.0f); }

int ensure_split_planes(otter_index* ix, hipStream_t st) {
    if (ix->xs_failed) return 1;
    if (ix->xs_cap < ix->cap_rows) {
        ix->xs.release();
        ix->xnorm2.release();
        ix->xs_rows = 0;
        ix->xs_cap = 0;
        hipError_t e = hipMalloc(&ix->xs.p, (size_t)ix->cap_rows * ix->ld * 4);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            ix->xs.p = nullptr;
            ix->xs_failed = true;           // not enough HBM for the planes: the exact kernel serves every batch
            fprintf(stderr, "[libotter_hip] note: no room in HBM for the bf16 planes of this %lld x %d index (%.1f GB more); "
                            "query batches of 33 and more are served by the exact fp32 kernel at about 1/6 of the "
                            "coarse pass's rate (otter_index_coarse_state() reports 2)\n",
                    ix->n, ix->d, (double)ix->cap_rows * ix->ld * 4 / 1e9);
            return 1;
        }
        ix->xs.bytes = (size_t)ix->cap_rows * ix->ld * 4;
        OTTER_HIP_TRY(hipMemsetAsync(ix->xs.p, 0, ix->xs.bytes, st));
        int rc;
        if ((rc = ix->xnorm2.reserve((size_t)ix->cap_rows * sizeof(float)))) return rc;
        if ((rc = ix->xnmax.reserve(sizeof(u32)))) return rc;
        OTTER_HIP_TRY(hipMemsetAsync(ix->xnmax.p, 0, sizeof(u32), st));
        ix->xs_cap = ix->cap_rows;
    }
    if (ix->xs_rows < ix->n) {
        const long long lo = ix->xs_rows, rows = ix->n - lo;
        hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)std::min<long long>((rows + 3) / 4, 16384)), dim3(256), 0,
                           st, ix->x + (size_t)lo * ix->ld, (long long)ix->ld,
                           (unsigned short*)ix->xs.p, rows, rows, ix->ld, ix->ld, (float*)ix->xnorm2.p + lo, lo);
        hipLaunchKernelGGL(max_norm_kernel, dim3((unsigned)std::min<long long>((rows + 255) / 256, 1024)), dim3(256), 0,
                           st, (const float*)ix->xnorm2.p + lo, rows, (u32*)ix->xnmax.p);
        OTTER_HIP_TRY(hipGetLastError());
        ix->xs_rows = ix->n;
    }
    return OTTER_OK;
}

// stage 0's corpus copy: bf16(x) rows in the bf16 index's tile-blocked plane format + max |x - bf16(x)|^2.
// 0: ready, 1: not available (no HBM for it), < 0: error
int ensure_hi_plane(otter_index* ix, hipStream_t st) {
    if (ix->xh_failed) return 1;
    int rc;
    if (ix->xh_cap < ix->cap_rows) {
        ix->xh.release();
        ix->xh_rows = 0;
        ix->xh_cap = 0;
        hipError_t e = hipMalloc(&ix->xh.p, (size_t)ix->cap_rows * ix->ld * 2);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            ix->xh.p = nullptr;
            ix->xh_failed = true;
            return 1;
        }
        ix->xh.bytes = (size_t)ix->cap_rows * ix->ld * 2;
        OTTER_HIP_TRY(hipMemsetAsync(ix->xh.p, 0, ix->xh.bytes, st));
        if ((rc = ix->xres2.reserve((size_t)ix->cap_rows * sizeof(float)))) return rc;
        if ((rc = ix->xrmax.reserve(sizeof(u32)))) return rc;
        OTTER_HIP_TRY(hipMemsetAsync(ix->xrmax.p, 0, sizeof(u32), st));
        ix->xh_cap = ix->cap_rows;
    }
    if (ix->xh_rows < ix->n) {
        const long long lo = ix->xh_rows, rows = ix->n - lo;
        const long long total = rows * ix->ld;
        hipLaunchKernelGGL(to_bf16_rows_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 8192)), dim3(256), 0,
                           st, (const float*)ix->x + (size_t)lo * ix->ld, (unsigned short*)ix->xh.p, rows, rows, ix->ld,
                           ix->ld, lo);
        // (hi plane only: the row norms the margin proof needs come from here; with the split planes, from their pass)
        float* nrm = nullptr;
        if (ix->planes == 1) {
            if (ix->xnorm2.bytes < (size_t)ix->cap_rows * sizeof(float)) {
                if ((rc = ix->xnorm2.reserve((size_t)ix->cap_rows * sizeof(float)))) return rc;
            }
            if (!ix->xnmax.p) {
                if ((rc = ix->xnmax.reserve(sizeof(u32)))) return rc;
                OTTER_HIP_TRY(hipMemsetAsync(ix->xnmax.p, 0, sizeof(u32), st));
            }
            nrm = (float*)ix->xnorm2.p + lo;
        }
        hipLaunchKernelGGL(residual_norm_kernel, dim3((unsigned)std::min<long long>((rows + 3) / 4, 16384)), dim3(256), 0, st,
                           (const float*)ix->x + (size_t)lo * ix->ld, rows, ix->ld, (float*)ix->xres2.p + lo, nrm);
        hipLaunchKernelGGL(max_norm_kernel, dim3((unsigned)std::min<long long>((rows + 255) / 256, 1024)), dim3(256), 0,
                           st, (const float*)ix->xres2.p + lo, rows, (u32*)ix->xrmax.p);
        if (nrm)
            hipLaunchKernelGGL(max_norm_kernel, dim3((unsigned)std::min<long long>((rows + 255) / 256, 1024)), dim3(256), 0,
                               st, (const float*)nrm, rows, (u32*)ix->xnmax.p);
        OTTER_HIP_TRY(hipGetLastError());
        ix->xh_rows = ix->n;
    }
    return OTTER_OK;
}

// Exact fp32 search of a large query batch through the bf16 matrix cores: coarse top-kc on hi/lo bf16
// planes, exact re-scoring of those candidates with scan_kernel's fp32 chain, and a per-query proof that
// no other row can reach the top k.  Results are identical to search_chunk's.
//   mode 3: coarse score x_hi q_hi alone (one product, on a third copy of the corpus that holds bf16(x) only --
//           skipped when HBM has no room for it).  q.x - q_hi.x_hi = (q - q_hi).x + q_hi.(x - x_hi), at most
//           |q - q_hi| max|x| + |q| max|x - x_hi|: kc = k + max(60, k max(1, 0.04 sqrt(d))).  Unproven queries
//           are searched again in mode 2.
//   mode 2: coarse score x_hi q_hi + x_lo q_hi (two products).  Left out: x . (q - q_hi), at most
//           |q - q_hi| max|x| (~1.6e-3 |q||x|, i.e. ~1.6e-3 sqrt(d) standard deviations of a gaussian score
//           distribution): kc = k + max(44, k max(0.6, 0.022 sqrt(d))).  Queries without a proof are
//           searched again in mode 1 (as one smaller batch), or by the exact kernel if they are few.
//   mode 1: x_hi q_hi + x_hi q_lo + x_lo q_hi, error <= 2e-4 |q||x|: kc = k + max(28, k/8); queries without a
//           proof (near-ties inside that bound, non-finite rows) go to the exact kernel.
// A corpus on which a mode fails for more than a quarter of a batch starts one mode further down from then on.
int split_kc(int k, int mode, int d) {
    if (mode == 1) return k + std::max(28, k / 8);
    const double sd = std::sqrt((double)d);
    if (mode == 2) return k + std::max(44, (int)std::ceil(k * std::max(0.6, 0.022 * sd)));
    return k + std::max(60, (int)std::ceil(k * std::max(1.0, 0.04 * sd)));   // mode 3: twice mode 2's bound
}

// The one host round trip of a coarse pass: the last kernel of the pass stores the count of unproven queries into a
// pinned, host-coherent word (armed with a sentinel before the launch) and the host polls it -- no copy, no stream
// synchronisation.  Every 4096 looks the stream is asked whether it has ended: a kernel that died must not hang the
// caller.
int arm_poll_word(otter_index* ix) {
    if (!ix->h_poll) OTTER_HIP_TRY(hipHostMalloc((void**)&ix->h_poll, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *(volatile u32*)ix->h_poll = 0xFFFFFFFFu;
    return OTTER_OK;
}
int wait_poll_word(otter_index* ix, hipStream_t st, long long* out) {
    volatile u32* hw = ix->h_poll;
    for (unsigned spins = 1; *hw == 0xFFFFFFFFu; ++spins) {
        if ((spins & 4095u) == 0u) {
            const hipError_t e = hipStreamQuery(st);
            if (e != hipErrorNotReady) {
                if (e != hipSuccess) OTTER_HIP_TRY(e);
                if (*hw == 0xFFFFFFFFu) return set_err(OTTER_EHIP, "search: the pass ended without its proof word");
            }
        }
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    *out = *hw;
    return OTTER_OK;
}
int reserve_fin(otter_index* ix, hipStream_t st) {
    const bool fresh = ix->fin.p == nullptr;
    int rc = ix->fin.reserve(128 * sizeof(u32));
    if (rc) return rc;
    if (fresh) OTTER_HIP_TRY(hipMemsetAsync(ix->fin.p, 0, 128 * sizeof(u32), st));   // (kernels leave the counters at zero)
    return OTTER_OK;
}

int search_chunk_split(otter_index* ix, const float* qd, long long nq, int k, float* D, long long* I, hipStream_t st,
                       int mode, bool first) {
    const int kc = split_kc(k, mode, ix->d);
    const long long nq_pad = round_up64(nq, S16_BQ);
    const int S2 = (kc + k - 1) / k;
    int rc;
    if ((rc = ix->qs.reserve((size_t)nq_pad * ix->ld * 4))) return rc;
    if ((rc = ix->qnorm2.reserve((size_t)nq_pad * sizeof(float)))) return rc;
    if ((rc = ix->qres2.reserve((size_t)nq_pad * sizeof(float)))) return rc;
    if ((rc = ix->qpad.reserve((size_t)nq_pad * ix->ld * sizeof(float)))) return rc;
    if ((rc = ix->cD.reserve((size_t)nq * kc * sizeof(float)))) return rc;
    if ((rc = ix->cI.reserve((size_t)nq * kc * sizeof(long long)))) return rc;
    if ((rc = ix->rs.reserve((size_t)S2 * nq_pad * k * sizeof(u64)))) return rc;
    if ((rc = ix->rscnt.reserve((size_t)S2 * nq_pad * sizeof(u32)))) return rc;
    if ((rc = ix->flags.reserve((size_t)(nq + 1) * sizeof(u32)))) return rc;
    if (mode == 3 && (rc = ix->qh16.reserve((size_t)nq_pad * ix->ld * 2))) return rc;
    // (one kernel: hi/lo planes, both norms and, for the one-product pass, the hi values alone as a plane of their own)
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)std::min<long long>((nq_pad + 3) / 4, 16384)), dim3(256), 0, st,
                       qd, (long long)ix->d, (unsigned short*)ix->qs.p, nq, nq_pad, ix->d, ix->ld, (float*)ix->qnorm2.p,
                       0ll, (float*)ix->qres2.p, mode == 3 ? (unsigned short*)ix->qh16.p : (unsigned short*)nullptr);
    // the re-scoring reads the fp32 query rows: in place when the caller's rows already have the padded length
    const bool q_in_place = ix->ld == ix->d && ((uintptr_t)qd & 15) == 0;
    if (!q_in_place) {
        const long long total = nq_pad * ix->ld;
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0,
                           st, qd, (float*)ix->qpad.p, nq, nq_pad, ix->d, ix->ld);
    }
    OTTER_HIP_TRY(hipGetLastError());
    bool optimistic = false, small = false;
    if (mode == 3) {
        static const char* no_stream16 = getenv("OTTER_NO_STREAM");
        if ((nq <= 32 || stream64_fits(ix, nq)) && kc <= 320 && !no_stream16) {
            small = true;
            // a small batch: the coarse pass is one read of the hi plane by the bf16 stream kernel (half the bytes of
            // the fp32 rows), its kc candidates go through the same re-scoring and proof
            Scratch own;
            rc = search_chunk_stream(ix, qd, nq, kc, (float*)ix->cD.p, (long long*)ix->cI.p, st, &own, ix->qh16.p, ix->xh.p);
        } else {
            // (measured, 1M x 768, one query tile: 65 queries 0.539 -> 0.507 ms, 256: 0.650 -> 0.616, 384: 1.06 -> 1.00)
            optimistic = nq > S64_NQP && !ix->opt_off;
            rc = scan16_and_merge<0>(ix, ix->xh.p, ix->ld, nq, kc, (float*)ix->cD.p, (long long*)ix->cI.p, st, ix->qh16.p,
                                     &optimistic);
        }
    } else if (mode == 2)
        rc = scan16_and_merge<2>(ix, ix->xs.p, 2 * ix->ld, nq, kc, (float*)ix->cD.p, (long long*)ix->cI.p, st);
    else
        rc = scan16_and_merge<1>(ix, ix->xs.p, 2 * ix->ld, nq, kc, (float*)ix->cD.p, (long long*)ix->cI.p, st);
    if (rc) return rc;
    u32* flags = (u32*)ix->flags.p;
    long long nf = 0;
    bool opt_missed = false;
    // (read per call: the tests compare the two forms)
    const char* no_fin = getenv("OTTER_NO_FINISH");
    if (small && kc <= FIN_MAX_KC && (size_t)ix->ld * 4 <= 32768 && !(no_fin && no_fin[0] == '1')) {
        // small batch: re-scoring, final selection and proof in one launch, the proof word polled from pinned memory
        if ((rc = reserve_fin(ix, st))) return rc;
        if ((rc = arm_poll_word(ix))) return rc;
        FinishArgs fa;
        fa.X = ix->x;
        fa.Q = q_in_place ? qd : (const float*)ix->qpad.p;
        fa.Ic = (const long long*)ix->cI.p;
        fa.Dc = (const float*)ix->cD.p;
        fa.D = D;
        fa.I = I;
        fa.comp = (u64*)ix->rs.p;
        fa.tick = (u32*)ix->fin.p;
        fa.flags = flags;
        fa.host_word = ix->h_poll;
        fa.qnorm2 = (const float*)ix->qnorm2.p;
        fa.xnorm2_max = (const u32*)ix->xnmax.p;
        fa.qres2 = (const float*)ix->qres2.p;
        fa.xres2_max = (const u32*)ix->xrmax.p;
        fa.rel = split_rel_eps(ix->d);
        fa.id_base = ix->id_base;
        fa.n = ix->n;
        fa.ld = ix->ld;
        fa.kc = kc;
        fa.k = k;
        fa.nq = (int)nq;
        fa.parts = (kc + FIN_ROWS - 1) / FIN_ROWS;
        const size_t lds = (size_t)FIN_STAGES * FIN_ROWS * FIN_PITCH + (size_t)ix->ld * 4;
        static bool attr_set[64] = {};
        if (ix->device >= 0 && ix->device < 64 && !attr_set[ix->device]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(finish_small_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, FIN_STAGES * FIN_ROWS * FIN_PITCH + 32768);
            attr_set[ix->device] = true;
        }
        hipLaunchKernelGGL(finish_small_kernel, dim3((unsigned)(nq * fa.parts)), dim3(256), lds, st, fa);
        OTTER_HIP_TRY(hipGetLastError());
        if ((rc = wait_poll_word(ix, st, &nf))) return rc;
    } else {
        RescoreArgs ra;
        ra.X = ix->x;
        ra.Q = q_in_place ? qd : (const float*)ix->qpad.p;
        ra.Ic = (const long long*)ix->cI.p;
        ra.out = (u64*)ix->rs.p;
        ra.out_cnt = (u32*)ix->rscnt.p;
        ra.id_base = ix->id_base;
        ra.ld = ix->ld;
        ra.kc = kc;
        ra.k = k;
        ra.nq_pad = (int)nq_pad;
        ra.zero_word = (u32*)ix->flags.p + nq;        // the margin check's count of unproven queries
        launch_rescore(ra, (unsigned)nq, st);
        OTTER_HIP_TRY(hipGetLastError());
        if ((rc = run_merge(ix, (const u64*)ix->rs.p, (const u32*)ix->rscnt.p, k, S2, nq, nq_pad, k, true, D, I, nullptr,
                            nullptr, nullptr, nullptr, nullptr, st)))
            return rc;
        if ((rc = reserve_fin(ix, st))) return rc;
        if ((rc = arm_poll_word(ix))) return rc;
        hipLaunchKernelGGL(margin_check_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, (const float*)D,
                           (const float*)ix->cD.p, (const long long*)ix->cI.p, (const float*)ix->qnorm2.p,
                           (const u32*)ix->xnmax.p, split_rel_eps(ix->d), nq, k, kc, ix->n, flags, flags + nq,
                           mode >= 2 ? (const float*)ix->qres2.p : (const float*)nullptr,
                           mode == 3 ? (const u32*)ix->xrmax.p : (const u32*)nullptr,
                           optimistic ? (const u32*)ix->thr0.p : (const u32*)nullptr, (u32*)ix->fin.p + 127, ix->h_poll);
        OTTER_HIP_TRY(hipGetLastError());
        // the one host round trip of this pass: how many queries are left without a proof (see wait_poll_word; the call
        // returns when the results are decided, the stream may still be retiring the kernel's tail)
        if ((rc = wait_poll_word(ix, st, &nf))) return rc;
        if (nf & 0x80000000ll) {
            // the optimistic thresholds were too high for more than a quarter of the batch (a corpus whose best rows sit
            // where the spread sample looks: the guess is not a verdict on the one-product bound): this index goes back to
            // the sample levels, the queries take the next pass as always
            nf &= 0x7FFFFFFFll;
            opt_missed = true;
            if (!ix->opt_off)
                fprintf(stderr, "[libotter_hip] note: the optimistic threshold of the coarse scan was too high for more than a "
                                "quarter of %lld queries; this index uses the sample levels until reset()\n", nq);
            ix->opt_off = true;
        }
    }
    if (mode >= 2 && first) ix->last_refined += nf;
    // (a verdict on the corpus needs a batch worth the name: one hard query of a one-query call is not one.  Small
    // calls -- the reference issues ONE query per call -- are judged over a running window of 64 queries per pass,
    // so a corpus whose proofs fail stops paying for the coarse pass there too)
    bool verdict = nq >= 32 && nf * 4 > nq && !opt_missed;
    if (nq < 32 && first) {
        ix->win_q[mode] += nq;
        ix->win_f[mode] += nf;
        if (ix->win_q[mode] >= 64) {
            verdict = ix->win_f[mode] * 4 > ix->win_q[mode];
            ix->win_q[mode] = ix->win_f[mode] = 0;
        }
    }
    if (mode == 3 && verdict) ix->one_prod_off = true;
    if (mode == 2 && verdict) ix->two_prod_off = true;
    if (mode == 1 && verdict) {
        if (!ix->coarse_off)
            fprintf(stderr, "[libotter_hip] note: the coarse pass could not prove %lld of %lld queries on this index (scores closer "
                            "than bf16 products resolve); batches are served by the exact fp32 kernel until reset() "
                            "(otter_index_coarse_state() reports 4)\n", nf, nq);
        ix->coarse_off = true;
    }
    if (mode == 1) ix->last_fallback += nf;
    if (nf == 0) return OTTER_OK;
    // unproven queries: gather them, search them again one level down, scatter the rows back
    otter::DevBuf& bq = mode == 3 ? ix->fb3_q : mode == 2 ? ix->fb2_q : ix->fb_q;
    otter::DevBuf& bD = mode == 3 ? ix->fb3_D : mode == 2 ? ix->fb2_D : ix->fb_D;
    otter::DevBuf& bI = mode == 3 ? ix->fb3_I : mode == 2 ? ix->fb2_I : ix->fb_I;
    if ((rc = bq.reserve((size_t)nf * ix->d * sizeof(float)))) return rc;
    if ((rc = bD.reserve((size_t)nf * k * sizeof(float)))) return rc;
    if ((rc = bI.reserve((size_t)nf * k * sizeof(long long)))) return rc;
    otter::DevBuf& bw = mode == 3 ? ix->fb3_w : mode == 2 ? ix->fb2_w : ix->fb_w;
    if ((rc = bw.reserve((size_t)nq * sizeof(int)))) return rc;
    hipLaunchKernelGGL(compact_flags_kernel, dim3(1), dim3(1024), 0, st, (const u32*)flags, nq, (int*)bw.p);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)nf), dim3(256), 0, st, qd, (const int*)bw.p, (float*)bq.p, ix->d);
    OTTER_HIP_TRY(hipGetLastError());
    if (mode >= 2 && nf >= 33 && ix->planes == 2) {
        const int next = (mode == 3 && !ix->two_prod_off && split_kc(k, 2, ix->d) <= S16_MAX_K) ? 2 : 1;
        rc = search_chunk_split(ix, (const float*)bq.p, nf, k, (float*)bD.p, (long long*)bI.p, st, next, false);
    } else {
        if (mode >= 2) ix->last_fallback += nf;
        ix->no_split = true;
        rc = search_chunk(ix, (const float*)bq.p, nf, k, (float*)bD.p, (long long*)bI.p, st, nullptr);
        ix->no_split = false;
    }
    if (rc) return rc;
    hipLaunchKernelGGL(scatter_results_kernel, dim3((unsigned)nf), dim3(256), 0, st, (const float*)bD.p,
                       (const long long*)bI.p, (const int*)bw.p, D, I, k);
    OTTER_HIP_TRY(hipGetLastError());
    return OTTER_OK;
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
    if (dtype != OTTER_F32 && dtype != OTTER_BF16)
        return set_err(OTTER_ENOTIMPL, "index_create: corpus storage is OTTER_F32 or OTTER_BF16 (dtype=%d)", dtype);
    int ndev = 0;
    OTTER_HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return set_err(OTTER_EINVAL, "index_create: device %d out of range (%d visible)", device, ndev);
    hipDeviceProp_t prop;
    OTTER_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(OTTER_ESTATE, "index_create: device %d is %s; this library is built for gfx950 only", device,
                       prop.gcnArchName);
    {
        // set-up of otter_index_search_noalloc: this device's address of the static scratch array, once per device
        DeviceGuard g(device);
        if (!g.ok) return set_err(OTTER_EHIP, "index_create: hipSetDevice(%d) failed", device);
        const int rs = resolve_scratch(device);
        if (rs) return rs;
    }
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
    if (ix->dtype == OTTER_BF16) {
        // rows arrive as float32 (the reference's .npy) and are rounded to bf16 on the device, slab by slab
        const long long slab = 65536;
        for (long long lo = 0; lo < n; lo += slab) {
            const long long c = std::min<long long>(slab, n - lo);
            const float* src = (const float*)x + (size_t)lo * ix->d;
            if (!x_is_device) {
                if ((rc = ix->io_q.reserve((size_t)c * ix->d * sizeof(float)))) return rc;
                OTTER_HIP_TRY(hipMemcpyAsync(ix->io_q.p, src, (size_t)c * ix->d * sizeof(float), hipMemcpyHostToDevice, st));
                src = (const float*)ix->io_q.p;
            }
            const long long total = c * ix->ld;
            hipLaunchKernelGGL(to_bf16_rows_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 8192)),
                               dim3(256), 0, st, src, ix->x16, c, c, ix->d, ix->ld, (long long)(ix->n + lo));
            OTTER_HIP_TRY(hipGetLastError());
            if (!x_is_device) OTTER_HIP_TRY(hipStreamSynchronize(st));   // the staging slab is reused
        }
        ix->n += n;
        return OTTER_OK;
    }
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

    ix->last_fallback = 0;      // diagnostic counters of this call, summed over its query chunks
    ix->last_refined = 0;
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

// otter_index_search for callers that cannot allocate: same kernels, scratch in the static per-GPU array.
int otter_index_search_noalloc(otter_index* ix, const float* q, int64_t nq, int k, float* D, int64_t* I, void* stream) {
    if (!ix) return set_err(OTTER_EINVAL, "search_noalloc: index is NULL");
    if (ix->dtype != OTTER_F32) return set_err(OTTER_ENOTIMPL, "search_noalloc: OTTER_F32 indexes only");
    if (nq < 0 || nq > NA_MAX_NQ)
        return set_err(OTTER_EINVAL, "search_noalloc: nq=%lld outside [0, %d] (use otter_index_search)", (long long)nq, NA_MAX_NQ);
    if (k < 1 || k > NA_MAX_K) return set_err(OTTER_EINVAL, "search_noalloc: k=%d outside [1, %d] (use otter_index_search)", k, NA_MAX_K);
    if (ix->ld > NA_MAX_LD) return set_err(OTTER_EINVAL, "search_noalloc: d=%d exceeds %d (use otter_index_search)", ix->d, NA_MAX_LD);
    if (nq == 0) return OTTER_OK;
    if (!q || !D || !I) return set_err(OTTER_EINVAL, "search_noalloc: q, D and I must be non-NULL device pointers");
    DeviceGuard g(ix->device);
    if (!g.ok) return set_err(OTTER_EHIP, "search_noalloc: hipSetDevice(%d) failed", ix->device);
    hipStream_t st = (hipStream_t)stream;
    if (ix->n == 0) {
        const long long total = (long long)nq * k;
        hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, st,
                           D, (long long*)I, total);
        OTTER_HIP_TRY(hipGetLastError());
        return OTTER_OK;
    }
    Scratch sc;
    sc.fixed = g_scratch_ptr[ix->device];       // resolved by otter_index_create on this device
    sc.fixed_bytes = NA_SCRATCH_BYTES;
    sc.used = 512;                              // the first 512 bytes hold the stream kernel's counters (ST_SYNC_WORDS)
    if (!sc.fixed) return set_err(OTTER_ESTATE, "search_noalloc: static scratch of device %d was not resolved", ix->device);
    return search_chunk(ix, q, nq, k, D, (long long*)I, st, &sc);
}

// Search restricted to per-query id subsets: the reference builds a throw-away
// faiss.IndexFlatIP over the TF-IDF-filtered blocks of every question and searches it with that one
// question (retrieval/eval_ottqa_retrieval_tfidffilter.py:530-546).  Here the corpus stays resident: the
// listed rows are gathered and scored with the exact kernel's fp32 chain (rescore_lds_kernel), then selected
// by the merge kernel; scores are bit-identical to those search() returns for the same rows.
int otter_index_search_subset(otter_index* ix, const void* q, int64_t nq, int k, const int64_t* ids, int m, float* D,
                              int64_t* I, int io_is_device, void* stream) {
    if (!ix) return set_err(OTTER_EINVAL, "search_subset: index is NULL");
    if (ix->dtype != OTTER_F32) return set_err(OTTER_ENOTIMPL, "search_subset: fp32 index only");
    if (nq < 0) return set_err(OTTER_EINVAL, "search_subset: nq=%lld < 0", (long long)nq);
    if (k < 1 || k > OTTER_MAX_K) return set_err(OTTER_EINVAL, "search_subset: k=%d outside [1, %d]", k, OTTER_MAX_K);
    if (m < 1 || m > (1 << 20)) return set_err(OTTER_EINVAL, "search_subset: m=%d outside [1, 2^20]", m);
    if (nq == 0) return OTTER_OK;
    if (!q || !D || !I || !ids) return set_err(OTTER_EINVAL, "search_subset: q, ids, D and I must be non-NULL");
    DeviceGuard g(ix->device);
    if (!g.ok) return set_err(OTTER_EHIP, "search_subset: hipSetDevice(%d) failed", ix->device);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const int S2 = (m + k - 1) / k;
    // chunk the queries so that the candidate lists stay below ~256 MB
    const long long per_q = (long long)S2 * k * (long long)sizeof(u64) + (long long)m * 8;
    const long long chunk = std::max<long long>(1, std::min<long long>(QCHUNK, (256ll << 20) / per_q));
    if ((rc = ix->flags.reserve(2 * sizeof(u32)))) return rc;
    for (long long q0 = 0; q0 < nq; q0 += chunk) {
        const long long c = std::min<long long>(chunk, nq - q0);
        const long long c_pad = c;      // row stride of the per-query scratch lists: nothing here reads a padded query
        if ((rc = ix->qpad.reserve((size_t)c_pad * ix->ld * sizeof(float)))) return rc;
        if ((rc = ix->cI.reserve((size_t)c * m * sizeof(long long)))) return rc;
        if ((rc = ix->rs.reserve((size_t)S2 * c_pad * k * sizeof(u64)))) return rc;
        if ((rc = ix->rscnt.reserve((size_t)S2 * c_pad * sizeof(u32)))) return rc;
        if ((rc = ix->io_D.reserve((size_t)c * k * sizeof(float)))) return rc;
        if ((rc = ix->io_I.reserve((size_t)c * k * sizeof(long long)))) return rc;
        const float* qd = (const float*)q + (size_t)q0 * ix->d;
        const long long* idd = (const long long*)ids + (size_t)q0 * m;
        float* Dd = D + (size_t)q0 * k;
        long long* Id = (long long*)I + (size_t)q0 * k;
        if (!io_is_device) {
            if ((rc = ix->io_q.reserve((size_t)c * ix->d * sizeof(float)))) return rc;
            if ((rc = ix->fb_I.reserve((size_t)c * m * sizeof(long long)))) return rc;
            OTTER_HIP_TRY(hipMemcpyAsync(ix->io_q.p, qd, (size_t)c * ix->d * sizeof(float), hipMemcpyHostToDevice, st));
            OTTER_HIP_TRY(hipMemcpyAsync(ix->fb_I.p, idd, (size_t)c * m * sizeof(long long), hipMemcpyHostToDevice, st));
            qd = (const float*)ix->io_q.p;
            idd = (const long long*)ix->fb_I.p;
            Dd = (float*)ix->io_D.p;
            Id = (long long*)ix->io_I.p;
        }
        u32* flag = (u32*)ix->flags.p;
        OTTER_HIP_TRY(hipMemsetAsync(flag, 0, sizeof(u32), st));
        // valid ids to the front of every list (the re-scoring kernel counts on that), bad ids flagged
        hipLaunchKernelGGL(subset_ids_kernel, dim3((unsigned)c), dim3(64), 0, st, idd, (long long*)ix->cI.p, m,
                           (long long)ix->id_base, ix->n, flag);
        {
            const long long total = c_pad * ix->ld;
            hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256),
                               0, st, qd, (float*)ix->qpad.p, c, c_pad, ix->d, ix->ld);
        }
        OTTER_HIP_TRY(hipGetLastError());
        u32 bad = 0;
        OTTER_HIP_TRY(hipMemcpyAsync(&bad, flag, sizeof(u32), hipMemcpyDeviceToHost, st));
        OTTER_HIP_TRY(hipStreamSynchronize(st));
        if (bad)
            return set_err(OTTER_EINVAL, "search_subset: %u ids outside [-1] U [%lld, %lld)", bad, (long long)ix->id_base,
                           (long long)(ix->id_base + ix->n));
        RescoreArgs ra;
        ra.X = ix->x;
        ra.Q = (const float*)ix->qpad.p;
        ra.Ic = (const long long*)ix->cI.p;
        ra.out = (u64*)ix->rs.p;
        ra.out_cnt = (u32*)ix->rscnt.p;
        ra.id_base = ix->id_base;
        ra.ld = ix->ld;
        ra.kc = m;
        ra.k = k;
        ra.nq_pad = (int)c_pad;
        ra.zero_word = nullptr;
        launch_rescore(ra, (unsigned)c, st);
        OTTER_HIP_TRY(hipGetLastError());
        if ((rc = run_merge(ix, (const u64*)ix->rs.p, (const u32*)ix->rscnt.p, k, S2, c, c_pad, k, true, Dd, Id, nullptr,
                            nullptr, nullptr, nullptr, nullptr, st)))
            return rc;
        if (!io_is_device) {
            OTTER_HIP_TRY(hipMemcpyAsync(D + (size_t)q0 * k, Dd, (size_t)c * k * sizeof(float), hipMemcpyDeviceToHost, st));
            OTTER_HIP_TRY(hipMemcpyAsync(I + (size_t)q0 * k, Id, (size_t)c * k * sizeof(long long), hipMemcpyDeviceToHost,
                                         st));
            OTTER_HIP_TRY(hipStreamSynchronize(st));
        }
    }
    return OTTER_OK;
}

// rows [i0, i0 + n) as float32 [n, d] (faiss: index.reconstruct_n(i0, n)); the bf16 index returns its rounded values
__global__ void unpack_rows_kernel(const float* __restrict__ x32, const unsigned short* __restrict__ x16, float* dst,
                                   long long i0, long long n, int d, int ld) {
    const long long total = n * d;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / d;
        const int c = (int)(i - r * d);
        dst[i] = x32 ? x32[(i0 + r) * ld + c] : __uint_as_float((u32)x16[plane_index(i0 + r, c, ld)] << 16);
    }
}

int otter_index_reconstruct_n(otter_index* ix, int64_t i0, int64_t n, float* out, int out_is_device, void* stream) {
    if (!ix) return set_err(OTTER_EINVAL, "reconstruct_n: index is NULL");
    if (i0 < 0 || n < 0 || i0 + n > ix->n) return set_err(OTTER_EINVAL, "reconstruct_n: rows [%lld, %lld) outside [0, %lld)",
                                                          (long long)i0, (long long)(i0 + n), (long long)ix->n);
    if (n == 0) return OTTER_OK;
    if (!out) return set_err(OTTER_EINVAL, "reconstruct_n: out is NULL");
    DeviceGuard g(ix->device);
    if (!g.ok) return set_err(OTTER_EHIP, "reconstruct_n: hipSetDevice(%d) failed", ix->device);
    hipStream_t st = (hipStream_t)stream;
    const long long slab = 1 << 16;
    for (long long lo = 0; lo < n; lo += slab) {
        const long long c = std::min<long long>(slab, n - lo);
        float* dst = out + (size_t)lo * ix->d;
        if (!out_is_device) {
            int rc = ix->io_q.reserve((size_t)c * ix->d * sizeof(float));
            if (rc) return rc;
            dst = (float*)ix->io_q.p;
        }
        const long long total = c * ix->d;
        hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 8192)), dim3(256), 0, st,
                           ix->dtype == OTTER_F32 ? (const float*)ix->x : (const float*)nullptr,
                           (const unsigned short*)ix->x16, dst, (long long)(i0 + lo), c, ix->d, ix->ld);
        OTTER_HIP_TRY(hipGetLastError());
        if (!out_is_device) {
            OTTER_HIP_TRY(hipMemcpyAsync(out + (size_t)lo * ix->d, dst, (size_t)c * ix->d * sizeof(float), hipMemcpyDeviceToHost,
                                         st));
            OTTER_HIP_TRY(hipStreamSynchronize(st));   // the staging slab is reused
        }
    }
    return OTTER_OK;
}

int64_t otter_index_ntotal(const otter_index* ix) { return ix ? ix->n : -1; }
int otter_index_d(const otter_index* ix) { return ix ? ix->d : -1; }

int otter_index_reset(otter_index* ix) {
    if (!ix) return set_err(OTTER_EINVAL, "reset: index is NULL");
    ix->n = 0;
    ix->xs_rows = 0;
    ix->two_prod_off = false;
    ix->one_prod_off = false;
    ix->opt_off = false;
    ix->coarse_off = false;
    for (int m = 0; m < 4; ++m) ix->win_q[m] = ix->win_f[m] = 0;
    ix->xh_rows = 0;
    if (ix->xrmax.p) {
        DeviceGuard g2(ix->device);
        OTTER_HIP_TRY(hipMemset(ix->xrmax.p, 0, sizeof(u32)));
    }
    if (ix->xnmax.p) {
        DeviceGuard g(ix->device);
        OTTER_HIP_TRY(hipMemset(ix->xnmax.p, 0, sizeof(u32)));
    }
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

int otter_index_set_planes(otter_index* ix, int planes) {
    if (!ix) return set_err(OTTER_EINVAL, "set_planes: index is NULL");
    if (planes < 0 || planes > 2) return set_err(OTTER_EINVAL, "set_planes: planes=%d outside {0, 1, 2}", planes);
    DeviceGuard g(ix->device);
    if (!g.ok) return set_err(OTTER_EHIP, "set_planes: hipSetDevice(%d) failed", ix->device);
    OTTER_HIP_TRY(hipDeviceSynchronize());      // nothing may still be reading a plane that is freed below
    if (planes < 2) {
        ix->xs.release();
        ix->xnorm2.release();
        ix->xs_rows = ix->xs_cap = 0;
        ix->xs_failed = false;
    }
    if (planes < 1) {
        ix->xh.release();
        ix->xres2.release();
        ix->xh_rows = ix->xh_cap = 0;
        ix->xh_failed = false;
    }
    ix->planes = planes;
    return OTTER_OK;
}

int otter_index_coarse_state(const otter_index* ix) {
    if (!ix) return set_err(OTTER_EINVAL, "coarse_state: index is NULL");
    if (ix->dtype != OTTER_F32) return 3;
    if (ix->xs_failed) return 2;
    if (ix->coarse_off) return 4;
    return ix->xs_rows >= ix->n && ix->n > 0 && ix->xs.p ? 1 : 0;
}

int64_t otter_index_last_fallback(const otter_index* ix) { return ix ? ix->last_fallback : -1; }
int64_t otter_index_last_refined(const otter_index* ix) { return ix ? ix->last_refined : -1; }

void otter_index_destroy(otter_index* ix) {
    if (!ix) return;
    DeviceGuard g(ix->device);
    for (auto& ev : ix->ev_pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : ix->ev_free) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (ix->x) (void)hipFree(ix->x);
    if (ix->x16) (void)hipFree(ix->x16);
    for (otter::DevBuf* b : {&ix->xs, &ix->xnorm2, &ix->xnmax, &ix->qs, &ix->qnorm2, &ix->cD, &ix->cI, &ix->rs,
                             &ix->rscnt, &ix->flags, &ix->fb_q, &ix->fb_D, &ix->fb_I, &ix->qres2, &ix->fb2_q, &ix->fb2_D, &ix->fb2_I, &ix->xh, &ix->xres2, &ix->xrmax, &ix->qh16,
                             &ix->fb3_q, &ix->fb3_D, &ix->fb3_I, &ix->fb_w, &ix->fb2_w, &ix->fb3_w})
        b->release();
    ix->qpad.release();
    ix->cand.release();
    ix->cnt.release();
    for (int i = 0; i < 2; ++i) { ix->lvl[i].release(); ix->lvlcnt[i].release(); }
    ix->smp.release();
    ix->smpcnt.release();
    ix->thr0.release();
    ix->bmax.release();
    ix->hold.release();
    ix->bmax16.release();
    ix->sync.release();
    ix->io_q.release();
    ix->io_D.release();
    ix->io_I.release();
    if (ix->h_poll) (void)hipHostFree(ix->h_poll);
    ix->fin.release();
    if (ix->sh_worker) {
        ix->sh_worker->stop();
        delete ix->sh_worker;
    }
    if (ix->sh_stream) (void)hipStreamDestroy(ix->sh_stream);
    for (otter::DevBuf* b : {&ix->sh_q, &ix->sh_D, &ix->sh_I, &ix->sh_gD, &ix->sh_gI, &ix->sh_tD, &ix->sh_tI}) b->release();
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

// One shard's part of otter_sharded_search, on its own host thread: queries to the shard's GPU, search on the
// shard's stream, candidates into slot g of the gathered buffers on the home GPU; returns once they have landed.
static int sharded_search_one(otter_index* ix, int home, const void* q, int q_is_device, int64_t nq, int k, float* gD,
                              long long* gI) {
    DeviceGuard g(ix->device);
    if (!g.ok) return set_err(OTTER_EHIP, "sharded_search: hipSetDevice(%d) failed", ix->device);
    if (!ix->sh_stream) OTTER_HIP_TRY(hipStreamCreateWithFlags(&ix->sh_stream, hipStreamNonBlocking));
    hipStream_t st = ix->sh_stream;
    const size_t qbytes = (size_t)nq * ix->d * sizeof(float), nk = (size_t)nq * k;
    int rc;
    const float* qd = (const float*)q;
    if (!q_is_device) {
        if ((rc = ix->sh_q.reserve(qbytes))) return rc;
        OTTER_HIP_TRY(hipMemcpyAsync(ix->sh_q.p, q, qbytes, hipMemcpyHostToDevice, st));
        qd = (const float*)ix->sh_q.p;
    } else if (ix->device != home) {
        if ((rc = ix->sh_q.reserve(qbytes))) return rc;
        OTTER_HIP_TRY(hipMemcpyPeerAsync(ix->sh_q.p, ix->device, q, home, qbytes, st));
        qd = (const float*)ix->sh_q.p;
    }
    if (ix->device == home) {
        // same GPU as the gathered buffers: the search writes its slot directly
        if ((rc = otter_index_search(ix, qd, nq, k, gD, (int64_t*)gI, 1, st))) return rc;
    } else {
        if ((rc = ix->sh_D.reserve(nk * sizeof(float)))) return rc;
        if ((rc = ix->sh_I.reserve(nk * sizeof(long long)))) return rc;
        if ((rc = otter_index_search(ix, qd, nq, k, (float*)ix->sh_D.p, (int64_t*)ix->sh_I.p, 1, st))) return rc;
        OTTER_HIP_TRY(hipMemcpyPeerAsync(gD, home, ix->sh_D.p, ix->device, nk * sizeof(float), st));
        OTTER_HIP_TRY(hipMemcpyPeerAsync(gI, home, ix->sh_I.p, ix->device, nk * sizeof(long long), st));
    }
    OTTER_HIP_TRY(hipStreamSynchronize(st));
    return OTTER_OK;
}

int otter_sharded_search(otter_index* const* shards, int G, const void* q, int64_t nq, int k, float* D, int64_t* I,
                         int io_is_device, void* stream) {
    if (!shards || G < 1 || G > 64) return set_err(OTTER_EINVAL, "sharded_search: need 1 ... 64 shards (G=%d)", G);
    for (int g = 0; g < G; ++g) {
        if (!shards[g]) return set_err(OTTER_EINVAL, "sharded_search: shard %d is NULL", g);
        if (shards[g]->d != shards[0]->d)
            return set_err(OTTER_EINVAL, "sharded_search: shard %d has d=%d, shard 0 has d=%d", g, shards[g]->d, shards[0]->d);
        for (int h = 0; h < g; ++h)
            if (shards[h] == shards[g]) return set_err(OTTER_EINVAL, "sharded_search: shards %d and %d are the same index", h, g);
    }
    if (nq < 0) return set_err(OTTER_EINVAL, "sharded_search: nq=%lld < 0", (long long)nq);
    if (k < 1 || k > OTTER_MAX_K) return set_err(OTTER_EINVAL, "sharded_search: k=%d outside [1, %d]", k, OTTER_MAX_K);
    if (nq == 0) return OTTER_OK;
    if (!q || !D || !I) return set_err(OTTER_EINVAL, "sharded_search: q, D and I must be non-NULL");
    otter_index* h0 = shards[0];
    const int home = h0->device;
    DeviceGuard guard(home);
    if (!guard.ok) return set_err(OTTER_EHIP, "sharded_search: hipSetDevice(%d) failed", home);
    hipStream_t st = (hipStream_t)stream;
    const size_t nk = (size_t)nq * k;
    int rc;
    if (G == 1) return otter_index_search(h0, q, nq, k, D, I, io_is_device, stream);
    if ((rc = h0->sh_gD.reserve((size_t)G * nk * sizeof(float)))) return rc;
    if ((rc = h0->sh_gI.reserve((size_t)G * nk * sizeof(long long)))) return rc;
    // peer copies go GPU to GPU over xGMI where the pair allows it (otherwise HIP stages them through the host);
    // a pair is asked for and enabled once per process
    {
        static std::mutex peer_m;
        static bool peer_done[64][64] = {};
        std::lock_guard<std::mutex> lk(peer_m);
        for (int g = 1; g < G; ++g) {
            const int dv = shards[g]->device;
            if (dv == home || dv < 0 || dv >= 64 || home >= 64 || peer_done[home][dv]) continue;
            peer_done[home][dv] = true;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, home, dv) == hipSuccess && can) {
                (void)hipDeviceEnablePeerAccess(dv, 0);          // "already enabled" is fine
                (void)hipGetLastError();
                DeviceGuard pg(dv);
                if (pg.ok) { (void)hipDeviceEnablePeerAccess(home, 0); (void)hipGetLastError(); }
            }
        }
    }
    // device queries: whatever produces them on the caller's stream has to be done before other GPUs read them
    if (io_is_device) OTTER_HIP_TRY(hipStreamSynchronize(st));
    // shards 1 .. G-1 on their own persistent host threads, shard 0 here.  Nothing may throw across the C ABI: a
    // shard whose thread cannot be created (or whose task cannot be queued) is searched inline instead.
    int rcs[64];
    char msgs[64][256];
    bool queued[64] = {};
    for (int g = 0; g < G; ++g) { rcs[g] = OTTER_OK; msgs[g][0] = 0; }
    auto run = [&](int g) {
        rcs[g] = sharded_search_one(shards[g], home, q, io_is_device, nq, k, (float*)h0->sh_gD.p + (size_t)g * nk,
                                    (long long*)h0->sh_gI.p + (size_t)g * nk);
        if (rcs[g]) snprintf(msgs[g], sizeof(msgs[g]), "%s", otter::err_buf());
    };
    for (int g = 1; g < G; ++g) {
        otter_index* ix = shards[g];
        try {
            if (!ix->sh_worker) {
                ShardWorker* w = new ShardWorker();
                try {
                    w->th = std::thread([w] { w->loop(); });
                } catch (...) {
                    delete w;
                    throw;
                }
                ix->sh_worker = w;
            }
            ix->sh_worker->submit([&run, g] { run(g); });
            queued[g] = true;
        } catch (...) {
            queued[g] = false;
        }
    }
    run(0);
    for (int g = 1; g < G; ++g) {
        if (queued[g]) shards[g]->sh_worker->wait();
        else run(g);
    }
    for (int g = 0; g < G; ++g)
        if (rcs[g]) return set_err(rcs[g], "sharded_search: shard %d (device %d): %s", g, shards[g]->device, msgs[g]);
    // G-way merge on the home GPU, hierarchical when G * k exceeds one workgroup's select
    float* oD = D;
    int64_t* oI = I;
    if (!io_is_device) {
        if ((rc = h0->io_D.reserve(nk * sizeof(float)))) return rc;
        if ((rc = h0->io_I.reserve(nk * sizeof(long long)))) return rc;
        oD = (float*)h0->io_D.p;
        oI = (int64_t*)h0->io_I.p;
    }
    const float* cD = (const float*)h0->sh_gD.p;
    const int64_t* cI = (const int64_t*)h0->sh_gI.p;
    int lists = G;
    const int step = std::max(2, MAX_SELECT / k);
    if (lists > step) {
        const size_t lvl = (size_t)((lists + step - 1) / step) * nk;
        if ((rc = h0->sh_tD.reserve(lvl * sizeof(float)))) return rc;
        if ((rc = h0->sh_tI.reserve(lvl * sizeof(long long)))) return rc;
    }
    while (lists > step) {
        // groups of `step` lists -> one list each: merged into the level buffer, then copied back to the front of
        // the gathered buffer, which the next round (or the final merge) reads
        const int groups = (lists + step - 1) / step;
        for (int j = 0; j < groups; ++j) {
            const int g0 = j * step, gn = std::min(step, lists - g0);
            if ((rc = otter_merge_topk(home, cD + (size_t)g0 * nk, cI + (size_t)g0 * nk, gn, nq, k,
                                       (float*)h0->sh_tD.p + (size_t)j * nk, (int64_t*)h0->sh_tI.p + (size_t)j * nk, stream)))
                return rc;
        }
        OTTER_HIP_TRY(hipMemcpyAsync(h0->sh_gD.p, h0->sh_tD.p, (size_t)groups * nk * sizeof(float), hipMemcpyDeviceToDevice, st));
        OTTER_HIP_TRY(hipMemcpyAsync(h0->sh_gI.p, h0->sh_tI.p, (size_t)groups * nk * sizeof(long long), hipMemcpyDeviceToDevice, st));
        lists = groups;
    }
    if ((rc = otter_merge_topk(home, cD, cI, lists, nq, k, oD, oI, stream))) return rc;
    if (!io_is_device) {
        OTTER_HIP_TRY(hipMemcpyAsync(D, oD, nk * sizeof(float), hipMemcpyDeviceToHost, st));
        OTTER_HIP_TRY(hipMemcpyAsync(I, oI, nk * sizeof(long long), hipMemcpyDeviceToHost, st));
    }
    OTTER_HIP_TRY(hipStreamSynchronize(st));
    return OTTER_OK;
}

int otter_inbatch_scores(int device, const float* q, const float* c, const float* neg_c, int B, int d, float* product,
                         int32_t* rank, void* stream) {
    if (B < 0 || d < 1) return set_err(OTTER_EINVAL, "inbatch_scores: bad B=%d d=%d", B, d);
    if (B == 0) return OTTER_OK;
    if (!q || !c || !product || !rank) return set_err(OTTER_EINVAL, "inbatch_scores: NULL buffer");
    if (B > 8192) return set_err(OTTER_EINVAL, "inbatch_scores: B=%d exceeds 8192 (one row of scores is kept in LDS)", B);
    DeviceGuard g(device);
    if (!g.ok) return set_err(OTTER_EHIP, "inbatch_scores: hipSetDevice(%d) failed", device);
    InbatchArgs a;
    a.q = q;
    a.c = c;
    a.neg = neg_c;
    a.product = product;
    a.rank = rank;
    a.B = B;
    a.d = d;
    a.ncol = B + (neg_c ? 1 : 0);
    hipLaunchKernelGGL(inbatch_kernel, dim3((unsigned)B), dim3(256), (size_t)(a.ncol + 1) * sizeof(float), (hipStream_t)stream, a);
    OTTER_HIP_TRY(hipGetLastError());
    return OTTER_OK;
}

}  // extern "C"
