// gemm_bf16.hip — the encoder's bf16 GEMM on gfx950: C = A W^T + bias (+ GELU | + residual).
//
// Serves the four projections of a BERT/RoBERTa layer that the reference reaches through HF's
// AutoModel (retrieval/models/tb_retriever.py:180-185): QKV, attention output, FFN up, FFN down.
//
// Structure (persistent, one 512-thread workgroup per CU, 256 x 256 output tiles, K-step 64):
//   * 8 waves = 2 groups x 4.  Group g owns token rows [128g, 128g+128) of the tile, wave wn of a
//     group owns feature columns [64wn, 64wn+64): a 128 x 64 wave tile held as 8 x 4 accumulators
//     of v_mfma_f32_16x16x32_bf16 (features on the accumulator rows, tokens on the lanes, so the
//     epilogue can store 16 B of consecutive features per lane).
//   * The two groups run the same program one barrier apart ("ping-pong"): while one group's
//     waves issue 16 MFMAs (one 64 x 32 quadrant over the whole K-step) the other group's waves,
//     which share the same four SIMDs, read fragments from LDS and issue LDS-DMA.
//   * LDS holds two K-steps, each as four 16 KiB half-tiles in the order they are consumed:
//     A-first (token rows of quadrant row 0 of both groups), B-first (feature rows of quadrant
//     column 0 of all four waves), B-second, A-second.  One half-tile is DMA'd per phase
//     (global_load_lds_dwordx4, 2 per wave, issued between the two halves of the phase's MFMA segment,
//     where a piece costs less issue time than in a load segment), seven half-tiles ahead of the
//     consumer, across output tiles (the stream of K-steps is flat over this workgroup's tiles), and
//     retired in the load segments with a counted s_waitcnt vmcnt(6): three half-tiles stay in flight,
//     the loop never drains to zero.
//   * Fragments are read from LDS one phase before the MFMAs that use them (a load segment only
//     issues reads and DMA; the counted lgkmcnt lands inside the MFMA segment).
//   * 128-byte LDS rows, 16-B chunks XOR-swizzled through the DMA *source* address
//     (chunk ^ ((row >> 1) & 7)): every ds_read_b128 lane group hits 16 distinct slots.
//   * Epilogue: bias is the accumulators' initial value; v_permlane16_swap pairs two feature
//     blocks so that a lane holds 8 consecutive features of one token -> one 16-B store
//     (and one 16-B residual load).  The two groups' epilogues run in the same barrier interval.
//
// Hazard bookkeeping (segments are the intervals between barriers; group 1 is one behind):
//   RAW  a half-tile retired by the wait in segment j is read no earlier than segment j + 2;
//   WAR  a slot is refilled no earlier than 3 segments after the MFMA segment that consumed it.
// With half-tile h = 4u + p + 7 issued in the MFMA segment of phase p of K-step u (one segment later than
// the load segment that used to issue it: WAR only gains), and the wait of load phase p covering everything
// up to 4u + p + 3, both hold by construction (table in DESIGN.md §4).
//
// LayerNorm folded around the GEMM (template parameter LN, GemmLN in otter_gemm.h): the producer
// (residual epilogue) stores the un-normalised sum and per-row (sum, sum of squares) partials of
// what it stored; the consumer multiplies the un-normalised rows with gamma-scaled weights and
// applies  C = rstd_m * acc - rstd_m * mean_m * s_n + c_n  in its epilogue
// (s_n = sum_k W'[n,k], c_n = bias_n + sum_k beta_k W[n,k]); a residual that is itself an
// un-normalised row is normalised on the fly.  The normalised activation is never written.
#include <algorithm>
#include <vector>

#include "otter_device.h"
#include "otter_gemm.h"

namespace {

using namespace otter;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct PPArgs {
    const char* A;      // [M_pad][lda] bf16
    const char* W;      // [N][ldw] bf16
    const float* bias;  // [N] or null
    const bf16_t* R;    // [M_pad][ldc] bf16 (EPI_BIAS_RESID)
    bf16_t* C;          // [M_pad][ldc] bf16
    int tiles_m, tiles_n, KT, N;
    int lda, ldw, ldc;  // elements
    // LayerNorm folding (LN != 0)
    const float2* stat; // [M_pad] (rstd, -mean * rstd) of the rows of A (EPI 0/1) or of R (EPI 2, LN 2)
    const float* v0;    // [N] s_n (EPI 0/1) or gamma (EPI 2, LN 2)
    const float* v1;    // [N] c_n (EPI 0/1) or beta  (EPI 2, LN 2)
    float2* part;       // EPI 2: [4 * tiles_n][M_pad] (sum, sum of squares) over 64 columns of what was stored
    long long M_pad;
    const int* tiles_m_dev;   // optional: number of 256-row tiles actually in use (<= tiles_m), read on the device
    unsigned long long* dbg;   // GEMM_DIAG builds only
};

#ifdef GEMM_DIAG
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define DG(...) __VA_ARGS__
#else
#define DG(...)
#endif

constexpr int HT = 16384;            // one half-tile: 128 rows x 128 B
constexpr int BUF = 4 * HT;          // one K-step
constexpr int OFF_AF = 0, OFF_BF = HT, OFF_BS = 2 * HT, OFF_AS = 3 * HT;

__device__ __forceinline__ void bar() { asm volatile("s_barrier" ::: "memory"); }

// gelu(x) = 0.5 x (1 + erf(x / sqrt 2)) for a bf16 result, two values per instruction:
// erf(z) ~ z P(z^2) on |z| <= 3.2 (degree-9 P, least-squares fit, |error| <= 9e-6) and clamped
// outside (1 - erf(3.2) = 6e-6).  No transcendental, so every step is a packed f32 op
// (v_pk_mul_f32 / v_pk_fma_f32); evaluated in fp32 the result is within 6.1e-5 absolute and
// 1.4e-5 relative (where |gelu| > 1) of the exact erf form -- far below the bf16 rounding of
// the store (tests/test_gemm_gpu.py compares against torch's erf GELU).
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 gelu_pk(f32x2 x) {
    f32x2 z = x * 0.70710678118654752440f;
    z[0] = __builtin_amdgcn_fmed3f(z[0], -3.2f, 3.2f);
    z[1] = __builtin_amdgcn_fmed3f(z[1], -3.2f, 3.2f);
    const f32x2 t = z * z;
    f32x2 p = t * -2.395521082e-09f + 1.418175454e-07f;
    p = p * t + -3.740854022e-06f;
    p = p * t + 5.852148138e-05f;
    p = p * t + -6.120049626e-04f;
    p = p * t + 4.588824678e-03f;
    p = p * t + -2.583134492e-02f;
    p = p * t + 1.118967519e-01f;
    p = p * t + -3.757361709e-01f;
    p = p * t + 1.128333932e+00f;
    const f32x2 hx = x * 0.5f;
    return hx * (z * p) + hx;
}

template <int EPI, int LN>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(PPArgs a) {
    constexpr bool LNIN = EPI != GEMM_EPI_BIAS_RESID && LN == 1;    // A rows are un-normalised
    constexpr bool STATS = EPI == GEMM_EPI_BIAS_RESID && LN >= 1;   // emit row statistics of C
    constexpr bool LNRES = EPI == GEMM_EPI_BIAS_RESID && LN == 2;   // R rows are un-normalised
    constexpr bool ROWSTAT = LNIN || LNRES;                          // the epilogue needs (rstd, -mean rstd) per row
    // VMEM operations a wave issues between a tile's last MFMA and the next tile's first half-tile
    constexpr int EPI_STORES = 16 + (STATS ? 8 : 0) + (ROWSTAT ? 1 : 0);
    constexpr int NVEC = LNIN ? 2 : LNRES ? 3 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sbias = reinterpret_cast<float*>(smem + 2 * BUF);   // [N] bias (zeros without one); c_n under LNIN
    float* sv0 = sbias + a.N;                                  // [N] s_n (LNIN) / gamma (LNRES)
    float* sv1 = sv0 + a.N;                                    // [N] beta (LNRES)
    // per wave: the 128 (rstd, -mean rstd) pairs of its group's token rows, DMA'd one tile ahead
    // (own region, own vmcnt: no barrier involved)
    float2* sstat = reinterpret_cast<float2*>(sbias + NVEC * a.N) + 128 * (threadIdx.x >> 6);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int bid = blockIdx.x, nblk = gridDim.x;
    const int KT = a.KT;

    // ---- tile schedule.  Workgroups b, b + 8, ... share an XCD and its L2: give every XCD a
    // contiguous range of tile rows and walk it in patches of 8 tile rows x all tile columns
    // (column-major inside a patch), so that the workgroups of an XCD, which take consecutive
    // list positions, work on ~8 A tiles x ~4 W tiles at a time instead of 32 x 1.
    // Any schedule is correct; this one is for L2 hits only.
    // packed (variable-length) batches: the row count is known on the device only
    const int tiles_m = a.tiles_m_dev ? min(__builtin_amdgcn_readfirstlane(*a.tiles_m_dev), a.tiles_m) : a.tiles_m;
    const bool xmode = (nblk & 7) == 0;
    const int wpx = xmode ? nblk >> 3 : nblk;                  // list stride of this workgroup
    const int idx = xmode ? bid >> 3 : bid;                    // first list position
    int tm0 = 0, n_tm = tiles_m;
    if (xmode) {
        const int x = bid & 7;
        tm0 = (int)(((long long)x * tiles_m) >> 3);
        n_tm = (int)(((long long)(x + 1) * tiles_m) >> 3) - tm0;
    }
    const int list_len = n_tm * a.tiles_n;
    const int n_my = idx < list_len ? (list_len - idx + wpx - 1) / wpx : 0;
    const int total_stages = n_my * KT;
    if (total_stages == 0) return;   // whole workgroup
    const int full_blocks = n_tm >> 3, per_block = 8 * a.tiles_n;
    auto coords = [&](int j, int& tm, int& tn) {
        const int pos = idx + wpx * j;
        const int blk = pos / per_block;
        if (blk < full_blocks) {
            const int rem = pos - blk * per_block;
            tn = rem >> 3;
            tm = tm0 + 8 * blk + (rem & 7);
        } else {
            const int bsz = n_tm - 8 * full_blocks;
            const int rem = pos - full_blocks * per_block;
            tn = rem / bsz;
            tm = tm0 + 8 * full_blocks + (rem - tn * bsz);
        }
    };

    if (LNIN) {
        for (int i = tid; i < a.N; i += 512) { sbias[i] = a.v1[i]; sv0[i] = a.v0[i]; }
    } else {
        for (int i = tid; i < a.N; i += 512) sbias[i] = a.bias ? a.bias[i] : 0.0f;
        if (LNRES)
            for (int i = tid; i < a.N; i += 512) { sv0[i] = a.v0[i]; sv1[i] = a.v1[i]; }
    }

    // ---- producer: per-lane source offsets (bytes), fixed for the kernel
    u32 voffA[2], voffB[2];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
        const int r = 8 * (wave * 2 + i2) + (lane >> 3);       // LDS row of the half-tile
        const int c = (lane & 7) ^ ((r >> 1) & 7);             // source chunk landing at position lane & 7
        const int arow = (r & 63) + 128 * (r >> 6);            // A-first: quadrant row 0 of group r >> 6
        const int wrow = 64 * (r >> 5) + (r & 31);             // B-first: quadrant column 0 of wave r >> 5
        voffA[i2] = ((u32)arow * (u32)a.lda + (u32)c * 8u) * 2u;
        voffB[i2] = ((u32)wrow * (u32)a.ldw + (u32)c * 8u) * 2u;
    }
    const u32 lds0 = (u32)(uintptr_t)(OTTER_AS3 char*)smem;
    const u32 dma_dst = (u32)(wave * 2) * 1024u;
    auto issue_rowstat = [&](int tm) {
        if (ROWSTAT)
            dma16s(reinterpret_cast<const char*>(a.stat) + ((size_t)tm * 256 + g * 128) * sizeof(float2), (u32)lane * 16u,
                   lds0 + (u32)(2 * BUF) + (u32)(NVEC * a.N) * 4u + (u32)wave * 1024u);
    };

    int is_stage = 0, is_t = 0, is_j = 0;
    bool is_valid = false;
    const char* is_tileA = a.A;
    const char* is_tileW = a.W;
    const char* is_gA = a.A;
    const char* is_gW = a.W;
    u32 is_lds = lds0;
    auto is_set_tile = [&]() {
        int tm, tn;
        coords(is_j, tm, tn);
        is_tileA = a.A + (size_t)tm * 256 * a.lda * 2;
        is_tileW = a.W + (size_t)tn * 256 * a.ldw * 2;
    };
    is_set_tile();
    // kind: 0 A-first, 1 B-first, 2 B-second, 3 A-second (compile-time at every call site)
    auto issue = [&](int kind) -> bool {
        if (kind == 0) {
            is_valid = is_stage < total_stages;
            if (is_valid) {
                is_gA = is_tileA + (size_t)is_t * 128;
                is_gW = is_tileW + (size_t)is_t * 128;
                is_lds = lds0 + (u32)(is_stage & 1) * BUF + dma_dst;
                ++is_stage;
                if (++is_t == KT) {
                    is_t = 0;
                    if (++is_j < n_my) is_set_tile();
                }
            }
        }
        if (is_valid) {
            if (kind == 0) {
                dma16s(is_gA, voffA[0], is_lds + OFF_AF);
                dma16s(is_gA, voffA[1], is_lds + OFF_AF + 1024);
            } else if (kind == 1) {
                dma16s(is_gW, voffB[0], is_lds + OFF_BF);
                dma16s(is_gW, voffB[1], is_lds + OFF_BF + 1024);
            } else if (kind == 2) {
                const char* b = is_gW + (size_t)32 * a.ldw * 2;
                dma16s(b, voffB[0], is_lds + OFF_BS);
                dma16s(b, voffB[1], is_lds + OFF_BS + 1024);
            } else {
                const char* b = is_gA + (size_t)64 * a.lda * 2;
                dma16s(b, voffA[0], is_lds + OFF_AS);
                dma16s(b, voffA[1], is_lds + OFF_AS + 1024);
            }
        }
        return is_valid;
    };

    // ---- consumer state
    f32x4 acc[8][4];
    int c_j = 0, c_t = 0, c_tm = 0, c_tn = 0;
    coords(0, c_tm, c_tn);
    auto init_acc = [&]() {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const f32x4 b = LNIN ? f32x4{0.f, 0.f, 0.f, 0.f}
                                 : *reinterpret_cast<const f32x4*>(sbias + c_tn * 256 + wn * 64 + nb * 16 + 4 * fq);
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) acc[mb][nb] = b;
        }
    };
    const u32 xrow = (u32)(64 * g + fr) * 128u;
    const u32 wrow = (u32)(32 * wn + fr) * 128u;
    const u32 coff0 = (u32)((fq) ^ (fr >> 1)) << 4;
    const u32 coff1 = (u32)((4 + fq) ^ (fr >> 1)) << 4;
    // fragments: X0 / X1 = token rows of quadrant row 0 / 1, Wf[nq] = feature rows of quadrant column nq
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
    DG(unsigned long long dW = 0, dI = 0;)
    bool last_issued = false;
    auto mma = [&](float4 (&X)[4][2], int mq, int nq, int kind) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[4 * mq + mb][2 * nq + nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, Wf[nq][nb][ks]), __builtin_bit_cast(bf16x8, X[mb][ks]),
                        acc[4 * mq + mb][2 * nq + nb], 0, 0, 0);
            // the phase's half-tile is issued between the two halves of the MFMA segment: a DMA piece costs about
            // 60 cycles of issue here against 100 and more in a load segment, and the load segment is the longer one
            if (ks == 0) {
                __builtin_amdgcn_sched_barrier(0);
#if !defined(GEMM_ABL) || GEMM_ABL != 1
                DG(unsigned long long ti0 = stamp();)
                last_issued = issue(kind);
                DG(dI += stamp() - ti0;)
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // `after_store`: the tile epilogue's 16 stores per lane sit inside the window of the four
    // youngest half-tiles for the next three phases; allow them to stay outstanding too
    // Load-segment wait of phase p of K-step u: half-tile 4u + p + 3 has landed once at most the three younger ones
    // (6 pieces) are outstanding; `after_store`: a tile's stores (and the row statistics) sit among them for the four
    // phases after an epilogue and may stay outstanding too.  Anything else in flight only makes the wait stricter.
    auto wait_landed = [&](bool after_store) {
        DG(unsigned long long tw0 = stamp();)
#if !defined(GEMM_ABL) || GEMM_ABL != 2
        if (!last_issued) wait_vmcnt<0>();
        else if (after_store) wait_vmcnt<6 + EPI_STORES>();
        else wait_vmcnt<6>();
#endif
        DG(dW += stamp() - tw0;)
    };
    auto epilogue = [&]() {
        const int col0 = c_tn * 256 + wn * 64 + 8 * (fq >> 1) + 16 * (fq & 1);
        const size_t row0 = (size_t)c_tm * 256 + g * 128 + fr;
        // two halves of 4 token blocks: the residual's 8 loads of a half are all issued before
        // the first is consumed (one memory latency per half instead of one per 16-B unit)
        // (2 token blocks per batch when the residual is normalised on the fly: registers)
        constexpr int MBB = LNRES ? 2 : 4;
#pragma unroll
        for (int half = 0; half < 8 / MBB; ++half) {
            uint4 rr[MBB][2];
            float2 stt[MBB];
            if (ROWSTAT) {
#pragma unroll
                for (int m4 = 0; m4 < MBB; ++m4) stt[m4] = sstat[(half * MBB + m4) * 16 + fr];
            }
            if (EPI == GEMM_EPI_BIAS_RESID) {
#pragma unroll
                for (int m4 = 0; m4 < MBB; ++m4)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
                        rr[m4][jn] = *reinterpret_cast<const uint4*>(
                            a.R + (row0 + (half * MBB + m4) * 16) * (size_t)a.ldc + col0 + jn * 32);
            }
#pragma unroll
            for (int m4 = 0; m4 < MBB; ++m4) {
                const int mb = half * MBB + m4;
                const size_t rowoff = (row0 + mb * 16) * (size_t)a.ldc;
                float rs = 0.f, rss = 0.f;
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x2 s = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[mb][2 * jn][i]),
                                                                         __float_as_uint(acc[mb][2 * jn + 1][i]),
                                                                         false, false);
                        v[i] = __uint_as_float(s[0]);
                        v[4 + i] = __uint_as_float(s[1]);
                    }
                    if (LNIN) {
                        // LN(A) W^T = rstd (A W'^T) - rstd mean s + c
                        const float* ps = sv0 + col0 + jn * 32;
                        const float* pc = sbias + col0 + jn * 32;
#pragma unroll
                        for (int e4 = 0; e4 < 2; ++e4) {
                            const f32x4 s4 = *reinterpret_cast<const f32x4*>(ps + 4 * e4);
                            const f32x4 c4 = *reinterpret_cast<const f32x4*>(pc + 4 * e4);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                v[4 * e4 + e] = fmaf(stt[m4].x, v[4 * e4 + e], fmaf(stt[m4].y, s4[e], c4[e]));
                        }
                    }
                    if (EPI == GEMM_EPI_BIAS_RESID) {
                        const u32 w[4] = {rr[m4][jn].x, rr[m4][jn].y, rr[m4][jn].z, rr[m4][jn].w};
                        float r[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            r[2 * e] = __uint_as_float(w[e] << 16);
                            r[2 * e + 1] = __uint_as_float(w[e] & 0xFFFF0000u);
                        }
                        if (LNRES) {
                            const float* pg = sv0 + col0 + jn * 32;
                            const float* pb = sv1 + col0 + jn * 32;
#pragma unroll
                            for (int e4 = 0; e4 < 2; ++e4) {
                                const f32x4 g4 = *reinterpret_cast<const f32x4*>(pg + 4 * e4);
                                const f32x4 b4 = *reinterpret_cast<const f32x4*>(pb + 4 * e4);
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    r[4 * e4 + e] = fmaf(fmaf(r[4 * e4 + e], stt[m4].x, stt[m4].y), g4[e], b4[e]);
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r[e];
                    }
                    if (EPI == GEMM_EPI_BIAS_GELU) {
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const f32x2 r = gelu_pk(f32x2{v[e], v[e + 1]});
                            v[e] = r[0];
                            v[e + 1] = r[1];
                        }
                    }
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
                    if (STATS) {
                        // statistics of the values as stored (bf16), so that the consumer's
                        // normalisation is consistent with the rows it reads
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = (float)o[e];
                            rs += f;
                            rss = fmaf(f, f, rss);
                        }
                    }
                    *reinterpret_cast<float4*>(a.C + rowoff + col0 + jn * 32) = __builtin_bit_cast(float4, o);
                }
                if (STATS) {
                    // the four lanes fr, fr+16, fr+32, fr+48 hold the other features of this token
                    u32x2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(rs), __float_as_uint(rs), false, false);
                    rs = __uint_as_float(t[0]) + __uint_as_float(t[1]);
                    t = __builtin_amdgcn_permlane16_swap(__float_as_uint(rss), __float_as_uint(rss), false, false);
                    rss = __uint_as_float(t[0]) + __uint_as_float(t[1]);
                    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(rs), __float_as_uint(rs), false, false);
                    rs = __uint_as_float(t[0]) + __uint_as_float(t[1]);
                    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(rss), __float_as_uint(rss), false, false);
                    rss = __uint_as_float(t[0]) + __uint_as_float(t[1]);
                    if (fq == 0) a.part[(size_t)(c_tn * 4 + wn) * a.M_pad + row0 + mb * 16] = float2{rs, rss};
                }
            }
        }
    };

    DG(unsigned long long dL = 0, dA = 0, dM = 0, dB = 0, dE = 0, dt0 = stamp(), dt = dt0;)
    // diagnostic barriers: time from segment start to barrier arrival, and time spent waiting in it
    auto barA = [&]() { DG(unsigned long long t1 = stamp(); dL += t1 - dt;) bar(); DG(dt = stamp(); dA += dt - t1;) };
    auto barB = [&]() { DG(unsigned long long t1 = stamp(); dM += t1 - dt;) bar(); DG(dt = stamp(); dB += dt - t1;) };

    // ---- prologue: row statistics of the first tile, half-tiles 0..6
    issue_rowstat(c_tm);
    issue(0); issue(1); issue(2); issue(3);
    issue(0); issue(1);
    last_issued = issue(2);
    if (last_issued) wait_vmcnt<8>();   // A-first(0), B-first(0), B-second(0) have landed
    else wait_vmcnt<0>();
    __syncthreads();                 // ... for every wave; the bias vector is in LDS
    init_acc();
    ldX(X0, smem + OFF_AF);
    if (g == 1) bar();               // group 1 runs one barrier behind

    // Fragments are read from LDS one phase before the MFMAs that use them, so a load segment only
    // issues its reads (their latency passes under the other group's MFMAs and this group's own);
    // the one exception is W0, which phase 3 still uses when the next K-step begins.
    // Load segment (u, p) waits until half-tile 4u + p + 3 has landed; MFMA segment (u, p) issues 4u + p + 7.
    for (int cs = 0; cs < total_stages; ++cs) {
        const char* lb = smem + (cs & 1) * BUF;
        const bool as = c_t == 0 && cs > 0;   // first K-step after an epilogue
        // phase 0: quadrant (0, 0)
        ldW(0, lb + OFF_BF);
        ldW(1, lb + OFF_BS);
        wait_landed(as);
        barA();
        mma(X0, 0, 0, 3);
        barB();
        // phase 1: quadrant (0, 1)
        ldX(X1, lb + OFF_AS);
        wait_landed(as);
        barA();
        mma(X0, 0, 1, 0);
        barB();
        // phase 2: quadrant (1, 1); token fragments of the next K-step's quadrant row 0
        if (cs + 1 < total_stages) ldX(X0, smem + ((cs + 1) & 1) * BUF + OFF_AF);
        wait_landed(as);
        barA();
        mma(X1, 1, 1, 1);
        barB();
        // phase 3: quadrant (1, 0)
        wait_landed(as);
        barA();
        mma(X1, 1, 0, 2);
        if (++c_t == KT) {
            // both groups write their tile in the same barrier interval
            c_t = 0;
            DG(unsigned long long te = stamp(); dM += te - dt;)
            if (g == 1) epilogue();
            DG(unsigned long long te1 = stamp();)
            bar();
            DG(unsigned long long te2 = stamp();)
            if (g == 0) epilogue();
            DG(dt = stamp(); dE += (te1 - te) + (dt - te2); dB += te2 - te1;)
            if (++c_j < n_my) {
                coords(c_j, c_tm, c_tn);
                init_acc();
                issue_rowstat(c_tm);
            }
        } else {
            barB();
        }
    }
    if (g == 0) bar();               // pairs with group 1's last barrier
#ifdef GEMM_DIAG
    if (a.dbg && lane == 0) {
        unsigned long long* o = a.dbg + ((size_t)bid * 8 + wave) * 8;
        o[0] = stamp() - dt0; o[1] = dL; o[2] = dA; o[3] = dM; o[4] = dB; o[5] = dE; o[6] = (unsigned long long)total_stages; o[7] = (dI << 32) | (dW & 0xFFFFFFFFull);
    }
#endif
}

// (rstd, -mean * rstd) per row from the producers' partials: one thread per row
__global__ __launch_bounds__(256) void ln_stats_kernel(const float2* __restrict__ part, int P, long long M_pad, int N,
                                                       float eps, float2* __restrict__ stat) {
    const long long r = blockIdx.x * 256ll + threadIdx.x;
    if (r >= M_pad) return;
    float s = 0.f, ss = 0.f;
    for (int p = 0; p < P; ++p) {
        const float2 v = part[(size_t)p * M_pad + r];
        s += v.x;
        ss += v.y;
    }
    const float mean = s / N;
    const float var = fmaxf(ss / N - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    stat[r] = float2{rstd, -mean * rstd};
}

// W'[n,k] = bf16(gamma_k W[n,k]);  s_n = sum_k W'[n,k];  c_n = bias_n + sum_k beta_k W[n,k].  One block per n.
__global__ __launch_bounds__(256) void fold_ln_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ bias,
                                                      bf16_t* __restrict__ Wout, float* __restrict__ s_out,
                                                      float* __restrict__ c_out, int K) {
    __shared__ float red[2][4];
    const int n = blockIdx.x, tid = threadIdx.x;
    float s = 0.f, c = 0.f;
    for (int k = tid; k < K; k += 256) {
        const float w = W[(size_t)n * K + k];
        const __bf16 wq = (__bf16)(gamma[k] * w);
        Wout[(size_t)n * K + k] = __builtin_bit_cast(bf16_t, wq);
        s += (float)wq;
        c = fmaf(beta[k], w, c);
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        c += __shfl_xor(c, o);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = c; }
    __syncthreads();
    if (tid == 0) {
        s_out[n] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        c_out[n] = (bias ? bias[n] : 0.f) + red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

}  // namespace

namespace otter {

int launch_ln_stats(const void* part, int P, long long M_pad, int N, float eps, void* stat, hipStream_t st) {
    hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)((M_pad + 255) / 256)), dim3(256), 0, st, (const float2*)part, P,
                       M_pad, N, eps, (float2*)stat);
    OTTER_HIP_TRY(hipGetLastError());
    return OTTER_OK;
}

int launch_fold_ln(const float* W, const float* gamma, const float* beta, const float* bias, void* Wout, float* s_out,
                   float* c_out, int N, int K, hipStream_t st) {
    hipLaunchKernelGGL(fold_ln_kernel, dim3(N), dim3(256), 0, st, W, gamma, beta, bias, (bf16_t*)Wout, s_out, c_out, K);
    OTTER_HIP_TRY(hipGetLastError());
    return OTTER_OK;
}

int launch_gemm_bf16(int epi, const void* A, const void* W, const float* bias, const void* R, void* C,
                     long long M_pad, int N, int K, int ncu, hipStream_t st) {
    GemmLN none;
    return launch_gemm_bf16_ln(epi, A, W, bias, R, C, M_pad, N, K, none, ncu, st);
}

int launch_gemm_bf16_ln(int epi, const void* A, const void* W, const float* bias, const void* R, void* C,
                        long long M_pad, int N, int K, const GemmLN& ln, int ncu, hipStream_t st) {
    if (M_pad <= 0 || M_pad % 256 || N <= 0 || N % 256 || K <= 0 || K % 64)
        return set_err(OTTER_EINVAL, "gemm_bf16: M_pad=%lld N=%d K=%d must be multiples of 256/256/64", M_pad, N, K);
    if (epi < 0 || epi > 2) return set_err(OTTER_EINVAL, "gemm_bf16: epilogue %d", epi);
    if (epi == GEMM_EPI_BIAS_RESID && !R) return set_err(OTTER_EINVAL, "gemm_bf16: residual epilogue without R");
    if ((long long)255 * std::max(N, K) * 2 + 128 > 0x7FFFFFFFll || M_pad / 256 * (N / 256) > 0x7FFFFFFFll)
        return set_err(OTTER_EINVAL, "gemm_bf16: shape too large");
    const bool resid = epi == GEMM_EPI_BIAS_RESID;
    if (ln.mode < 0 || ln.mode > 2 || (!resid && ln.mode == 2))
        return set_err(OTTER_EINVAL, "gemm_bf16: LayerNorm mode %d with epilogue %d", ln.mode, epi);
    if (ln.mode && resid && !ln.part) return set_err(OTTER_EINVAL, "gemm_bf16: statistics output missing");
    if (((ln.mode == 1 && !resid) || ln.mode == 2) && (!ln.stat || !ln.v0 || !ln.v1))
        return set_err(OTTER_EINVAL, "gemm_bf16: LayerNorm folding needs stat, v0, v1");
    PPArgs a;
    a.A = (const char*)A;
    a.W = (const char*)W;
    a.bias = bias;
    a.R = (const bf16_t*)R;
    a.C = (bf16_t*)C;
    a.tiles_m = (int)(M_pad / 256);
    a.tiles_n = N / 256;
    a.KT = K / 64;
    a.N = N;
    a.lda = K;
    a.ldw = K;
    a.ldc = N;
    a.stat = (const float2*)ln.stat;
    a.v0 = ln.v0;
    a.v1 = ln.v1;
    a.part = (float2*)ln.part;
    a.M_pad = M_pad;
    a.tiles_m_dev = ln.tiles_m_dev;
    void (*kern)(PPArgs) = nullptr;
    int nvec = 1;
    if (epi == GEMM_EPI_BIAS) {
        kern = ln.mode ? gemm_pp_kernel<GEMM_EPI_BIAS, 1> : gemm_pp_kernel<GEMM_EPI_BIAS, 0>;
        nvec = ln.mode ? 2 : 1;
    } else if (epi == GEMM_EPI_BIAS_GELU) {
        kern = ln.mode ? gemm_pp_kernel<GEMM_EPI_BIAS_GELU, 1> : gemm_pp_kernel<GEMM_EPI_BIAS_GELU, 0>;
        nvec = ln.mode ? 2 : 1;
    } else {
        kern = ln.mode == 2 ? gemm_pp_kernel<GEMM_EPI_BIAS_RESID, 2>
               : ln.mode == 1 ? gemm_pp_kernel<GEMM_EPI_BIAS_RESID, 1> : gemm_pp_kernel<GEMM_EPI_BIAS_RESID, 0>;
        nvec = ln.mode == 2 ? 3 : 1;
    }
    const bool rowstat = ln.mode == 2 || (ln.mode == 1 && !resid);
    const int lds = 2 * BUF + nvec * N * (int)sizeof(float) + (rowstat ? 8 * 1024 : 0);
    if (lds > 160 * 1024) return set_err(OTTER_EINVAL, "gemm_bf16: N=%d too wide for the LDS epilogue vectors", N);
    const int ntiles = a.tiles_m * a.tiles_n;
    OTTER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = std::min(ntiles, std::max(1, ncu));
    a.dbg = nullptr;
#ifdef GEMM_DIAG
    static DevBuf dbgbuf;
    const size_t dbg_n = (size_t)grid * 8 * 8;
    { int rc = dbgbuf.reserve(dbg_n * 8); if (rc) return rc; }
    OTTER_HIP_TRY(hipMemsetAsync(dbgbuf.p, 0, dbg_n * 8, st));
    a.dbg = (unsigned long long*)dbgbuf.p;
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
    OTTER_HIP_TRY(hipGetLastError());
#ifdef GEMM_DIAG
    {
        std::vector<unsigned long long> h(dbg_n);
        OTTER_HIP_TRY(hipStreamSynchronize(st));
        OTTER_HIP_TRY(hipMemcpy(h.data(), dbgbuf.p, dbg_n * 8, hipMemcpyDeviceToHost));
        for (int grp = 0; grp < 2; ++grp) {
            double t[7] = {0, 0, 0, 0, 0, 0, 0}, n = 0, tI = 0, tW = 0;
            for (size_t i = 0; i < dbg_n; i += 8) {
                if (!h[i] || (int)((i / 8) % 8) / 4 != grp) continue;
                for (int k = 0; k < 7; ++k) t[k] += (double)h[i + k];
                tI += (double)(h[i + 7] >> 32); tW += (double)(h[i + 7] & 0xFFFFFFFFull);
                n += 1;
            }
            if (n > 0)
                fprintf(stderr, "[gemm diag] epi=%d N=%d K=%d group %d: per K-step cycles total %.0f = load-seg %.0f + barA-wait %.0f + "
                        "mfma-seg %.0f + barB-wait %.0f + epilogue %.0f  (stages/wg %.0f); in load-seg: dma issue %.0f vmcnt wait %.0f\n", epi, N, K, grp, t[0] / t[6],
                        t[1] / t[6], t[2] / t[6], t[3] / t[6], t[4] / t[6], t[5] / t[6], t[6] / n, tI / t[6], tW / t[6]);
        }
    }
#endif
    return OTTER_OK;
}

}  // namespace otter
