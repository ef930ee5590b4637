// encoder.hip — BERT/RoBERTa-base bi-encoder forward on gfx950 (fp32 parity path and bf16
// throughput path share every kernel through a storage-type template).
//
// Replaces `model(batch)['embed']` of the reference's encoder classes
// (retrieval/models/tb_retriever.py:150-261, retrieval/models/retriever.py:249-345) including
// the HF AutoModel forward they call (tb_retriever.py:180-185).
//
// Activations live in HBM as row-major [M_pad, width] in the compute type T (float or bf16),
// M = B*L tokens, M_pad = M rounded up to the 256-row GEMM tile (+64 rows of slack for the
// attention key tiles); padding rows are zero and are never written with non-finite values.
//
// Per layer:  QKV = X Wqkv^T + b                (gemm, 256x256 tile, LDS-DMA ring, MFMA)
//             CTX = softmax(Q K^T / 8 + mask) V  (flash-style, scores never leave registers)
//             T1  = CTX Wo^T + b + X ; X1 = LN(T1)
//             MID = gelu(X1 W1^T + b)
//             T2  = MID W2^T + b + X1 ; X = LN(T2)
#include <math.h>
#include <stdlib.h>

#include <map>
#include <new>
#include <string>
#include <vector>

#include "otter_common.h"
#include "otter_device.h"
#include "otter_gemm.h"

namespace {

using namespace otter;

// ------------------------------------------------------------------ GEMM (NT)
// C[M_pad, N] = A[M_pad, K] * W[N, K]^T (+ bias[N]) (+ gelu | + R[M_pad, N]); fp32 accumulate.
// 256 x 256 tile, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA blocks of 32 x 32.
// LDS stage: 128-byte rows (32 floats / 64 bf16 of k) of the 256 A rows and 256 W rows = 64 KB,
// two stages.  Fragment reads are 16 B per lane (chunk 2j + h of the row): for fp32 that is four
// k-steps of v_mfma_f32_32x32x2_f32, for bf16 one v_mfma_f32_32x32x16_bf16.
enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESID = 2 };

struct GemmArgs {
    const void* A;
    const void* W;
    const float* bias;
    const void* R;
    void* C;
    int tiles_m, tiles_n;
    int K;              // elements
    int lda, ldw, ldc;  // elements
};

template <typename T>
__device__ __forceinline__ void mma_chunk(f32x16& acc, const float4& a, const float4& b);
template <>
__device__ __forceinline__ void mma_chunk<float>(f32x16& acc, const float4& a, const float4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<bf16_t>(f32x16& acc, const float4& a, const float4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0,
                                                  0, 0);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_kernel(GemmArgs g) {
    constexpr int BM = 256, BN = 256, ROWB = 128, CH = 8, RPI = 8, NW = 8;
    constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int A_INSTR = BM / RPI / NW, B_INSTR = BN / RPI / NW, IPS = A_INSTR + B_INSTR;
    constexpr int KE = ROWB / (int)sizeof(T);   // k elements per stage
    constexpr int EPC = 16 / (int)sizeof(T);    // elements per 16-byte chunk
    constexpr int MB = 4, NB = 2, NG = CH / 2;  // blocks per wave, k-groups per stage

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = wg / g.tiles_n, tn = wg - tm * g.tiles_n;
    const int KS = g.K / KE;

    auto swz = [](int r) { return (r >> 1) & 7; };
    const int dma_r = lane / CH, dma_p = lane % CH;
    const u32 lds0 = (u32)(uintptr_t)(OTTER_AS3 char*)smem;
    u32 voff[IPS];
#pragma unroll
    for (int t = 0; t < IPS; ++t) {
        const int i = (t < A_INSTR ? t : t - A_INSTR) * NW + wave;
        const int r = RPI * i + dma_r;
        const u32 ld = t < A_INSTR ? (u32)g.lda : (u32)g.ldw;
        voff[t] = ((u32)r * ld + (u32)((dma_p ^ swz(r)) * EPC)) * (u32)sizeof(T);
    }
    const T* gA0 = (const T*)g.A + (size_t)tm * BM * g.lda;
    const T* gW0 = (const T*)g.W + (size_t)tn * BN * g.ldw;
    auto produce = [&](int ks, int slot, int t) {
        const u32 base = lds0 + slot * STAGE_BYTES;
        if (t < A_INSTR) dma16s(gA0 + (size_t)ks * KE, voff[t], base + (t * NW + wave) * 1024);
        else dma16s(gW0 + (size_t)ks * KE, voff[t], base + A_BYTES + ((t - A_INSTR) * NW + wave) * 1024);
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

#pragma unroll
    for (int t = 0; t < IPS; ++t) produce(0, 0, t);

    const int h = lane >> 5, l31 = lane & 31, lswz = swz(l31);
    int slot = 0;
    for (int ks = 0; ks < KS; ++ks) {
        wait_vmcnt<0>();   // stage ks (issued one stage ago) has landed for this wave
        wg_barrier();      // ... and for every wave; slot^1 is free again
        const char* sA = smem + slot * STAGE_BYTES;
        const char* sB = sA + A_BYTES;
        float4 af[2][MB], bf[2][NB];
        auto load_frags = [&](int j, int buf) {
            const int coff = ((2 * j + h) ^ lswz) << 4;
#pragma unroll
            for (int m = 0; m < MB; ++m)
                af[buf][m] = *reinterpret_cast<const float4*>(sA + ((wm * MB + m) * 32 + l31) * ROWB + coff);
#pragma unroll
            for (int n = 0; n < NB; ++n)
                bf[buf][n] = *reinterpret_cast<const float4*>(sB + ((wn * NB + n) * 32 + l31) * ROWB + coff);
        };
        load_frags(0, 0);
        const bool more = ks + 1 < KS;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int cb = j & 1;
            if (j + 1 < NG) load_frags(j + 1, cb ^ 1);
            if (more) {
#pragma unroll
                for (int t = j; t < IPS; t += NG) produce(ks + 1, slot ^ 1, t);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) mma_chunk<T>(acc[m][n], af[cb][m], bf[cb][n]);
        }
        slot ^= 1;
    }

    // epilogue: lane owns column (l31) of each block, 16 rows per block
    T* C = (T*)g.C;
    const T* R = (const T*)g.R;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int col = tn * BN + (wn * NB + n) * 32 + l31;
        const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * BM + (wm * MB + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = acc[m][n][r] + bv;
                if (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
                if (EPI == EPI_BIAS_RESID) v += to_f<T>(R[(size_t)row * g.ldc + col]);
                C[(size_t)row * g.ldc + col] = from_f<T>(v);
            }
        }
    }
}

// ------------------------------------------------------------------ attention
// One workgroup = 128 query rows of one (batch, head); 4 waves x 32 queries.  Scores are computed
// transposed (S^T = K Q^T: keys on the accumulator rows, the query on the lane), so the softmax
// statistics of a query are lane-local and the probability tile is already the B operand of the
// second product O^T = V^T P^T ("accumulator tile as the next MFMA's operand").
struct AttnArgs {
    const void* QKV;    // [M_pad + 64, 3H]: q | k | v
    void* CTX;          // [M_pad, H]
    const float* kbias; // [B][Lp]: 0 for tokens, -1e30 for padding and for keys in [L, Lp)
    int B, L, Lp, H, heads;
    float scale;
    // packed (variable-length) batches, bf16 kernel only: sequence b occupies rows
    // [seq_off[b], seq_off[b] + seq_len[b]) of QKV / CTX, seq_len a multiple of 64 (<= Lp)
    const int* seq_off = nullptr;
    const int* seq_len = nullptr;
    // attn_pipe_kernel -> attn_kernel: one byte per 32-query block, [B][heads][16], non-zero = "recompute this block"
    // (attn_kernel with flags set works on the 128-query groups that hold a flagged block and exits on the others)
    unsigned char* flags = nullptr;
    int ncu = 256;
};

template <typename T>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs a) {
    constexpr int DH = 64, KB = 64;
    constexpr int ROWB = DH * (int)sizeof(T);   // 128 (bf16) / 256 (fp32) bytes per K/V row
    constexpr int CH = ROWB / 16;               // 16-byte chunks per row
    constexpr int NQ = CH / 2;                  // fragment chunks per lane
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr bool BF = sizeof(T) == 2;
    __shared__ __attribute__((aligned(16))) char sK[KB * ROWB];
    __shared__ __attribute__((aligned(16))) char sV[KB * ROWB];
    __shared__ float sBias[KB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, head = blockIdx.y;
    // as the repair pass behind attn_pipe_kernel: only the 128-query groups that hold a flagged block
    if (a.flags && reinterpret_cast<const unsigned*>(a.flags)[((size_t)b * a.heads + head) * 4 + blockIdx.x] == 0u) return;
    const int L = a.seq_len ? a.seq_len[b] : a.L;                       // rows of this sequence (packed: a multiple of 64)
    const size_t row0 = a.seq_off ? (size_t)a.seq_off[b] : (size_t)b * a.L;
    if ((int)blockIdx.x * 128 >= L) return;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int ld = 3 * a.H;
    const T* base = (const T*)a.QKV + row0 * ld + head * DH;
    auto swz = [](int r) { return CH == 16 ? (r & 15) : ((r >> 1) & 7); };

    // this lane's query row (clamped: rows past L are computed and dropped)
    const int qrow = min(q0 + l31, L - 1);
    float4 qf[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j)
        qf[j] = *reinterpret_cast<const float4*>(base + (size_t)qrow * ld + (2 * j + h) * EPC);

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.0f;
    float mrun = -INFINITY, lrun = 0.0f;

    for (int kb0 = 0; kb0 < L; kb0 += KB) {
        __syncthreads();
        // stage K (swizzled chunks) and V (row-major) of keys kb0 .. kb0+63
        for (int c = tid; c < KB * CH; c += 256) {
            const int r = c / CH, p = c % CH;
            const T* src = base + (size_t)(kb0 + r) * ld;
            *reinterpret_cast<float4*>(sK + r * ROWB + ((p ^ swz(r)) << 4)) =
                *reinterpret_cast<const float4*>(src + a.H + p * EPC);
            *reinterpret_cast<float4*>(sV + r * ROWB + (p << 4)) =
                *reinterpret_cast<const float4*>(src + 2 * a.H + p * EPC);
        }
        if (tid < KB) sBias[tid] = a.kbias[(size_t)b * a.Lp + kb0 + tid];
        __syncthreads();

        // S^T = K Q^T for the two 32-key halves
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = 0.0f;
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int krow = sub * 32 + l31;
                const float4 kf =
                    *reinterpret_cast<const float4*>(sK + krow * ROWB + (((2 * j + h) ^ swz(krow)) << 4));
                mma_chunk<T>(s[sub], kf, qf[j]);
            }
        }
        // scale, key-padding bias, running max
        float mx = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = s[sub][r] * a.scale + sBias[key];
                s[sub][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __expf(mrun - mnew);
        float psum = 0.0f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(s[sub][r] - mnew);
                s[sub][r] = p;
                psum += p;
            }
        psum += __shfl_xor(psum, 32);
        lrun = lrun * alpha + psum;
        mrun = mnew;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

        // O^T += V^T P^T
        if constexpr (BF) {
            const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3, dcol0 = 16 * ((lane >> 4) & 1);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    // B fragment: registers 8*s2 .. 8*s2+7 of the probability tile, as bf16
                    bf16x8 pb;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pb[j] = (__bf16)s[sub][8 * s2 + j];
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        // A fragment: V^T[d][key], key = 32 sub + 16 s2 + 8 (j>>2) + 4 h + (j&3),
                        // two transposed LDS reads of 4 keys x 16 columns per 16-lane group
                        const int kbase = sub * 32 + s2 * 16 + 4 * h;
                        const char* p0 = sV + (kbase + qq) * ROWB + (d * 32 + dcol0 + 4 * pp) * 2;
                        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((OTTER_AS3 s16x4*)p0);
                        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((OTTER_AS3 s16x4*)(p0 + 8 * ROWB));
                        short va[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        bf16x8 av;
                        __builtin_memcpy(&av, va, 16);
                        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, pb, o[d], 0, 0, 0);
                    }
                }
        } else {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int key = sub * 32 + (t & 3) + 8 * (t >> 2) + 4 * h;
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const float av = *reinterpret_cast<const float*>(sV + key * ROWB + (d * 32 + l31) * 4);
                        o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s[sub][t], o[d], 0, 0, 0);
                    }
                }
        }
    }

    // normalise and store: lane = query, registers = head dimension
    const int q = q0 + l31;
    if (q < L) {
        const float inv = 1.0f / lrun;
        T* dst = (T*)a.CTX + (row0 + q) * a.H + head * DH;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[d * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = from_f<T>(o[d][r] * inv);
    }
}

// ------------------------------------------------------------------ attention, bf16 throughput path
// One 512-thread workgroup per (batch, head); L <= 512.  All of K and V of the head are brought
// into LDS once by LDS-DMA (64-key tiles, 2 pieces per wave per tile, issued back to back) and
// released tile by tile with a counted s_waitcnt vmcnt + one barrier each; after the fill there is
// no further synchronisation: every wave owns 64 queries (two 32-query blocks sharing each K / V
// fragment read) and walks all keys with an online softmax in base 2
// (p = exp2(s * scale * log2 e + bias - max)).  Layout of the products as in attn_kernel: scores
// transposed (keys on accumulator rows, the query on the lane), the probability tile is the next
// MFMA's B operand, V read transposed with ds_read_b64_tr_b16.  The context row is stored 16 B
// per lane after a v_permlane32_swap of the two lane halves.
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void dma4s(const void* sbase, u32 voff, u32 lds_dst) {   // 64 lanes x 4 B
    u32 keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

__device__ __forceinline__ void wait_vmcnt_upto7(int n) {   // waits for AT MOST n outstanding: n rounded DOWN to a listed value
    if (n > 16) n = 16;
    if (n > 8) n &= ~1;
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 7: wait_vmcnt<7>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 10: wait_vmcnt<10>(); break;
        case 12: wait_vmcnt<12>(); break;
        case 14: wait_vmcnt<14>(); break;
        case 16: wait_vmcnt<16>(); break;
        default: wait_vmcnt<0>(); break;    // (unreachable; waiting for everything is always safe)
    }
}

template <int QB>
__global__ __launch_bounds__(1024 / QB, QB == 2 ? 2 : 4) void attn_full_kernel(AttnArgs a) {
    constexpr int NW = 16 / QB;      // waves; each moves PPW of the 16 1-KiB pieces of a 64-key tile
    constexpr int PPW = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y;
    const int Lb = a.seq_len ? __builtin_amdgcn_readfirstlane(a.seq_len[b]) : a.L;       // rows of this sequence
    const int nt = a.seq_len ? Lb >> 6 : a.Lp >> 6;                                      // 64-key tiles
    const size_t row0 = a.seq_off ? (size_t)__builtin_amdgcn_readfirstlane(a.seq_off[b]) : (size_t)b * a.L;
    char* sK = smem;                                // [Lp][128 B], chunks swizzled by (row >> 1) & 7
    char* sV = smem + nt * 8192;                    // [Lp][128 B], linear
    float* sB = reinterpret_cast<float*>(smem + 2 * nt * 8192);   // [Lp] additive key bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int head = blockIdx.x;
    const int ld = 3 * a.H;
    const bf16_t* base = (const bf16_t*)a.QKV + row0 * ld + head * 64;
    const int q0 = wave * 32 * QB;
    const bool active = q0 < Lb;                    // wave-uniform

    // fill: key bias first (the only conditional issue: everything after it is counted the same in every wave)
    const u32 lds0 = (u32)(uintptr_t)(OTTER_AS3 char*)smem;
    if (wave < nt) dma4s(a.kbias + (size_t)b * a.Lp + wave * 64, (u32)lane * 4u, lds0 + 2 * nt * 8192 + wave * 256);
    // this lane's two query rows: loads hidden from hipcc (it would drain the DMA queue for them)
    f32x4v qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = min(q0 + qb * 32 + l31, Lb - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16_t* p = base + (size_t)qrow * ld + (2 * j + h) * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[qb][j]) : "v"(p) : "memory");
        }
    }
    // then K and V tile by tile (wave w moves rows 8w .. 8w+7 of every tile)
    {
        // piece p of a tile: p < 8 -> K rows 8p .. 8p+7 (swizzled), else V rows 8(p-8) .. (linear)
        u32 voff[PPW], dst[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave * PPW + i, isv = p >> 3, r = 8 * (p & 7) + (lane >> 3);
            const int c = isv ? (lane & 7) : ((lane & 7) ^ ((r >> 1) & 7));
            voff[i] = ((u32)r * (u32)ld + (u32)(isv ? 2 * a.H : a.H) + (u32)(c * 8)) * 2u;
            dst[i] = lds0 + (u32)(isv ? nt * 8192 : 0) + (u32)(p & 7) * 1024u;
        }
        for (int t = 0; t < nt; ++t) {
            const char* tb = reinterpret_cast<const char*>(base) + (size_t)t * 64 * ld * 2;
#pragma unroll
            for (int i = 0; i < PPW; ++i) dma16s(tb, voff[i], dst[i] + t * 8192);
        }
    }
    // The query rows are older than the PPW * nt pieces just issued: wait for exactly them, here, in straight-line
    // code, and only then let the compiler see the registers as written.  (Without this the register copies it
    // places ahead of the tile loop read the destinations before the loads have landed.)
    wait_vmcnt_upto7(PPW * nt);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
        asm volatile("" : "+v"(qf[qb][0]), "+v"(qf[qb][1]), "+v"(qf[qb][2]), "+v"(qf[qb][3]));

    f32x16 o[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][d][r] = 0.0f;
    float mrun[QB], lrun[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { mrun[qb] = -INFINITY; lrun[qb] = 0.0f; }
    const float scale2 = a.scale * 1.4426950408889634f;
    const int lswz = (l31 >> 1) & 7;
    const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3, dcol0 = 16 * ((lane >> 4) & 1);

    // Optional (-DATTN_STAGGER): the two waves that share a SIMD (w and w + 4) run one phase apart: waves 0-3 do
    // QK^T, softmax and PV of tile t in interval t; waves 4-7 do softmax and PV of tile t - 1 and then QK^T of tile t,
    // so one partner's softmax (vector ALU) runs beside the other's products (matrix cores).  Same instructions on the
    // same data per query: results unchanged.  It measured no faster than running in step (the hardware already
    // interleaves the two waves of a SIMD), so it is off by default.
    f32x16 s[QB][2];
    auto qk = [&](int t) {
        const char* kt = sK + t * 8192;

        // S^T = K Q^T: [qb][sub] = 32 keys x 32 queries
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[qb][sub][r] = 0.0f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 kf =
                    *reinterpret_cast<const float4*>(kt + (sub * 32 + l31) * 128 + (((2 * j + h) ^ lswz) << 4));
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    s[qb][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[qb][j]), s[qb][sub], 0, 0, 0);
            }
    };
    auto sm_pv = [&](int t) {
        const char* vt = sV + t * 8192;
        // does this tile hold a masked key (padding)?  wave-uniform; bias values are 0 or -1e30.  One bias word per
        // lane (64 keys, 64 lanes); the lane's own 32 bias values are read only on the masked path
        const bool masked = __ballot(sB[t * 64 + lane] != 0.0f) != 0ull;

#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            // row maximum in the exponent's units (scale2 > 0: max commutes with the scaling)
            float mx = -INFINITY;
            if (masked) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float4 bb = *reinterpret_cast<const float4*>(sB + t * 64 + sub * 32 + 8 * (r >> 2) + 4 * h);
                        const float bias = (r & 3) == 0 ? bb.x : (r & 3) == 1 ? bb.y : (r & 3) == 2 ? bb.z : bb.w;
                        const float v = fmaf(s[qb][sub][r], scale2, bias);
                        s[qb][sub][r] = v;
                        mx = fmaxf(mx, v);
                    }
            } else {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][sub][r]);
                mx *= scale2;
            }
            {
                const u32x2v sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            // deferred rescale: the reference maximum moves only when some row's maximum grew by
            // more than 2^8 (p <= 256 in between: safe in fp32 sums and in the bf16 P operand);
            // the first tile always takes it (mrun = -inf)
            if (__ballot(mx > mrun[qb] + 8.0f) != 0ull) {
                const float mnew = fmaxf(mrun[qb], mx);
                const float alpha = __builtin_amdgcn_exp2f(mrun[qb] - mnew);
                lrun[qb] *= alpha;
                mrun[qb] = mnew;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][d][r] *= alpha;
            }
            const float mref = mrun[qb];
            float psum = 0.0f;
            if (masked) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(s[qb][sub][r] - mref);
                        s[qb][sub][r] = pv;
                        psum += pv;
                    }
            } else {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(fmaf(s[qb][sub][r], scale2, -mref));
                        s[qb][sub][r] = pv;
                        psum += pv;
                    }
            }
            {
                const u32x2v sw =
                    __builtin_amdgcn_permlane32_swap(__float_as_uint(psum), __float_as_uint(psum), false, false);
                psum = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            lrun[qb] += psum;
        }

        // O^T += V^T P^T
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pb[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pb[qb][j] = (__bf16)s[qb][sub][8 * s2 + j];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int kbase = sub * 32 + s2 * 16 + 4 * h;
                    const char* p0 = vt + (kbase + qq) * 128 + (d * 32 + dcol0 + 4 * pp) * 2;
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((OTTER_AS3 s16x4*)p0);
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((OTTER_AS3 s16x4*)(p0 + 8 * 128));
                    short va[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    bf16x8 av;
                    __builtin_memcpy(&av, va, 16);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
                        o[qb][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, pb[qb], o[qb][d], 0, 0, 0);
                }
            }
    };
#ifdef ATTN_STAGGER
    const bool late = QB == 2 && wave >= NW / 2;      // wave-uniform
#else
    const bool late = false;   // measured (B = 128, L = 512): 14.06 ms per batch staggered, 13.97 ms in step -- off
#endif
    auto tile_landed = [&](int t) {
        // tile t (and everything older: Q, bias) has landed for this wave, then for all waves
        wait_vmcnt_upto7(PPW * (nt - 1 - t));
        asm volatile("s_barrier" ::: "memory");
    };
    // (two loops, not one loop with a branch inside: each keeps its own, smaller set of live registers -- this kernel
    // must not spill: its hidden asynchronous loads land in registers the compiler believes are already written)
    if (!late) {
        for (int t = 0; t < nt; ++t) {
            tile_landed(t);
            if (!active) continue;
            qk(t);
            sm_pv(t);
        }
    } else {
        for (int t = 0; t < nt; ++t) {
            tile_landed(t);
            if (!active) continue;
            if (t > 0) sm_pv(t - 1);
            qk(t);
        }
        if (active && nt > 0) sm_pv(nt - 1);
    }

    // normalise and store: lanes l and l + 32 hold one query; after the swap a lane holds 8
    // consecutive head dimensions (16 B): half 0 -> [8 rq, 8 rq + 8), half 1 -> the next 8
    if (active) {
        // (the lane index is made opaque here so that the store addresses are derived after the loops: computed at
        // kernel entry they would be spilled around them, and this kernel must not spill)
        int l31e = l31;
        asm volatile("" : "+v"(l31e));
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int q = q0 + qb * 32 + l31e;
            const float inv = 1.0f / lrun[qb];
            bf16_t* dst = (bf16_t*)a.CTX + (row0 + min(q, Lb - 1)) * a.H + head * 64;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int rq = 0; rq < 4; rq += 2) {
                    bf16x8 ov;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x2v sw = __builtin_amdgcn_permlane32_swap(
                            __float_as_uint(o[qb][d][4 * rq + i] * inv), __float_as_uint(o[qb][d][4 * rq + 4 + i] * inv),
                            false, false);
                        ov[i] = (__bf16)__uint_as_float(sw[0]);
                        ov[4 + i] = (__bf16)__uint_as_float(sw[1]);
                    }
                    if (q < Lb)
                        *reinterpret_cast<float4*>(dst + d * 32 + 8 * (rq + h)) = __builtin_bit_cast(float4, ov);
                }
        }
    }
}

// ------------------------------------------------------------------ attention, bf16, software-pipelined, persistent
// 512 threads = 8 waves, two per SIMD, 256 arch VGPRs each and NO accumulator file (with any AGPR use a 512-thread
// kernel gets a fixed 128 / 128 split, which spills); min(B * heads, CUs) workgroups, each walking (batch, head) items
// blockIdx, blockIdx + grid, ...  All K and V of an item's head are resident in LDS (64 KiB + 64 KiB + key bias),
// same image, fill pieces and store as attn_full_kernel.  Wave w owns the 32-query blocks w and w + 8.
// The unit of work is one 32-query block against one 64-key tile; the steps of a unit are spread over three
// consecutive stages, so that ONE stage (one basic block, 16 slots of one MFMA + a slice of vector work each,
// hand-ordered and pinned with sched_barrier) holds
//     the score products of unit u + 1          (8 MFMAs, 8 K fragment reads)
//     the context products of unit u - 1        (8 MFMAs, 16 transposed V reads)
//     exponentials, row sums, bf16 packing of u (vector ALU)
// i.e. the matrix pipe and the vector ALU always have independent work from the same instruction stream; the second
// wave of the SIMD fills what issue slots are left (one wave per SIMD with the whole 512-register file was
// issue-bound: its MFMA and vector parts simply added up).
// Per score the vector ALU does one v_exp, one add and half a packed convert; nothing else:
//   * the scale: the query rows arrive pre-multiplied by (1/8) log2(e) -- folded into W_q / b_q at finalize;
//   * the key-padding bias is the INITIAL VALUE of the score accumulators on tiles that hold a masked key (one ballot
//     over the tile's 64 bias words per item), the MFMA's constant 0 elsewhere;
//   * no running maximum: softmax is invariant under a per-row factor -- p = 2^s and the row sum carry the same
//     one -- and fp32 / bf16 hold 2^-126 .. 2^127, while scores of a trained encoder are a few tens at most.
//     Instead of watching the maximum, the RESULT is checked: a block of queries is accepted iff every row sum
//     lies in (1e-30, 1e30) (then no exponential overflowed, no context sum can have, and nothing underflowed);
//     otherwise -- scores beyond +-100, or rows with every key masked -- the block is flagged (a.flags, one byte per
//     block) and the launch that follows, attn_kernel<bf16> with the flags as its work list, recomputes it with the
//     plain online-softmax loop (5 us when nothing is flagged; never more on sane inputs).
// Fill and compute overlap ACROSS items: while item i is computed, the tiles of item i + grid stream by LDS-DMA into
// the slots item i has finished with (tile t - 2 is refilled in iteration t, the last two after the item's barrier),
// so a workgroup waits for a fill only at its first item.  Without that every CU filled (65 us per layer in all) and
// computed in phase with all others.
// Loads: K/V/bias are LDS-DMA (invisible to hipcc, ordered by hand: vmcnt(0) + barrier once per item); the query rows
// are PLAIN loads -- hipcc's own wait for them can only be stronger than needed, never weaker, whatever the register
// allocator does (round 2 hid them in asm and once read a register before it had landed).
// Two hazards met on the way, both invisible to hipcc: an asm v_add directly behind the v_exp whose result it reads
// (a transcendental's result needs a wait state: wrong row sums) -- the sums are plain C++ now, one chain of adds and
// one of fmas by an opaque 1.0 so that they are not packed into v_pk_add --; and per-lane constants computed at kernel
// entry, which were spilled and reloaded inside the tile loop behind an s_waitcnt vmcnt(0) that drains the refill DMA
// -- they are derived from the lane id (asm volatile v_mbcnt) where they are used.
__device__ __forceinline__ void mfma_qk(f32x16& acc, const f32x4v& k, const f32x4v& q) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k), __builtin_bit_cast(bf16x8, q), acc, 0, 0, 0);
}
__device__ __forceinline__ void mfma_qk0(f32x16& acc, const f32x4v& k, const f32x4v& q) {   // C = 0: starts a chain
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k), __builtin_bit_cast(bf16x8, q), z, 0, 0, 0);
}
__device__ __forceinline__ void scores_ready(f32x16 (&)[2]) {}
struct AttnItem {
    int b, head, Lb, nt;
    size_t row0;
    const bf16_t* base;   // q | k | v of the head, row 0 of the sequence
};

__global__ __launch_bounds__(512, 2) void attn_pipe_kernel(AttnArgs a) {
    constexpr int LDS_V = 65536, LDS_B = 131072;   // K tiles | V tiles | key bias (8 tiles each: L <= 512)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char* sK = smem;
    const char* sV = smem + LDS_V;
    const char* sBias = smem + LDS_B;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ld = 3 * a.H;
    const int n_items = a.B * a.heads;
    const u32 lds0 = (u32)(uintptr_t)(OTTER_AS3 char*)smem;
    // Per-lane constants are derived from the lane id where they are used (v_mbcnt: two instructions), per item, not
    // once at kernel entry: kept alive across an item's prologue and store they were spilled, and the s_waitcnt vmcnt(0)
    // of a reload inside the tile loop drains the refill DMA the loop is there to hide.
    // (asm volatile: as a builtin the value is loop-invariant, gets hoisted to kernel entry and spilled all the same)
    auto lane_id = [&]() {
        int x;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
        return x;
    };

    // --- the fill: LDS-DMA pieces of a 64-key tile (16 x 1 KiB, 2 per wave), key bias, query rows ---------------
    // piece p of a tile: p < 8 -> K rows 8p .. 8p+7 (chunks swizzled by (row >> 1) & 7), else V rows (linear)
    u32 dst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = wave * 2 + i, isv = p >> 3;
        dst[i] = lds0 + (u32)(isv ? LDS_V : 0) + (u32)(p & 7) * 1024u;
    }
    auto piece_offsets = [&](u32 (&voff)[2]) {
        const int lane = lane_id();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = wave * 2 + i, isv = p >> 3, r = 8 * (p & 7) + (lane >> 3);
            const int c = isv ? (lane & 7) : ((lane & 7) ^ ((r >> 1) & 7));
            voff[i] = ((u32)r * (u32)ld + (u32)(isv ? 2 * a.H : a.H) + (u32)(c * 8)) * 2u;
        }
    };
    auto geom = [&](int it) {
        AttnItem g;
        g.b = it / a.heads;
        g.head = it - g.b * a.heads;
        g.Lb = a.seq_len ? __builtin_amdgcn_readfirstlane(a.seq_len[g.b]) : a.L;
        g.nt = a.seq_len ? g.Lb >> 6 : a.Lp >> 6;   // 1 .. 8
        g.row0 = a.seq_off ? (size_t)__builtin_amdgcn_readfirstlane(a.seq_off[g.b]) : (size_t)g.b * a.L;
        g.base = (const bf16_t*)a.QKV + g.row0 * ld + g.head * 64;
        return g;
    };
    auto issue_tile = [&](const AttnItem& g, int t, const u32 (&voff)[2]) {
        const char* tb = reinterpret_cast<const char*>(g.base) + (size_t)t * 64 * ld * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16s(tb, voff[i], dst[i] + t * 8192);
    };
    auto issue_bias = [&](const AttnItem& g) {
        if (wave < g.nt) dma4s(a.kbias + (size_t)g.b * a.Lp + wave * 64, (u32)lane_id() * 4u, lds0 + LDS_B + wave * 256);
    };
    // query rows of this wave's blocks (wave, wave + 8); rows past the sequence are clamped (computed, never stored).
    // Plain loads, which hipcc counts: its own s_waitcnt for them can only be stronger than needed (the LDS-DMA
    // pieces issued around them are invisible to it and only add to the real count), never weaker -- unlike loads
    // hidden in asm, no register is ever read before it has landed, whatever the register allocator does.
    auto issue_q = [&](const AttnItem& g, f32x4v (&q)[2][4]) {
        const int lane = lane_id(), h = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qrow = min((8 * i + wave) * 32 + l31, g.Lb - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                q[i][j] = *reinterpret_cast<const f32x4v*>(g.base + (size_t)qrow * ld + (2 * j + h) * 8);
        }
    };
    // everything this wave has issued is done, then for every wave
    auto fill_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto barrier = [&]() { asm volatile("s_barrier" ::: "memory"); };

    // per-lane LDS byte offsets inside a tile (everything else is an immediate); set per item, see lane_id()
    u32 kaddr0 = 0, vaddr = 0, baddr = 0;
    float one = 1.0f;   // made opaque per item (exp_slice)
    auto lds_offsets = [&]() {
        const int lane = lane_id(), h = lane >> 5, l31 = lane & 31, i16 = lane & 15;
        const int lswz = (l31 >> 1) & 7;
        // K fragment of k-step j: chunk (2 j + h) ^ lswz of row l31 = kaddr0 ^ (j << 5)  (2 j + h = 2 j ^ h)
        kaddr0 = (u32)(l31 * 128 + ((h ^ lswz) << 4));
        vaddr = (u32)((4 * h + (i16 >> 2)) * 128 + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2);
        baddr = (u32)(16 * h);
        one = 1.0f;
        asm volatile("" : "+v"(one));
    };

    typedef int i32x2v __attribute__((ext_vector_type(2)));
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    struct VFrag { i32x2v lo, hi; };   // keys kbase .. +3 and kbase + 8 .. +11 of 16 head dimensions, transposed

    // --- pieces of a stage -------------------------------------------------------------------------------
    auto ld_k = [&](const char* kt, int i) {   // fragment of score product i: keys 32 (i >> 2) .. +31, k-step i & 3
        return *reinterpret_cast<const f32x4v*>(kt + (i >> 2) * 4096 + (kaddr0 ^ (u32)((i & 3) << 5)));
    };
    auto ld_bias = [&](const char* bt, int sub, f32x16& acc) {   // accumulator rows = keys 32 sub + 8 g + 4 h + (0 .. 3)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4v bb = *reinterpret_cast<const f32x4v*>(bt + baddr + sub * 128 + g * 32);
            acc[4 * g + 0] = bb.x;
            acc[4 * g + 1] = bb.y;
            acc[4 * g + 2] = bb.z;
            acc[4 * g + 3] = bb.w;
        }
    };
    auto ld_v = [&](const char* vt, int i) {   // operand of context product i: sub = i >> 2, s2 = (i >> 1) & 1, d = i & 1
        const char* p0 = vt + vaddr + ((i >> 2) * 32 + ((i >> 1) & 1) * 16) * 128 + (i & 1) * 64;
        VFrag f;
        f.lo = __builtin_bit_cast(i32x2v, __builtin_amdgcn_ds_read_tr16_b64_v4i16((OTTER_AS3 s16x4*)p0));
        f.hi = __builtin_bit_cast(i32x2v, __builtin_amdgcn_ds_read_tr16_b64_v4i16((OTTER_AS3 s16x4*)(p0 + 8 * 128)));
        return f;
    };
    auto mfma_pv = [&](f32x16& acc, const VFrag& f, const bf16x8& pb) {
        const i32x4v av = {f.lo.x, f.lo.y, f.hi.x, f.hi.y};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), pb, acc, 0, 0, 0);
    };
    // slice k (0 .. 15) of the exponentials of a unit: two scores -> two probabilities, partial sums, one packed pair
    auto exp_slice = [&](int k, const f32x16 (&sc)[2], float (&ps)[2], bf16x8 (&pc)[4]) {
        const int sub = k >> 3, r0 = 2 * (k & 7);
        const float e0 = __builtin_amdgcn_exp2f(sc[sub][r0]);
        const float e1 = __builtin_amdgcn_exp2f(sc[sub][r0 + 1]);
        // Two dependent chains in plain C++, one of adds and one of fmas by an opaque 1.0: two chains of adds get
        // packed into v_pk_add_f32 (slower beside MFMAs), and an asm v_add right behind the v_exp whose result it reads
        // is a hazard hipcc cannot pad (a transcendental's result needs a wait state before the next vector
        // instruction reads it: seen as wrong row sums)
        ps[0] += e0;
        ps[1] = __builtin_fmaf(e1, one, ps[1]);
        pc[sub * 2 + (r0 >> 3)][r0 & 7] = (__bf16)e0;
        pc[sub * 2 + (r0 >> 3)][(r0 & 7) + 1] = (__bf16)e1;
    };
    // One stage.  Scores of unit u + 1 -> sn (tile t_qk, queries qn); context products of unit u - 1 (tile t_pv,
    // probabilities pprev) -> oprev; exponentials of unit u (scores sc) -> pc, ls.  MFMA slot k: even = score
    // product k / 2, odd = context product k / 2; the LDS reads of a slot are issued PF slots ahead.  `mk`: tile
    // t_qk holds a masked key -- then the score accumulators start from the key bias (read from LDS straight into
    // them), else from the MFMA's constant 0.
    // The scores read here were written by the MFMAs of the previous stage: its chain of keys 0 .. 31 ended in
    // slot 6 there, and the chain of keys 32 .. 63 (slot 14) is first read in slot 4 here -- far beyond the 12 wait
    // states an MFMA result needs before the vector ALU may read it.
    auto stage = [&](int t_qk, bool mk, const f32x4v (&qn)[4], f32x16 (&sn)[2], int t_pv, const bf16x8 (&pprev)[4],
                     f32x16 (&oprev)[2], const f32x16 (&sc)[2], float& ls, bf16x8 (&pc)[4]) {
        const char* kt = sK + t_qk * 8192;
        const char* vt = sV + t_pv * 8192;
        const char* bt = sBias + t_qk * 256;
        f32x4v kf[8];
        VFrag vf[8];
        float ps[2] = {0.0f, 0.0f};
        constexpr int PF = 2;    // LDS reads run this many slots ahead of the MFMA that consumes them
        constexpr int PRE = 2;   // exponential slices done before the first MFMA (while its operands are on their way)
        auto ld_slot = [&](int k) {
            if ((k & 1) == 0) kf[k >> 1] = ld_k(kt, k >> 1);
            else vf[k >> 1] = ld_v(vt, k >> 1);
        };
        if (mk) ld_bias(bt, 0, sn[0]);
#pragma unroll
        for (int k = 0; k < PF; ++k) ld_slot(k);
#pragma unroll
        for (int k = 0; k < PRE; ++k) exp_slice(k, sc, ps, pc);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k + PF < 16) ld_slot(k + PF);
            if (k == 8 - PF && mk) ld_bias(bt, 1, sn[1]);   // (the second chain starts in slot 8; its registers are the
                                                            // ones the first half of `sc` dies in)
            if ((k & 1) == 0) {
                const int i = k >> 1;   // score product i: chain i >> 2, k-step i & 3
                if ((i & 3) == 0 && !mk) mfma_qk0(sn[i >> 2], kf[i], qn[i & 3]);
                else mfma_qk(sn[i >> 2], kf[i], qn[i & 3]);
            } else {
                mfma_pv(oprev[(k >> 1) & 1], vf[k >> 1], pprev[k >> 2]);
            }
            if (k + PRE < 16) exp_slice(k + PRE, sc, ps, pc);
            __builtin_amdgcn_sched_barrier(0);
        }
        ls += ps[0] + ps[1];
    };
    // plain (unpipelined) pieces for the first unit of an item and the drain
    auto qk = [&](int t, const f32x4v (&q)[4], f32x16 (&s)[2]) {
        const char* kt = sK + t * 8192;
        const char* bt = sBias + t * 256;
        ld_bias(bt, 0, s[0]);
        ld_bias(bt, 1, s[1]);
#pragma unroll
        for (int i = 0; i < 8; ++i) mfma_qk(s[i >> 2], ld_k(kt, i), q[i & 3]);
        scores_ready(s);
    };
    auto pv = [&](int t, const bf16x8 (&p)[4], f32x16 (&oc)[2]) {
        const char* vt = sV + t * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) mfma_pv(oc[i & 1], ld_v(vt, i), p[i >> 1]);
    };
    // normalise and store one block: lanes l and l + 32 hold one query; after the swap a lane holds 8 consecutive
    // head dimensions (16 B): half 0 -> [8 rq, 8 rq + 8), half 1 -> the next 8.  l: the full row sum.
    auto store_block = [&](const AttnItem& g, int blk, const f32x16 (&oc)[2], float l) {
        const int lane = lane_id(), h = lane >> 5;
        const int q = blk * 32 + (lane & 31);
        const float inv = 1.0f / l;
        bf16_t* dstp = (bf16_t*)a.CTX + (g.row0 + min(q, g.Lb - 1)) * a.H + g.head * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; rq += 2) {
                bf16x8 ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x2v sw = __builtin_amdgcn_permlane32_swap(
                        __float_as_uint(oc[d][4 * rq + i] * inv), __float_as_uint(oc[d][4 * rq + 4 + i] * inv), false, false);
                    ov[i] = (__bf16)__uint_as_float(sw[0]);
                    ov[4 + i] = (__bf16)__uint_as_float(sw[1]);
                }
                if (q < g.Lb) *reinterpret_cast<float4*>(dstp + d * 32 + 8 * (rq + h)) = __builtin_bit_cast(float4, ov);
            }
    };

    // --- the items of this workgroup ------------------------------------------------------------------------
    int it = blockIdx.x;
    if (it >= n_items) return;
    AttnItem cur = geom(it);
    f32x4v qf[2][4];
    issue_bias(cur);
    issue_q(cur, qf);
    {
        u32 voff[2];
        piece_offsets(voff);
        for (int t = 0; t < cur.nt; ++t) issue_tile(cur, t, voff);
    }
#ifdef ATTN_STAMP
    unsigned long long st_[8][6];
    int sti_ = 0;
#define STAMP(k) do { if (sti_ < 8) st_[sti_][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
    for (;;) {
        // the last tiles, the bias and the query rows of `cur` were issued a moment ago (its first tiles long before)
        STAMP(0);
        fill_landed();
        STAMP(1);
        const int nxt = it + (int)gridDim.x;
        const bool has_next = nxt < n_items;
        const AttnItem nx = geom(has_next ? nxt : it);
        // which tiles hold a masked key (bias values are 0 or -1e30)
        unsigned maskbits = 0;
        for (int t = 0; t < cur.nt; ++t)
            if (__ballot(reinterpret_cast<const float*>(sBias)[t * 64 + lane_id()] != 0.0f) != 0ull) maskbits |= 1u << t;
        u32 voff[2];
        piece_offsets(voff);
        lds_offsets();

        f32x16 o[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][d][r] = 0.0f;
        float lsum[2] = {0.0f, 0.0f};   // this lane's half of a row sum
        f32x16 s[2][2];                 // scores of this wave's block 0 / block 1
        bf16x8 p[2][4];                 // P operand of block 0 / block 1
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) p[1][i][j] = (__bf16)0.0f;   // "the unit before the first": adds nothing

        qk(0, qf[0], s[0]);
        STAMP(2);
        for (int t = 0; t < cur.nt; ++t) {
            if (t > 0) barrier();   // every wave is through iteration t - 1 ...
            // ... whose first stage was the last reader of tile t - 2 (its V; its K was last read in iteration t - 2):
            // the next item's tile t - 2 streams into that slot while this item is being computed
            if (has_next && t >= 2 && t - 2 < nx.nt) issue_tile(nx, t - 2, voff);
            const int tn = t + 1 < cur.nt ? t + 1 : t;   // (no next tile: the last one again, result unused)
            const int tp = t > 0 ? t - 1 : 0;            // (no previous tile: P is zero)
            // unit (t, 0): beside the scores of (t, 1) and the context products of (t - 1, 1)
            stage(t, (maskbits >> t) & 1u, qf[1], s[1], tp, p[1], o[1], s[0], lsum[0], p[0]);
            // unit (t, 1): beside the scores of (t + 1, 0) and the context products of (t, 0)
            stage(tn, (maskbits >> tn) & 1u, qf[0], s[0], t, p[0], o[0], s[1], lsum[1], p[1]);
        }
        pv(cur.nt - 1, p[1], o[1]);
        STAMP(3);

        // accept a block iff all its row sums are of sane magnitude; the others are flagged for attn_kernel
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float l = lsum[i];
            const u32x2v sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
            l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            const bool good = __ballot(!(l > 1e-30f && l < 1e30f)) == 0ull;
            if (lane_id() == 0) a.flags[(size_t)it * 16 + 8 * i + wave] = good ? 0 : 1;
            if (good) store_block(cur, 8 * i + wave, o[i], l);
        }
        STAMP(4);
#ifdef ATTN_STAMP
        if (!has_next) { ++sti_; break; }
#else
        if (!has_next) break;
#endif
        barrier();   // every wave is done with the LDS image of `cur`
        piece_offsets(voff);
        for (int t = cur.nt >= 2 ? cur.nt - 2 : 0; t < nx.nt; ++t) issue_tile(nx, t, voff);
        issue_bias(nx);
        issue_q(nx, qf);
        STAMP(5);
#ifdef ATTN_STAMP
        ++sti_;
#endif
        cur = nx;
        it = nxt;
    }
#ifdef ATTN_STAMP
    if ((blockIdx.x == 0 || blockIdx.x == 100) && threadIdx.x == 0 && a.B >= 64) {
        static __device__ int once_[2];
        if (atomicAdd(&once_[blockIdx.x == 0 ? 0 : 1], 1) == 20)
            for (int i = 0; i < sti_ && i < 8; ++i)
                printf("wg %d item %d: wait %llu qk0 %llu loop %llu store %llu issue %llu (cycles)\n", (int)blockIdx.x, i,
                       st_[i][1] - st_[i][0], st_[i][2] - st_[i][1], st_[i][3] - st_[i][2], st_[i][4] - st_[i][3],
                       i + 1 < sti_ ? st_[i][5] - st_[i][4] : 0ull);
    }
#endif
}

int launch_attn_bf16(const AttnArgs& a, hipStream_t st) {
    // switches for A/B runs and tests, read per call: OTTER_ATTN_OLD=1 the round-2 kernel, OTTER_ATTN_SAFE=1 the plain
    // online-softmax kernel on every block (what the repair pass runs on the blocks attn_pipe_kernel rejects)
    const char* ev = getenv("OTTER_ATTN_OLD");
    const bool old_kernel = ev && ev[0] == '1';
    ev = getenv("OTTER_ATTN_SAFE");
    if (ev && ev[0] == '1') {
        AttnArgs b = a;
        b.flags = nullptr;
        hipLaunchKernelGGL(attn_kernel<bf16_t>, dim3((a.L + 127) / 128, a.heads, a.B), dim3(256), 0, st, b);
        OTTER_HIP_TRY(hipGetLastError());
        return OTTER_OK;
    }
    if (!old_kernel && a.flags) {
        // persistent: one workgroup per CU walks (batch, head) items; the blocks it rejects (see the kernel's header)
        // are recomputed by the plain online-softmax kernel, whose workgroups exit at once when nothing is flagged
        const int lds = 131072 + 8 * 256;
        const int items = a.B * a.heads;
        OTTER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_pipe_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(attn_pipe_kernel, dim3(items < a.ncu ? items : a.ncu), dim3(512), lds, st, a);
        hipLaunchKernelGGL(attn_kernel<bf16_t>, dim3((a.L + 127) / 128, a.heads, a.B), dim3(256), 0, st, a);
        OTTER_HIP_TRY(hipGetLastError());
        return OTTER_OK;
    }
    const int nt = a.Lp >> 6;
    const int lds = 2 * nt * 8192 + nt * 256;
    // 8 waves x 64 queries.  (A 16-wave x 32-query form exists as attn_full_kernel<1>; it needs 128 registers per
    // lane, spills, and is not launched: a kernel whose hidden asynchronous loads land in registers must not spill.)
    AttnArgs b = a;
    b.flags = nullptr;
    OTTER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_full_kernel<2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(attn_full_kernel<2>, dim3(a.heads, a.B), dim3(512), lds, st, b);
    OTTER_HIP_TRY(hipGetLastError());
    return OTTER_OK;
}

// ------------------------------------------------------------------ row kernels
// RoBERTa position ids: cumsum(ids != pad) * (ids != pad) + pad  (modeling_roberta.py:142-155);
// BERT: arange(L).  Also writes the additive key-padding bias from input_mask.
// kbias is [B][Lp], Lp = L rounded up to 64: keys past L are masked like padding tokens.
// seq_len (optional): rows of sequence b that anything reads = 1 + the last position set in
// input_mask, part2_mask or part3_mask, rounded up to 64 (the packed path computes only those).
__global__ void prep_kernel(const long long* ids, const long long* mask, int* pos, float* kbias, int L, int Lp, int pad,
                            int roberta, const long long* p2, const long long* p3, int* seq_len) {
    const int b = blockIdx.x, lane = threadIdx.x;  // one wave per sequence
    int run = 0, last = 0;
    for (int c0 = 0; c0 < Lp; c0 += 64) {
        const int l = c0 + lane;
        const bool in = l < L;
        const bool tok = in && ids[(size_t)b * L + l] != pad;
        const unsigned long long bal = __ballot(tok);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (in) pos[(size_t)b * L + l] = roberta ? (tok ? run + before + 1 + pad : pad) : l;
        const bool key = in && mask[(size_t)b * L + l] != 0;
        kbias[(size_t)b * Lp + l] = key ? 0.0f : -1e30f;
        run += __popcll(bal);
        if (seq_len) {
            const bool used = key || (in && p2 && p2[(size_t)b * L + l] != 0) || (in && p3 && p3[(size_t)b * L + l] != 0);
            const unsigned long long ub = __ballot(used);
            if (ub) last = c0 + 63 - __builtin_clzll(ub);
        }
    }
    if (seq_len && lane == 0) seq_len[b] = min(Lp, ((last + 1 + 63) >> 6) << 6);
}

// exclusive prefix sum of the sequence lengths; info[0] = 256-row tiles in use, info[1] = rows in use
__global__ void seq_offsets_kernel(const int* seq_len, int B, int* seq_off, int* info) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int acc = 0;
    for (int b = 0; b < B; ++b) {
        seq_off[b] = acc;
        acc += seq_len[b];
    }
    info[0] = (acc + 255) >> 8;
    info[1] = acc;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// X[m] = LN(word[ids] + pos[pos_ids] + type[type_ids]); one wave per token, H = 256 * NV
template <typename T, int NV>
__global__ __launch_bounds__(256) void embed_ln_kernel(const long long* ids, const int* pos, const long long* types,
                                                       const float* word, const float* pemb, const float* temb,
                                                       const float* gamma, const float* beta, T* X, int M, int H,
                                                       int vocab, int npos, int ntype, float eps,
                                                       const int* seq_off = nullptr, const int* seq_len = nullptr,
                                                       int L = 1) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    size_t mo = m;                  // output row: dense, or packed (token l of sequence b -> seq_off[b] + l)
    if (seq_off) {
        const int b = m / L, l = m - b * L;
        if (l >= seq_len[b]) return;
        mo = (size_t)seq_off[b] + l;
    }
    long long id = ids[m];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    int p = pos[m];
    p = p < 0 ? 0 : (p >= npos ? npos - 1 : p);
    long long ty = types ? types[m] : 0;
    ty = ty < 0 ? 0 : (ty >= ntype ? ntype - 1 : ty);
    float v[NV * 4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        const float4 w = *reinterpret_cast<const float4*>(word + (size_t)id * H + c);
        const float4 q = *reinterpret_cast<const float4*>(pemb + (size_t)p * H + c);
        const float4 t = *reinterpret_cast<const float4*>(temb + (size_t)ty * H + c);
        v[4 * i + 0] = w.x + q.x + t.x;
        v[4 * i + 1] = w.y + q.y + t.y;
        v[4 * i + 2] = w.z + q.z + t.z;
        v[4 * i + 3] = w.w + q.w + t.w;
        sum += v[4 * i] + v[4 * i + 1] + v[4 * i + 2] + v[4 * i + 3];
    }
    const float mean = wave_sum(sum) / H;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < NV * 4; ++i) var += (v[i] - mean) * (v[i] - mean);
    const float rstd = rsqrtf(wave_sum(var) / H + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            X[mo * H + c + e] = from_f<T>((v[4 * i + e] - mean) * rstd * gamma[c + e] + beta[c + e]);
    }
}

// Y[m] = LN(X[m]); one wave per row.  With seq_off: X is packed, Y dense (rows nothing reads are skipped)
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_kernel(const T* X, const float* gamma, const float* beta, T* Y, int M, int H,
                                                 float eps, const int* seq_off = nullptr, const int* seq_len = nullptr,
                                                 int L = 1) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    size_t mi = m;
    if (seq_off) {
        const int b = m / L, l = m - b * L;
        if (l >= seq_len[b]) return;
        mi = (size_t)seq_off[b] + l;
    }
    float v[NV * 4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[4 * i + e] = to_f<T>(X[mi * H + c + e]);
            sum += v[4 * i + e];
        }
    }
    const float mean = wave_sum(sum) / H;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < NV * 4; ++i) var += (v[i] - mean) * (v[i] - mean);
    const float rstd = rsqrtf(wave_sum(var) / H + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            Y[(size_t)m * H + c + e] = from_f<T>((v[4 * i + e] - mean) * rstd * gamma[c + e] + beta[c + e]);
    }
}

// pooling_masked_part (tb_retriever.py:42-58) + concat: grid (B, parts); out fp32 [B, D]
// part 0 = h[:,0] (skipped when the projection kernel writes it), parts 1,2 = pooled by mask.
template <typename T>
__global__ __launch_bounds__(256) void pool_kernel(const T* X, const long long* mask2, const long long* mask3,
                                                   float* out, int L, int H, int D, int method, int part_lo) {
    const int b = blockIdx.x, part = blockIdx.y + part_lo, tid = threadIdx.x;
    const T* hb = X + (size_t)b * L * H;
    float* ob = out + (size_t)b * D + (size_t)part * H;
    __shared__ int sIdx;
    __shared__ float sCnt[4];
    if (part == 0 || method == OTTER_POOL_CLS || method < 0 || method > OTTER_POOL_FIRST) {
        for (int c = tid; c < H; c += 256) ob[c] = to_f<T>(hb[c]);
        return;
    }
    const long long* mk = (part == 1 ? mask2 : mask3) + (size_t)b * L;
    if (method == OTTER_POOL_FIRST) {
        // argmax(mask * arange(L, 0, -1)): the first position holding the largest mask value;
        // for a 0/1 mask the first 1, and row 0 when the mask is empty
        // (256 lanes, not one walking the row: 57 -> 6 us per 128 x 512 batch.  Largest value first, lowest position
        // among equals: composite (value, L - l) maximised; all-zero or negative rows keep position 0, as the serial
        // rule `v > best` from best = 0 did)
        __shared__ long long sBest[4];
        __shared__ int sArg[4];
        long long best = 0;
        int idx = 0;
        for (int l = tid; l < L; l += 256) {
            const long long v = mk[l] * (long long)(L - l);
            if (v > best) { best = v; idx = l; }      // (ascending l per lane: the first of equals stays)
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const long long ob2 = __shfl_xor(best, o);
            const int oi = __shfl_xor(idx, o);
            if (ob2 > best || (ob2 == best && ob2 > 0 && oi < idx)) { best = ob2; idx = oi; }
        }
        if ((tid & 63) == 0) { sBest[tid >> 6] = best; sArg[tid >> 6] = idx; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (sBest[w] > best || (sBest[w] == best && best > 0 && sArg[w] < idx)) { best = sBest[w]; idx = sArg[w]; }
            sIdx = idx;
        }
        __syncthreads();
        const T* row = hb + (size_t)sIdx * H;
        for (int c = tid; c < H; c += 256) ob[c] = to_f<T>(row[c]);
        return;
    }
    for (int c = tid; c < H; c += 256) {
        float acc = method == OTTER_POOL_MAX ? -INFINITY : 0.f;
        float cnt = 0.f;
        for (int l = 0; l < L; ++l) {
            const float mv = (float)mk[l];
            const float hv = to_f<T>(hb[(size_t)l * H + c]);
            if (method == OTTER_POOL_MAX) acc = fmaxf(acc, hv + (1.0f - mv) * -1e5f);
            else acc += hv * mv;
            cnt += mv;
        }
        if (method == OTTER_POOL_MEAN) acc = acc / cnt;   // empty mask -> NaN, as the reference
        ob[c] = acc;
    }
    (void)sCnt;
}

// att_self_pool (tb_retriever.py:510-523): out = sum_l softmax_l(w.h[l] + b + (1 - m[l]) * -1e6) * h[l]
// grid (B, 2): part 1 = table side (self_proj_tab, part2_mask), part 2 = passage side (self_proj_psg, part3_mask)
template <typename T>
__global__ __launch_bounds__(256) void att_pool_kernel(const T* X, const long long* mask2, const long long* mask3,
                                                       const float* w_tab, const float* b_tab, const float* w_psg,
                                                       const float* b_psg, float* out, int L, int H, int D) {
    extern __shared__ float sl[];   // [L] logits -> weights, [8] reduce
    float* red = sl + L;
    const int b = blockIdx.x, part = blockIdx.y + 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* hb = X + (size_t)b * L * H;
    const long long* mk = (part == 1 ? mask2 : mask3) + (size_t)b * L;
    const float* w = part == 1 ? w_tab : w_psg;
    const float bias = part == 1 ? b_tab[0] : b_psg[0];
    for (int l = wave; l < L; l += 4) {
        float acc = 0.f;
        for (int c = lane; c < H; c += 64) acc = fmaf(to_f<T>(hb[(size_t)l * H + c]), w[c], acc);
        acc = wave_sum(acc);
        if (lane == 0) sl[l] = acc + bias + (1.0f - (float)mk[l]) * -1e6f;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int l = tid; l < L; l += 256) mx = fmaxf(mx, sl[l]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int l = tid; l < L; l += 256) {
        const float e = expf(sl[l] - mx);
        sl[l] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    float* ob = out + (size_t)b * D + (size_t)part * H;
    for (int c = tid; c < H; c += 256) {
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(sl[l] * inv, to_f<T>(hb[(size_t)l * H + c]), acc);
        ob[c] = acc;
    }
}

// part1 = LN(Linear(h[:,0]))  (project = Sequential(Linear, LayerNorm), tb_retriever.py:162-163)
template <typename T>
__global__ __launch_bounds__(256) void proj_ln_kernel(const T* X, const float* W, const float* bias,
                                                      const float* gamma, const float* beta, float* out, int L, int H,
                                                      int D, float eps) {
    extern __shared__ float sh[];   // [H] input, [H] output, [8] reduce
    float* xin = sh;
    float* y = sh + H;
    float* red = sh + 2 * H;
    const int b = blockIdx.x, tid = threadIdx.x;
    const T* row = X + (size_t)b * L * H;
    for (int c = tid; c < H; c += 256) xin[c] = to_f<T>(row[c]);
    __syncthreads();
    float lsum = 0.f;
    for (int n = tid; n < H; n += 256) {
        const float* w = W + (size_t)n * H;
        float acc = 0.f;
        for (int k = 0; k < H; ++k) acc = fmaf(w[k], xin[k], acc);
        acc += bias[n];
        y[n] = acc;
        lsum += acc;
    }
    lsum = wave_sum(lsum);
    if ((tid & 63) == 0) red[tid >> 6] = lsum;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / H;
    float lvar = 0.f;
    for (int n = tid; n < H; n += 256) lvar += (y[n] - mean) * (y[n] - mean);
    lvar = wave_sum(lvar);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = lvar;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / H + eps);
    for (int n = tid; n < H; n += 256) out[(size_t)b * D + n] = (y[n] - mean) * rstd * gamma[n] + beta[n];
}

template <typename T>
__global__ void to_f32_kernel(const T* src, float* dst, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = to_f<T>(src[i]);
}

__global__ void scale_copy_kernel(const float* src, float* dst, long long n, float f) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = src[i] * f;
}

template <typename T>
__global__ void from_f32_kernel(const float* src, T* dst, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = from_f<T>(src[i]);
}

// ------------------------------------------------------------------ host side

struct Tensor {
    float* d = nullptr;
    std::vector<int64_t> shape;
    size_t numel() const {
        size_t n = 1;
        for (auto s : shape) n *= (size_t)s;
        return n;
    }
};

struct LayerW {
    void* wqkv = nullptr;   // [3H, H] in T
    float* bqkv = nullptr;  // [3H]
    void *wo = nullptr, *w1 = nullptr, *w2 = nullptr;
    const float *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
    const float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
    // bf16 path, LayerNorm folded into the consumers (gemm_bf16.hip): gamma-scaled matrices, s_n, c_n
    void *wqkv_f = nullptr, *w1_f = nullptr;   // wqkv_f: folded with the PREVIOUS layer's output LayerNorm (layers >= 1)
    float *sqkv = nullptr, *cqkv = nullptr, *s1 = nullptr, *c1 = nullptr;
};

}  // namespace

struct otter_encoder {
    otter_encoder_cfg cfg;
    int device = 0;
    int ncu = 256;
    int dtype = -1;   // compute type after finalize
    std::map<std::string, Tensor> w;
    std::vector<LayerW> layers;
    std::vector<void*> owned;   // packed buffers
    // workspace (grow-only)
    otter::DevBuf X, X1, QKV, CTX, TMP, MID, pos, kbias;
    otter::DevBuf part, stat1, stat2;   // LayerNorm statistics of the fused bf16 path
    otter::DevBuf seq;                  // packed batches: seq_len[B], seq_off[B], info[2]
    otter::DevBuf aflags;               // attention: blocks attn_pipe_kernel rejected, [B][heads][16] bytes
    bool ln_fused = false;
    // attn_kernel's score scale.  fp32: 1/sqrt(64).  bf16: the query projection is stored pre-multiplied by
    // (1/sqrt(64)) log2(e) (attn_pipe_kernel exponentiates its scores in base 2 as they come out of the MFMA chain),
    // so the natural-exponent fallback kernel multiplies by ln 2.
    float attn_scale = 0.125f;
    long long ws_rows = 0;
};

namespace {

size_t esize(int dtype) { return dtype == OTTER_BF16 ? 2 : 4; }

int expected_shape(const otter_encoder_cfg& c, const std::string& name, std::vector<int64_t>* shp, bool* ignored) {
    const int64_t H = c.hidden_size, I = c.intermediate_size;
    *ignored = false;
    auto ends = [&](const char* s) {
        const size_t n = strlen(s);
        return name.size() >= n && name.compare(name.size() - n, n, s) == 0;
    };
    if (ends("position_ids") || ends("token_type_ids")) { *ignored = true; return 0; }   // HF buffers
    if (name.rfind("pooler.", 0) == 0) { *ignored = true; return 0; }                    // unused by the path ([0] only)
    if (name == "embeddings.word_embeddings.weight") { *shp = {c.vocab_size, H}; return 0; }
    if (name == "embeddings.position_embeddings.weight") { *shp = {c.max_position_embeddings, H}; return 0; }
    if (name == "embeddings.token_type_embeddings.weight") { *shp = {c.type_vocab_size, H}; return 0; }
    if (name == "embeddings.LayerNorm.weight" || name == "embeddings.LayerNorm.bias") { *shp = {H}; return 0; }
    if (name == "project.0.weight") { *shp = {H, H}; return 0; }
    if (name == "self_proj_tab.weight" || name == "self_proj_psg.weight") { *shp = {1, H}; return 0; }
    if (name == "self_proj_tab.bias" || name == "self_proj_psg.bias") { *shp = {1}; return 0; }
    if (name == "project.0.bias" || name == "project.1.weight" || name == "project.1.bias") { *shp = {H}; return 0; }
    if (name.rfind("encoder.layer.", 0) == 0) {
        const char* p = name.c_str() + 14;
        char* end = nullptr;
        const long li = strtol(p, &end, 10);
        if (end == p || *end != '.' || li < 0 || li >= c.num_hidden_layers) return -1;
        const std::string rest(end + 1);
        if (rest == "attention.self.query.weight" || rest == "attention.self.key.weight" ||
            rest == "attention.self.value.weight" || rest == "attention.output.dense.weight") { *shp = {H, H}; return 0; }
        if (rest == "intermediate.dense.weight") { *shp = {I, H}; return 0; }
        if (rest == "output.dense.weight") { *shp = {H, I}; return 0; }
        if (rest == "intermediate.dense.bias") { *shp = {I}; return 0; }
        if (rest == "attention.self.query.bias" || rest == "attention.self.key.bias" ||
            rest == "attention.self.value.bias" || rest == "attention.output.dense.bias" ||
            rest == "output.dense.bias" || rest == "attention.output.LayerNorm.weight" ||
            rest == "attention.output.LayerNorm.bias" || rest == "output.LayerNorm.weight" ||
            rest == "output.LayerNorm.bias") { *shp = {H}; return 0; }
    }
    return -1;
}

template <typename T>
int convert_to(const float* src, void* dst, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(from_f32_kernel<T>, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, st, src,
                       (T*)dst, (long long)n);
    OTTER_HIP_TRY(hipGetLastError());
    return OTTER_OK;
}

int dev_alloc(otter_encoder* e, void** p, size_t bytes) {
    OTTER_HIP_TRY(hipMalloc(p, bytes));
    e->owned.push_back(*p);
    return OTTER_OK;
}

template <typename T>
int finalize_t(otter_encoder* e) {
    const auto& c = e->cfg;
    const size_t H = c.hidden_size, I = c.intermediate_size;
    hipStream_t st = nullptr;
    e->layers.resize(c.num_hidden_layers);
    // bf16: W_q and b_q times (1/sqrt(d_head)) log2(e), in fp32, before anything is derived from them
    const float dh_scale = 1.0f / sqrtf((float)(H / c.num_attention_heads));
    float *wq_s = nullptr, *bq_s = nullptr;
    e->attn_scale = dh_scale;
    if (sizeof(T) == 2) {
        OTTER_HIP_TRY(hipMalloc((void**)&wq_s, H * H * sizeof(float)));
        OTTER_HIP_TRY(hipMalloc((void**)&bq_s, H * sizeof(float)));
        e->attn_scale = 0.69314718055994531f;
    }
    struct TmpFree {
        float *a, *b;
        ~TmpFree() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); }
    } tmp_free{wq_s, bq_s};
    for (int li = 0; li < c.num_hidden_layers; ++li) {
        const std::string p = "encoder.layer." + std::to_string(li) + ".";
        LayerW& L = e->layers[li];
        int rc;
        const float* wsrc[3] = {e->w[p + "attention.self.query.weight"].d, e->w[p + "attention.self.key.weight"].d,
                                e->w[p + "attention.self.value.weight"].d};
        const float* bsrc[3] = {e->w[p + "attention.self.query.bias"].d, e->w[p + "attention.self.key.bias"].d,
                                e->w[p + "attention.self.value.bias"].d};
        if (sizeof(T) == 2) {
            const float f = dh_scale * 1.4426950408889634f;
            hipLaunchKernelGGL(scale_copy_kernel, dim3(1024), dim3(256), 0, st, wsrc[0], wq_s, (long long)(H * H), f);
            hipLaunchKernelGGL(scale_copy_kernel, dim3(4), dim3(256), 0, st, bsrc[0], bq_s, (long long)H, f);
            OTTER_HIP_TRY(hipGetLastError());
            wsrc[0] = wq_s;
            bsrc[0] = bq_s;
        }
        if ((rc = dev_alloc(e, &L.wqkv, 3 * H * H * sizeof(T)))) return rc;
        if ((rc = dev_alloc(e, (void**)&L.bqkv, 3 * H * sizeof(float)))) return rc;
        for (int t = 0; t < 3; ++t) {
            if ((rc = convert_to<T>(wsrc[t], (char*)L.wqkv + t * H * H * sizeof(T), H * H, st))) return rc;
            OTTER_HIP_TRY(hipMemcpyAsync(L.bqkv + t * H, bsrc[t], H * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        struct { const char* n; void** dst; size_t cnt; } mats[3] = {
            {"attention.output.dense.weight", &L.wo, H * H},
            {"intermediate.dense.weight", &L.w1, I * H},
            {"output.dense.weight", &L.w2, H * I}};
        for (auto& m : mats) {
            if (sizeof(T) == 4) {
                *m.dst = e->w[p + m.n].d;   // fp32: use the uploaded tensor in place
            } else {
                if ((rc = dev_alloc(e, m.dst, m.cnt * sizeof(T)))) return rc;
                if ((rc = convert_to<T>(e->w[p + m.n].d, *m.dst, m.cnt, st))) return rc;
            }
        }
        L.bo = e->w[p + "attention.output.dense.bias"].d;
        L.b1 = e->w[p + "intermediate.dense.bias"].d;
        L.b2 = e->w[p + "output.dense.bias"].d;
        L.ln1g = e->w[p + "attention.output.LayerNorm.weight"].d;
        L.ln1b = e->w[p + "attention.output.LayerNorm.bias"].d;
        L.ln2g = e->w[p + "output.LayerNorm.weight"].d;
        L.ln2b = e->w[p + "output.LayerNorm.bias"].d;
        if (sizeof(T) == 2 && e->ln_fused) {
            // FFN up reads LN1's rows: fold (gamma1, beta1) into W1
            if ((rc = dev_alloc(e, &L.w1_f, I * H * sizeof(T)))) return rc;
            if ((rc = dev_alloc(e, (void**)&L.s1, I * sizeof(float)))) return rc;
            if ((rc = dev_alloc(e, (void**)&L.c1, I * sizeof(float)))) return rc;
            if ((rc = launch_fold_ln(e->w[p + "intermediate.dense.weight"].d, L.ln1g, L.ln1b, L.b1, L.w1_f, L.s1, L.c1,
                                     (int)I, (int)H, st)))
                return rc;
            if (li > 0) {
                // QKV reads the previous layer's LN2 rows
                const LayerW& P = e->layers[li - 1];
                if ((rc = dev_alloc(e, &L.wqkv_f, 3 * H * H * sizeof(T)))) return rc;
                if ((rc = dev_alloc(e, (void**)&L.sqkv, 3 * H * sizeof(float)))) return rc;
                if ((rc = dev_alloc(e, (void**)&L.cqkv, 3 * H * sizeof(float)))) return rc;
                for (int t = 0; t < 3; ++t)
                    if ((rc = launch_fold_ln(wsrc[t], P.ln2g, P.ln2b, L.bqkv + t * H,
                                             (char*)L.wqkv_f + t * H * H * sizeof(T), L.sqkv + t * H, L.cqkv + t * H,
                                             (int)H, (int)H, st)))
                        return rc;
            }
        }
    }
    OTTER_HIP_TRY(hipStreamSynchronize(st));
    return OTTER_OK;
}

int ensure_ws(otter_encoder* e, int B, int L) {
    const long long M = (long long)B * L;
    const long long rows = round_up64(M, 256) + 64;
    int rc0 = e->kbias.reserve((size_t)B * round_up(L, 64) * sizeof(float));   // [B][Lp], fully written by prep_kernel
    if (rc0) return rc0;
    if ((rc0 = e->aflags.reserve((size_t)B * e->cfg.num_attention_heads * 16))) return rc0;   // fully written per launch
    if (rows <= e->ws_rows) return OTTER_OK;
    const size_t H = e->cfg.hidden_size, I = e->cfg.intermediate_size, es = esize(e->dtype);
    int rc;
    struct { otter::DevBuf* b; size_t bytes; } bufs[] = {
        {&e->X, (size_t)rows * H * es},   {&e->X1, (size_t)rows * H * es},  {&e->QKV, (size_t)rows * 3 * H * es},
        {&e->CTX, (size_t)rows * H * es}, {&e->TMP, (size_t)rows * H * es}, {&e->MID, (size_t)rows * I * es},
        {&e->pos, (size_t)rows * 4}};
    for (auto& b : bufs) {
        if ((rc = b.b->reserve(b.bytes))) return rc;
        // padding rows must stay finite (they are read as GEMM rows and as masked attention keys)
        OTTER_HIP_TRY(hipMemset(b.b->p, 0, b.b->bytes));
    }
    if (e->ln_fused) {
        const size_t parts = (size_t)4 * (H / 256);
        if ((rc = e->part.reserve(parts * rows * 8))) return rc;
        if ((rc = e->stat1.reserve((size_t)rows * 8))) return rc;
        if ((rc = e->stat2.reserve((size_t)rows * 8))) return rc;
    }
    e->ws_rows = rows;
    return OTTER_OK;
}

template <typename T>
int launch_gemm(int epi, const void* A, const void* W, const float* bias, const void* R, void* C, long long M, int N,
                int K, int ncu, hipStream_t st) {
    if (sizeof(T) == 2 && K % 64 == 0)   // bf16 throughput path: persistent ping-pong kernel (gemm_bf16.hip)
        return launch_gemm_bf16(epi, A, W, bias, R, C, round_up64(M, 256), N, K, ncu, st);
    GemmArgs g;
    g.A = A; g.W = W; g.bias = bias; g.R = R; g.C = C;
    g.tiles_m = (int)(round_up64(M, 256) / 256);
    g.tiles_n = N / 256;
    g.K = K; g.lda = K; g.ldw = K; g.ldc = N;
    const size_t lds = 2 * 512 * 128;
    void (*kern)(GemmArgs) = epi == EPI_BIAS ? gemm_kernel<T, EPI_BIAS>
                             : epi == EPI_BIAS_GELU ? gemm_kernel<T, EPI_BIAS_GELU>
                                                    : gemm_kernel<T, EPI_BIAS_RESID>;
    OTTER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(512), lds, st, g);
    OTTER_HIP_TRY(hipGetLastError());
    return OTTER_OK;
}

template <typename T, int NV>
int run_layers_nv(otter_encoder* e, const int64_t* ids, const int64_t* mask, const int64_t* types, int B, int L,
                  int n_layers, hipStream_t st) {
    const auto& c = e->cfg;
    const long long M = (long long)B * L;
    const int H = c.hidden_size, I = c.intermediate_size;
    const int Lp = round_up(L, 64);
    int rc;
    hipLaunchKernelGGL(prep_kernel, dim3(B), dim3(64), 0, st, (const long long*)ids, (const long long*)mask,
                       (int*)e->pos.p, (float*)e->kbias.p, L, Lp, c.pad_token_id, c.position_mode == OTTER_POS_ROBERTA,
                       (const long long*)nullptr, (const long long*)nullptr, (int*)nullptr);
    const unsigned rowgrid = (unsigned)((M + 3) / 4);
    hipLaunchKernelGGL((embed_ln_kernel<T, NV>), dim3(rowgrid), dim3(256), 0, st, (const long long*)ids,
                       (const int*)e->pos.p, (const long long*)types, e->w["embeddings.word_embeddings.weight"].d,
                       e->w["embeddings.position_embeddings.weight"].d, e->w["embeddings.token_type_embeddings.weight"].d,
                       e->w["embeddings.LayerNorm.weight"].d, e->w["embeddings.LayerNorm.bias"].d, (T*)e->X.p, (int)M, H,
                       c.vocab_size, c.max_position_embeddings, c.type_vocab_size, c.layer_norm_eps,
                       (const int*)nullptr, (const int*)nullptr, 1);
    OTTER_HIP_TRY(hipGetLastError());
    for (int li = 0; li < n_layers; ++li) {
        const LayerW& W = e->layers[li];
        if ((rc = launch_gemm<T>(EPI_BIAS, e->X.p, W.wqkv, W.bqkv, nullptr, e->QKV.p, M, 3 * H, H, e->ncu, st))) return rc;
        AttnArgs a;
        a.QKV = e->QKV.p; a.CTX = e->CTX.p; a.kbias = (const float*)e->kbias.p;
        a.B = B; a.L = L; a.Lp = Lp; a.H = H; a.heads = c.num_attention_heads;
        a.scale = e->attn_scale;
        a.flags = (unsigned char*)e->aflags.p; a.ncu = e->ncu;
        if (sizeof(T) == 2 && L <= 512) {
            if ((rc = launch_attn_bf16(a, st))) return rc;
        } else {
            a.flags = nullptr;   // every block
            hipLaunchKernelGGL(attn_kernel<T>, dim3((L + 127) / 128, c.num_attention_heads, B), dim3(256), 0, st, a);
            OTTER_HIP_TRY(hipGetLastError());
        }
        if ((rc = launch_gemm<T>(EPI_BIAS_RESID, e->CTX.p, W.wo, W.bo, e->X.p, e->TMP.p, M, H, H, e->ncu, st))) return rc;
        hipLaunchKernelGGL((ln_kernel<T, NV>), dim3(rowgrid), dim3(256), 0, st, (const T*)e->TMP.p, W.ln1g, W.ln1b,
                           (T*)e->X1.p, (int)M, H, c.layer_norm_eps, (const int*)nullptr, (const int*)nullptr, 1);
        if ((rc = launch_gemm<T>(EPI_BIAS_GELU, e->X1.p, W.w1, W.b1, nullptr, e->MID.p, M, I, H, e->ncu, st))) return rc;
        if ((rc = launch_gemm<T>(EPI_BIAS_RESID, e->MID.p, W.w2, W.b2, e->X1.p, e->TMP.p, M, H, I, e->ncu, st))) return rc;
        hipLaunchKernelGGL((ln_kernel<T, NV>), dim3(rowgrid), dim3(256), 0, st, (const T*)e->TMP.p, W.ln2g, W.ln2b,
                           (T*)e->X.p, (int)M, H, c.layer_norm_eps, (const int*)nullptr, (const int*)nullptr, 1);
        OTTER_HIP_TRY(hipGetLastError());
    }
    return OTTER_OK;
}

// bf16 path with both LayerNorms of a layer folded into the GEMMs around them (gemm_bf16.hip):
// the attention-output and FFN-down GEMMs store the un-normalised sums T1 (X1) and T2 (TMP) plus row
// statistics; QKV and FFN-up read them through gamma-scaled weights; residuals are normalised on the
// fly.  Only the last layer's output is materialised (pooling / hidden-state readers need it).
//
// Packed batches (pk != null): the rows that nothing reads -- padding past the last position set in
// input_mask / part2_mask / part3_mask -- are not computed at all.  Sequence b occupies rows
// [seq_off[b], seq_off[b] + seq_len[b]) of every activation, seq_len a multiple of 64 (attention tiles);
// the lengths live on the device only (no host synchronisation): the GEMMs read their tile count from
// seq_info[0], attention its offset and length per sequence, and the final LayerNorm writes the dense
// [B, L, H] layout the pooling kernels read.  The reference computes every padded position
// (scripts/encode_corpus.py:139-144 feeds max_length-padded batches); the embeddings are the same.
struct PackHint {
    const int64_t* p2;
    const int64_t* p3;
};

template <int NV>
int run_layers_fused(otter_encoder* e, const int64_t* ids, const int64_t* mask, const int64_t* types, int B, int L,
                     int n_layers, hipStream_t st, const PackHint* pk) {
    typedef bf16_t T;
    const auto& c = e->cfg;
    const long long M = (long long)B * L, M_pad = round_up64(M, 256);
    const int H = c.hidden_size, I = c.intermediate_size;
    const int Lp = round_up(L, 64), P = 4 * (H / 256);
    int rc;
    int *seq_len = nullptr, *seq_off = nullptr, *seq_info = nullptr;
    if (pk) {
        if ((rc = e->seq.reserve((size_t)(2 * B + 2) * sizeof(int)))) return rc;
        seq_len = (int*)e->seq.p;
        seq_off = seq_len + B;
        seq_info = seq_off + B;
    }
    hipLaunchKernelGGL(prep_kernel, dim3(B), dim3(64), 0, st, (const long long*)ids, (const long long*)mask,
                       (int*)e->pos.p, (float*)e->kbias.p, L, Lp, c.pad_token_id, c.position_mode == OTTER_POS_ROBERTA,
                       (const long long*)(pk ? pk->p2 : nullptr), (const long long*)(pk ? pk->p3 : nullptr), seq_len);
    if (pk) hipLaunchKernelGGL(seq_offsets_kernel, dim3(1), dim3(64), 0, st, (const int*)seq_len, B, seq_off, seq_info);
    const unsigned rowgrid = (unsigned)((M + 3) / 4);
    hipLaunchKernelGGL((embed_ln_kernel<T, NV>), dim3(rowgrid), dim3(256), 0, st, (const long long*)ids,
                       (const int*)e->pos.p, (const long long*)types, e->w["embeddings.word_embeddings.weight"].d,
                       e->w["embeddings.position_embeddings.weight"].d, e->w["embeddings.token_type_embeddings.weight"].d,
                       e->w["embeddings.LayerNorm.weight"].d, e->w["embeddings.LayerNorm.bias"].d, (T*)e->X.p, (int)M, H,
                       c.vocab_size, c.max_position_embeddings, c.type_vocab_size, c.layer_norm_eps,
                       (const int*)seq_off, (const int*)seq_len, L);
    OTTER_HIP_TRY(hipGetLastError());
    for (int li = 0; li < n_layers; ++li) {
        const LayerW& W = e->layers[li];
        GemmLN ln;
        if (li == 0) {
            ln.tiles_m_dev = seq_info;
            if ((rc = launch_gemm_bf16_ln(GEMM_EPI_BIAS, e->X.p, W.wqkv, W.bqkv, nullptr, e->QKV.p, M_pad, 3 * H, H, ln, e->ncu,
                                          st)))
                return rc;
        } else {
            ln = GemmLN();
            ln.tiles_m_dev = seq_info;
            ln.mode = 1; ln.stat = e->stat2.p; ln.v0 = W.sqkv; ln.v1 = W.cqkv;
            if ((rc = launch_gemm_bf16_ln(GEMM_EPI_BIAS, e->TMP.p, W.wqkv_f, nullptr, nullptr, e->QKV.p, M_pad, 3 * H, H, ln,
                                          e->ncu, st)))
                return rc;
        }
        AttnArgs a;
        a.QKV = e->QKV.p; a.CTX = e->CTX.p; a.kbias = (const float*)e->kbias.p;
        a.B = B; a.L = L; a.Lp = Lp; a.H = H; a.heads = c.num_attention_heads;
        a.scale = e->attn_scale;
        a.seq_off = seq_off; a.seq_len = seq_len;
        a.flags = (unsigned char*)e->aflags.p; a.ncu = e->ncu;
        if (L <= 512) {
            if ((rc = launch_attn_bf16(a, st))) return rc;
        } else {
            a.flags = nullptr;   // every block
            hipLaunchKernelGGL(attn_kernel<T>, dim3((L + 127) / 128, c.num_attention_heads, B), dim3(256), 0, st, a);
            OTTER_HIP_TRY(hipGetLastError());
        }
        // T1 = x + ctx Wo^T + bo, x = the embedding output (layer 0) or LN2 of the previous layer's T2
        ln = GemmLN();
        ln.tiles_m_dev = seq_info;
        ln.part = e->part.p;
        if (li == 0) {
            ln.mode = 1;
            rc = launch_gemm_bf16_ln(GEMM_EPI_BIAS_RESID, e->CTX.p, W.wo, W.bo, e->X.p, e->X1.p, M_pad, H, H, ln, e->ncu, st);
        } else {
            const LayerW& Pv = e->layers[li - 1];
            ln.mode = 2; ln.stat = e->stat2.p; ln.v0 = Pv.ln2g; ln.v1 = Pv.ln2b;
            rc = launch_gemm_bf16_ln(GEMM_EPI_BIAS_RESID, e->CTX.p, W.wo, W.bo, e->TMP.p, e->X1.p, M_pad, H, H, ln, e->ncu, st);
        }
        if (rc) return rc;
        if ((rc = launch_ln_stats(e->part.p, P, M_pad, H, c.layer_norm_eps, e->stat1.p, st))) return rc;
        // MID = gelu(LN1(T1) W1^T + b1)
        ln = GemmLN();
        ln.tiles_m_dev = seq_info;
        ln.mode = 1; ln.stat = e->stat1.p; ln.v0 = W.s1; ln.v1 = W.c1;
        if ((rc = launch_gemm_bf16_ln(GEMM_EPI_BIAS_GELU, e->X1.p, W.w1_f, nullptr, nullptr, e->MID.p, M_pad, I, H, ln, e->ncu,
                                      st)))
            return rc;
        // T2 = LN1(T1) + MID W2^T + b2   (T2 overwrites the previous layer's T2: its last reader was this
        // layer's attention-output GEMM)
        ln = GemmLN();
        ln.tiles_m_dev = seq_info;
        ln.mode = 2; ln.stat = e->stat1.p; ln.v0 = W.ln1g; ln.v1 = W.ln1b; ln.part = e->part.p;
        if ((rc = launch_gemm_bf16_ln(GEMM_EPI_BIAS_RESID, e->MID.p, W.w2, W.b2, e->X1.p, e->TMP.p, M_pad, H, I, ln, e->ncu,
                                      st)))
            return rc;
        if (li + 1 < n_layers) {
            if ((rc = launch_ln_stats(e->part.p, P, M_pad, H, c.layer_norm_eps, e->stat2.p, st))) return rc;
        } else {
            hipLaunchKernelGGL((ln_kernel<T, NV>), dim3(rowgrid), dim3(256), 0, st, (const T*)e->TMP.p, W.ln2g, W.ln2b,
                               (T*)e->X.p, (int)M, H, c.layer_norm_eps, (const int*)seq_off, (const int*)seq_len, L);
            OTTER_HIP_TRY(hipGetLastError());
        }
    }
    if (pk && n_layers == 0) return set_err(OTTER_EINVAL, "packed path needs at least one layer");
    return OTTER_OK;
}

template <typename T>
int run_layers(otter_encoder* e, const int64_t* ids, const int64_t* mask, const int64_t* types, int B, int L,
               int n_layers, hipStream_t st, const PackHint* pk = nullptr) {
    if (sizeof(T) == 2 && e->ln_fused) {
        // packing pays when whole 64-row tiles can be dropped; the 'max' pooling of an EMPTY part mask reads
        // padded positions (tb_retriever.py:50-52), so that method keeps the dense layout
        const char* nopack = getenv("OTTER_NO_PACK");
        if (pk && (L % 64 != 0 || L < 128 || L > 512 || n_layers < 1 || e->cfg.part_pooling == OTTER_POOL_MAX ||
                   e->cfg.part_pooling == OTTER_POOL_ATT_SELF || (nopack && nopack[0] == '1')))
            pk = nullptr;
        switch (e->cfg.hidden_size / 256) {
            case 1: return run_layers_fused<1>(e, ids, mask, types, B, L, n_layers, st, pk);
            case 2: return run_layers_fused<2>(e, ids, mask, types, B, L, n_layers, st, pk);
            case 3: return run_layers_fused<3>(e, ids, mask, types, B, L, n_layers, st, pk);
            case 4: return run_layers_fused<4>(e, ids, mask, types, B, L, n_layers, st, pk);
        }
    }
    switch (e->cfg.hidden_size / 256) {
        case 1: return run_layers_nv<T, 1>(e, ids, mask, types, B, L, n_layers, st);
        case 2: return run_layers_nv<T, 2>(e, ids, mask, types, B, L, n_layers, st);
        case 3: return run_layers_nv<T, 3>(e, ids, mask, types, B, L, n_layers, st);
        case 4: return run_layers_nv<T, 4>(e, ids, mask, types, B, L, n_layers, st);
    }
    return set_err(OTTER_ENOTIMPL, "hidden_size %d not supported", e->cfg.hidden_size);
}

template <typename T>
int pool_out(otter_encoder* e, const int64_t* p2, const int64_t* p3, int B, int L, float* out, hipStream_t st) {
    const auto& c = e->cfg;
    const int H = c.hidden_size, D = c.three_cat ? 3 * H : H;
    int part_lo = 0;
    if (!c.no_proj) {
        hipLaunchKernelGGL(proj_ln_kernel<T>, dim3(B), dim3(256), (2 * H + 8) * sizeof(float), st, (const T*)e->X.p,
                           e->w["project.0.weight"].d, e->w["project.0.bias"].d, e->w["project.1.weight"].d,
                           e->w["project.1.bias"].d, out, L, H, D, c.layer_norm_eps);
        part_lo = 1;
    }
    int parts = (c.three_cat ? 3 : 1) - part_lo;
    if (c.three_cat && c.part_pooling == OTTER_POOL_ATT_SELF) {
        hipLaunchKernelGGL(att_pool_kernel<T>, dim3(B, 2), dim3(256), (size_t)(L + 8) * sizeof(float), st, (const T*)e->X.p,
                           (const long long*)p2, (const long long*)p3, e->w["self_proj_tab.weight"].d,
                           e->w["self_proj_tab.bias"].d, e->w["self_proj_psg.weight"].d, e->w["self_proj_psg.bias"].d, out,
                           L, H, D);
        parts = 1 - part_lo;             // part 0 (h[:,0]) below unless the projection wrote it
    }
    if (parts > 0)
        hipLaunchKernelGGL(pool_kernel<T>, dim3(B, parts), dim3(256), 0, st, (const T*)e->X.p, (const long long*)p2,
                           (const long long*)p3, out, L, H, D, c.part_pooling, part_lo);
    OTTER_HIP_TRY(hipGetLastError());
    return OTTER_OK;
}

int check_call(otter_encoder* e, const void* ids, const void* mask, int B, int L, const char* what) {
    if (!e) return set_err(OTTER_EINVAL, "%s: encoder is NULL", what);
    if (e->dtype < 0) return set_err(OTTER_ESTATE, "%s: call otter_encoder_finalize first", what);
    if (!ids || !mask) return set_err(OTTER_EINVAL, "%s: input_ids and input_mask must be non-NULL", what);
    if (B < 1 || L < 1) return set_err(OTTER_EINVAL, "%s: bad batch shape B=%d L=%d", what, B, L);
    const int maxL = e->cfg.position_mode == OTTER_POS_ROBERTA
                         ? e->cfg.max_position_embeddings - e->cfg.pad_token_id - 1
                         : e->cfg.max_position_embeddings;
    if (L > maxL) return set_err(OTTER_EINVAL, "%s: L=%d exceeds the position table (max %d)", what, L, maxL);
    if ((long long)B * L > (1ll << 30)) return set_err(OTTER_EINVAL, "%s: B*L too large", what);
    return OTTER_OK;
}

}  // namespace

extern "C" {

int otter_encoder_create(const otter_encoder_cfg* cfg, int device, otter_encoder** out) {
    if (!out) return set_err(OTTER_EINVAL, "encoder_create: out is NULL");
    *out = nullptr;
    if (!cfg) return set_err(OTTER_EINVAL, "encoder_create: cfg is NULL");
    const otter_encoder_cfg& c = *cfg;
    if (c.hidden_size <= 0 || c.hidden_size % 256 != 0 || c.hidden_size > 1024)
        return set_err(OTTER_ENOTIMPL, "encoder_create: hidden_size %d must be 256, 512, 768 or 1024", c.hidden_size);
    if (c.num_attention_heads <= 0 || c.hidden_size != c.num_attention_heads * 64)
        return set_err(OTTER_ENOTIMPL, "encoder_create: head dimension must be 64 (hidden %d, heads %d)", c.hidden_size,
                       c.num_attention_heads);
    if (c.intermediate_size <= 0 || c.intermediate_size % 256 != 0)
        return set_err(OTTER_ENOTIMPL, "encoder_create: intermediate_size %d must be a multiple of 256",
                       c.intermediate_size);
    if (c.num_hidden_layers < 1 || c.vocab_size < 1 || c.max_position_embeddings < 1 || c.type_vocab_size < 1)
        return set_err(OTTER_EINVAL, "encoder_create: bad layer/vocab/position/type sizes");
    if (c.part_pooling < OTTER_POOL_MEAN || c.part_pooling > OTTER_POOL_ATT_SELF)
        return set_err(OTTER_EINVAL, "encoder_create: unknown pooling %d", c.part_pooling);
    if (c.position_mode != OTTER_POS_BERT && c.position_mode != OTTER_POS_ROBERTA)
        return set_err(OTTER_EINVAL, "encoder_create: unknown position_mode %d", c.position_mode);
    int ndev = 0;
    OTTER_HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return set_err(OTTER_EINVAL, "encoder_create: device %d out of range (%d visible)", device, ndev);
    hipDeviceProp_t prop;
    OTTER_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(OTTER_ESTATE, "encoder_create: device %d is %s; this library is built for gfx950 only", device,
                       prop.gcnArchName);
    otter_encoder* e = new (std::nothrow) otter_encoder();
    if (!e) return set_err(OTTER_ENOMEM, "encoder_create: host allocation failed");
    e->cfg = c;
    e->device = device;
    e->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = e;
    return OTTER_OK;
}

int otter_encoder_set_weight(otter_encoder* e, const char* name, const float* host_ptr, const int64_t* shape, int ndim) {
    if (!e || !name || !host_ptr || !shape) return set_err(OTTER_EINVAL, "set_weight: NULL argument");
    if (e->dtype >= 0) return set_err(OTTER_ESTATE, "set_weight: encoder already finalized");
    std::vector<int64_t> want;
    bool ignored = false;
    if (expected_shape(e->cfg, name, &want, &ignored)) return set_err(OTTER_EINVAL, "set_weight: unknown tensor '%s'", name);
    if (ignored) return OTTER_OK;
    bool ok = (int)want.size() == ndim;
    for (int i = 0; ok && i < ndim; ++i) ok = want[i] == shape[i];
    if (!ok) {
        std::string got, exp;
        for (int i = 0; i < ndim; ++i) got += std::to_string(shape[i]) + (i + 1 < ndim ? "," : "");
        for (size_t i = 0; i < want.size(); ++i) exp += std::to_string(want[i]) + (i + 1 < want.size() ? "," : "");
        return set_err(OTTER_EINVAL, "set_weight: '%s' has shape [%s], expected [%s]", name, got.c_str(), exp.c_str());
    }
    DeviceGuard g(e->device);
    if (!g.ok) return set_err(OTTER_EHIP, "set_weight: hipSetDevice(%d) failed", e->device);
    Tensor& t = e->w[name];
    t.shape = want;
    const size_t bytes = t.numel() * sizeof(float);
    if (!t.d) OTTER_HIP_TRY(hipMalloc((void**)&t.d, bytes));
    OTTER_HIP_TRY(hipMemcpy(t.d, host_ptr, bytes, hipMemcpyHostToDevice));
    return OTTER_OK;
}

int otter_encoder_finalize(otter_encoder* e, int compute_dtype) {
    if (!e) return set_err(OTTER_EINVAL, "finalize: encoder is NULL");
    if (e->dtype >= 0) return set_err(OTTER_ESTATE, "finalize: already finalized");
    if (compute_dtype != OTTER_F32 && compute_dtype != OTTER_BF16)
        return set_err(OTTER_ENOTIMPL, "finalize: compute dtype %d not built (OTTER_F32 or OTTER_BF16)", compute_dtype);
    // every tensor the configured path reads must be present
    std::vector<std::string> need = {"embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
                                     "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight",
                                     "embeddings.LayerNorm.bias"};
    const char* per_layer[] = {"attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight",
                               "attention.self.key.bias", "attention.self.value.weight", "attention.self.value.bias",
                               "attention.output.dense.weight", "attention.output.dense.bias",
                               "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
                               "intermediate.dense.weight", "intermediate.dense.bias", "output.dense.weight",
                               "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias"};
    for (int li = 0; li < e->cfg.num_hidden_layers; ++li)
        for (const char* n : per_layer) need.push_back("encoder.layer." + std::to_string(li) + "." + n);
    if (!e->cfg.no_proj)
        for (const char* n : {"project.0.weight", "project.0.bias", "project.1.weight", "project.1.bias"}) need.push_back(n);
    if (e->cfg.three_cat && e->cfg.part_pooling == OTTER_POOL_ATT_SELF)
        for (const char* n : {"self_proj_tab.weight", "self_proj_tab.bias", "self_proj_psg.weight", "self_proj_psg.bias"})
            need.push_back(n);
    for (const auto& n : need)
        if (e->w.find(n) == e->w.end()) return set_err(OTTER_ESTATE, "finalize: tensor '%s' was never set", n.c_str());
    DeviceGuard g(e->device);
    if (!g.ok) return set_err(OTTER_EHIP, "finalize: hipSetDevice(%d) failed", e->device);
    // bf16: LayerNorm folded into the GEMMs around it (OTTER_NO_LN_FUSE=1 keeps the separate kernels, for A/B runs)
    const char* nofuse = getenv("OTTER_NO_LN_FUSE");
    // (the folded consumers keep two epilogue vectors of their output width in LDS: width <= 3072)
    e->ln_fused = compute_dtype == OTTER_BF16 && !(nofuse && nofuse[0] == '1') && e->cfg.intermediate_size <= 3072 &&
                  3 * e->cfg.hidden_size <= 3072;
    int rc = compute_dtype == OTTER_BF16 ? finalize_t<bf16_t>(e) : finalize_t<float>(e);
    if (rc) return rc;
    e->dtype = compute_dtype;
    return OTTER_OK;
}

int otter_encoder_forward(otter_encoder* e, const int64_t* input_ids, const int64_t* input_mask, const int64_t* type_ids,
                          const int64_t* part2_mask, const int64_t* part3_mask, int B, int L, float* out, void* stream) {
    int rc = check_call(e, input_ids, input_mask, B, L, "forward");
    if (rc) return rc;
    if (!out) return set_err(OTTER_EINVAL, "forward: out is NULL");
    if (e->cfg.three_cat && e->cfg.part_pooling != OTTER_POOL_CLS && (!part2_mask || !part3_mask))
        return set_err(OTTER_EINVAL, "forward: part2_mask and part3_mask are required for the three-vector encoder");
    DeviceGuard g(e->device);
    if (!g.ok) return set_err(OTTER_EHIP, "forward: hipSetDevice(%d) failed", e->device);
    if ((rc = ensure_ws(e, B, L))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int nl = e->cfg.num_hidden_layers;
    if (e->dtype == OTTER_BF16) {
        const PackHint pk = {part2_mask, part3_mask};
        if ((rc = run_layers<bf16_t>(e, input_ids, input_mask, type_ids, B, L, nl, st, &pk))) return rc;
        return pool_out<bf16_t>(e, part2_mask, part3_mask, B, L, out, st);
    }
    if ((rc = run_layers<float>(e, input_ids, input_mask, type_ids, B, L, nl, st))) return rc;
    return pool_out<float>(e, part2_mask, part3_mask, B, L, out, st);
}

int otter_encoder_forward_hidden(otter_encoder* e, const int64_t* input_ids, const int64_t* input_mask,
                                 const int64_t* type_ids, int B, int L, int n_layers, float* hidden, void* stream) {
    int rc = check_call(e, input_ids, input_mask, B, L, "forward_hidden");
    if (rc) return rc;
    if (!hidden) return set_err(OTTER_EINVAL, "forward_hidden: hidden is NULL");
    if (n_layers < 0 || n_layers > e->cfg.num_hidden_layers)
        return set_err(OTTER_EINVAL, "forward_hidden: n_layers=%d outside [0, %d]", n_layers, e->cfg.num_hidden_layers);
    DeviceGuard g(e->device);
    if (!g.ok) return set_err(OTTER_EHIP, "forward_hidden: hipSetDevice(%d) failed", e->device);
    if ((rc = ensure_ws(e, B, L))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)B * L * e->cfg.hidden_size;
    const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 8192);
    if (e->dtype == OTTER_BF16) {
        if ((rc = run_layers<bf16_t>(e, input_ids, input_mask, type_ids, B, L, n_layers, st))) return rc;
        hipLaunchKernelGGL(to_f32_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)e->X.p, hidden, n);
    } else {
        if ((rc = run_layers<float>(e, input_ids, input_mask, type_ids, B, L, n_layers, st))) return rc;
        hipLaunchKernelGGL(to_f32_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)e->X.p, hidden, n);
    }
    OTTER_HIP_TRY(hipGetLastError());
    return OTTER_OK;
}

int otter_gemm_bf16(int device, int epi, const void* A, const void* W, const float* bias, const void* R, void* C,
                    int64_t M, int N, int K, void* stream) {
    if (!A || !W || !C) return set_err(OTTER_EINVAL, "gemm_bf16: NULL operand");
    if (epi < GEMM_EPI_BIAS || epi > GEMM_EPI_BIAS_RESID) return set_err(OTTER_EINVAL, "gemm_bf16: unknown epilogue %d", epi);
    DeviceGuard g(device);
    if (!g.ok) return set_err(OTTER_EHIP, "gemm_bf16: hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    OTTER_HIP_TRY(hipGetDeviceProperties(&prop, device));
    return launch_gemm_bf16(epi, A, W, bias, R, C, M, N, K, prop.multiProcessorCount, (hipStream_t)stream);
}

int otter_gemm_bf16_ln(int device, int epi, int ln_mode, const void* A, const void* W, const float* bias, const void* R,
                       void* C, const void* stat, const float* v0, const float* v1, void* part, int64_t M, int N, int K,
                       void* stream) {
    if (!A || !W || !C) return set_err(OTTER_EINVAL, "gemm_bf16_ln: NULL operand");
    DeviceGuard g(device);
    if (!g.ok) return set_err(OTTER_EHIP, "gemm_bf16_ln: hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    OTTER_HIP_TRY(hipGetDeviceProperties(&prop, device));
    GemmLN ln;
    ln.mode = ln_mode; ln.stat = stat; ln.v0 = v0; ln.v1 = v1; ln.part = part;
    return launch_gemm_bf16_ln(epi, A, W, bias, R, C, M, N, K, ln, prop.multiProcessorCount, (hipStream_t)stream);
}

int otter_ln_stats(int device, const void* part, int P, int64_t M, int N, float eps, void* stat, void* stream) {
    if (!part || !stat || P <= 0 || M <= 0 || N <= 0) return set_err(OTTER_EINVAL, "ln_stats: bad argument");
    DeviceGuard g(device);
    if (!g.ok) return set_err(OTTER_EHIP, "ln_stats: hipSetDevice(%d) failed", device);
    return launch_ln_stats(part, P, M, N, eps, stat, (hipStream_t)stream);
}

int otter_ln_fold(int device, const float* W, const float* gamma, const float* beta, const float* bias, void* Wout,
                  float* s_out, float* c_out, int N, int K, void* stream) {
    if (!W || !gamma || !beta || !Wout || !s_out || !c_out || N <= 0 || K <= 0)
        return set_err(OTTER_EINVAL, "ln_fold: bad argument");
    DeviceGuard g(device);
    if (!g.ok) return set_err(OTTER_EHIP, "ln_fold: hipSetDevice(%d) failed", device);
    return launch_fold_ln(W, gamma, beta, bias, Wout, s_out, c_out, N, K, (hipStream_t)stream);
}

int otter_encoder_out_dim(const otter_encoder* e) {
    if (!e) return -1;
    return e->cfg.three_cat ? 3 * e->cfg.hidden_size : e->cfg.hidden_size;
}

void otter_encoder_destroy(otter_encoder* e) {
    if (!e) return;
    DeviceGuard g(e->device);
    for (auto& kv : e->w)
        if (kv.second.d) (void)hipFree(kv.second.d);
    for (void* p : e->owned) (void)hipFree(p);
    e->X.release(); e->X1.release(); e->QKV.release(); e->CTX.release();
    e->TMP.release(); e->MID.release(); e->pos.release(); e->kbias.release();
    e->part.release(); e->stat1.release(); e->stat2.release(); e->seq.release(); e->aflags.release();
    delete e;
}

}  // extern "C"
