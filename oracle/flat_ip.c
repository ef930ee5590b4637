/*
 * oracle/flat_ip.c — CPU restatement of faiss.IndexFlatIP add()/search().
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under otter_amd/ may import, link or call
 * this file; it is the checker for tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.
 *
 * PARITY UNPINNED: the algorithm lives in the un-vendored third-party package
 * `faiss-gpu` (version not pinned; /root/reference/README.md:23), which is not
 * installed and whose sources are not under /root/reference.  The reference has
 * no tests, fixtures or golden vectors for this path.  This
 * file restates faiss' published contract for IndexFlatIP, anchored on the
 * reference's call sites:
 *     index = faiss.IndexFlatIP(d); index.add(xb)     scripts/eval_ottqa_retrieval.py:108-109
 *     D, I = index.search(q, k)                       scripts/eval_ottqa_retrieval.py:129
 *     `-1` ids when fewer than k rows exist           retrieval/eval_ottqa_retrieval_tfidffilter.py:282-284
 * Contract: S = q . xb^T in float32; per query the k largest scores in
 * descending order; ids are insertion order (int64); missing entries are
 * I = -1, D = -FLT_MAX.  faiss leaves the order of equal scores undefined; this
 * oracle (and the HIP path) use (score descending, id ascending).
 *
 * Three accumulation modes, because float32 summation order is not part of the
 * contract:
 *   mode 0  plain float32 accumulation, k ascending (naive loop)
 *   mode 1  the exact order of the gfx950 kernel: one fmaf chain per output,
 *           k visited as 8-groups in the order 0,4,1,5,2,6,3,7 (the k-pairing of
 *           v_mfma_f32_32x32x2_f32 fed with 16-byte fragments) — bit-exact oracle
 *   mode 2  float64 accumulation (arbiter for near-ties), score rounded to float32
 *   mode 3  eight float32 partial sums over k mod 8, added pairwise at the end: the shape of a
 *           SIMD inner-product loop such as faiss' fvec_inner_product (AVX2, 8 lanes) -- used as
 *           the timed "reference-style" CPU baseline (nq = 1 per scan), never as a parity target
 *
 * Build: gcc -O2 -mavx2 -mfma -fopenmp -shared -fPIC oracle/flat_ip.c -o oracle/liboracle_flat_ip.so -lm
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float s; int64_t id; } cand_t;

/* "a ranks before b": score descending, id ascending */
static inline int before(const cand_t a, const cand_t b) {
    return (a.s > b.s) || (a.s == b.s && a.id < b.id);
}

/* binary heap whose root is the WORST kept candidate */
static void sift_down(cand_t* h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, w = i;
        if (l < n && before(h[w], h[l])) w = l;
        if (r < n && before(h[w], h[r])) w = r;
        if (w == i) return;
        cand_t t = h[i]; h[i] = h[w]; h[w] = t;
        i = w;
    }
}

static int cmp_cand(const void* pa, const void* pb) {
    const cand_t a = *(const cand_t*)pa, b = *(const cand_t*)pb;
    if (before(a, b)) return -1;
    if (before(b, a)) return 1;
    return 0;
}

static inline float dot_mode0(const float* x, const float* q, int d) {
    float acc = 0.0f;
    for (int j = 0; j < d; ++j) acc += x[j] * q[j];
    return acc;
}

/* order of k inside each 8-group as the HIP kernel feeds the MFMA */
static const int kperm8[8] = {0, 4, 1, 5, 2, 6, 3, 7};

static inline float dot_mode1(const float* x, const float* q, int d) {
    float acc = 0.0f;
    int j = 0;
    for (; j + 8 <= d; j += 8)
        for (int t = 0; t < 8; ++t) acc = fmaf(x[j + kperm8[t]], q[j + kperm8[t]], acc);
    /* tail: the kernel pads d with zeros up to a multiple of 64; fmaf(0,0,acc) = acc,
     * so the remaining terms are visited in the same permuted order with the
     * missing ones skipped */
    if (j < d) {
        for (int t = 0; t < 8; ++t) {
            int kk = j + kperm8[t];
            if (kk < d) acc = fmaf(x[kk], q[kk], acc);
        }
    }
    return acc;
}

static inline float dot_mode2(const float* x, const float* q, int d) {
    double acc = 0.0;
    for (int j = 0; j < d; ++j) acc += (double)x[j] * (double)q[j];
    return (float)acc;
}

static inline float dot_mode3(const float* x, const float* q, int d) {
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int j = 0;
    for (; j + 8 <= d; j += 8)
        for (int t = 0; t < 8; ++t) p[t] += x[j + t] * q[j + t];
    for (int t = 0; j < d; ++j, ++t) p[t] += x[j] * q[j];
    return ((p[0] + p[4]) + (p[2] + p[6])) + ((p[1] + p[5]) + (p[3] + p[7]));
}

/* returns 0 on success */
int oracle_flat_ip_search(const float* xb, int64_t n, int d, const float* q, int64_t nq, int k,
                          float* D, int64_t* I, int mode, int64_t id_base, int nthreads) {
    if (d <= 0 || k <= 0 || n < 0 || nq < 0) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    int fail = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t iq = 0; iq < nq; ++iq) {
        cand_t* heap = (cand_t*)malloc(sizeof(cand_t) * (size_t)k);
        if (!heap) { fail = 1; continue; }
        int hn = 0;
        const float* qq = q + iq * (int64_t)d;
        for (int64_t i = 0; i < n; ++i) {
            const float* x = xb + i * (int64_t)d;
            float s = mode == 1 ? dot_mode1(x, qq, d) : mode == 2 ? dot_mode2(x, qq, d) : mode == 3 ? dot_mode3(x, qq, d) : dot_mode0(x, qq, d);
            cand_t c = {s, i};
            if (hn < k) {
                heap[hn++] = c;
                if (hn == k) for (int t = k / 2 - 1; t >= 0; --t) sift_down(heap, k, t);
            } else if (before(c, heap[0])) {
                heap[0] = c;
                sift_down(heap, k, 0);
            }
        }
        qsort(heap, (size_t)hn, sizeof(cand_t), cmp_cand);
        for (int t = 0; t < k; ++t) {
            if (t < hn) { D[iq * k + t] = heap[t].s; I[iq * k + t] = heap[t].id + id_base; }
            else        { D[iq * k + t] = -FLT_MAX;  I[iq * k + t] = -1; }
        }
        free(heap);
    }
    return fail ? -2 : 0;
}

/* merge of G sorted candidate lists per query (the sharded search's last step):
 * Dc [G, nq, k], Ic [G, nq, k] -> D [nq, k], I [nq, k]; ids < 0 are padding. */
int oracle_merge_topk(const float* Dc, const int64_t* Ic, int G, int64_t nq, int k,
                      float* D, int64_t* I) {
    if (G <= 0 || k <= 0 || nq < 0) return -1;
    cand_t* all = (cand_t*)malloc(sizeof(cand_t) * (size_t)G * (size_t)k);
    if (!all) return -2;
    for (int64_t iq = 0; iq < nq; ++iq) {
        int m = 0;
        for (int g = 0; g < G; ++g)
            for (int t = 0; t < k; ++t) {
                int64_t id = Ic[((int64_t)g * nq + iq) * k + t];
                if (id >= 0) { all[m].s = Dc[((int64_t)g * nq + iq) * k + t]; all[m].id = id; ++m; }
            }
        qsort(all, (size_t)m, sizeof(cand_t), cmp_cand);
        for (int t = 0; t < k; ++t) {
            if (t < m) { D[iq * k + t] = all[t].s; I[iq * k + t] = all[t].id; }
            else       { D[iq * k + t] = -FLT_MAX; I[iq * k + t] = -1; }
        }
    }
    free(all);
    return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
