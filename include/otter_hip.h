/*
 * otter_hip.h — C ABI of libotter_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for OTTeR's dense-retrieval hot path.  The reference has no
 * FFI of its own: its boundary is two Python call surfaces.
 * Each entry point below cites the reference call it replaces
 * (paths relative to the reference tree).
 *
 *   index   : faiss.IndexFlatIP(d) / .add(xb) / .search(q, k)
 *             scripts/eval_ottqa_retrieval.py:108-109,129
 *   encoder : model = Cls(config, args); load_saved(...); model(batch)['embed']
 *             scripts/encode_corpus.py:54-68,98,139-144
 *             retrieval/models/tb_retriever.py:150-261, retrieval/models/retriever.py:249-345
 *
 * Conventions
 *   - every function returns int: 0 = OK, negative = error class below; the
 *     message is kept per thread and read with otter_last_error().
 *   - the caller owns every buffer passed in or out; an index owns its private
 *     HBM copy of the corpus; an encoder owns its private HBM copy of the weights.
 *   - handles are not thread-safe; one handle per device.
 *   - `stream` is a hipStream_t passed as void* (NULL = the device's default
 *     stream).  Calls with host buffers synchronise the stream before returning.
 *     Calls with device buffers enqueue their work on the stream and return, with
 *     these exceptions, which also wait for it: otter_index_search on an
 *     OTTER_F32 index of 65,536 and more rows whose coarse-pass planes fit HBM
 *     (one 4-byte word per pass of the coarse search: how many queries still
 *     lack a proof; usually one pass.  The last kernel of the pass stores it into
 *     pinned memory and the host polls it -- the call returns when the results
 *     are decided, the kernel's tail may still be retiring on the stream), otter_index_search_subset (validation of
 *     the id lists), otter_index_add when the allocation grows, and
 *     otter_index_timing.  otter_index_search_noalloc, searches of an OTTER_BF16
 *     index and searches of smaller fp32 indexes only enqueue.
 *   - no torch types anywhere; plain pointers and sizes only.
 */
#ifndef OTTER_HIP_H
#define OTTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error classes */
#define OTTER_OK        0
#define OTTER_EINVAL   (-1)   /* bad argument (shape, k, dtype, NULL)            */
#define OTTER_ENOMEM   (-2)   /* hipMalloc failed                                */
#define OTTER_EHIP     (-3)   /* any other HIP runtime error                     */
#define OTTER_ESTATE   (-4)   /* call not valid in the handle's current state    */
#define OTTER_ENOTIMPL (-5)   /* configuration not built into this library       */

/* element types */
#define OTTER_F32   0
#define OTTER_BF16  1
#define OTTER_F16   2
#define OTTER_I64   3

/* largest k the fused scan keeps per query (same bound as faiss' GPU k-select) */
#define OTTER_MAX_K 2048

const char* otter_last_error(void);
/* library ABI version (bumped on any signature change) */
int otter_version(void);
/* number of visible HIP devices, or a negative error class */
int otter_device_count(void);

/* ------------------------------------------------------------------ index */

typedef struct otter_index otter_index;

/* faiss.IndexFlatIP(d)  — scripts/eval_ottqa_retrieval.py:108.
 * dtype = storage type of the corpus in HBM (OTTER_F32: exact IndexFlatIP
 * arithmetic in fp32).  The index lives on HIP device `device`. */
int otter_index_create(int d, int dtype, int device, otter_index** out);
/* dtype OTTER_BF16 (no counterpart in the reference; BASELINE's 100M-row configuration): rows are
 * rounded to bfloat16 on add(), queries on search(); scores are accumulated in float32 and are
 * NOT IndexFlatIP's float32 scores (reported separately: recall@k against the OTTER_F32 index).
 * k <= 1280 for this storage type. */

/* index.add(xb) — scripts/eval_ottqa_retrieval.py:109.
 * x: float32 row-major [n, d], host (x_is_device = 0) or device pointer.
 * ids are insertion order, continuing from ntotal. */
int otter_index_add(otter_index* ix, const void* x, int64_t n, int x_is_device, void* stream);

/* D, I = index.search(q, k) — scripts/eval_ottqa_retrieval.py:129.
 * q: float32 [nq, d]; D: float32 [nq, k]; I: int64 [nq, k]; all three host
 * (io_is_device = 0) or all three device pointers.  Rows are sorted by
 * (score descending, id ascending); when fewer than k rows exist the tail is
 * I = -1, D = -FLT_MAX (faiss convention; the reference guards for it at
 * retrieval/eval_ottqa_retrieval_tfidffilter.py:282-284). 1 <= k <= OTTER_MAX_K. */
int otter_index_search(otter_index* ix, const void* q, int64_t nq, int k,
                       float* D, int64_t* I, int io_is_device, void* stream);

/* otter_index_search for callers that cannot allocate.  NO counterpart in the reference: it exists for callers
 * that cannot call hipMalloc -- code under HIP graph capture, tight serving loops (the reference's real call
 * pattern is one query per call, scripts/eval_ottqa_retrieval.py:126-129), harnesses that forbid allocation.
 * Same kernels and bit-identical results as otter_index_search; every byte of device scratch (per-strip
 * candidate lists and counts, block maxima and sync words of the threshold exchange) lives in ONE static 34 MiB
 * __device__ array per GPU that is part of the library's code object (each GPU the library runs on holds its
 * copy until the process ends).  The call issues no hipMalloc / hipFree and no allocating library routine, and
 * only enqueues work on `stream`.
 * Limits (OTTER_EINVAL beyond them): OTTER_F32 index, 0 <= nq <= 32, 1 <= k <= 128, d <= 4096; q, D, I are
 * device pointers (float32 [nq, d], float32 [nq, k], int64 [nq, k]).  The measurement hook
 * (otter_index_set_timing) does not cover this entry point.
 * All calls on one GPU share that GPU's array: two calls that overlap in time on the same GPU (different
 * streams or threads, any index) race; calls on one stream are ordered and safe. */
int otter_index_search_noalloc(otter_index* ix, const float* q, int64_t nq, int k,
                               float* D, int64_t* I, void* stream);

/* index.ntotal / index.d */
/* Search restricted to per-query row subsets.  Replaces the reference's per-question throw-away index
 * (retrieval/eval_ottqa_retrieval_tfidffilter.py:530-546: faiss.IndexFlatIP(d); index.add(qid2emb[qid]);
 * index.search(q[idx:idx+1], beam_size)) with one batched call on the resident corpus.
 * ids: int64 [nq, m], row ids of this index (insertion order + id_base), lists shorter than m padded with -1;
 * any other value is an error (OTTER_EINVAL).  D float32 [nq, k], I int64 [nq, k]: the k best listed rows per
 * query by (score descending, id ascending), I = the row id as listed (not the position in the list);
 * fewer than k valid ids leave I = -1, D = -FLT_MAX.  Scores are bit-identical to otter_index_search's for
 * the same rows.  fp32 indexes only.  q, ids, D, I are all host or all device pointers (io_is_device);
 * the call synchronises the stream (id validation). */
int otter_index_search_subset(otter_index* index, const void* q, int64_t nq, int k, const int64_t* ids, int m,
                              float* D, int64_t* I, int io_is_device, void* stream);

/* rows [i0, i0 + n) of the index as float32 [n, d] (faiss: index.reconstruct_n(i0, n, out); what
 * faiss.write_index serialises, scripts/eval_ottqa_retrieval.py:115).  A bf16 index returns its rounded values. */
int otter_index_reconstruct_n(otter_index* index, int64_t i0, int64_t n, float* out, int out_is_device, void* stream);

int64_t otter_index_ntotal(const otter_index* ix);
int     otter_index_d(const otter_index* ix);

/* index.reset(): drop all rows, keep the allocation */
int otter_index_reset(otter_index* ix);

/* Row-sharded use (replaces faiss.index_cpu_to_all_gpus,
 * scripts/eval_ottqa_retrieval.py:110-112): ids reported by search() become
 * id_base + local row, so that per-shard results carry global ids. */
int otter_index_set_id_base(otter_index* ix, int64_t id_base);

void otter_index_destroy(otter_index* ix);

/* Measurement hook (no counterpart in the reference; used by bench.py for the
 * roofline figure).  While enabled, search() brackets its scan (the scan kernel
 * and, for small query batches, the sample scan + sample merge ahead of it: every
 * corpus byte is read inside the bracket exactly once) with HIP events on the
 * caller's stream.  otter_index_timing() waits for the recorded
 * events and returns the summed scan-kernel time and the number of launches
 * since the previous call (at most 4096 launches are kept between calls). */
int otter_index_set_timing(otter_index* ix, int enable);
int otter_index_timing(otter_index* ix, double* scan_ms, int64_t* scan_launches);

/* State of the coarse pass of an index (no counterpart in the reference): 0 = its bf16 planes are not built yet
 * (they are built by the first search: once the corpus' bytes again for the hi/lo planes,
 * half of them again for the one-product plane), 1 = built and in use, 2 = HBM had no room for them, so the exact
 * fp32 kernel serves every batch size (about 1/6 of the rate at 2,214 queries; a note goes to stderr once),
 * 3 = not an OTTER_F32 index, 4 = switched off for this corpus until reset(): even the three-product pass left more
 * than a quarter of a batch unproven (scores closer together than bf16 products resolve, e.g. near-duplicate rows),
 * so the exact kernel serves every batch instead of paying for coarse passes that prove nothing. */
int otter_index_coarse_state(const otter_index* ix);

/* HBM economy (no counterpart in the reference).  Besides its float32 rows an OTTER_F32 index builds, at its first
 * search, bf16 planes for the coarse pass: planes = 2 (default) the hi/lo planes (the corpus' bytes again: two- and
 * three-product passes, the fallbacks of large batches) AND the hi plane alone (half of them: the one-product pass
 * every batch size starts with) -- 2.5x the corpus in all; planes = 1 the hi plane only -- 1.5x: a query the
 * one-product pass cannot prove goes to the exact float32 kernels, which costs nothing on corpora whose proofs
 * succeed (every corpus measured except near-duplicate rows) and is the right choice for a process that serves
 * small query batches; planes = 0 none -- 1.0x: the exact float32 kernels serve everything (batches of 33 and more
 * at about 1/6 of the rate).  Results are the same bits under every setting.  Planes no longer wanted are freed
 * (the device is synchronised first); may be called at any time between searches. */
int otter_index_set_planes(otter_index* ix, int planes);

/* Diagnostic (no counterpart in the reference).  Large query batches on an OTTER_F32 index run a
 * coarse pass on the bf16 matrix cores (hi/lo split of both operands), re-score its k + margin
 * candidates with the exact float32 chain and PROVE per query that no other row can reach the top
 * k; a query without a proof is searched again by the exact float32 kernel.  Returns how many
 * queries of the last search() call (summed over its 16,384-query chunks) took that second route (0 in the usual case). */
int64_t otter_index_last_fallback(const otter_index* ix);
/* Queries of the last search() that the two-product coarse pass could not prove and that were searched again
 * with the three-product pass (or, if fewer than 33, by the exact kernel); 0 on the other paths. */
int64_t otter_index_last_refined(const otter_index* index);

/* Final merge of the row-sharded search: G sorted candidate lists per query
 * (Dc float32 [G, nq, k], Ic int64 [G, nq, k], device pointers, as returned
 * by G per-shard search() calls and exchanged with an all-gather) are merged
 * into D [nq, k], I [nq, k] (device) under the same (score desc, id asc)
 * order; entries with id < 0 are padding.  G * k <= 8192 (merge hierarchically
 * beyond that: the output has the layout of one input list). */
int otter_merge_topk(int device, const float* Dc, const int64_t* Ic, int G, int64_t nq, int k,
                     float* D, int64_t* I, void* stream);

/* index = faiss.index_cpu_to_all_gpus(index); index.search(q, k) -- scripts/eval_ottqa_retrieval.py:110-112,129:
 * ONE process driving every GPU.  (The reference replicates the whole index on each GPU; here the rows are
 * sharded, so G GPUs hold G times the corpus and each query reads every row once.)
 * shards[0..G): indexes created on their own devices (several may share a device), each holding a contiguous
 * row range with otter_index_set_id_base(first global row).  q, D, I as in otter_index_search: all host
 * pointers, or all device pointers on the device of shards[0] (the "home" GPU; `stream` is a stream there).
 * Every shard is searched from its own host thread on its own stream (a search may wait for its GPU, and the
 * GPUs must not wait for each other); queries and candidates travel as peer copies (xGMI), 12 bytes per
 * candidate; the G-way merge (otter_merge_topk, hierarchical past G * k = 8192) runs on the home GPU.  No
 * communicator is involved: inside one process peer copies need none.  The multi-process form (one rank per
 * GPU, RCCL all-gather of the packed candidates) is otter_amd/sharded.py over otter_index_search and
 * otter_merge_topk.  The call waits for `stream` and for every shard before it returns.  G <= 64.
 * Not thread-safe per handle, like every call here: one otter_sharded_search at a time per set of shards. */
int otter_sharded_search(otter_index* const* shards, int G, const void* q, int64_t nq, int k,
                         float* D, int64_t* I, int io_is_device, void* stream);

/* In-batch score matrix of the fine-tuning script's dev-set evaluation (reference
 * scripts/train_1hop_tb_retrieval.py:304-319; the same matrix is the input of loss_single,
 * retrieval/criterions.py:15-17): product[i][j] = q_i . c_j for j < B and, when neg_c is given,
 * product[i][B] = q_i . neg_c_i (torch.mm + cat, float32).  rank[i] = how many columns sort before the
 * target column i in descending order (equal scores: lower column first, as argmax picks), so
 * accuracy = mean(rank == 0) and MRR = mean(1 / (rank + 1)).  All pointers are device pointers:
 * q, c, neg_c float32 [B, d]; product float32 [B, B + (neg_c != NULL)]; rank int32 [B].  B <= 8192. */
int otter_inbatch_scores(int device, const float* q, const float* c, const float* neg_c, int B, int d,
                         float* product, int32_t* rank, void* stream);

/* ---------------------------------------------------------------- encoder */

typedef struct otter_encoder otter_encoder;

/* pooling_masked_part methods — retrieval/models/tb_retriever.py:42-58 */
#define OTTER_POOL_MEAN  0
#define OTTER_POOL_SUM   1
#define OTTER_POOL_MAX   2
#define OTTER_POOL_CLS   3
#define OTTER_POOL_FIRST 4
/* att_self_pool of the ...ThreeCatAtt encoders (retrieval/models/tb_retriever.py:510-523, selected by
 * scripts/encode_corpus.py:50-66 when part_pooling starts with 'att'): softmax over positions of
 * Linear(H,1)(h) + (1 - mask) * -1e6, weighted sum of h.  Needs the tensors self_proj_tab.{weight,bias}
 * (part 2) and self_proj_psg.{weight,bias} (part 3).  'att_self_multihead' needs a module the reference
 * does not ship and is not built. */
#define OTTER_POOL_ATT_SELF 5

/* position-id conventions of the HF embedding layer */
#define OTTER_POS_BERT    0   /* position = arange(L)                                   */
#define OTTER_POS_ROBERTA 1   /* position = cumsum(ids != pad) * (ids != pad) + pad_id  */

typedef struct otter_encoder_cfg {
    int32_t vocab_size;
    int32_t hidden_size;              /* 768 */
    int32_t num_hidden_layers;        /* 12  */
    int32_t num_attention_heads;      /* 12  */
    int32_t intermediate_size;        /* 3072 */
    int32_t max_position_embeddings;  /* 514 roberta / 512 bert */
    int32_t type_vocab_size;          /* 1 roberta / 2 bert */
    int32_t pad_token_id;             /* 1 roberta / 0 bert */
    int32_t position_mode;            /* OTTER_POS_* */
    int32_t three_cat;                /* 1: [B, 3H] (tb_retriever.py), 0: [B, H] (retriever.py) */
    int32_t no_proj;                  /* 1: part1 = h[:,0]; 0: part1 = LN(Linear(h[:,0])) */
    int32_t part_pooling;             /* OTTER_POOL_* for part2/part3 */
    float   layer_norm_eps;           /* 1e-5 roberta / 1e-12 bert */
} otter_encoder_cfg;

/* model = Cls(bert_config, args) — scripts/encode_corpus.py:54-68 */
int otter_encoder_create(const otter_encoder_cfg* cfg, int device, otter_encoder** out);

/* load_saved(model, ckpt, exact=False) — retrieval/utils/utils.py:12-28.
 * name is the state_dict key without the leading "encoder." of the wrapper
 * (e.g. "embeddings.word_embeddings.weight", "encoder.layer.0.attention.self.query.weight",
 * "project.0.weight"); host_ptr is float32 row-major of the given shape.
 * Returns OTTER_EINVAL for an unknown name or a wrong shape. */
int otter_encoder_set_weight(otter_encoder* enc, const char* name, const float* host_ptr,
                             const int64_t* shape, int ndim);

/* model.to(device); model.eval() — scripts/encode_corpus.py:99,130.
 * Checks that every tensor was set and repacks the weights for the compute
 * type: OTTER_F32 (parity path) or OTTER_BF16 (throughput path, fp32 accumulate). */
int otter_encoder_finalize(otter_encoder* enc, int compute_dtype);

/* out = model(batch)['embed'] — retrieval/models/tb_retriever.py:256-261,172-195;
 * retrieval/models/retriever.py:340-345,270-286.
 * All inputs are device pointers to int64 [B, L] (the reference batch dict of
 * retrieval/data/encode_datasets.py:393-417); type_ids/part2_mask/part3_mask
 * may be NULL (type 0 / masks required only when three_cat = 1).
 * out: device float32 [B, D], D = 3H when three_cat else H. */
int otter_encoder_forward(otter_encoder* enc,
                          const int64_t* input_ids, const int64_t* input_mask,
                          const int64_t* type_ids, const int64_t* part2_mask,
                          const int64_t* part3_mask, int B, int L,
                          float* out, void* stream);

/* Parity hook (no counterpart call in the reference; it exposes what HF's
 * AutoModel returns as hidden_states[n_layers], tb_retriever.py:180-185):
 * runs the embedding layer and the first n_layers transformer layers
 * (0 <= n_layers <= num_hidden_layers) and writes the hidden state,
 * device float32 [B, L, H]. */
int otter_encoder_forward_hidden(otter_encoder* enc,
                                 const int64_t* input_ids, const int64_t* input_mask,
                                 const int64_t* type_ids, int B, int L, int n_layers,
                                 float* hidden, void* stream);

/* Parity hook (no counterpart call in the reference): the bf16 GEMM every projection of the
 * encoder runs on, exposed so that tests can compare it with a plain fp32 matmul.
 * C[M, N] = A[M, K] * W[N, K]^T + bias[N]   (epi 0)
 *                                 ... then GELU(erf)          (epi 1)
 *                                 ... + R[M, N]               (epi 2)
 * A, W, R, C: device bf16 row-major; bias: device float32 or NULL.
 * M and N multiples of 256, K a multiple of 64. */
int otter_gemm_bf16(int device, int epi, const void* A, const void* W, const float* bias, const void* R,
                    void* C, int64_t M, int N, int K, void* stream);

/* Parity hooks for the LayerNorm folding of the bf16 path (no counterpart call in the reference; the
 * reference's LayerNorms are torch.nn.LayerNorm inside HF's BertLayer, reached from
 * retrieval/models/tb_retriever.py:180-185).  All pointers are device pointers.
 *
 * otter_ln_fold:  Wout[n,k] = bf16(gamma[k] * W[n,k]);  s_out[n] = sum_k Wout[n,k];
 *                 c_out[n] = bias[n] + sum_k beta[k] * W[n,k]          (W float32 [N, K]; bias may be NULL)
 * otter_gemm_bf16_ln, epi 0 / 1, ln_mode 1:  C = LN(A) W^T + bias (then GELU for epi 1) computed as
 *                 rstd_m * (A Wout^T)[m,n] - rstd_m * mean_m * s_n + c_n  with W = Wout, v0 = s, v1 = c,
 *                 stat[m] = (rstd_m, -mean_m * rstd_m) float32 pairs; `bias` unused.
 *                 epi 2, ln_mode 1:  C = A W^T + bias + R and part[4 * N / 256][M] float32 pairs
 *                 (sum, sum of squares) of the stored rows over 64 columns each.
 *                 epi 2, ln_mode 2:  as mode 1 with R replaced by (R * rstd + (-mean rstd)) * v0 + v1
 *                 (v0 = gamma, v1 = beta, stat of the rows of R).
 * otter_ln_stats: stat[m] = (rstd, -mean * rstd) over N features from P partial pairs per row. */
int otter_ln_fold(int device, const float* W, const float* gamma, const float* beta, const float* bias, void* Wout,
                  float* s_out, float* c_out, int N, int K, void* stream);
int otter_gemm_bf16_ln(int device, int epi, int ln_mode, const void* A, const void* W, const float* bias, const void* R,
                       void* C, const void* stat, const float* v0, const float* v1, void* part, int64_t M, int N, int K,
                       void* stream);
int otter_ln_stats(int device, const void* part, int P, int64_t M, int N, float eps, void* stat, void* stream);

int  otter_encoder_out_dim(const otter_encoder* enc);
void otter_encoder_destroy(otter_encoder* enc);

#ifdef __cplusplus
}
#endif
#endif /* OTTER_HIP_H */
