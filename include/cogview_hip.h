/*
 * cogview_hip.h -- C ABI of libcogview_hip.so, the MI355X (gfx950) implementation of the CogView
 * training/inference hot path.
 *
 * The reference (THUDM/CogView) has no FFI of its own: its hot path reaches native code through
 * torch/apex/cuBLAS Python calls.  Each entry point below replaces one such call site; the citation
 * (file:line under the reference tree) names the Python interface whose arithmetic it implements.  The
 * Python package `cogview_amd` binds these symbols with ctypes and re-exposes the reference's own module
 * surface (mpu.*, model.GPT2Model, fp16.*, vqvae.*) on top of them; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name says "host"; nothing is allocated inside;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and is stream-ordered;
 *   - dtype codes: COGV_F16=0 (IEEE half), COGV_BF16=1, COGV_F32=2; "T" below is the 16-bit storage type;
 *   - return value: 0 = ok, 1 = bad argument (shape / alignment / dtype), 2 = launch failure,
 *     3 = unsupported combination.  No exceptions, no errno, no global state.
 *   - row-major everywhere; leading dimensions are in ELEMENTS.
 *   - dropout masks are a pure function of (seed, stream_id, element index): a counter-based generator
 *     (PCG hash of the 8-element group counter, xorshift-expanded), 16 random bits per element, keep iff
 *     bits >= round(p*65536), kept values scaled by 65536/(65536-round(p*65536)).  The backward kernels
 *     regenerate masks from the same triple, so nothing is stored (and activation-checkpoint recompute
 *     replays identical masks: reference mpu/random.py:308-310,353-355 does this by saving RNG states).
 */
#ifndef COGVIEW_HIP_H
#define COGVIEW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COGV_F16 0
#define COGV_BF16 1
#define COGV_F32 2

/* ------------------------------------------------------------------ library */
int cogv_version(void);                 /* ABI version, currently 1 */
const char* cogv_arch(void);            /* "gfx950" */

/* ------------------------------------------------------------------ GEMM (MFMA)
 * C[M,N] = epilogue(A_op[M,K] . B_op[N,K]^T), fp32 accumulation.
 *   trans_a = 0: A stored [M][K] (lda >= K);  trans_a = 1: A stored [K][M] (lda >= M)
 *   trans_b = 0: B stored [N][K] (ldb >= K);  trans_b = 1: B stored [K][N] (ldb >= N)
 * replaces F.linear at mpu/layers.py:243 (ColumnParallelLinear.forward), mpu/layers.py:319
 * (RowParallelLinear.forward), model/gpt2_modeling.py:117 (tied logits) and their autograd
 * (dgrad: trans_b=1; wgrad: trans_a=trans_b=1).
 * epilogue order: +bias -> [store pre-activation or gelu' to aux] -> GeLU | x gelu'(aux) | x aux -> dropout -> +C -> round
 * -> abs-max.   GeLU is the tanh form of mpu/sparse_transformer.py:172-176.
 */
#define COGV_EPI_BIAS 1     /* + bias[n]                                                             */
#define COGV_EPI_GELU 2     /* out = gelu(x); if aux != NULL the rounded pre-activation is stored     */
#define COGV_EPI_DGELU 4    /* out = x * gelu'(aux[m][n])                                             */
#define COGV_EPI_DROPOUT 8  /* out = dropout(out), element index m*N+n                                */
#define COGV_EPI_ABSMAX 16  /* atomicMax(*absmax, max|out|) -- feeds the Sandwich-LN scale            */
#define COGV_EPI_ACCUM 32   /* out += C (gradient accumulation / tied embedding)                      */
#define COGV_EPI_COLSUM 64  /* column sums of the (rounded) output per 128-row slab -> colsum_partial:      *
                             * the bias gradient of the layer that produced the GEMM's A operand, without  *
                             * re-reading the output (generation-3 kernel only: M, N >= 256, K % 64 == 0)  */

#define COGV_EPI_GELU_DAUX 128 /* with COGV_EPI_GELU: aux receives gelu'(pre-activation) instead of the pre-activation --  *
                                * same bytes; the backward GEMM then uses COGV_EPI_MULAUX (one multiply per element)  *
                                * where COGV_EPI_DGELU re-evaluates the sigmoid (exp2 + rcp per element)              */
#define COGV_EPI_MULAUX 256    /* out = x * aux[m][n] (autograd of gelu, mpu/sparse_transformer.py:172-179, through   *
                                * the stored derivative); exclusive with COGV_EPI_DGELU                               */

typedef struct cogv_gemm_desc {
  int dtype;            /* COGV_F16 | COGV_BF16 : type of A, B, bias, aux and (unless out_f32) C */
  int trans_a, trans_b;
  int M, N, K;
  const void* A; int lda;
  const void* B; int ldb;
  void* C; int ldc;
  int out_f32;          /* 1: C is float */
  int flags;            /* COGV_EPI_* */
  const void* bias;     /* [N] */
  void* aux; int ldaux; /* [M][ldaux] */
  float* absmax;        /* device scalar, caller zeroes it */
  float dropout_p; uint64_t seed; uint64_t stream_id;
  int splitk;           /* >1: contraction split over this many workgroups + reduce pass */
  int kernel_variant;   /* 0 = auto; 1 = generation 1 (register-staged 128x128x64); 3 = generation 2 (256x128x32 LDS-DMA ring); 9 = generation 3 (256x256x64, 8 waves ping-pong, persistent); 10 = generation 4 (256x256x64, 4 waves of 128x128, persistent: what auto picks for M, N >= 256) */
  void* workspace; size_t workspace_bytes;   /* >= cogv_gemm_workspace_bytes() when splitk > 1 */
  float* colsum_partial;                     /* COGV_EPI_COLSUM: [cogv_gemm_colsum_rows(M)][N] fp32, fully written */
  /* COGV_EPI_DROPOUT: this call computes rows [dropout_row0, dropout_row0 + M) of a larger [rows][N] tensor and draws the mask
   * of element (m, n) at linear index (dropout_row0 + m) * N + n -- a row-parallel Linear (mpu/layers.py:312-326) computed in
   * row chunks, each chunk's all-reduce running under the next chunk's GEMM, drops exactly what the whole-tensor call drops. */
  long long dropout_row0;
} cogv_gemm_desc;

int cogv_gemm(const cogv_gemm_desc* d, void* stream);
size_t cogv_gemm_workspace_bytes(const cogv_gemm_desc* d);
int cogv_gemm_pick_splitk(int M, int N, int K);
int cogv_gemm_colsum_rows(int M);            /* slabs of COGV_EPI_COLSUM partial sums: 2 * ceil(M / 256) */
/* out[n] (+)= sum over `rows` partial rows; the second half of cogv_colsum, also used after COGV_EPI_COLSUM */
int cogv_colsum_finalize(int dtype, const float* partial, int rows, int N, void* out, int accumulate, void* stream);
/* Up to 16 GEMMs of one dtype and layout (same trans_a / trans_b) in ONE persistent launch: the weight gradients
 * dW = dY^T X of the four linears of a layer -- or of several layers -- (autograd of mpu/layers.py:243,319) fill the
 * 256 CUs together where each alone would leave a partial last round.  COGV_ERR_UNSUPPORTED when a problem does not fit the 256x256x64
 * kernel (M, N >= 256, K % 64 == 0): issue them one by one then.  Split-K as in cogv_gemm, per problem. */
int cogv_gemm_grouped(const cogv_gemm_desc* descs, int count, void* stream);
/* Note: the persistent GEMM kernel distributes tiles through per-XCD atomic work queues; the library keeps their
 * counters in 4 KiB of device memory per GPU that it allocates itself on the first GEMM call (its only allocation). */
int cogv_gemm_pick_splitk_tiles(int tiles_256x256, int K);
/* Model parallelism: the persistent GEMM kernels (one workgroup per CU, each owning its CU's register file) leave `n` CUs
 * free from now on (n = 0: none, the default; n < 0: query only) -- room for the channel workgroups of an all-reduce that
 * runs concurrently with the launch: the row-parallel Linear of mpu/layers.py:312-326 computed in row chunks, chunk i's
 * reduce (mpu/mappings.py:22-31) under chunk i + 1's GEMM.  Process-wide, not thread safe; returns the previous value. */
int cogv_gemm_reserve_cus(int n);

/* Decode-step matrix-vector product with the layer's LayerNorms as prologue (M <= 8 rows, K = hidden size <= 4096,
 * K % 512 == 0; trans_a = trans_b = 0; epilogue flags BIAS | GELU | ABSMAX only; d->A is ignored):
 *     t    = gamma_post ? residual + LN_{eps (|z|max/8)^2}(z) * gamma_post + beta_post : z        (t_out, if given, receives t)
 *     x_in = LN_{eps (|t|max/8)^2}(t) * gamma + beta ;   C = epilogue(x_in . B^T)
 * i.e. mpu/sparse_transformer.py:326-341 -- `x + LN3(attn)` feeding `LN2`, or `y + LN4(mlp)` feeding the next layer's
 * `LN1` / the final LayerNorm -- fused into the GEMV that consumes it (every workgroup recomputes the few-KB vectors;
 * four Sandwich-LN launches per layer disappear from a decode step).  |t|max is taken over all M rows.  z_absmax: device
 * scalar with max|z| as z's producer published it (COGV_EPI_ABSMAX), or NULL: the kernel then takes max|z| itself (post-LN
 * form; same value, no atomics in the producer's tail) / max|t| (plain input). */
typedef struct cogv_ln_prologue {
  const void* z; const float* z_absmax;
  const void* gamma_post; const void* beta_post; const void* residual; void* t_out;
  const void* gamma; const void* beta;
  float eps;
  int stream_f32;      /* 1: residual, t_out and (without a post-LN) z are fp32 rows -- the fp32 residual stream (see
                          cogv_sandwich_ln_fwd); t is then formed and normalised without intermediate roundings */
} cogv_ln_prologue;
int cogv_gemv_ln(const cogv_gemm_desc* d, const cogv_ln_prologue* ln, void* stream);
/* Decode step, attention-output projection (mpu/sparse_transformer.py:163-166 after the attention of :652-673): the M <= 8
 * row matrix-vector product C = epilogue(att . B^T) whose input att [M][heads * 64] is COMBINED inside the kernel from the
 * split partials cogv_attention_decode left in its workspace (skip_combine = 1; `capacity` as in that call) -- one launch
 * per layer less than combine + GEMV.  d->A is ignored; K = heads * 64, K % 512 == 0; flags BIAS | ABSMAX only. */
int cogv_gemv_attn(const cogv_gemm_desc* d, const void* partials, int heads, int capacity, void* stream);

/* ------------------------------------------------------------------ Sandwich-LN
 * y = [residual +] LayerNorm_{eps*(amax/8)^2}(x) * gamma + beta ; amax = *absmax_in (NULL: plain LN).
 * replaces mpu/sparse_transformer.py:40-44 (LayerNorm = FusedLayerNorm(x / (x.abs().max()/8))) and the
 * residual adds at :329 and :340.  mean/rstd ([rows] fp32, may be NULL) are saved for backward.
 * absmax_out (may be NULL): atomicMax of |y| -- the next LayerNorm's scale.
 * stream_mode -- which tensors are the fp32 RESIDUAL STREAM of the transformer (the hidden state that runs through the
 * layer loop mpu/sparse_transformer.py:571-613 and its gradient; kept in fp32 so that depth does not cost precision):
 *   COGV_LN_ALL_T       every tensor in the storage type T (stand-alone LayerNorm; with a residual the LN output is
 *                       rounded to T before the add, the reference's rounding point at :326-329)
 *   COGV_LN_STREAM_IN   x (backward: x, add_in, dx) fp32, y (backward: dy) T -- LN1, LN2, final LN; no residual
 *   COGV_LN_STREAM_OUT  x (backward: x, dx) T, residual and y (backward: dy) fp32 -- LN3, LN4; residual required,
 *                       y = residual + LN(x) evaluated in fp32 without an intermediate rounding
 */
#define COGV_LN_ALL_T 0
#define COGV_LN_STREAM_IN 1
#define COGV_LN_STREAM_OUT 2
int cogv_sandwich_ln_fwd(int dtype, const void* x, const void* gamma, const void* beta, const void* residual,
                         void* y, float* mean, float* rstd, const float* absmax_in, float* absmax_out,
                         int rows, int h, float eps, int stream_mode, void* stream);
/* dx = [add_in +] dropout_mask( LN'(dy) ) ; dgamma/dbeta/colsum (T, [h], any may be NULL) get the column
 * reductions (colsum = column sums of the written dx: the bias gradient of the Linear that produced x). */
int cogv_sandwich_ln_bwd(int dtype, const void* dy, const void* x, const void* gamma, const float* mean,
                         const float* rstd, const void* add_in, void* dx, void* dgamma, void* dbeta, void* colsum,
                         int accumulate_param_grads, int rows, int h, float dropout_p, uint64_t seed,
                         uint64_t stream_id, void* workspace, size_t workspace_bytes, int stream_mode, void* stream);
/* The same with the dropout mask READ FROM x (round 6; COGV_LN_ALL_T / COGV_LN_STREAM_OUT: x is 16-bit): cogv_gemm's dropout
 * epilogue (COGV_EPI_DROPOUT) writes every dropped element as -0.0 and never writes a kept element as -0.0 ("marked zeros";
 * numerically the reference's dropout(x), mpu/sparse_transformer.py:163-168,231-233), so "16-bit pattern 0x8000" IS the forward
 * mask and the backward of the Sandwich-LN that follows that GEMM (mpu/sparse_transformer.py:326-329,337-340) needs neither
 * the generator nor stored bits.  dropout_p only supplies the scale 1 / (1 - p) (same 16-bit threshold arithmetic as the
 * replaying form; 0 = no dropout).  Bit-identical to cogv_sandwich_ln_bwd with the producing GEMM's (p, seed, stream). */
int cogv_sandwich_ln_bwd_marked(int dtype, const void* dy, const void* x, const void* gamma, const float* mean,
                                const float* rstd, const void* add_in, void* dx, void* dgamma, void* dbeta, void* colsum,
                                int accumulate_param_grads, int rows, int h, float dropout_p, void* workspace,
                                size_t workspace_bytes, int stream_mode, void* stream);
/* LN2' and LN3' of a transformer layer in ONE pass over their rows (round 6).  In backward the two Sandwich-LNs are neighbours
 * with no GEMM between them (mpu/sparse_transformer.py:326-341 read backwards: y = x + LN3(ao) feeds LN2 and the second
 * residual add):
 *     dy   = dout + LN2'(dc ; y)        dc [rows][h] T, y / dout / dy fp32 (the residual stream and its gradient)
 *     d_ao = mask( LN3'(dy ; ao) )      ao / d_ao T; dropout_p > 0: ao carries cogv_gemm's marked zeros (see above), 0: no mask
 * dy still goes to memory (LN1' adds it later) but is not read back: 18 instead of 22 bytes per element for the pair.  dy and
 * d_ao are bit-identical to cogv_sandwich_ln_bwd(STREAM_IN, add_in = dout) followed by cogv_sandwich_ln_bwd_marked(STREAM_OUT);
 * the five column reductions (dgamma2, dbeta2, dgamma3, dbeta3, colsum of d_ao; T, [h], any may be NULL) equal theirs up to the
 * fp32 summation order.  h >= 2048 (narrower rows: COGV_ERR_UNSUPPORTED, issue the two launches). */
int cogv_sandwich_ln_bwd_pair(int dtype, const void* dc, const void* y, const void* gamma2, const float* mean2,
                              const float* rstd2, const void* dout, void* dy, void* dgamma2, void* dbeta2,
                              const void* ao, const void* gamma3, const float* mean3, const float* rstd3, void* d_ao,
                              void* dgamma3, void* dbeta3, void* colsum, int accumulate_param_grads, int rows, int h,
                              float dropout_p, void* workspace, size_t workspace_bytes, void* stream);
size_t cogv_ln_bwd_pair_workspace_bytes(int rows, int h);
size_t cogv_ln_bwd_workspace_bytes(int rows, int h);
int cogv_ln_bwd_num_blocks(int rows);   /* upper bound of the backward kernel's workgroup count (workspace sizing) */

/* ------------------------------------------------------------------ attention (head dim 64)
 * replaces standard_attention, mpu/sparse_transformer.py:652-673, plus the head permutes at :112-120,:159.
 * Tensor element (b, row, head, d) lives at base + b*bs + row*rs + head*64 + d.
 * Mask: key j visible to query i iff j <= i + (s_k - s_q) or j < sep + (s_k - s_q) (sep = 0: causal);
 * masked scores are exactly -10000 as in the reference.  lse: [B][H][s_q] fp32; dvec: backward workspace
 * [2][B][H][s_q] 4-byte words (softmax-backward row term D, then the per-row dropout keys), written by the dQ kernel.
 */
typedef struct cogv_attn_desc {
  int dtype; int B, H, s_q, s_k, head_dim; int sep;
  float scale;                       /* 1/sqrt(head_dim), applied to QK^T */
  float dropout_p; uint64_t seed; uint64_t stream_id;
  const void* q; const void* k; const void* v; void* o;
  const void* dout; void* dq; void* dk; void* dv;
  float* lse; float* dvec;
  long long q_bs, k_bs, v_bs, o_bs, do_bs, dq_bs, dk_bs, dv_bs;
  int q_rs, k_rs, v_rs, o_rs, do_rs, dq_rs, dk_rs, dv_rs;
  /* backward only, optional: per-workgroup column sums of the stored dq | dk | dv (the bias gradient of the fused
   * QKV projection, mpu/sparse_transformer.py:101-110) -> colsum_partial[B * ceil(s/128)][3 * H * 64] fp32, columns
   * ordered [q heads | k heads | v heads]; requires s_q == s_k; finish with cogv_colsum_finalize. */
  float* colsum_partial;
  /* optional (forward only unless sparse_window > 0): gathered keys (sparse_attention_inference, mpu/sparse_transformer.py:727-750): key slot j of
   * batch b is row kv_index[b * kv_index_bs + j] of k and v; s_k is the number of slots (<= 4096).  The left-to-right
   * rule applies to SLOTS: the last s_q slots are the queries' own positions.  Bit 31 of an entry marks a masked slot
   * (score -10000): a decode step over a fixed-capacity key/value cache flags the slots it has not written yet, so the
   * launch parameters stay constant from step to step (HIP-graph replay, cogview_amd/generation/decoder.py). */
  const int* kv_index; long long kv_index_bs;
  /* sparse TRAINING form (sparse_attention, mpu/sparse_transformer.py:675-725) in "slot space": sparse_window > 0 (the
   * reference's query_window, a multiple of 128 dividing s_q).  Each block of sparse_window queries has its own index
   * row (kv_index_gs entries apart) of s_k slots: first sparse_pivots pivot slots, then the key_window_times *
   * sparse_window window slots ending with the block's own positions.  Bit 31 of an entry marks a masked slot (an
   * invisible pivot, front padding): score -10000.  Pivot slots add sparse_pivot_bias = log(s // n_pivots) to the
   * scaled score; the left-to-right rule applies to the last sparse_window slots. */
  long long kv_index_gs; int sparse_window, sparse_pivots; float sparse_pivot_bias;
  /* optional (dense form, dropout_p > 0): cogv_attention_keep_bits_bytes(B, H, s_q, s_k) bytes, 4-byte aligned.  The forward
   * call WRITES the keep decisions of its attention dropout there (1 bit per score the kernel evaluated: the reference's
   * dropout mask at mpu/sparse_transformer.py:667-669, which autograd keeps for the backward pass); the backward call given
   * the same buffer READS them instead of regenerating the draws.  Results are bit-identical with and without it.  NULL:
   * backward regenerates (also for the gathered / sparse forms, which ignore the field). */
  void* keep_bits;
  /* optional (dense form only; excludes kv_index and keep_bits): an ARBITRARY mask tensor M in the storage type, [B][s_q][s_k]
   * (mask_bs = s_q * s_k) or [s_q][s_k] shared by the batch (mask_bs = 0), applied exactly as the reference does for any
   * real M -- scaled score * M - 10000 * (1 - M), mpu/sparse_transformer.py:661-663 -- in forward and backward; `sep` is ignored
   * (no left-to-right rule: the tensor decides).  The left-to-right masks the reference's callers build go through `sep`
   * instead (causal block skipping); this is the general path of the module surface, not a hot path. */
  const void* mask; long long mask_bs;
} cogv_attn_desc;
int cogv_attention_fwd(const cogv_attn_desc* d, void* stream);
int cogv_attention_bwd(const cogv_attn_desc* d, void* stream);
size_t cogv_attention_keep_bits_bytes(int B, int H, int s_q, int s_k);
/* Decode step (generation/sampling.py:139-148: one model call per generated token; mems mpu/sparse_transformer.py:526-546):
 * ONE query token per batch row against a fixed-capacity key/value cache.  qkv: the QKV projection of the new token,
 * [B][3 * H * 64] = q | k | v (qkv_bs elements between batch rows); cache: [B][capacity][2 * H * 64] keys | values
 * (cache_bs / cache_rs: batch / slot strides in elements).  *pos (device int64) = slot of the new token = number of
 * slots already valid: the kernel attends slots [0, *pos] -- the new token's key / value are taken from qkv and WRITTEN
 * into slot *pos -- so no launch parameter depends on the length (HIP-graph replay).  out: [B][H * 64] (out_bs elements
 * between rows).  Keys are split over workgroups (capacity / 128 per head); a second launch combines the partial softmax
 * results in split order.  workspace: cogv_attention_decode_workspace_bytes() bytes (partials; no initialisation needed). */
typedef struct cogv_attn_decode_desc {
  int dtype; int B, H, capacity, head_dim;
  float scale;                       /* 1/sqrt(head_dim) */
  const void* qkv; long long qkv_bs;
  void* cache; long long cache_bs; int cache_rs;
  void* out; long long out_bs;
  const long long* pos;
  void* workspace; size_t workspace_bytes;
  int skip_combine;                  /* 1: leave the split partials in `workspace` (66 floats per (row, head, split): max,
                                        sum, 64 outputs) for cogv_gemv_attn to combine; `out` is not written and may be NULL */
} cogv_attn_decode_desc;
size_t cogv_attention_decode_workspace_bytes(int B, int H, int capacity);
int cogv_attention_decode(const cogv_attn_decode_desc* d, void* stream);
/* Sparse training form, backward: cogv_attention_bwd with sparse_window > 0 writes dq as usual, but dk / dv are
 * SLOT-SPACE buffers [B * s_q / sparse_window][s_k slots][H][64] (dk_bs / dv_bs = the stride of one (batch, query block)
 * plane) -- the gradient of each gathered copy of a key, the same quantity the reference's autograd holds for pivot_k
 * / window_k before torch.gather / the padded-window view scatter it back (mpu/sparse_transformer.py:689-705).
 * cogv_sparse_slot_reduce folds them onto the keys: dk[b][r] = sum over the window slots of r (blocks r/w ... r/w +
 * times - 1) + the pivot slot pivot_inv[b][r] (-1: r is no pivot; pivots are distinct per sample, as the reference's
 * random.sample draws them) of the blocks that see it.  fp32 sums in a fixed order (deterministic).
 * dk_slots / dv_slots: contiguous [B][s / window][n_pivots + times * window][H * 64]. */
int cogv_sparse_slot_reduce(int dtype, const void* dk_slots, const void* dv_slots, const int* pivot_inv, void* dk, void* dv,
                            long long dk_bs, int dk_rs, long long dv_bs, int dv_rs, int B, int s, int H, int window,
                            int times, int n_pivots, void* stream);

/* ------------------------------------------------------------------ embedding
 * out = dropout( table[ids - vocab_start] (0 outside the shard) [+ pos_table[pos_ids]] ), abs-max of out.
 * replaces VocabParallelEmbedding.forward mpu/layers.py:117-133 and the position add + dropout at
 * mpu/sparse_transformer.py:522-524.  ids == NULL: the word part is read from x_in (model-parallel path,
 * after the all-reduce).  Backward: dtable[id] += the fp32 sum, in ascending token order, of the (dropout-masked)
 * gradient rows of the tokens carrying id, rounded once (torch's embedding_dense_backward under
 * mpu/layers.py:117-133 accumulates in fp32 too); dpos likewise per position id.  Deterministic: no floating-point
 * atomics.  workspace: cogv_embedding_bwd_workspace_bytes(vocab_end - vocab_start (0 without dtable), n_pos (0 without
 * dpos)) bytes of scratch, 16-byte aligned (cleared by the call itself).
 */
int cogv_embedding_fwd(int dtype, const int64_t* ids, const void* table, int64_t vocab_start, int64_t vocab_end,
                       const void* x_in, const int64_t* pos_ids, const void* pos_table, int64_t n_pos, void* out,
                       float* absmax_out, int64_t n_tok, int h, float dropout_p, uint64_t seed, uint64_t stream_id,
                       int out_f32, void* stream);
int cogv_embedding_bwd(int dtype, const void* dout, const int64_t* ids, void* dtable, int64_t vocab_start,
                       int64_t vocab_end, const int64_t* pos_ids, void* dpos, int64_t n_pos, void* dx,
                       int64_t n_tok, int h, float dropout_p, uint64_t seed, uint64_t stream_id, void* workspace,
                       size_t workspace_bytes, int dout_f32, void* stream);
size_t cogv_embedding_bwd_workspace_bytes(int64_t table_rows, int64_t n_pos);

/* ------------------------------------------------------------------ element-wise (n % 8 == 0)
 * gelu: mpu/sparse_transformer.py:172-179; dropout: torch.nn.Dropout call sites :167,:233,:524 */
int cogv_gelu_fwd(int dtype, const void* x, void* y, size_t n, void* stream);
int cogv_gelu_bwd(int dtype, const void* dy, const void* x, void* dx, size_t n, void* stream);
int cogv_dropout(int dtype, const void* x, void* y, size_t n, float p, uint64_t seed, uint64_t stream_id,
                 float* absmax_out, void* stream);
int cogv_add(int dtype, const void* a, const void* b, void* out, size_t n, float* absmax_out, void* stream);
int cogv_scale(int dtype, const void* x, void* y, size_t n, float scale, void* stream);
int cogv_absmax(int dtype, const void* x, size_t n, float* out, void* stream);      /* atomicMax into *out; dtype may be COGV_F32 (n % 4 == 0) */
/* out(fp32) = a(fp32) + b(T): a branch output joins the fp32 residual stream (the residual adds of
 * mpu/sparse_transformer.py:329,:340 when the layer is composed op by op); abs-max of out like cogv_add */
int cogv_add_stream(int dtype, const float* a, const void* b, float* out, size_t n, float* absmax_out, void* stream);
/* out[n] (+)= sum_m dy[m][n]  -- bias gradients of the Linear layers */
int cogv_colsum(int dtype, const void* dy, int M, int N, int ld, void* out, int accumulate, void* workspace,
                size_t workspace_bytes, void* stream);
size_t cogv_colsum_workspace_bytes(int M, int N);

/* ------------------------------------------------------------------ vocab-parallel cross entropy
 * replaces _VocabParallelCrossEntropy, mpu/cross_entropy.py:25-104.  logits [rows][V_local] in
 * logits_dtype (F16/BF16/F32), target int64 global ids, shard = [vocab_start, vocab_start+V_local).
 * fwd writes per-row shard statistics: rowmax, sumexp (relative to rowmax), predicted logit (0 when the
 * target is outside the shard) and, for a single shard, loss = log(sumexp) + rowmax - predicted.
 * bwd writes dlogits = (exp(logit - gmax)/gsum - onehot) * grad[row] in logits_dtype (may alias logits).
 */
int cogv_ce_fwd(int logits_dtype, const void* logits, const int64_t* target, int64_t vocab_start, int rows,
                int v_local, float* rowmax, float* sumexp, float* predicted, float* loss, void* stream);
int cogv_ce_bwd(int logits_dtype, const void* logits, const int64_t* target, int64_t vocab_start, int rows,
                int v_local, const float* gmax, const float* gsum, const float* grad, void* dlogits, void* stream);

/* ------------------------------------------------------------------ optimizer (flat buffers, chunk table)
 * One launch over the whole flat parameter space replaces FP16_Optimizer's per-tensor passes
 * (fp16/fp16.py:399-453,556-567), the per-tensor overflow check (fp16/loss_scaler.py:107-145), the
 * per-tensor norm of mpu/grads.py:59-73 and apex FusedAdam (call site pretrain_gpt2.py:139-140).
 * Chunk c covers flat elements [chunk_start[c], chunk_start[c]+chunk_len[c]), belongs to hyper-parameter
 * group chunk_group[c] (< 8) and counts toward the norm iff chunk_norm[c] != 0 (model-parallel dedup of
 * mpu/grads.py:61).  chunk_start % 8 == 0.
 */
/* stats[0] += sum of squares of grads in counted chunks (double; per-workgroup partials summed in a fixed order: the same
 * bits run after run), stats[1] = 1.0 if any grad is inf/nan.  workspace: cogv_grad_stats_workspace_bytes() bytes of scratch
 * owned by the caller (8-byte aligned, no initialisation; one per stream that may run the pass concurrently).  dtype may be
 * COGV_F32: the fp32 master gradients of FP16_Optimizer's generic path / of an fp32 model (mpu/grads.py:62-84). */
int cogv_grad_stats(int dtype, const void* grads, const int64_t* chunk_start, const int32_t* chunk_len,
                    const uint8_t* chunk_norm, int nchunks, double* stats, void* workspace, size_t workspace_bytes,
                    void* stream);
size_t cogv_grad_stats_workspace_bytes(void);
typedef struct cogv_adam_desc {
  int dtype;                       /* dtype of model params / grads (F16|BF16) */
  void* params; const void* grads; /* flat model params (written) and their (scaled) grads */
  float* master; float* exp_avg; float* exp_avg_sq;   /* flat fp32 */
  const int64_t* chunk_start; const int32_t* chunk_len; const uint8_t* chunk_group; int nchunks;
  float lr[8]; float weight_decay[8];
  float beta1, beta2, eps; int step; int bias_correction; int adam_w_mode;
  double beta1_d, beta2_d;         /* the same betas in double (1-beta and beta^step are formed in double) */
  float inv_loss_scale;            /* grads are multiplied by this */
  float max_grad_norm;             /* > 0: clip by global norm computed from stats */
  const double* stats;             /* from cogv_grad_stats (device); stats[1] != 0 => whole step skipped */
  const double* norm_sumsq_override; /* optional device scalar with the MP-reduced sum of squares */
} cogv_adam_desc;
int cogv_adamw_step(const cogv_adam_desc* d, void* stream);
/* master[i] = (float)params[i]  /  params[i] = (T)master[i]  over the chunk table */
int cogv_cast_flat(int dtype, const void* src_half, float* dst_f32, size_t n, void* stream);
int cogv_cast_flat_back(int dtype, const float* src_f32, void* dst_half, size_t n, void* stream);

/* ------------------------------------------------------------------ VQ-VAE tokenizer (fp32, exact-fp32 MFMA)
 * replaces the conv stacks of vqvae/vqvae_zc.py:121-129,159-164 (Encoder) and :172-192 (Decoder), the
 * nearest-code search :41-54 and embed_code :95-96 for the production config of vqvae/api.py:12-20.
 * Activations NHWC fp32; weights repacked by the caller to [parity][Cout][tap][Cin]:
 *   COGV_CONV_4X4_S2  (Conv2d k4 s2 p1)        taps = ky*4+kx, 1 parity,  W_packed[co][ky*4+kx][ci] = W[co][ci][ky][kx]
 *   COGV_CONV_1X1                              1 tap
 *   COGV_CONV_3X3_S1  (Conv2d k3 s1 p1)        taps = ky*3+kx, 1 parity,  W_packed[co][ky*3+kx][ci] = W[co][ci][ky][kx]
 *                                              (the non-production encoders :147, :154 and ResBlock :105)
 *   COGV_CONVT_4X4_S2 (ConvTranspose2d k4 s2 p1) 4 parities z = py*2+px (output pixel (2y+py, 2x+px)), 4 taps
 *        W_packed[z][co][ty*2+tx][ci] = W[ci][co][ky][kx],  ky = (py ? 2*ty : 1+2*ty),  kx = (px ? 2*tx : 1+2*tx)
 */
#define COGV_CONV_4X4_S2 0
#define COGV_CONV_1X1 1
#define COGV_CONVT_4X4_S2 2
#define COGV_CONV_3X3_S1 3
typedef struct cogv_conv_desc {
  int kind; int B, IH, IW, Cin, Cout; int relu;     /* relu: applied to the output */
  const void* in; const void* w; const void* bias; void* out;
  /* optional: the decoder's final 1x1 convolution to RGB (vqvae/vqvae_zc.py:190) fused into this layer's epilogue.
   * rgb_w = its weights [3][Cout] (fp32); rgb_partial receives [Cout / 128][B * OH * OW][4] fp32 partial sums INSTEAD
   * of the output tensor (out may be NULL; relu must be set; Cout % 128 == 0); finish with cogv_rgb_finalize_f32. */
  const void* rgb_w; void* rgb_partial;
  /* ResBlock support (vqvae/vqvae_zc.py:99-114: ReLU, conv3x3, ReLU, conv1x1, `out += input`): relu_in applies ReLU to the INPUT
   * activations inside the kernel (the block's leading ReLU; also the ReLU that follows the last block, :159); residual
   * (same NHWC shape as out, 16-byte aligned, may be NULL) is added after bias / ReLU, through max(., 0) when relu_residual is
   * set -- the reference's leading nn.ReLU(inplace=True) overwrites the block's input, so what it adds back is relu(input). */
  int relu_in; const void* residual; int relu_residual;
} cogv_conv_desc;
int cogv_conv2d_nhwc_f32(const cogv_conv_desc* d, void* stream);
/* out NCHW [B,3,H,W] = (sum over the ntiles partial planes + bias[c]) * scale[c] + shift[c] (scale/shift: 3 HOST floats or NULL) */
int cogv_rgb_finalize_f32(const float* partial, int ntiles, const float* bias, float* out, int B, int H, int W,
                          const float* scale3_host, const float* shift3_host, void* stream);
/* ids[m] = argmin_j (|x_m|^2 - 2 x_m.E_j) + |E_j|^2, first minimum on ties; embed_t = E^T [n_embed][D], embed_sq = |E_j|^2 */
int cogv_vq_argmin_f32(const float* x, const float* embed_t, const float* embed_sq, int64_t* ids, int M, int D,
                       int n_embed, void* stream);
int cogv_nchw3_to_nhwc4_f32(const float* in, float* out, int B, int H, int W, void* stream);
int cogv_embed_code_f32(const int64_t* ids, const float* embed_t, float* out, int64_t npix, int D, int n_embed,
                        void* stream);
/* NHWC [B,H,W,Cin] -> NCHW [B,3,H,W]: (w[3][Cin] . x + bias) * scale + shift (scale/shift: 3 HOST floats or NULL) */
int cogv_conv1x1_to_rgb_f32(const float* in, const float* w, const float* bias, float* out, int B, int H, int W,
                            int Cin, const float* scale3_host, const float* shift3_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COGVIEW_HIP_H */
