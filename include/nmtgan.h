/* nmtgan.h - C ABI of libnmtgan_hip.so: the MI355X (gfx950) rollout-reward path of NMT_GAN.
 *
 * The reference (ZhenYangIACAS/NMT_GAN) has no FFI/plugin interface: its boundary is the Python
 * method surface gan_train.py calls (Model.get_reward model.py:291, Model.generate_step
 * model.py:268, the DisCNN feed/fetch of model.py:311-313).  nmt_gan_amd/ keeps that surface and
 * binds the entry points below with ctypes; every entry point cites the reference code it
 * replaces (paths relative to the reference root).
 *
 * Conventions
 *  - all pointers are BORROWED device pointers unless a field says "host"; nothing is retained
 *    past the call, nothing is freed.  The caller owns the workspace.
 *  - every call enqueues asynchronously on `stream` (a hipStream_t passed as void*) and never
 *    synchronises; no hipMalloc/hipFree inside.
 *  - return 0 on success, a negative code on error; nmtgan_last_error() gives the thread-local
 *    message.  Shapes are validated on the host before any launch.
 *  - state kept by the library is thread-local and set only for the duration of a call (error text, the
 *    workgroup cap and the split-K scratch the pipelines set around their own GEMM launches) plus one event
 *    pool per device for the two-stream schedule: calls from different threads / on different streams do not
 *    interfere, as long as concurrent calls use different workspaces.
 *  - "op dtype" = the MFMA operand / KV-cache / activation-copy element type of a packed model:
 *    NMTGAN_F16 (IEEE half operands, fp32 accumulate: the throughput tier - 2^-12 operand rounding, values
 *    saturate at +-65504), NMTGAN_BF16 (bf16 operands at the same MFMA rate, 2^-9 operand rounding, fp32 range)
 *    or NMTGAN_F32 (exact fp32 MFMA; the 1e-5 parity tier).  Residual stream, LayerNorm, softmax, CDF are
 *    fp32/fp64 always.
 *  - weight matrices are stored TRANSPOSED, Wt[N][K] K-contiguous (TF stores kernel[K][N]); the
 *    Python facade repacks once at load time.
 */
#ifndef NMTGAN_H_
#define NMTGAN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMTGAN_VERSION 1
#define NMTGAN_BF16 0
#define NMTGAN_F32 1
#define NMTGAN_F16 2
#define NMTGAN_MAX_BLOCKS 16
#define NMTGAN_MAX_WIDTHS 32

int nmtgan_version(void);
const char* nmtgan_last_error(void);

/* ------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = A[M,K] . Wt[N,K]^T  (+ epilogue).  Replaces tf.layers.conv2d 1x1 (= dense+bias,
 * common_layers.py:284-290; call sites common_attention.py:372-387,397, common_layers.py:1093-1106),
 * tf.matmul(.,E^T) (model.py:811/818), tf.nn.conv2d VALID over time (cnn_discriminator.py:94-99)
 * and the highway/linear matmuls (cnn_discriminator.py:35-48).
 * A row r lives at A + ((r / a_group) * a_group_stride + (r % a_group) * lda) elements; a_group<=0
 * means plain rows (r*lda).  For the discriminator conv of width w: a_group = T-w+1, lda = E,
 * a_group_stride = T*E, K = w*E (a window is a contiguous slice of the [T,E] sentence image).
 * epilogue NMTGAN_EPI_LINEAR : v = (acc + bias[n]) * seg_scale[s]; relu optional; column segment
 *      s = n / seg_width goes to out[s] (row stride ldo[s], dtype out_f32[s] ? fp32 : op dtype).
 *      out_f32[s] == 2: the segment is written as the split-precision operand image nmtgan_split3 makes of it,
 *      fp16 [M][3 W] = [hi | lo | hi] with W = the segment's width and ldo[s] = 3 W (N % 64 == 0, 16-byte rows).
 * epilogue NMTGAN_EPI_CONVMAX: v = relu(acc*scale[n] + shift[n]) (conv bias + inference BN folded,
 *      cnn_discriminator.py:100-105,22-33); pooled[(r / a_group)*ld_pool + n] = max(., v) - the
 *      max over time of cnn_discriminator.py:469.  pooled must be >= 0 on entry (memset 0).
 * epilogue NMTGAN_EPI_ADDLN  : ln_x <- LayerNorm(ln_x + acc + bias) * ln_gamma + ln_beta (eps 1e-6, biased
 *      variance), also written to ln_out in the op dtype when ln_out != NULL: the residual() of model.py:725-739
 *      fused into the projection that feeds it.  16-bit op dtypes, N = 512, plain rows, K*esize % 64 == 0.
 *      ln_x is [M][512] fp32, read and rewritten in place; every row's result depends on that row alone.
 */
#define NMTGAN_EPI_LINEAR 0
#define NMTGAN_EPI_CONVMAX 1
#define NMTGAN_EPI_ADDLN 2
typedef struct {
  int op_dtype;              /* NMTGAN_BF16 / NMTGAN_F32 for A and Wt */
  int epilogue;
  const void* A;
  const void* Wt;            /* [N][K] */
  int M, N, K;
  int64_t lda;
  int a_group;
  int64_t a_group_stride;
  /* LINEAR */
  const float* bias;         /* [N] or NULL */
  int relu;
  int n_seg;                 /* 1..3 */
  int seg_width;             /* columns per segment (multiple of 32 when n_seg>1) */
  void* out[3];
  int64_t ldo[3];
  int out_f32[3];
  float seg_scale[3];
  /* CONVMAX */
  const float* scale;        /* [N] */
  const float* shift;        /* [N] */
  float* pooled;
  int64_t ld_pool;
  int64_t ldw;               /* elements per Wt row; 0 = K.  > K selects a column slice of a wider matrix (the taps
                                [k, w) of a conv filter bank: Wt + k*E, K = (w-k)*E, ldw = w*E) */
  /* ADDLN (bias as in LINEAR) */
  float* ln_x;               /* [M][N] fp32, in and out */
  const float* ln_gamma;     /* [N] */
  const float* ln_beta;      /* [N] */
  void* ln_out;              /* [M][N] op dtype or NULL */
  /* CONVMAX with an additive term (optional, add != NULL): v = relu((acc + add[(add_blk[r / a_group] * add_rows +
   * add_row0 + r % a_group) * N + n]) * scale[n] + shift[n]).  The part of a conv window's sum that does not depend on
   * the rollout - its leading taps lie in the given prefix, equal to y_sample's - comes from a table computed once per
   * get_reward; add is [blocks][add_rows][N] fp32, add_blk[M / a_group] int32 names the block of each sentence.
   * Needs K*esize % 128 == 0 and >= 512 bytes. */
  const float* add; const int32_t* add_blk; int add_rows; int add_row0;
} nmtgan_gemm_args;
int nmtgan_gemm(const nmtgan_gemm_args* a, void* stream);
/* n independent problems in as few launches as possible: one persistent kernel walks the tiles of up to 24 problems
 * (longest k-loop first), so problems with fewer tiles than the chip has CUs - the per-width, per-row-range conv
 * products of the discriminator, cnn_discriminator.py:461-477 - fill it together.  All problems share op dtype and
 * epilogue: CONVMAX (with or without the additive term) or LINEAR with n_seg = 1, an fp32 or op-dtype output (out_f32
 * 1 or 0) and seg_scale 1.
 * Results are those of n nmtgan_gemm calls (same k order per output element). */
int nmtgan_gemm_grouped(const nmtgan_gemm_args* a, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Embedding: out[r] = mult * table[tok(r)] + timing[pos(r)]   (fp32 and op-dtype copies).
 * Replaces tf.gather + scale + add_timing_signal_1d (model.py:777-801, 506, 565;
 * common_attention.py:55-68).  Decoder passes mult = 2*sqrt(d) (model.py:565 adds x twice).
 * tok(r) = tok[r*tok_ld + (tok_col ? tok_col[r] : 0) + tok_col_add]
 * pos(r) = (pos ? pos[r] : r % pos_mod) + pos_add
 */
typedef struct {
  const float* table;        /* [V][d] fp32 */
  const float* timing;       /* [max_pos][d] fp32 */
  int vocab, max_pos;
  const int32_t* tok; int64_t tok_ld; const int32_t* tok_col; int tok_col_add;
  const int32_t* pos; int pos_mod; int pos_add;
  float mult;
  int M, d;
  float* out_f32;            /* [M][d] */
  void* out_op;              /* [M][d] op dtype, may be NULL */
  int op_dtype;
  void* out_split;           /* optional fp16 [M][3d]: the output as a split-precision GEMM operand (nmtgan_split3) */
} nmtgan_embed_args;
int nmtgan_embed(const nmtgan_embed_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * out = LayerNorm(x + y) * gamma + beta, eps 1e-6, biased variance
 * (model.py:725-739 residual(); common_layers.py:336-341,359-374).  Optional zero_row[r] != 0
 * zeroes the output row (encoder padding, model.py:542).
 */
typedef struct {
  const float* x; const float* y;      /* [M][d] fp32; y may be NULL */
  const float* gamma; const float* beta;
  int M, d;
  float* out_f32; void* out_op; int op_dtype;
  const uint8_t* zero_row;
  void* out_split;           /* optional fp16 [M][3d]: the output as a split-precision GEMM operand (nmtgan_split3) */
} nmtgan_addln_args;
int nmtgan_add_layernorm(const nmtgan_addln_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Split-precision GEMM operand: A [M][K] fp32 (row stride lda) -> out [M][3K] fp16 = [A_hi | A_lo | A_hi] with
 * A_hi = fp16(A), A_lo = fp16(A - A_hi).  Against weights packed [N][3K] = [W_hi | W_hi | W_lo] one fp16 nmtgan_gemm over
 * 3K computes A_hi.W_hi + A_lo.W_hi + A_hi.W_lo: tf.matmul / conv2d 1x1 in fp32 (same call sites as nmtgan_gemm) to about
 * 21 significand bits per operand at the fp16 matrix-core rate.  What the generator pipelines do per GEMM when
 * nmtgan_generator.gemm_split is set.  K % 8 == 0, lda % 4 == 0, 16-byte aligned pointers.
 */
int nmtgan_split3(const float* A, int64_t lda, int M, int K, void* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row attention: one query row against (1) a block of shared keys and (2) its private suffix.
 * Replaces dot_product_attention (common_attention.py:294-331) for: encoder self-attention
 * (padding mask), decoder causal self-attention (prefill and the incremental rollout step),
 * decoder-encoder attention.  q is already scaled by dk^-0.5 (common_attention.py:393).
 * Segment 1: keys j in [0, cnt1(r)) at k1 + (blk(r)*len1 + j)*d, skipped where
 *            mask1[blk(r)*len1 + j] != 0 (the -1e9 padding bias, common_attention.py:192-202).
 *            blk(r) = seg1_blk ? seg1_blk[r] : r / rows_per_blk
 *            cnt1(r) = seg1_cnt ? seg1_cnt[r] : (causal ? r % rows_per_blk + 1 : len1)
 *            (the causal form is the lower-triangle bias, common_attention.py:178-189)
 * Segment 2: keys s in [0, n_steps2) at k2 + (step_off[s] + r)*d  (step-major private cache).
 */
typedef struct {
  const void* q;             /* [M][d] fp32 (or kv_dtype, see q_op_dtype) */
  int M, d, heads;
  int kv_dtype;              /* dtype of k1,v1,k2,v2 and out */
  const void* k1; const void* v1; int len1;
  const int32_t* seg1_blk; const int32_t* seg1_cnt; int rows_per_blk; int causal;
  const uint8_t* mask1;
  const void* k2; const void* v2; const int64_t* step_off; int n_steps2;
  void* out;                 /* [M][d] kv_dtype */
  /* optional running softmax state to continue from (written by nmtgan_attn_group in partial mode):
   * un-normalised o_init[M][d], m_init[M][heads], l_init[M][heads]; all three or none */
  const float* o_init; const float* m_init; const float* l_init;
  int q_op_dtype;            /* != 0: q is [M][d] in kv_dtype (as written by a GEMM output segment) instead of fp32 */
  int out_split;             /* kv_dtype NMTGAN_F32 only, != 0: out is fp16 [M][3d], the split-precision operand image
                              * of the fp32 result (what nmtgan_split3 would make of it) */
} nmtgan_attn_args;
int nmtgan_attn_rows(const nmtgan_attn_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Group attention (16-bit op dtypes only): the rows_per_group (<= 32) consecutive query rows of a group share one
 * block of keys (the rollouts of one (sentence, prefix) pair share the teacher-forced prefix and the
 * encoder memory), so K/V are loaded once per group and the products run on MFMA.  Same math as
 * nmtgan_attn_rows segment 1 (common_attention.py:294-331).  k[nblk][len][d] row-major,
 * vt[nblk][d][len_padded] TRANSPOSED values (nmtgan_transpose_v), len_padded a multiple of 32.
 * row_blk / row_cnt are per-row arrays read at the first row of each group (row_cnt NULL -> len).
 * Exactly one output mode: `out` (normalised, bf16 [M][d]) or the running state (o_part un-normalised
 * fp32 [M][d], m_part/l_part [M][heads]) that nmtgan_attn_rows continues over private keys.
 */
typedef struct {
  const void* q; int M, d, heads, rows_per_group;   /* q fp32, or kv_dtype when q_bf16 != 0 */
  const void* k; const void* vt; int len, len_padded;
  const int32_t* row_blk; const int32_t* row_cnt; const uint8_t* mask;
  void* out;
  float* o_part; float* m_part; float* l_part;
  int q_bf16;
  /* Slot form (optional): one workgroup serves `slots` groups that share ONE block of keys (the rollout
   * groups (b, g), (b, g+1), .. of a sentence: with rollnum 16 two groups fill the 32 MFMA columns).
   * slot_table[n_entries][2 + 2*slots] int32 on the device: {blk, 0, row0[slots], cnt[slots]}; a slot is
   * served when 0 <= row0 < M, its rows are row0 .. row0+rows_per_group-1, its key count is cnt (len when
   * row_cnt is NULL).  row_blk / row_cnt are then only used as "counts present" flags. */
  const int32_t* slot_table; int slots; int n_entries;
  int kv_dtype;              /* NMTGAN_BF16 (0) or NMTGAN_F16: element type of k, vt, out and of q when q_bf16 != 0.
                              * NMTGAN_F32: k, vt, q and out are all fp32 (q_bf16 is ignored), the products run on the
                              * fp32 MFMA, head dim 32 or 64, every pointer 16-byte aligned */
  int out_split;             /* NMTGAN_F32 with `out` only, != 0: out is fp16 [M][3d], the split-precision operand image */
} nmtgan_attn_group_args;
int nmtgan_attn_group(const nmtgan_attn_group_args* a, void* stream);
/* v[nblk*len][d] -> vt[nblk][d][len_padded], 16-bit elements (bf16 or fp16), zero padded */
int nmtgan_transpose_v(const void* v, void* vt, int nblk, int len, int len_padded, int d, void* stream);
/* the same for fp32 values (the layout nmtgan_attn_group reads with kv_dtype NMTGAN_F32) */
int nmtgan_transpose_v_f32(const float* v, float* vt, int nblk, int len, int len_padded, int d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Categorical sampler with injected uniforms.  Replaces softmax -> log -> tf.multinomial(.,1)
 * (model.py:235-241).  out[r] = min{k : c_k > u(r) * c_{V-1}}, c_k = sum_{j<=k} exp(l_j - max l)
 * accumulated in fp64 (TF CPU kernel semantics), clamped to V-1.  exp is evaluated in fp32 and
 * widened (the CPU kernel evaluates it in fp64): buckets agree except within ~1e-7 relative of a
 * CDF boundary.
 * logits row of r = row_map ? row_map[r] : r;  u(r) = u[(u_off ? u_off[r] : r) + u_add].
 * greedy != 0: out[r] = first argmax (model.py:153-154) and u is ignored.
 */
typedef struct {
  const float* logits; int64_t ld; int V; int M;
  const int32_t* row_map;
  const float* u; const int64_t* u_off; int64_t u_add;
  int greedy;
  int32_t* out;
} nmtgan_sample_args;
int nmtgan_sample_rows(const nmtgan_sample_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused logits + sampling: out[r] = sample(softmax(x[r] . w^T), u(r)) without writing the logits
 * (model.py:226-241: top() -> slice -> softmax -> log -> tf.multinomial).  x[M][d], w[V][d] in op
 * dtype; V % 64 == 0, d*esize % 128 == 0 and >= 512 bytes (else use nmtgan_gemm + nmtgan_sample_rows).
 * Same CDF semantics as nmtgan_sample_rows, accumulated per 64-entry vocabulary tile in fp64.
 */
size_t nmtgan_logits_sample_workspace_bytes(int M, int V);
int nmtgan_logits_sample(int op_dtype, const void* x, const void* w, int M, int V, int d, const float* u,
                         const int64_t* u_off, int64_t u_add, int32_t* out, void* workspace,
                         size_t workspace_bytes, void* stream);

/* nmtgan_logits_sample without a workspace argument, for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * Same kernels, same results as nmtgan_logits_sample; the scratch (per-tile statistics, selection arrays, counters,
 * work list) is a static device array of NMTGAN_LOGITS_SAMPLE_STATIC_BYTES that every GPU the library runs on holds
 * until the process ends.  The call allocates nothing; it fails with a message when
 * nmtgan_logits_sample_workspace_bytes(M, V) exceeds the array (about 8,300 rows at V = 32000).  All calls on one
 * GPU share the array: calls that overlap in time on the same GPU (different streams or threads) race.
 * The array's address is resolved on the first call on each device (a symbol lookup, which may load the code object):
 * make that first call OUTSIDE any stream capture - a first call that arrives under capture is refused with a message;
 * later calls can be captured and replayed.  `stream` must belong to the current device. */
#define NMTGAN_LOGITS_SAMPLE_STATIC_BYTES (32u << 20)
int nmtgan_logits_sample_static(int op_dtype, const void* x, const void* w, int M, int V, int d, const float* u,
                                const int64_t* u_off, int64_t u_add, int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Packed models.  All matrices Wt[N][K] in op dtype; biases / LN / BN-folded vectors fp32.
 */
typedef struct {
  const void* w; const float* b;       /* Wt[N][K], bias[N] */
} nmtgan_dense;
typedef struct { const float* g; const float* b; } nmtgan_ln;
typedef struct {
  nmtgan_dense qkv, o, ffn1, ffn2;     /* [3d,d] [d,d] [4d,d] [d,4d] */
  nmtgan_ln ln0, ln1;
} nmtgan_enc_block;
typedef struct {
  nmtgan_dense qkv, o, cq, ckv, co, ffn1, ffn2;   /* self / cross(q [d,d], kv [2d,d], o) / ffn */
  nmtgan_ln ln0, ln1, ln2;
} nmtgan_dec_block;
typedef struct {                       /* Model weights (model.py:486-618, 772-820) */
  int op_dtype;
  int d, heads, n_blocks, vocab_src, vocab_dst;
  float emb_mult;                      /* sqrt(d) if scale_embedding else 1 (model.py:494,553) */
  const float* emb_src;                /* [Vs][d] fp32 */
  const float* emb_dst;                /* [Vt][d] fp32 */
  const void* softmax_w;               /* [Vt][d] op dtype (already [N][K]) */
  const float* timing; int max_pos;    /* [max_pos][d] fp32 */
  nmtgan_enc_block enc[NMTGAN_MAX_BLOCKS];
  nmtgan_dec_block dec[NMTGAN_MAX_BLOCKS];
  int gemm_split;                      /* != 0 (needs op_dtype NMTGAN_F32): every dense layer and the softmax table are
                                          packed as fp16 [N][3K] = [W_hi | W_hi | W_lo] (W_hi = fp16(W), W_lo = fp16(W - W_hi));
                                          the pipelines split each fp32 GEMM input the same way on the fly,
                                          [A_hi | A_lo | A_hi], and run the product as ONE fp16 MFMA GEMM over 3K:
                                          A_hi.W_hi + A_lo.W_hi + A_hi.W_lo, about 21 significand bits per operand at
                                          three times the fp16 cost instead of the fp32 matrix-core rate.  Activations,
                                          K/V caches, attention and LayerNorm stay fp32 as in the fp32 tier. */
} nmtgan_generator;

typedef struct {                       /* one tower of DisCNN (cnn_discriminator.py:454-482) */
  const void* emb;                     /* [V][E] op dtype */
  const void* conv_w[NMTGAN_MAX_WIDTHS];     /* Wt[n_w][w*E] */
  const float* conv_scale[NMTGAN_MAX_WIDTHS];/* gamma*rsqrt(var+1e-3) */
  const float* conv_shift[NMTGAN_MAX_WIDTHS];/* (b-mean)*scale+beta */
  const void* hw_out;                  /* Wt[Fp][Fp] zero padded */
  const void* hw_tr;                   /* Wt[Fp][Fp] */
} nmtgan_disc_tower;
typedef struct {                       /* DisCNN inference weights */
  int op_dtype;
  int max_len, dim_word, vocab, vocab_s;
  int n_widths; int widths[NMTGAN_MAX_WIDTHS]; int n_filters[NMTGAN_MAX_WIDTHS];
  int F, Fp;                           /* sum n_filters; Fp = F rounded up to 64 */
  nmtgan_disc_tower tgt, src;
  const float* head_w;                 /* [2][2*Fp] fp32: class-major, [tgt Fp | src Fp] */
  const float* head_b;                 /* [2] */
  int gemm_split;                      /* != 0 (needs op_dtype NMTGAN_F16): the split-precision form of the scoring path -
                                          cnn_discriminator.py:440-522 computes in fp32, and with |score| ~ 100 a 16-bit
                                          operand rounding moves P(real) by percents - with every product as ONE fp16 MFMA
                                          GEMM over hi/lo operand splits, interleaved PER TOKEN / PER TAP so that a conv
                                          window stays one contiguous slice of the sentence image:
                                          emb      fp16 [V][3E]        row  = [e_hi | e_lo | e_hi]
                                          conv_w   fp16 Wt[n_w][w][3E] tap  = [W_hi | W_hi | W_lo]
                                          hw_*     fp16 Wt[Fp][3Fp]         = [W_hi | W_hi | W_lo]   (the pooled features
                                          are split on the fly, nmtgan_split3)
                                          with x_hi = fp16(x), x_lo = fp16(x - x_hi); dim_word stays E.  Accumulation,
                                          BatchNorm, max-pool, gate and head are fp32 as in every tier.  Scoring only
                                          (nmtgan_disc_train_grads refuses it). */
} nmtgan_discriminator;

/* ------------------------------------------------------------------------------------------
 * Discriminator inference.  Replaces sess.run(disc.dis_ypred_for_auc, {dis_input_x, dis_input_xs,
 * dis_dropout_keep_prob: 1.0}) (model.py:311-313 -> cnn_discriminator.py:529-558, 440-522).
 * nmtgan_disc_features: ids[M][T] (batch-major) -> highway features feat[M][Fp] fp32 of one tower.
 * nmtgan_disc_head: p_real[r] = softmax([feat_t[r] | feat_s[src_row ? src_row[r] : r]].W + b)[1].
 * Both chunk internally; workspace from nmtgan_disc_workspace_bytes(disc, chunk_rows).
 */
size_t nmtgan_disc_workspace_bytes(const nmtgan_discriminator* d, int chunk_rows);
int nmtgan_disc_features(const nmtgan_discriminator* d, int source_tower, const int32_t* ids, int M,
                         float* feat, int chunk_rows, void* workspace, size_t workspace_bytes, void* stream);
int nmtgan_disc_head(const nmtgan_discriminator* d, const float* feat_t, const float* feat_s,
                     const int32_t* src_row, int M, float* p_real, float* probs2, void* stream);

/* ------------------------------------------------------------------------------------------
 * Table form of the target tower's conv + BatchNorm + ReLU + max-over-time (cnn_discriminator.py:88-109, 461-477) for
 * the rollouts of one get_reward (model.py:303-313) - the building blocks nmtgan_rollout_reward runs when
 * nmtgan_reward_plan.conv_table selects the form, exposed for callers that schedule the discriminator themselves.
 * A filter of width w at window p is sum_{j<w} E[tok[p+j]] . W_w[j]: every term depends on one token and one tap, so
 *   tab[i][v][j][0..nfp) = E[v] . W_w[j]   for the whole vocabulary (nmtgan_conv_table_build: one GEMM per width),
 * and a window of a rollout is w gathered rows of tab added in tap order.  A rollout (b, g) keeps y[b][0..g):
 *   S[i][b][p][j-1][0..nfp) = sum of the first j taps (1 <= j < w) of window p of y[b]   (nmtgan_conv_table_prefix,
 *   which also writes relu(bn(conv)) of every window of y to win_vals[b*T + p][Fp] - the values behind a prefix max),
 * and nmtgan_conv_table_rows visits the windows p >= g - w + 1 of each row, starts a window that begins inside the
 * prefix from S[i][b][p][g-p-1], gathers the remaining taps with the row's own tokens and folds
 * max_p relu(sum * scale + shift) into pooled[r][col_i ..] (pooled must hold the max over the windows that end inside
 * the prefix, or zeros; col_i = filters of the widths before i).  Entries are `dtype` (NMTGAN_F16 / BF16 / F32; fp32
 * entries are added in fp64), S is fp32, nfp[i] >= nf[i] pads a table row to whole 16-byte vectors (the library uses
 * 128-byte lines); every table 16-byte aligned.  wt[i] is scratch of the build: w*nfp operand rows of the discriminator.
 * Token ids are clamped to [0, vocab) on the device; row_b[r] must name a sentence of the preceding
 * nmtgan_conv_table_prefix call (0 <= row_b[r] < B: it indexes S, which the kernels cannot bound) and row_g[r] the row's
 * give point, 1 <= row_g[r] <= T (a row whose give point is T has no window of its own and keeps pooled as passed).
 */
typedef struct {
  int n_widths, T, Fp, vocab, dtype;
  void* tab[NMTGAN_MAX_WIDTHS];          /* [vocab][w][nfp] */
  float* S[NMTGAN_MAX_WIDTHS];           /* [B][T-w+1][w-1][nfp] (w = 1: unused, any non-null pointer) */
  const float* scale[NMTGAN_MAX_WIDTHS]; /* [nf] folded BatchNorm, as nmtgan_disc_tower.conv_scale / conv_shift */
  const float* shift[NMTGAN_MAX_WIDTHS];
  int w[NMTGAN_MAX_WIDTHS], nf[NMTGAN_MAX_WIDTHS], nfp[NMTGAN_MAX_WIDTHS];
  void* wt[NMTGAN_MAX_WIDTHS];           /* build scratch [w*nfp][dim_word (x3 for gemm_split)] op dtype */
} nmtgan_conv_table;
int nmtgan_conv_table_build(const nmtgan_discriminator* d, const nmtgan_conv_table* t, void* stream);
int nmtgan_conv_table_prefix(const nmtgan_conv_table* t, const int32_t* y, int B, float* win_vals, void* stream);
int nmtgan_conv_table_rows(const nmtgan_conv_table* t, const int32_t* ids, const int32_t* row_b,
                           const int32_t* row_g, int rows, float* pooled, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-prefix reward averaging.  Replaces the numpy tail of Model.get_reward (model.py:315-345):
 * Q[b,g-1] = sum_r p[grp_first[b,g-1] + r] (rolls added in r order) or rollnum adds of p_base[b]
 * when grp_first < 0 (prefix >= sentence length); Q[b,T-1] = rollnum adds of p_base[b];
 * rewards = ((Q - bias_num*rollnum) * mask) / rollnum   (bias skipped when use_bias == 0).
 */
int nmtgan_reward_reduce(const float* p, const float* p_base, const int32_t* grp_first,
                         const float* mask, int B, int T, int rollnum, int use_bias, double bias_num,
                         float* rewards, void* stream);

/* ------------------------------------------------------------------------------------------
 * The hot path: Model.get_reward (model.py:291-346) = rollout graph (model.py:203-266) +
 * discriminator graph, restructured: encoder once, teacher-forced prefix pass once, KV-cached
 * lock-step rollouts over all live (sentence, roll, prefix) rows, source tower once.
 * The plan (row order, live counts) is built by the host facade; rows are sorted by remaining
 * private steps (descending) so the live rows of step s are the prefix [0, live[s]).
 */
typedef struct {
  int B, Ls, T, rollnum;
  int n_rows;                       /* live rollouts: rows (b, r, g) with 1 <= g < len_b */
  int n_steps;                      /* max private steps = max_b(len_b) - 2 (>= 0) */
  const int32_t* live;              /* HOST [n_steps+1]: live[0] = n_rows, live[s] rows at step s */
  const int64_t* step_off_host;     /* HOST [n_steps+1]: cumulative rows of steps 1..s */
  const int64_t* step_off;          /* device copy of step_off_host */
  const int32_t* x;                 /* [B][Ls] source ids */
  const uint8_t* x_pad;             /* [B][Ls] 1 where x == 0 */
  const int32_t* x_pad_to_T;        /* [B][T] source ids zero-extended to T (share_function.py:240-245) */
  const int32_t* y;                 /* [B][T] y_sample (zero from </S> on) */
  const int32_t* dec_in;            /* [B][T] shift_right(y) (model.py:764-766) */
  const float* mask;                /* [B][T] y_sample_mask (prefix of ones) */
  const int32_t* row_b;             /* [n_rows] sentence of row */
  const int32_t* row_g;             /* [n_rows] give_num of row */
  const int32_t* row_gp1;           /* [n_rows] give_num + 1 (shared keys of row) */
  const int32_t* row_lrow;          /* [n_rows] b*T + g: prefix-pass logits row of the first sample */
  const int64_t* row_uoff;          /* [n_rows] offset of u[r, g-1, b, g] in uniforms */
  const int32_t* grp_first;         /* [B][T-1] first row of group (b,g) or -1 */
  const float* uniforms;            /* [R][T-1][B_global][T] */
  int use_bias; double bias_num;    /* bias array = float32(bias_num * rollnum), model.py:340-341 */
  int logits_chunk_rows;            /* rows per logits GEMM + sampler chunk */
  int disc_chunk_rows;
  float* rewards;                   /* out [B][T] fp32 */
  int32_t* rollouts;                /* out [n_rows][T] masked rollouts (debug/parity), may be NULL */
  float* p_rows;                    /* out [n_rows] P(real) per rollout, may be NULL */
  float* p_base;                    /* out [B] P(real) of y_sample itself, may be NULL */
  float* prefix_logits;             /* out [B][T][V] teacher-forced logits, may be NULL */
  const int32_t* row_g_host;        /* HOST copy of row_g, may be NULL.  When given, the discriminator skips
                                       the conv windows that end inside a rollout's given prefix (same value
                                       as in y_sample) and takes their max from one pass over y_sample */
  /* Two-stream schedule (optional).  A rollout (b, r, g) is complete T-g steps into the lock-step loop,
   * so the discriminator can score finished rollouts on `aux_stream` while the remaining, ever smaller
   * decode steps run on `stream`: the conv GEMMs are MFMA-bound, the late decode steps latency/HBM-bound.
   * The library forks/joins the two streams with events (thread-local pool, created once per device);
   * results are identical to the single-stream schedule.  NULL or == stream: single stream. */
  /* Slot table of nmtgan_attn_group for the rollout steps (optional, see there): entries sorted by the row
   * index of their first slot, grp_live[s] (HOST, [n_steps+1]) = entries whose first slot is live at step s,
   * cnt = give_num + 1. */
  const int32_t* grp_table; int grp_slots; const int32_t* grp_live;
  void* aux_stream;
  int aux_gemm_wgs;                 /* workgroup cap of the persistent GEMMs on aux_stream while decode steps are
                                       still being issued (leaves CUs to the decoder); 0 = default (224) */
  int aux_min_rows;                 /* finished rollouts are handed to aux_stream in batches of at least this
                                       many rows; 0 = default (6144) */
  int conv_bucket_taps;             /* partial-window reuse in the discriminator (needs row_g_host): a conv window that
                                       straddles a rollout's give point takes the sum over its leading prefix taps, in
                                       blocks of this many taps, from one pass over y_sample and only multiplies the
                                       rest.  0 = default (4; 8 when NMTGAN_CONV_GROUP=0 launches every product on its
                                       own), < 0 = off (every such window is multiplied in full) */
  int conv_table;                   /* table form of the target tower's conv for the rollouts (needs row_g_host): the call
                                       first multiplies the WHOLE embedding table with every filter tap,
                                       P_w[v][j][:] = E[v] . W_w[j] (workspace: vocab * sum_w w * n_w entries - fp32 for a
                                       gemm_split or fp32 discriminator, else the op dtype), and a window of a rollout is
                                       then w gathered rows of P added in tap order (its leading prefix taps come as one
                                       partial sum from a pass over y_sample) - HBM-bound reads in place of the conv GEMMs,
                                       which recompute every (token, tap) product ~ n_rows * T / vocab times.
                                       0 = default (on for a 16-bit or gemm_split discriminator when the tables fit
                                       conv_table_max_bytes and n_rows * T >= 4 * vocab - fewer rollout tokens than that
                                       do not repay a pass over the vocabulary; an fp32 discriminator keeps the GEMM
                                       form), > 0 = on for every op dtype and size (still subject to the cap), < 0 = off */
  int64_t conv_table_max_bytes;     /* 0 = default (96 GiB) */
} nmtgan_reward_plan;
size_t nmtgan_rollout_reward_workspace_bytes(const nmtgan_generator* g, const nmtgan_discriminator* d,
                                             const nmtgan_reward_plan* plan);
int nmtgan_rollout_reward(const nmtgan_generator* g, const nmtgan_discriminator* d,
                          const nmtgan_reward_plan* plan, void* workspace, size_t workspace_bytes,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * Greedy generation.  Replaces sess.run(generate_sample) (model.py:140-176, 268-271): KV-cached,
 * argmax per position, no early stop.  x [B][Ls], x_pad [B][Ls] -> y [B][T].
 */
size_t nmtgan_generate_workspace_bytes(const nmtgan_generator* g, int B, int Ls, int T);
int nmtgan_generate_greedy(const nmtgan_generator* g, const int32_t* x, const uint8_t* x_pad, int B, int Ls,
                           int T, int32_t* y, void* workspace, size_t workspace_bytes, void* stream);

/* Teacher-forced logits [B][T][V] of decoder(shift_right(y)) - parity hook for the encoder/decoder
 * stack (model.py:486-618 + top, model.py:804-820). */
size_t nmtgan_teacher_forced_workspace_bytes(const nmtgan_generator* g, int B, int Ls, int T);
int nmtgan_teacher_forced_logits(const nmtgan_generator* g, const int32_t* x, const uint8_t* x_pad,
                                 const int32_t* dec_in, int B, int Ls, int T, float* logits,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Policy-gradient step of the generator: what gan_train.py:112 runs right after
 * get_reward.  Replaces sess.run([generate_g_loss, generate_g_grad, generate_g_optm]) of
 * Model.generate_step_and_update (model.py:273-276): decoder(shift_right(Y)) (model.py:178-184) ->
 * gan_output (model.py:649-666)  loss = -sum_{b,t} softmax(logits)[b,t,Y[b,t]] * reward[b,t] / B
 * -> compute_gradients over every trainable variable -> RMSProp (model.py:119).
 * nmtgan_pg_grads: forward + loss + backward.  `grads` mirrors the nmtgan_generator it differentiates: every
 *   pointer names a WRITABLE fp32 buffer with the layout of the weight it belongs to (matrices Wt[N][K],
 *   vectors [N], tables emb_src/emb_dst/softmax_w [V][d]; op_dtype/timing/sizes of `grads` are ignored) and
 *   is overwritten.  loss: 1 float on the device; p_target (optional): [B][T] probability of Y[b,t].
 *   x [B][Ls], x_pad [B][Ls] (1 where x == 0), y [B][T], dec_in = shift_right(y), reward [B][T].
 *   Residual dropout is neutralised as on the rollout path; for a shared_embedding model the three table
 *   gradients (emb_src, emb_dst, softmax_w of `grads`) name ONE buffer and are accumulated into it;
 *   max(Ls, T) is bounded by the 160 KiB of LDS the attention backward keeps its head slices and two score
 *   matrices in (90 at a head width of 64).
 * nmtgan_rmsprop_update: dense ApplyRMSProp over n contiguous parameters (ms += (g*g - ms)(1 - decay);
 *   mom = mom*momentum + lr*g/sqrt(ms + eps); w -= mom) with g = grad * grad_scale (the tower average of
 *   model.py:689-722), then clipped to [-grad_clip, grad_clip] when grad_clip > 0 (the discriminator's
 *   average_clip_gradient_by_value, share_function.py:413-441); mom may be NULL (momentum 0); w_op (optional)
 *   receives the updated parameters in op_dtype (the MFMA operand copy of a bf16 model).
 */
size_t nmtgan_pg_workspace_bytes(const nmtgan_generator* g, int B, int Ls, int T);
int nmtgan_pg_grads(const nmtgan_generator* g, const nmtgan_generator* grads, const int32_t* x, const uint8_t* x_pad,
                    const int32_t* y, const int32_t* dec_in, const float* reward, int B, int Ls, int T, float* loss,
                    float* p_target, void* workspace, size_t workspace_bytes, void* stream);
int nmtgan_rmsprop_update(float* w, float* ms, float* mom, const float* grad, int64_t n, float lr, float decay,
                          float momentum, float eps, float grad_scale, float grad_clip, void* w_op, int op_dtype,
                          void* stream);
/* tf.train.AdadeltaOptimizer() - the only other optimizer build_generate can select (model.py:111-119; defaults
 * lr 0.001, rho 0.95, eps 1e-8, both slots start at 0): accum = rho accum + (1-rho) g^2; upd = sqrt(accum_update + eps)
 * rsqrt(accum + eps) g; accum_update = rho accum_update + (1-rho) upd^2; w -= lr upd.  Same conventions as above.
 * The discriminator's default optimizer is AdadeltaOptimizer(learning_rate=1., rho=0.95, epsilon=1e-6)
 * (cnn_discriminator.py:156-158). */
int nmtgan_adadelta_update(float* w, float* accum, float* accum_update, const float* grad, int64_t n, float lr,
                           float rho, float eps, float grad_scale, float grad_clip, void* w_op, int op_dtype,
                           void* stream);

/* The discriminator's remaining optimizer choices (cnn_discriminator.py:153-161), same conventions:
 * tf.train.GradientDescentOptimizer(0.0001): w -= lr g;  tf.train.AdamOptimizer() (lr 0.001, beta1 0.9, beta2 0.999,
 * epsilon 1e-8): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps), t = the
 * 1-based step count, slots start at 0. */
int nmtgan_sgd_update(float* w, const float* grad, int64_t n, float lr, float grad_scale, float grad_clip, void* w_op,
                      int op_dtype, void* stream);
int nmtgan_adam_update(float* w, float* m, float* v, const float* grad, int64_t n, float lr, float beta1, float beta2,
                       float eps, int t, float grad_scale, float grad_clip, void* w_op, int op_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Discriminator training step: one iteration of DisCNN.train (cnn_discriminator.py:664-691),
 * i.e. sess.run([train_optm, train_loss, train_accuracy, train_grads_and_vars]) for one tower: build_model
 * (cnn_discriminator.py:248-345) in TRAINING mode - BatchNorm with batch moments over (sentence, window) per filter,
 * moving statistics moved in place with decay 0.9 (conv_batch_norm, :22-33), dropout keep 1.0 (train() feeds
 * drop_prob = 1.0) - softmax cross-entropy, and the gradients of sum(losses) (compute_gradients on the per-example
 * vector, :334) of every trainable variable.
 * `d` supplies the MFMA-operand copies the forward reads (tables, conv filters Wt[n][w*E], highway Wt[Fp][Fp]) and
 * the fp32 head; `p` the fp32 per-filter vectors of each tower, concatenated over the widths in filter order ([F]
 * each: conv bias, BN gamma/beta, BN moving mean/variance - the moving statistics are UPDATED); `grads` has the
 * same shape and receives fp32 gradients in the packed layouts (emb [V][E], conv_w[i] Wt[n][w*E], conv_b/gamma/beta
 * [F], hw_* Wt[Fp][Fp], head_w [2][2Fp], head_b [2]; its moving_* fields are ignored).
 * x / xs: [M][T] batch-major ids of the target / source tower; labels [M][2] one-hot rows; loss (mean of the
 * per-example losses, :343) and accuracy (:326-327): 1 float each on the device.
 * nmtgan_disc_fold_bn recomputes the inference constants (conv_scale / conv_shift of nmtgan_disc_tower) from the
 * trained vectors; nmtgan_clip_values is the one-off weight clip to [-1, 1] after the first step (:583-591, 690);
 * with infinite bounds it is a plain refresh of an operand copy in any op dtype.
 * The two gradient entry points (nmtgan_pg_grads, nmtgan_disc_train_grads) and the optimizer kernels' w_op take
 * NMTGAN_BF16 or NMTGAN_F32 models: the backward pass keeps d(logits)-sized values in the operand type, which
 * underflow in fp16 without loss scaling.  A model that scores in NMTGAN_F16 keeps a bf16 operand copy for its update
 * steps and refreshes the fp16 copy from the fp32 master weights afterwards (the Python facade does).
 */
typedef struct {
  float* emb;                                  /* [V][E] */
  float* conv_w[NMTGAN_MAX_WIDTHS];            /* Wt[n_w][w*E] */
  float* conv_b; float* gamma; float* beta;    /* [F] each, widths concatenated */
  float* moving_mean; float* moving_var;       /* [F] each, not trainable */
  float* hw_out; float* hw_tr;                 /* Wt[Fp][Fp] */
} nmtgan_disc_tower_params;
typedef struct {
  nmtgan_disc_tower_params tgt, src;
  float* head_w;                               /* [2][2*Fp] */
  float* head_b;                               /* [2] */
} nmtgan_disc_params;
size_t nmtgan_disc_train_workspace_bytes(const nmtgan_discriminator* d, int M);
int nmtgan_disc_train_grads(const nmtgan_discriminator* d, const nmtgan_disc_params* p, const nmtgan_disc_params* grads,
                            const int32_t* x, const int32_t* xs, const float* labels, int M, float* loss,
                            float* accuracy, void* workspace, size_t workspace_bytes, void* stream);
int nmtgan_disc_fold_bn(const float* conv_b, const float* gamma, const float* beta, const float* moving_mean,
                        const float* moving_var, int n, float* scale, float* shift, void* stream);
int nmtgan_clip_values(float* w, int64_t n, float lo, float hi, void* w_op, int op_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement hook (no counterpart in the reference; used by bench.py's roofline leg only).
 * Between start and stop every kernel launch is bracketed by HIP events on its own stream;
 * stop synchronises the device and returns summed milliseconds and launch counts per class:
 * 0 other, 1 gemm_proj (QKV/out/cross projections), 2 gemm_ffn, 3 gemm_logits, 4 gemm_conv
 * (discriminator conv+BN+ReLU+max), 5 gemm_highway, 6 attn_self_step, 7 attn_cross_step,
 * 8 attn_prefill (encoder + teacher-forced pass), 9 sampler, 10 add_layernorm, 11 embed,
 * 12 disc_small (gather/cast/gate/head/reduce), 13 attn_self_private (the per-row pass over the private
 * suffix keys of a rollout step when the shared prefix went through nmtgan_attn_group; class 6 then holds
 * only the shared-prefix pass), 14 sampler_rows (nmtgan_sample_rows over materialised logits: the first token of
 * every rollout; class 9 then holds the fused logits+sampling tail of the rollout steps), 15 conv_table (the gathers
 * of the table form of the rollouts' conv + max; class 4 then holds the GEMMs that build the tables and the conv of
 * y_sample and of the source sentences).  Allocates events: never use inside timed runs.
 */
int nmtgan_profile_classes(void);
int nmtgan_profile_start(void);
int nmtgan_profile_stop(double* ms_per_class, int64_t* launches_per_class, int n_classes);

#ifdef __cplusplus
}
#endif
#endif /* NMTGAN_H_ */
