// Host-side pipelines of libnmtgan_hip.so: they only enqueue kernels on the caller's stream.
//
//   encode            model.py:486-543      once per batch (the reference re-runs it per sess.run)
//   cross K/V         common_attention.py:383-388, once per sentence and decoder block
//   prefill           model.py:545-618 over shift_right(y_sample): shared K/V + prefix logits
//   rollout steps     model.py:224-245 restructured: lock-step over all live (b, r, g) rows,
//                     one position per step, private step-major KV cache
//   discriminator     cnn_discriminator.py:440-522, source tower once per sentence
//   reward reduce     model.py:315-345
#include <vector>

#include "common.h"
#include "kernels.h"

namespace nmtgan {

static inline char* at(const void* base, int64_t elems, int es) { return (char*)base + elems * es; }

// Split-precision GEMM form (nmtgan_generator.gemm_split): the scratch that receives the [hi | lo | hi] image of a GEMM
// input whose producer did not write it.  A pipeline passes its own context to the GEMMs of the GENERATOR only - the
// discriminator's launches never see it - and everything that uses one context is enqueued on ONE stream, in order,
// so one buffer serves every GEMM.  nullptr = plain GEMM in the model's op dtype.
struct SplitCtx { void* buf; size_t bytes; };
// A [M][K] fp32 -> the context's scratch as fp16 [M][3K] = [hi | lo | hi]
static int split_operand(const SplitCtx* sp, const void* A, int64_t lda, int M, int K, hipStream_t st) {
  NMTGAN_REQUIRE((size_t)M * 3 * K * sizeof(f16_t) <= sp->bytes, "split GEMM: scratch too small for %d x %d", M, K);
  return launch_split3((const float*)A, lda, M, K, sp->buf, st);
}

// Error budget of the 16-bit tiers, measured from the split-precision tier by taking the precision away from ONE
// component at a time (tools/error_budget.py): NMTGAN_SPLIT_ABLATE is a bit mask - 1 projections (qkv, o, cross q / kv /
// o), 2 feed-forward pair, 4 logits, 8 attention inputs (q, K, V rounded to fp16 as the fp16 caches hold them).  A masked
// GEMM multiplies only the hi halves of its operands (columns 0 .. K-1 of the [hi | lo | hi] and [hi | hi | lo] images):
// plain fp16 operands with fp32 accumulation, exactly the fp16 tier's product.  0 (default) = the tier as shipped.
static int split_ablate() {
  static const int v = [] { const char* e = getenv("NMTGAN_SPLIT_ABLATE"); return e ? atoi(e) : 0; }();
  return v;
}

// out[s] = (A . Wt^T + b) with up to 3 column segments
static int dense(int op, const SplitCtx* sp, const void* A, int64_t lda, int M, int K, int N, const nmtgan_dense& w,
                 int n_seg, int seg_width, void* const* outs, const int64_t* ldos, const int* out_f32,
                 const float* scales, int relu, hipStream_t st, const void* presplit = nullptr) {
  nmtgan_gemm_args a{};
  a.op_dtype = op; a.epilogue = NMTGAN_EPI_LINEAR;
  a.A = A; a.Wt = w.w; a.M = M; a.N = N; a.K = K; a.lda = lda; a.a_group = 0;
  a.bias = w.b; a.relu = relu; a.n_seg = n_seg; a.seg_width = seg_width;
  for (int s = 0; s < n_seg; ++s) {
    a.out[s] = outs[s]; a.ldo[s] = ldos[s]; a.out_f32[s] = out_f32[s]; a.seg_scale[s] = scales ? scales[s] : 1.f;
  }
  if (sp) {                                        // w.w is fp16 [N][3K] = [hi | hi | lo]; outputs stay fp32
    NMTGAN_REQUIRE(op == NMTGAN_F32, "split GEMM: the model must be packed with fp32 storage");
    // presplit: the producer of A already wrote its [hi | lo | hi] image (add_layernorm, attention, the first
    // feed-forward layer); else one pass over A makes it in the thread's scratch
    if (!presplit) NMTGAN_TRY(split_operand(sp, A, lda, M, K, st));
    a.op_dtype = NMTGAN_F16; a.A = presplit ? presplit : sp->buf; a.lda = 3 * (int64_t)K; a.K = 3 * K;
    for (int s = 0; s < n_seg; ++s) a.out_f32[s] = out_f32[s] == 2 ? 2 : 1;    // 2: the consumer's operand image
    const int role = prof::g_hint == prof::GEMM_FFN ? 2 : (prof::g_hint == prof::GEMM_LOGITS ? 4 : 1);
    if (split_ablate() & role) { a.K = K; a.ldw = 3 * (int64_t)K; }            // hi halves only (error budget)
  }
  return launch_gemm(&a, st);
}
static int dense1(int op, const SplitCtx* sp, const void* A, int64_t lda, int M, int K, int N, const nmtgan_dense& w,
                  void* out, int64_t ldo, int out_is_f32, float scale, int relu, hipStream_t st,
                  const void* presplit = nullptr) {
  void* outs[1] = {out};
  int64_t ld[1] = {ldo};
  int f[1] = {out_is_f32};
  float sc[1] = {scale};
  return dense(op, sp, A, lda, M, K, N, w, 1, N, outs, ld, f, sc, relu, st, presplit);
}

static int add_ln(const nmtgan_generator* g, float* xf, const float* yf, const nmtgan_ln& ln, int M, void* xo,
                  const uint8_t* zero_row, hipStream_t st, void* x_img = nullptr) {
  nmtgan_addln_args a{};
  a.x = xf; a.y = yf; a.gamma = ln.g; a.beta = ln.b; a.M = M; a.d = g->d;
  a.out_f32 = xf; a.out_op = xo; a.op_dtype = g->op_dtype; a.zero_row = zero_row; a.out_split = x_img;
  return launch_add_layernorm(&a, st);
}

// xf <- LayerNorm(xf + A . w^T + b), xo = its op-dtype copy: one kernel where the fused epilogue takes the shape
// (16-bit tiers at d = 512) and the projection is bound by its output traffic (K <= d: the attention output
// projections), else the projection into yf followed by add_layernorm (K = 4 d, the second feed-forward layer, is
// bound by the matrix cores and runs faster on the 256x256 kernel even with the extra round trip of yf;
// NMTGAN_FUSE_LN=2 fuses it too, 0 fuses nothing).  The split-K pipelines (B-row greedy decode) keep the two-kernel
// form: their projections are cut into k-slices to fill the CUs.
static int dense_add_ln(const nmtgan_generator* g, const SplitCtx* sp, const void* A, int64_t lda, int M, int K,
                        const nmtgan_dense& w, const nmtgan_ln& ln, float* xf, float* yf, void* xo, hipStream_t st,
                        const void* a_img = nullptr, void* x_img = nullptr) {
  static const int fuse = [] { const char* e = getenv("NMTGAN_FUSE_LN"); return e ? atoi(e) : 1; }();
  if (fuse > 0 && (K <= g->d || fuse > 1) && !gemm_splitk_active() && gemm_addln_supported(g->op_dtype, g->d, K)) {
    nmtgan_gemm_args a{};
    a.op_dtype = g->op_dtype; a.epilogue = NMTGAN_EPI_ADDLN;
    a.A = A; a.Wt = w.w; a.M = M; a.N = g->d; a.K = K; a.lda = lda;
    a.bias = w.b; a.ln_x = xf; a.ln_gamma = ln.g; a.ln_beta = ln.b; a.ln_out = xo;
    return launch_gemm(&a, st);
  }
  // split-precision form: a_img = A's operand image written by its producer, x_img = where to leave the image of x
  NMTGAN_TRY(dense1(g->op_dtype, sp, A, lda, M, K, g->d, w, yf, g->d, 1, 1.f, 0, st, a_img));
  prof::Hint own_class(-1);                      // the add_layernorm launch is timed under its own class
  return add_ln(g, xf, yf, ln, M, xo, nullptr, st, x_img);
}

static int check_generator(const nmtgan_generator* g) {
  NMTGAN_REQUIRE(g != nullptr, "generator: null");
  NMTGAN_REQUIRE(valid_op_dtype(g->op_dtype), "generator: bad op_dtype");
  NMTGAN_REQUIRE(g->d > 0 && g->d % 64 == 0 && g->d <= 1024, "generator: hidden_units must be a multiple of 64, <= 1024");
  NMTGAN_REQUIRE(g->heads > 0 && 64 % g->heads == 0 && g->d % g->heads == 0, "generator: num_heads must divide 64");
  NMTGAN_REQUIRE(g->n_blocks > 0 && g->n_blocks <= NMTGAN_MAX_BLOCKS, "generator: bad num_blocks");
  NMTGAN_REQUIRE(g->vocab_dst % 4 == 0, "generator: dst_vocab_size must be a multiple of 4 (pad the softmax table)");
  NMTGAN_REQUIRE(g->emb_src && g->emb_dst && g->softmax_w && g->timing, "generator: null table");
  NMTGAN_REQUIRE(!g->gemm_split || g->op_dtype == NMTGAN_F32, "generator: gemm_split needs op_dtype NMTGAN_F32");
  return 0;
}

// ------------------------------------------------------------------------------------------ activations
struct Act {
  float* xf; void* xo; float* yf; float* q; void* ao; void* h;
  void* split; size_t split_bytes;            // split A operand of the widest GEMM (K = 4 d): rows x 12 d fp16
  void* img;                                  // split-precision tier: [hi | lo | hi] image (rows x 3 d fp16) of the
                                              // current d-wide GEMM operand (x after embed / LayerNorm, or the
                                              // attention output), written by the kernel that produces it
  SplitCtx sc; bool use_split;                // the generator's split context (sc = {split, split_bytes})
  const SplitCtx* sp() const { return use_split ? &sc : nullptr; }
};
static void carve_act(Arena& ar, int64_t rows, int d, int es, Act& a, bool gemm_split = false) {
  a.xf = ar.get<float>(rows * d);
  a.xo = es == 4 ? (void*)a.xf : ar.take(rows * d * es);     // fp32 operands: the residual stream IS the GEMM operand
  a.yf = ar.get<float>(rows * d);
  a.q = ar.get<float>(rows * d);
  a.ao = ar.take(rows * d * es);
  a.h = ar.take(rows * 4 * d * es);
  a.split_bytes = gemm_split ? (size_t)rows * 12 * d * sizeof(f16_t) : 0;
  a.split = gemm_split ? ar.take(a.split_bytes) : nullptr;
  a.img = gemm_split ? ar.take((size_t)rows * 3 * d * sizeof(f16_t)) : nullptr;
  a.sc = SplitCtx{a.split, a.split_bytes};
  a.use_split = gemm_split;
}

// ------------------------------------------------------------------------------------------ encoder
static int encode(const nmtgan_generator* g, const int32_t* x, const uint8_t* x_pad, int B, int Ls, Act& a,
                  void* kbuf, void* vbuf, hipStream_t st) {
  const int op = g->op_dtype, d = g->d, M = B * Ls;
  const float qscale = 1.0f / sqrtf((float)(d / g->heads));
  nmtgan_embed_args e{};
  e.table = g->emb_src; e.timing = g->timing; e.vocab = g->vocab_src; e.max_pos = g->max_pos;
  e.tok = x; e.tok_ld = 1; e.pos_mod = Ls; e.mult = g->emb_mult; e.M = M; e.d = d;
  e.out_f32 = a.xf; e.out_op = a.xo; e.op_dtype = op;
  NMTGAN_REQUIRE(Ls <= g->max_pos, "encode: source length %d exceeds timing table %d", Ls, g->max_pos);
  NMTGAN_TRY(launch_embed(&e, st));
  for (int l = 0; l < g->n_blocks; ++l) {
    const nmtgan_enc_block& b = g->enc[l];
    void* outs[3] = {a.q, kbuf, vbuf};
    int64_t ld[3] = {d, d, d};
    int f32[3] = {0, 0, 0};                       // q in op dtype like K/V (the attention kernels take it as such)
    float sc[3] = {qscale, 1.f, 1.f};
    NMTGAN_TRY(dense(op, a.sp(), a.xo, d, M, d, 3 * d, b.qkv, 3, d, outs, ld, f32, sc, 0, st));
    if (a.sp() && (split_ablate() & 8))
      for (int s = 0; s < 3; ++s) NMTGAN_TRY(launch_round_f16((float*)outs[s], (int64_t)M * d, st));
    nmtgan_attn_args t{};
    t.q = a.q; t.q_op_dtype = 1; t.M = M; t.d = d; t.heads = g->heads; t.kv_dtype = op;
    t.k1 = kbuf; t.v1 = vbuf; t.len1 = Ls; t.rows_per_blk = Ls; t.causal = 0; t.mask1 = x_pad; t.out = a.ao;
    NMTGAN_TRY(launch_attn_rows(&t, st));
    NMTGAN_TRY(dense1(op, a.sp(), a.ao, d, M, d, d, b.o, a.yf, d, 1, 1.f, 0, st));
    NMTGAN_TRY(add_ln(g, a.xf, a.yf, b.ln0, M, a.xo, nullptr, st));
    {
      prof::Hint ph(prof::GEMM_FFN);
      NMTGAN_TRY(dense1(op, a.sp(), a.xo, d, M, d, 4 * d, b.ffn1, a.h, 4 * d, 0, 1.f, 1, st));
      NMTGAN_TRY(dense1(op, a.sp(), a.h, 4 * d, M, 4 * d, d, b.ffn2, a.yf, d, 1, 1.f, 0, st));
    }
    NMTGAN_TRY(add_ln(g, a.xf, a.yf, b.ln1, M, a.xo, l == g->n_blocks - 1 ? x_pad : nullptr, st));   // model.py:542
  }
  return 0;
}

// cross-attention K/V of every decoder block from the encoder output (op dtype, [B*Ls][d])
static int cross_kv(const nmtgan_generator* g, const SplitCtx* sp, const void* enc_op, int M, void* crossK, void* crossV,
                    hipStream_t st) {
  const int op = g->op_dtype, d = g->d, es = dtype_size(op);
  for (int l = 0; l < g->n_blocks; ++l) {
    void* outs[2] = {at(crossK, (int64_t)l * M * d, es), at(crossV, (int64_t)l * M * d, es)};
    int64_t ld[2] = {d, d};
    int f32[2] = {0, 0};
    float sc[2] = {1.f, 1.f};
    NMTGAN_TRY(dense(op, sp, enc_op, d, M, d, 2 * d, g->dec[l].ckv, 2, d, outs, ld, f32, sc, 0, st));
    if (sp && (split_ablate() & 8))
      for (int s = 0; s < 2; ++s) NMTGAN_TRY(launch_round_f16((float*)outs[s], (int64_t)M * d, st));
  }
  return 0;
}

// Self-attention of the rollout steps.  Default: ONE launch of the per-row kernel over both key segments - the shared
// teacher-forced prefix (L2-resident; the four rows of a workgroup share it) and the row's private suffix.
// NMTGAN_SELF_GROUP=1 restores the earlier two-launch form (shared prefix per group on the matrix cores, running
// (o, m, l) state through HBM, then the per-row kernel over the private suffix): 66.0 against 59.8 ms per C2
// get_reward in the fp16 tier, 122.0 against 116.8 ms in the split-precision tier, 288 launches more.  Cross attention
// (every row attends to all Ls encoder positions) keeps the group kernel.
static bool self_group_pass() {
  static const bool v = [] { const char* e = getenv("NMTGAN_SELF_GROUP"); return e && atoi(e) != 0; }();
  return v;
}

// One decoder pass over M rows that are already embedded in a.xf/a.xo.  Self-attention keys:
// shared segment (shK/shV, per-row block/count or the causal rows_per_blk form) + private segment.
struct DecAttn {
  // self attention
  const void* shK; const void* shV; int len_sh; int64_t sh_layer_stride;      // elements per block layer
  const int32_t* seg1_blk; const int32_t* seg1_cnt; int rows_per_blk; int causal;
  void* prK; void* prV; int64_t pr_layer_stride; const int64_t* step_off; int n_steps2;
  // where this pass writes its own K/V (element offset inside a layer)
  void* wrK; void* wrV; int64_t wr_layer_stride; int64_t wr_off;
  // cross attention
  const void* crossK; const void* crossV; int Ls; int64_t cross_layer_stride;
  const int32_t* cross_blk; int cross_rows_per_blk; const uint8_t* x_pad;
  int step_mode;   // 1: rollout step (profile classes attn_self/attn_cross), 0: prefill
  // group path (bf16 rollout steps): R consecutive rows share their segment-1 keys -> MFMA kernel
  int group_R; const void* shVt; int sh_Lp; int64_t sht_layer_stride;
  const void* crossVt; int cross_Lp; int64_t crosst_layer_stride;
  float* o_part; float* m_part; float* l_part;
  const int32_t* slot_table; int slots; int n_entries;      // several groups of a sentence per workgroup
};

static int decoder_pass(const nmtgan_generator* g, int M, Act& a, const DecAttn& c, hipStream_t st) {
  const int op = g->op_dtype, d = g->d, es = dtype_size(op);
  const float qscale = 1.0f / sqrtf((float)(d / g->heads));
  // Split-precision tier: every producer of a GEMM operand writes the operand's fp16 [hi | lo | hi] image itself
  // (the embedding before this pass, add_layernorm, the attention kernels, the first feed-forward layer), so no
  // stand-alone split pass runs between the layers.  a.img holds x's image when this function is entered and left.
  const SplitCtx* const sp = a.sp();
  void* const img = sp ? a.img : nullptr;
  void* const attn_out = img ? img : a.ao;
  for (int l = 0; l < g->n_blocks; ++l) {
    const nmtgan_dec_block& b = g->dec[l];
    {
      void* outs[3] = {a.q, at(c.wrK, l * c.wr_layer_stride + c.wr_off, es), at(c.wrV, l * c.wr_layer_stride + c.wr_off, es)};
      int64_t ld[3] = {d, d, d};
      int f32[3] = {0, 0, 0};
      float sc[3] = {qscale, 1.f, 1.f};
      NMTGAN_TRY(dense(op, sp, a.xo, d, M, d, 3 * d, b.qkv, 3, d, outs, ld, f32, sc, 0, st, img));
      if (img && (split_ablate() & 8))
        for (int s = 0; s < 3; ++s) NMTGAN_TRY(launch_round_f16((float*)outs[s], (int64_t)M * d, st));
    }
    if (c.group_R > 0 && self_group_pass()) {
      // shared prefix keys once per group on MFMA -> running (o, m, l); private suffix per row after
      nmtgan_attn_group_args ga{};
      ga.q = a.q; ga.q_bf16 = 1; ga.kv_dtype = op; ga.M = M; ga.d = d; ga.heads = g->heads; ga.rows_per_group = c.group_R;
      ga.k = at(c.shK, l * c.sh_layer_stride, es); ga.vt = at(c.shVt, l * c.sht_layer_stride, es);
      ga.len = c.len_sh; ga.len_padded = c.sh_Lp; ga.row_blk = c.seg1_blk; ga.row_cnt = c.seg1_cnt;
      ga.o_part = c.o_part; ga.m_part = c.m_part; ga.l_part = c.l_part;
      ga.slot_table = c.slot_table; ga.slots = c.slots; ga.n_entries = c.n_entries;
      prof::Hint ph(prof::ATTN_SELF);
      NMTGAN_TRY(launch_attn_group(&ga, st));
      nmtgan_attn_args t{};
      t.q = a.q; t.q_op_dtype = 1; t.M = M; t.d = d; t.heads = g->heads; t.kv_dtype = op;
      t.k2 = at(c.prK, l * c.pr_layer_stride, es); t.v2 = at(c.prV, l * c.pr_layer_stride, es);
      t.step_off = c.step_off; t.n_steps2 = c.n_steps2;
      t.o_init = c.o_part; t.m_init = c.m_part; t.l_init = c.l_part; t.out = attn_out; t.out_split = img != nullptr;
      prof::Hint ph2(prof::ATTN_SELF_PRIVATE);
      NMTGAN_TRY(launch_attn_rows(&t, st));
    } else {
      nmtgan_attn_args t{};
      t.q = a.q; t.q_op_dtype = 1; t.M = M; t.d = d; t.heads = g->heads; t.kv_dtype = op;
      if (c.shK) {
        t.k1 = at(c.shK, l * c.sh_layer_stride, es); t.v1 = at(c.shV, l * c.sh_layer_stride, es);
        t.len1 = c.len_sh; t.seg1_blk = c.seg1_blk; t.seg1_cnt = c.seg1_cnt; t.rows_per_blk = c.rows_per_blk;
        t.causal = c.causal;
      }
      if (c.n_steps2 > 0) {
        t.k2 = at(c.prK, l * c.pr_layer_stride, es); t.v2 = at(c.prV, l * c.pr_layer_stride, es);
        t.step_off = c.step_off; t.n_steps2 = c.n_steps2;
      }
      t.out = attn_out; t.out_split = img != nullptr;
      prof::Hint ph(c.step_mode ? prof::ATTN_SELF : prof::ATTN_PREFILL);
      NMTGAN_TRY(launch_attn_rows(&t, st));
    }
    NMTGAN_TRY(dense_add_ln(g, sp, a.ao, d, M, d, b.o, b.ln0, a.xf, a.yf, a.xo, st, img, img));
    NMTGAN_TRY(dense1(op, sp, a.xo, d, M, d, d, b.cq, a.q, d, 0, qscale, 0, st, img));
    if (img && (split_ablate() & 8)) NMTGAN_TRY(launch_round_f16(a.q, (int64_t)M * d, st));
    if (c.group_R > 0) {
      nmtgan_attn_group_args ga{};
      ga.q = a.q; ga.q_bf16 = 1; ga.kv_dtype = op; ga.M = M; ga.d = d; ga.heads = g->heads; ga.rows_per_group = c.group_R;
      ga.k = at(c.crossK, l * c.cross_layer_stride, es); ga.vt = at(c.crossVt, l * c.crosst_layer_stride, es);
      ga.len = c.Ls; ga.len_padded = c.cross_Lp; ga.row_blk = c.cross_blk; ga.mask = c.x_pad;
      ga.out = attn_out; ga.out_split = img != nullptr;
      ga.slot_table = c.slot_table; ga.slots = c.slots; ga.n_entries = c.n_entries;
      prof::Hint ph(prof::ATTN_CROSS);
      NMTGAN_TRY(launch_attn_group(&ga, st));
    } else {
      nmtgan_attn_args t{};
      t.q = a.q; t.q_op_dtype = 1; t.M = M; t.d = d; t.heads = g->heads; t.kv_dtype = op;
      t.k1 = at(c.crossK, l * c.cross_layer_stride, es); t.v1 = at(c.crossV, l * c.cross_layer_stride, es);
      t.len1 = c.Ls; t.seg1_blk = c.cross_blk; t.rows_per_blk = c.cross_rows_per_blk; t.causal = 0;
      t.mask1 = c.x_pad; t.out = attn_out; t.out_split = img != nullptr;
      prof::Hint ph(c.step_mode ? prof::ATTN_CROSS : prof::ATTN_PREFILL);
      NMTGAN_TRY(launch_attn_rows(&t, st));
    }
    NMTGAN_TRY(dense_add_ln(g, sp, a.ao, d, M, d, b.co, b.ln1, a.xf, a.yf, a.xo, st, img, img));
    {
      prof::Hint ph(prof::GEMM_FFN);
      if (img) {                                   // hidden layer straight into its operand image [M][12 d]
        NMTGAN_REQUIRE((size_t)M * 12 * d * sizeof(f16_t) <= a.split_bytes, "decoder: split scratch too small");
        NMTGAN_TRY(dense1(op, sp, a.xo, d, M, d, 4 * d, b.ffn1, a.split, 12 * (int64_t)d, 2, 1.f, 1, st, img));
        NMTGAN_TRY(dense_add_ln(g, sp, a.h, 4 * d, M, 4 * d, b.ffn2, b.ln2, a.xf, a.yf, a.xo, st, a.split, img));
      } else {
        NMTGAN_TRY(dense1(op, sp, a.xo, d, M, d, 4 * d, b.ffn1, a.h, 4 * d, 0, 1.f, 1, st));
        NMTGAN_TRY(dense_add_ln(g, sp, a.h, 4 * d, M, 4 * d, b.ffn2, b.ln2, a.xf, a.yf, a.xo, st));
      }
    }
  }
  return 0;
}

static int logits_gemm(const nmtgan_generator* g, const SplitCtx* sp, const void* x_op, int M, float* logits,
                       hipStream_t st) {
  nmtgan_dense w{g->softmax_w, nullptr};
  prof::Hint ph(prof::GEMM_LOGITS);
  return dense1(g->op_dtype, sp, x_op, g->d, M, g->d, g->vocab_dst, w, logits, g->vocab_dst, 1, 1.f, 0, st);
}

// teacher-forced decoder over dec_in[B][T]; writes shared K/V and leaves the final states in a.xo
static int prefill(const nmtgan_generator* g, const int32_t* dec_in, const uint8_t* x_pad, int B, int Ls, int T,
                   Act& a, void* shK, void* shV, const void* crossK, const void* crossV, hipStream_t st) {
  const int d = g->d, M = B * T;
  NMTGAN_REQUIRE(T <= g->max_pos, "prefill: max_len %d exceeds timing table %d", T, g->max_pos);
  nmtgan_embed_args e{};
  e.table = g->emb_dst; e.timing = g->timing; e.vocab = g->vocab_dst; e.max_pos = g->max_pos;
  e.tok = dec_in; e.tok_ld = 1; e.pos_mod = T; e.mult = 2.0f * g->emb_mult;      // model.py:565 (quirk B.2)
  e.M = M; e.d = d; e.out_f32 = a.xf; e.out_op = a.xo; e.op_dtype = g->op_dtype;
  e.out_split = a.sp() ? a.img : nullptr;
  NMTGAN_TRY(launch_embed(&e, st));
  DecAttn c{};
  c.shK = shK; c.shV = shV; c.len_sh = T; c.sh_layer_stride = (int64_t)M * d; c.rows_per_blk = T; c.causal = 1;
  c.wrK = shK; c.wrV = shV; c.wr_layer_stride = (int64_t)M * d; c.wr_off = 0;
  c.crossK = crossK; c.crossV = crossV; c.Ls = Ls; c.cross_layer_stride = (int64_t)B * Ls * d;
  c.cross_rows_per_blk = T; c.x_pad = x_pad;
  return decoder_pass(g, M, a, c, st);
}

// ------------------------------------------------------------------------------------------ discriminator
static int check_disc(const nmtgan_discriminator* d) {
  NMTGAN_REQUIRE(d != nullptr, "discriminator: null");
  NMTGAN_REQUIRE(valid_op_dtype(d->op_dtype), "discriminator: bad op_dtype");
  NMTGAN_REQUIRE(d->n_widths > 0 && d->n_widths <= NMTGAN_MAX_WIDTHS, "discriminator: bad n_widths");
  NMTGAN_REQUIRE(d->Fp % 64 == 0 && d->Fp >= d->F && d->F > 0, "discriminator: Fp must be F rounded up to 64");
  NMTGAN_REQUIRE((d->dim_word * dtype_size(d->op_dtype)) % 16 == 0, "discriminator: dim_word misaligned");
  int f = 0;
  for (int i = 0; i < d->n_widths; ++i) {
    NMTGAN_REQUIRE(d->widths[i] >= 1 && d->widths[i] <= d->max_len, "discriminator: filter width %d > max_len", d->widths[i]);
    f += d->n_filters[i];
  }
  NMTGAN_REQUIRE(f == d->F, "discriminator: sum(num_filters) != F");
  NMTGAN_REQUIRE(!d->gemm_split || d->op_dtype == NMTGAN_F16, "discriminator: gemm_split needs op_dtype NMTGAN_F16");
  return 0;
}
// elements per token of the sentence image (and per tap of a filter): the split-precision form interleaves the
// [hi | lo | hi] / [hi | hi | lo] parts per token / tap, so a window of w tokens stays one contiguous slice
static inline int disc_ew(const nmtgan_discriminator* d) { return d->gemm_split ? 3 * d->dim_word : d->dim_word; }

struct DiscBufs { void* e; float* pooled; void* pooled_op; float* ho; float* ht; };
static void carve_disc(Arena& ar, const nmtgan_discriminator* d, int chunk, DiscBufs& b) {
  const int es = dtype_size(d->op_dtype);
  b.e = ar.take((int64_t)chunk * d->max_len * disc_ew(d) * es);
  b.pooled = ar.get<float>((int64_t)chunk * d->Fp);
  b.pooled_op = ar.take((int64_t)chunk * d->Fp * es * (d->gemm_split ? 3 : 1));
  b.ho = ar.get<float>((int64_t)chunk * d->Fp);
  b.ht = ar.get<float>((int64_t)chunk * d->Fp);
}

// Windows a rollout shares with y_sample (those ending inside its given prefix) are not recomputed: their
// max comes from pm[b*T + g][Fp] (prefix_max over one pass on y_sample); g_min = smallest give_num of the chunk.
struct PrefixReuse { const float* pm; const int32_t* row_b; const int32_t* row_g; const int32_t* row_g_host; };
static const int REUSE_ROWS = [] { const char* e = getenv("NMTGAN_REUSE_ROWS"); return e && atoi(e) > 0 ? atoi(e) : 8192; }();   // granularity at which the first computed window is chosen

// Windows that STRADDLE the give point (p < g < p + w) start with g - p tokens of y_sample.  Their taps are cut
// into blocks of `delta`: a window whose first c*delta taps lie in the prefix takes that part of its sum from
// S[w][c][b][p] = sum_{j < c*delta} E[y_b[p+j]] . W_w[j]  (one pass over y_sample per get_reward, prefix sums
// over tap blocks) and only multiplies the taps from c*delta on with the rollout's tokens: a plain GEMM over the
// windows p in (g - (c+1)*delta, g - c*delta] with K = (w - c*delta)*E into `raw`, then bucket_finish adds S,
// applies BN + ReLU and folds the max into the pooled features.  Windows closer than delta to the give point (and
// those behind it) keep the fused conv + max GEMM with all their taps.  -28 % conv FLOPs at T = 50, delta = 8.
constexpr int BUCKET_MAX_C = 64;
// Grouped launches of the discriminator's conv products (NMTGAN_CONV_GROUP: 0 = one launch per product as before;
// default: products of fewer than this many 256x128 tiles share persistent launches, and every partial-window
// product runs there with the table term added in the epilogue - no raw buffer, no finish kernel)
static int conv_group_tiles() {
  static const int v = [] { const char* e = getenv("NMTGAN_CONV_GROUP"); return e ? atoi(e) : 1024; }();
  return v;
}
struct BucketReuse {
  int delta;
  void* ye;                                    // y_sample sentence image [B*T][E], op dtype
  float* S; float* raw;
  int64_t s_off[NMTGAN_MAX_WIDTHS][BUCKET_MAX_C];   // float offset of S[width i][c], c = 1..; [i][0] unused
};
static int bucket_delta(const nmtgan_reward_plan* p) {          // NMTGAN_CONV_BUCKET overrides the plan field
  static const int env = [] { const char* e = getenv("NMTGAN_CONV_BUCKET"); return e ? atoi(e) : -1000; }();
  if (env != -1000) return env;
  // default: blocks of 4 taps with grouped launches (a finer cut multiplies fewer prefix taps, and the many small
  // products it makes share launches: conv 81 -> 79 ms at T = 50; blocks of 2 or 1 lose more to ragged tiles than they
  // save), blocks of 8 when every product is a launch of its own
  return p->conv_bucket_taps == 0 ? (conv_group_tiles() > 0 ? 4 : 8) : p->conv_bucket_taps;
}

// highway features of `rows` sentences ids[rows][T] -> feat[rows][Fp]
static int disc_features_chunk(const nmtgan_discriminator* d, const nmtgan_disc_tower& tw, const int32_t* ids,
                               int rows, float* feat, DiscBufs& b, hipStream_t st, const PrefixReuse* reuse = nullptr,
                               const BucketReuse* bk = nullptr, const ConvTable* table = nullptr) {
  const int T = d->max_len, E = disc_ew(d), Fp = d->Fp, op = d->op_dtype;
  if (table) NMTGAN_REQUIRE(reuse != nullptr, "disc: the table form scores rollouts (needs their give points)");
  else NMTGAN_TRY(launch_disc_embed(tw.emb, op, &tw == &d->src ? d->vocab_s : d->vocab, ids, (int64_t)rows * T, E, b.e, st));
  if (reuse) {
    NMTGAN_TRY(launch_pool_init(reuse->pm, reuse->row_b, reuse->row_g, rows, T, Fp, b.pooled, st));
  } else {
    NMTGAN_TRY(launch_zero(b.pooled, (size_t)rows * Fp * sizeof(float), st));
  }
  const int group_tiles = conv_group_tiles();
  std::vector<nmtgan_gemm_args> grouped;          // conv products of this chunk that share persistent launches
  auto conv_max = [&](int i, int s0, int s1, int p0, int col) -> int {      // fused conv + BN + ReLU + max, all taps
    const int w = d->widths[i], P = T - w + 1 - p0;
    if (P <= 0 || s1 <= s0) return 0;
    nmtgan_gemm_args a{};
    a.op_dtype = op; a.epilogue = NMTGAN_EPI_CONVMAX;
    a.A = at(b.e, ((int64_t)s0 * T + p0) * E, dtype_size(op)); a.Wt = tw.conv_w[i];
    a.M = (s1 - s0) * P; a.N = d->n_filters[i]; a.K = w * E;
    a.lda = E; a.a_group = P; a.a_group_stride = (int64_t)T * E;
    a.scale = tw.conv_scale[i]; a.shift = tw.conv_shift[i]; a.pooled = b.pooled + (int64_t)s0 * Fp + col; a.ld_pool = Fp;
    if (group_tiles > 0) { grouped.push_back(a); return 0; }
    prof::Hint ph(prof::GEMM_CONV);
    return launch_gemm(&a, st);
  };
  if (table) {
    // every window that holds a sampled token, as gathered rows of the per-(token, tap) tables
    NMTGAN_TRY(launch_table_conv(*table, ids, reuse->row_b, reuse->row_g, rows, b.pooled, st));
  }
  int col = 0;
  for (int i = 0; i < (table ? 0 : d->n_widths); ++i) {
    const int w = d->widths[i];
    const bool buckets = reuse && bk && bk->delta > 0 && w > bk->delta;
    // sub-ranges of at most REUSE_ROWS rows.  Windows every row of a sub-range shares with y_sample are skipped:
    // p0 = first window that can contain a sampled token of any of its rows (p + w > smallest give_num); their
    // max comes from the prefix-max table, and a max does not care which of its arguments were recomputed.
    // With the partial-window reuse the SUM of a straddling window is split at a tap block, which rounds
    // differently from the unsplit sum, so there the split must not depend on which rows happen to share a
    // sub-range: it follows gq = (give_num / delta) * delta of the row itself, and a sub-range is a run of rows
    // with equal gq - every row's score is the same under any chunking, schedule or sharding.
    for (int s0 = 0; s0 < rows;) {
      auto g_low = [&](int lo, int hi) {
        int g_min = T;
        for (int r = lo; r < hi; ++r) g_min = reuse->row_g_host[r] < g_min ? reuse->row_g_host[r] : g_min;
        return g_min;
      };
      int s1 = s0 + REUSE_ROWS < rows ? s0 + REUSE_ROWS : rows;
      if (!reuse) s1 = rows;
      if (!buckets) {
        auto first_window = [&](int lo, int hi) {
          if (!reuse) return 0;
          const int g_min = g_low(lo, hi);
          return g_min - w + 1 > 0 ? g_min - w + 1 : 0;
        };
        const int p0 = first_window(s0, s1);
        while (s1 < rows) {
          const int nx = s1 + REUSE_ROWS < rows ? s1 + REUSE_ROWS : rows;
          if (first_window(s1, nx) != p0) break;
          s1 = nx;
        }
        NMTGAN_TRY(conv_max(i, s0, s1, p0, col));
      } else {
        const int dl = bk->delta, es = dtype_size(op), N = d->n_filters[i], Pw = T - w + 1;
        const int g_lo = reuse->row_g_host[s0] / dl * dl;             // gq of the run
        s1 = s0 + 1;
        while (s1 < rows && s1 - s0 < REUSE_ROWS && reuse->row_g_host[s1] / dl * dl == g_lo) ++s1;
        NMTGAN_TRY(conv_max(i, s0, s1, g_lo - dl + 1 > 0 ? g_lo - dl + 1 : 0, col));     // windows within delta of gq, or behind
        for (int c = 1; c * dl <= w - 1 && c < BUCKET_MAX_C; ++c) {
          int pa = g_lo - (c + 1) * dl + 1, pb = g_lo - c * dl;
          if (pa < g_lo - w + 1) pa = g_lo - w + 1;                   // windows below hold no sampled token
          if (pa < 0) pa = 0;
          if (pb > T - w) pb = T - w;
          if (pb < pa) continue;
          const int nq = pb - pa + 1;
          nmtgan_gemm_args a{};
          a.op_dtype = op; a.epilogue = NMTGAN_EPI_LINEAR;
          a.A = at(b.e, ((int64_t)s0 * T + pa + c * dl) * E, es); a.Wt = at(tw.conv_w[i], (int64_t)c * dl * E, es);
          a.M = (s1 - s0) * nq; a.N = N; a.K = (w - c * dl) * E; a.ldw = (int64_t)w * E;
          a.lda = E; a.a_group = nq; a.a_group_stride = (int64_t)T * E;
          if (group_tiles > 0 && ((int64_t)a.K * es) % 128 == 0 && (int64_t)a.K * es >= 512) {
            // the taps taken from y_sample (table S) are added in the epilogue, then BN + ReLU + max as for a whole
            // window (the grouped kernel's k-tiles are 128 bytes, at least four of them; a shorter product - one or two
            // taps of a narrow embedding - keeps the raw buffer + finish kernel; which form a product takes depends
            // on its width and tap block alone, and both add the same fp32 terms in the same order)
            a.epilogue = NMTGAN_EPI_CONVMAX;
            a.scale = tw.conv_scale[i]; a.shift = tw.conv_shift[i]; a.pooled = b.pooled + (int64_t)s0 * Fp + col; a.ld_pool = Fp;
            a.add = bk->S + bk->s_off[i][c]; a.add_blk = reuse->row_b + s0; a.add_rows = Pw; a.add_row0 = pa;
            grouped.push_back(a);
            continue;
          }
          a.n_seg = 1; a.seg_width = N; a.out[0] = bk->raw; a.ldo[0] = N; a.out_f32[0] = 1; a.seg_scale[0] = 1.f;
          {
            prof::Hint ph(prof::GEMM_CONV);
            NMTGAN_TRY(launch_gemm(&a, st));
          }
          prof::Hint ph(prof::GEMM_CONV);
          NMTGAN_TRY(launch_bucket_finish(bk->raw, bk->S + bk->s_off[i][c], reuse->row_b + s0, s1 - s0, nq, N, Pw, pa,
                                          tw.conv_scale[i], tw.conv_shift[i], b.pooled + (int64_t)s0 * Fp + col, Fp, st));
        }
      }
      s0 = s1;
    }
    col += d->n_filters[i];
  }
  if (!grouped.empty()) {
    prof::Hint ph(prof::GEMM_CONV);
    NMTGAN_TRY(launch_gemm_group(grouped.data(), (int)grouped.size(), st, group_tiles));
  }
  // highway input in the operand type; split form: pooled as [hi | lo | hi] against Wt[Fp][3 Fp] = [hi | hi | lo]
  const int Kh = d->gemm_split ? 3 * Fp : Fp;
  if (d->gemm_split) {
    prof::Hint ph(prof::DISC_SMALL);
    NMTGAN_TRY(launch_split3(b.pooled, Fp, rows, Fp, b.pooled_op, st));
  } else {
    NMTGAN_TRY(launch_cast_rows(b.pooled, b.pooled_op, op, (int64_t)rows * Fp, st));
  }
  nmtgan_dense wo{tw.hw_out, nullptr}, wt{tw.hw_tr, nullptr};
  {
    prof::Hint ph(prof::GEMM_HIGHWAY);
    NMTGAN_TRY(dense1(op, nullptr, b.pooled_op, Kh, rows, Kh, Fp, wo, b.ho, Fp, 1, 1.f, 0, st));
    NMTGAN_TRY(dense1(op, nullptr, b.pooled_op, Kh, rows, Kh, Fp, wt, b.ht, Fp, 1, 1.f, 0, st));
  }
  return launch_highway_gate(b.pooled, b.ho, b.ht, feat, (int64_t)rows * Fp, st);
}

// S tables of the partial-window reuse from the y_sample image: per width and tap block c one GEMM over the
// taps [(c-1)*delta, c*delta) of every window, then prefix sums over c
static int bucket_tables(const nmtgan_discriminator* d, const nmtgan_disc_tower& tw, const int32_t* y, int B,
                         const BucketReuse& bk, hipStream_t st) {
  const int T = d->max_len, E = disc_ew(d), op = d->op_dtype, es = dtype_size(op), dl = bk.delta;
  NMTGAN_TRY(launch_disc_embed(tw.emb, op, d->vocab, y, (int64_t)B * T, E, bk.ye, st));
  const int group_tiles = conv_group_tiles();
  std::vector<nmtgan_gemm_args> grouped;          // every (width, tap block) product first, then the prefix sums over blocks
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < d->n_widths; ++i) {
      const int w = d->widths[i], N = d->n_filters[i], Pw = T - w + 1;
      if (w <= dl) continue;
      for (int c = 1; c * dl <= w - 1 && c < BUCKET_MAX_C; ++c) {
        float* Sc = bk.S + bk.s_off[i][c];
        if (pass == 1) {
          if (c > 1) NMTGAN_TRY(launch_add_inplace(Sc, bk.S + bk.s_off[i][c - 1], (int64_t)B * Pw * N, st));
          continue;
        }
        nmtgan_gemm_args a{};
        a.op_dtype = op; a.epilogue = NMTGAN_EPI_LINEAR;
        a.A = at(bk.ye, (int64_t)(c - 1) * dl * E, es); a.Wt = at(tw.conv_w[i], (int64_t)(c - 1) * dl * E, es);
        a.M = B * Pw; a.N = N; a.K = dl * E; a.ldw = (int64_t)w * E;
        a.lda = E; a.a_group = Pw; a.a_group_stride = (int64_t)T * E;
        a.n_seg = 1; a.seg_width = N; a.out[0] = Sc; a.ldo[0] = N; a.out_f32[0] = 1; a.seg_scale[0] = 1.f;
        if (group_tiles > 0) { grouped.push_back(a); continue; }
        prof::Hint ph(prof::GEMM_CONV);
        NMTGAN_TRY(launch_gemm(&a, st));
      }
    }
    if (pass == 0 && !grouped.empty()) {
      prof::Hint ph(prof::GEMM_CONV);
      NMTGAN_TRY(launch_gemm_group(grouped.data(), (int)grouped.size(), st, group_tiles));
    }
  }
  return 0;
}

// Table form of the target tower's conv for the rollouts (disc_table.hip): P_w[v][j][:] = E[v] . W_w[j] for the whole
// vocabulary, one GEMM per width over the embedding table (M = vocab, K = one token's operand row, N = w * nfp: the
// filter bank re-laid in tap-major rows, so an output row is a token's whole entry [w][nfp], pad columns zero).
// Split-precision operands give an fp32 table, 16-bit operands a table of their own type, fp32 operands an fp32 table.
static int conv_table_dtype(const nmtgan_discriminator* d) { return d->gemm_split ? NMTGAN_F32 : d->op_dtype; }
static int conv_table_nfp(const nmtgan_discriminator* d, int nf) {        // table rows padded to whole 128-byte lines
  const int per = 128 / dtype_size(conv_table_dtype(d));
  return (nf + per - 1) / per * per;
}
static int64_t conv_table_bytes(const nmtgan_discriminator* d) {
  int64_t n = 0;
  for (int i = 0; i < d->n_widths; ++i)
    n += align_up((size_t)d->vocab * d->widths[i] * conv_table_nfp(d, d->n_filters[i]) * dtype_size(conv_table_dtype(d)), 256);
  return n;
}
static int build_conv_table(const nmtgan_discriminator* d, const nmtgan_disc_tower& tw, const ConvTable& t, hipStream_t st) {
  const int E = disc_ew(d), op = d->op_dtype, es = dtype_size(op);
  for (int i = 0; i < d->n_widths; ++i) {
    const int w = d->widths[i];
    NMTGAN_TRY(launch_table_weights(tw.conv_w[i], w, d->n_filters[i], t.nfp[i], E * es, t.wt[i], st));
    nmtgan_gemm_args a{};
    a.op_dtype = op; a.epilogue = NMTGAN_EPI_LINEAR;
    a.A = tw.emb; a.lda = E; a.M = d->vocab; a.K = E; a.N = w * t.nfp[i]; a.Wt = t.wt[i];
    a.n_seg = 1; a.seg_width = a.N; a.out[0] = t.tab[i]; a.ldo[0] = a.N;
    a.out_f32[0] = t.dtype == NMTGAN_F32 ? 1 : 0; a.seg_scale[0] = 1.f;
    prof::Hint ph(prof::GEMM_CONV);
    NMTGAN_TRY(launch_gemm(&a, st));
  }
  return 0;
}

int conv_table_build(const nmtgan_discriminator* d, const ConvTable* t, hipStream_t st) {
  NMTGAN_TRY(check_disc(d));
  NMTGAN_REQUIRE(t != nullptr && t->n_widths == d->n_widths && t->vocab == d->vocab && t->dtype == conv_table_dtype(d),
                 "conv_table_build: the table's widths, vocabulary or entry type do not match the discriminator");
  for (int i = 0; i < d->n_widths; ++i) {
    NMTGAN_REQUIRE(t->w[i] == d->widths[i] && t->nf[i] == d->n_filters[i] && t->nfp[i] >= t->nf[i] &&
                       (t->nfp[i] * dtype_size(t->dtype)) % 16 == 0 && t->tab[i] && t->wt[i] &&
                       (uintptr_t)t->tab[i] % 16 == 0 && (uintptr_t)t->wt[i] % 16 == 0,
                   "conv_table_build: bad entry for width %d", d->widths[i]);
  }
  return build_conv_table(d, d->tgt, *t, st);
}

// C[b*T + p][Fp] = relu(bn(conv)) of every window p of rows sentences (one "sentence" of rows*T tokens with
// a pooling group of one window; windows that run over a sentence end are computed and never read)
static int disc_window_values(const nmtgan_discriminator* d, const nmtgan_disc_tower& tw, const int32_t* ids, int rows,
                              float* Cw, DiscBufs& b, hipStream_t st) {
  const int T = d->max_len, E = disc_ew(d), Fp = d->Fp, op = d->op_dtype;
  NMTGAN_TRY(launch_disc_embed(tw.emb, op, d->vocab, ids, (int64_t)rows * T, E, b.e, st));
  NMTGAN_TRY(launch_zero(Cw, (size_t)rows * T * Fp * sizeof(float), st));
  int col = 0;
  const int group_tiles = conv_group_tiles();
  std::vector<nmtgan_gemm_args> grouped;
  for (int i = 0; i < d->n_widths; ++i) {
    const int w = d->widths[i];
    nmtgan_gemm_args a{};
    a.op_dtype = op; a.epilogue = NMTGAN_EPI_CONVMAX;
    a.A = b.e; a.Wt = tw.conv_w[i]; a.M = rows * T - w + 1; a.N = d->n_filters[i]; a.K = w * E;
    a.lda = E; a.a_group = 1; a.a_group_stride = E;
    a.scale = tw.conv_scale[i]; a.shift = tw.conv_shift[i]; a.pooled = Cw + col; a.ld_pool = Fp;
    col += d->n_filters[i];
    if (group_tiles > 0) { grouped.push_back(a); continue; }
    prof::Hint ph(prof::GEMM_CONV);
    NMTGAN_TRY(launch_gemm(&a, st));
  }
  if (!grouped.empty()) {
    prof::Hint ph(prof::GEMM_CONV);
    NMTGAN_TRY(launch_gemm_group(grouped.data(), (int)grouped.size(), st, group_tiles));
  }
  return 0;
}

int disc_workspace(const nmtgan_discriminator* d, int chunk, size_t* bytes) {
  NMTGAN_TRY(check_disc(d));
  NMTGAN_REQUIRE(chunk > 0, "disc: chunk_rows must be positive");
  Arena ar(nullptr, 0);
  DiscBufs b;
  carve_disc(ar, d, chunk, b);
  *bytes = align_up(ar.off, 256);
  return 0;
}

int disc_features(const nmtgan_discriminator* d, int source, const int32_t* ids, int M, float* feat, int chunk,
                  void* ws, size_t ws_bytes, hipStream_t st) {
  NMTGAN_TRY(check_disc(d));
  NMTGAN_REQUIRE(chunk > 0 && M >= 0 && ids && feat, "disc_features: bad arguments");
  Arena ar(ws, ws_bytes);
  DiscBufs b;
  carve_disc(ar, d, chunk, b);
  NMTGAN_REQUIRE(ws != nullptr && !ar.overflow, "disc_features: workspace too small (%zu < %zu)", ws_bytes, ar.off);
  const nmtgan_disc_tower& tw = source ? d->src : d->tgt;
  for (int r0 = 0; r0 < M; r0 += chunk) {
    const int rows = M - r0 < chunk ? M - r0 : chunk;
    NMTGAN_TRY(disc_features_chunk(d, tw, ids + (int64_t)r0 * d->max_len, rows, feat + (int64_t)r0 * d->Fp, b, st));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------ rollout reward
// Fork/join events of the two-stream schedule: a thread-local pool per device, created on first use (host
// objects, no device memory), re-recorded call after call (a wait captures the record that precedes it).
struct SyncEvents {
  size_t used = 0;
  int order(hipStream_t first, hipStream_t then) {        // work enqueued on `then` from now on runs after
    static thread_local std::vector<hipEvent_t> pool[16]; //   everything enqueued on `first` so far
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) NMTGAN_FAIL(-3, "two-stream: bad device");
    std::vector<hipEvent_t>& ev = pool[dev];
    if (used == ev.size()) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) NMTGAN_FAIL(-3, "two-stream: event create failed");
      ev.push_back(e);
    }
    hipEvent_t e = ev[used++];
    if (hipEventRecord(e, first) != hipSuccess || hipStreamWaitEvent(then, e, 0) != hipSuccess)
      NMTGAN_FAIL(-3, "two-stream: event record/wait failed");
    return 0;
  }
};

struct RewardBufs {
  Act act;
  void *kbuf, *vbuf, *crossK, *crossV, *shK, *shV, *prK, *prV;
  void *shVt, *crossVt; float *o_part, *m_part, *l_part; int group_R, Tp, Lsp;
  float* logits; float* prefix_logits;
  void* fused_ws; bool fused;                 // logits never materialised during the rollout steps
  int32_t* Y; int32_t* tok_cur;
  float *feat_s, *feat_t, *p_rows, *p_base;
  float *win_vals, *win_pm; bool reuse;        // prefix-window reuse in the discriminator
  BucketReuse bucket;                          // partial-window reuse (delta = 0: off)
  ConvTable table; bool use_table;             // table form of the rollouts' conv (replaces the two above's GEMMs)
  DiscBufs disc;
  int64_t act_rows, logits_rows, pr_rows;
  int disc_chunk;
};

static int carve_reward(Arena& ar, const nmtgan_generator* g, const nmtgan_discriminator* d,
                        const nmtgan_reward_plan* p, RewardBufs& b) {
  const int es = dtype_size(g->op_dtype), dm = g->d, N = g->n_blocks;
  const int64_t enc_rows = (int64_t)p->B * p->Ls, pre_rows = (int64_t)p->B * p->T;
  int64_t act_rows = enc_rows > pre_rows ? enc_rows : pre_rows;
  if (p->n_steps >= 1 && p->live[1] > act_rows) act_rows = p->live[1];
  b.act_rows = act_rows;
  carve_act(ar, act_rows, dm, es, b.act, g->gemm_split != 0);
  b.kbuf = ar.take(enc_rows * dm * es);
  b.vbuf = ar.take(enc_rows * dm * es);
  b.crossK = ar.take(N * enc_rows * dm * es);
  b.crossV = ar.take(N * enc_rows * dm * es);
  b.shK = ar.take(N * pre_rows * dm * es);
  b.shV = ar.take(N * pre_rows * dm * es);
  // group attention path: 16-bit operands, <= 32 rollouts per (sentence, prefix) group, MFMA-friendly head dim
  const int dk = dm / g->heads;
  // (the split-precision tier runs the fp32 form of the group kernel; the fp32 parity tier keeps the per-row kernel)
  const bool group_dtype = dtype_size(g->op_dtype) == 2 ? (dk == 32 || dk == 64 || dk == 128)
                                                        : (g->gemm_split && (dk == 32 || dk == 64));
  b.group_R = (group_dtype && p->rollnum <= 32 && p->n_steps >= 1) ? p->rollnum : 0;
  b.Tp = (p->T + 31) / 32 * 32;
  b.Lsp = (p->Ls + 31) / 32 * 32;
  b.shVt = b.crossVt = nullptr;
  b.o_part = b.m_part = b.l_part = nullptr;
  if (b.group_R > 0) {
    b.crossVt = ar.take((size_t)N * p->B * dm * b.Lsp * es);
    if (self_group_pass()) {
      b.shVt = ar.take((size_t)N * p->B * dm * b.Tp * es);
      b.o_part = ar.get<float>(act_rows * dm);
      b.m_part = ar.get<float>(act_rows * g->heads);
      b.l_part = ar.get<float>(act_rows * g->heads);
    }
  }
  b.pr_rows = p->n_steps >= 1 ? p->step_off_host[p->n_steps] : 0;
  b.prK = ar.take((size_t)N * b.pr_rows * dm * es);
  b.prV = ar.take((size_t)N * b.pr_rows * dm * es);
  int64_t lrows = p->logits_chunk_rows > 0 ? p->logits_chunk_rows : 4096;
  if (p->n_steps >= 1 && p->live[1] < lrows) lrows = p->live[1];
  if (lrows < 1) lrows = 1;
  static const bool fused_env = [] { const char* e = getenv("NMTGAN_FUSED_SAMPLE"); return !e || atoi(e) != 0; }();
  b.fused = fused_env && p->n_steps >= 1 &&
            (g->gemm_split ? gemm_stats_supported(NMTGAN_F16, g->vocab_dst, 3 * dm) : gemm_stats_supported(g->op_dtype, g->vocab_dst, dm));
  b.fused_ws = b.fused ? ar.take(logits_sample_workspace(p->live[1], g->vocab_dst)) : nullptr;
  if (b.fused) lrows = 1;
  b.logits_rows = lrows;
  b.logits = ar.get<float>(lrows * g->vocab_dst);
  b.prefix_logits = p->prefix_logits ? p->prefix_logits : ar.get<float>(pre_rows * g->vocab_dst);
  b.Y = p->rollouts ? p->rollouts : ar.get<int32_t>((int64_t)(p->n_rows > 0 ? p->n_rows : 1) * p->T);
  b.tok_cur = ar.get<int32_t>(p->n_rows > 0 ? p->n_rows : 1);
  b.disc_chunk = p->disc_chunk_rows > 0 ? p->disc_chunk_rows : 8192;
  const int64_t most = p->n_rows > p->B ? p->n_rows : p->B;
  if (b.disc_chunk > most) b.disc_chunk = (int)most;
  b.feat_s = ar.get<float>((int64_t)p->B * d->Fp);
  b.feat_t = ar.get<float>((int64_t)b.disc_chunk * d->Fp);
  b.p_rows = p->p_rows ? p->p_rows : ar.get<float>(p->n_rows > 0 ? p->n_rows : 1);
  b.p_base = p->p_base ? p->p_base : ar.get<float>(p->B);
  carve_disc(ar, d, b.disc_chunk, b.disc);
  static const bool reuse_env = [] { const char* e = getenv("NMTGAN_CONV_REUSE"); return !e || atoi(e) != 0; }();
  b.reuse = reuse_env && p->row_g_host != nullptr && p->n_rows > 0;
  b.win_vals = b.win_pm = nullptr;
  if (b.reuse) {
    const int64_t wrows = (int64_t)(p->B < b.disc_chunk ? p->B : b.disc_chunk) * p->T;
    b.win_vals = ar.get<float>(wrows * d->Fp);
    b.win_pm = ar.get<float>((int64_t)p->B * p->T * d->Fp);
  }
  // Table form of the rollouts' conv (NMTGAN_CONV_TABLE=0 or plan.conv_table < 0 switch it off; it needs the give
  // points on the host like the prefix reuse, and tables that fit the cap).  By default only
  //  - when the rollouts hold at least four tokens per vocabulary entry: the tables cost one product per (vocabulary
  //    entry, tap), the conv GEMMs one per (rollout token, tap) less what the prefix reuse saves;
  //  - for a 16-bit or split discriminator: with fp32 operands - the parity tier - one fp32 rounding per (token, tap)
  //    entry is a little more than the matrix cores' own accumulation loses, so that tier keeps the GEMMs unless the
  //    plan asks for the tables.
  static const bool table_env = [] { const char* e = getenv("NMTGAN_CONV_TABLE"); return !e || atoi(e) != 0; }();
  const int64_t table_cap = p->conv_table_max_bytes > 0 ? p->conv_table_max_bytes : (int64_t)96 << 30;
  b.table = ConvTable{};
  const bool table_pays = (int64_t)p->n_rows * p->T >= 4 * (int64_t)d->vocab;
  b.use_table = table_env && (p->conv_table > 0 || (p->conv_table == 0 && d->op_dtype != NMTGAN_F32 && table_pays)) &&
                b.reuse && p->T <= TABLE_MAX_LEN && conv_table_bytes(d) <= table_cap;
  if (b.use_table) {
    ConvTable& t = b.table;
    t.n_widths = d->n_widths; t.T = p->T; t.Fp = d->Fp; t.vocab = d->vocab; t.dtype = conv_table_dtype(d);
    for (int i = 0; i < d->n_widths; ++i) {
      const int w = d->widths[i];
      t.w[i] = w; t.nf[i] = d->n_filters[i]; t.nfp[i] = conv_table_nfp(d, d->n_filters[i]);
      t.scale[i] = d->tgt.conv_scale[i]; t.shift[i] = d->tgt.conv_shift[i];
      t.tab[i] = ar.take((size_t)d->vocab * w * t.nfp[i] * dtype_size(t.dtype));
      t.S[i] = ar.get<float>((size_t)p->B * (p->T - w + 1) * (w > 1 ? w - 1 : 1) * t.nfp[i]);
      t.wt[i] = ar.take((size_t)w * t.nfp[i] * disc_ew(d) * dtype_size(d->op_dtype));
    }
  }
  b.bucket = BucketReuse{};
  int max_w = 0, max_n = 0;
  for (int i = 0; i < d->n_widths; ++i) {
    max_w = d->widths[i] > max_w ? d->widths[i] : max_w;
    max_n = d->n_filters[i] > max_n ? d->n_filters[i] : max_n;
  }
  const int dl = bucket_delta(p);
  const int des = dtype_size(d->op_dtype);
  if (b.reuse && !b.use_table && dl > 0 && (d->dim_word * des) % 128 == 0 && (max_w - 1) / dl < BUCKET_MAX_C && max_w > dl) {
    b.bucket.delta = dl;
    b.bucket.ye = ar.take((size_t)p->B * p->T * disc_ew(d) * des);
    int64_t off = 0;
    for (int i = 0; i < d->n_widths; ++i)
      for (int c = 1; c * dl <= d->widths[i] - 1 && c < BUCKET_MAX_C; ++c) {
        b.bucket.s_off[i][c] = off;
        off += (int64_t)p->B * (p->T - d->widths[i] + 1) * d->n_filters[i];
        off = (off + 63) / 64 * 64;
      }
    b.bucket.S = ar.get<float>(off > 0 ? off : 1);
    b.bucket.raw = ar.get<float>((int64_t)REUSE_ROWS * dl * max_n);
  }
  return 0;
}

static int check_plan(const nmtgan_generator* g, const nmtgan_discriminator* d, const nmtgan_reward_plan* p) {
  NMTGAN_TRY(check_generator(g));
  NMTGAN_TRY(check_disc(d));
  NMTGAN_REQUIRE(p != nullptr, "plan: null");
  // the two networks share no activation buffer and no scratch; the split-precision generator (fp32 storage, fp16
  // products) pairs with a 16-bit discriminator (plain or split form), every other tier packs both alike
  if (g->gemm_split)
    NMTGAN_REQUIRE(dtype_size(d->op_dtype) == 2, "a gemm_split generator pairs with a 16-bit discriminator (plain or gemm_split)");
  else
    NMTGAN_REQUIRE(g->op_dtype == d->op_dtype, "generator and discriminator must be packed in the same op dtype");
  NMTGAN_REQUIRE(p->B > 0 && p->Ls > 0 && p->T > 1 && p->rollnum > 0, "plan: bad B/Ls/T/rollnum");
  NMTGAN_REQUIRE(p->T == d->max_len, "plan: max_len %d != discriminator max_len %d", p->T, d->max_len);
  NMTGAN_REQUIRE(p->Ls <= g->max_pos && p->T <= g->max_pos, "plan: length exceeds timing table");
  NMTGAN_REQUIRE(p->n_rows >= 0 && p->n_rows % p->rollnum == 0, "plan: n_rows must be a multiple of rollnum");
  NMTGAN_REQUIRE(p->n_steps >= 0 && p->n_steps <= (p->T > 2 ? p->T - 2 : 0), "plan: n_steps must be in [0, T-2]");
  NMTGAN_REQUIRE(p->live && p->step_off_host, "plan: null host tables");
  NMTGAN_REQUIRE(p->live[0] == p->n_rows, "plan: live[0] != n_rows");
  int64_t cum = 0;
  NMTGAN_REQUIRE(p->step_off_host[0] == 0, "plan: step_off[0] must be 0");
  for (int s = 1; s <= p->n_steps; ++s) {
    NMTGAN_REQUIRE(p->live[s] > 0 && p->live[s] <= p->live[s - 1], "plan: live[] must be positive and non-increasing");
    cum += p->live[s];
    NMTGAN_REQUIRE(p->step_off_host[s] == cum, "plan: step_off[%d] inconsistent with live[]", s);
  }
  return 0;
}

int reward_workspace(const nmtgan_generator* g, const nmtgan_discriminator* d, const nmtgan_reward_plan* p,
                     size_t* bytes) {
  NMTGAN_TRY(check_plan(g, d, p));
  Arena ar(nullptr, 0);
  RewardBufs b;
  NMTGAN_TRY(carve_reward(ar, g, d, p, b));
  *bytes = align_up(ar.off, 256);
  return 0;
}

int rollout_reward(const nmtgan_generator* g, const nmtgan_discriminator* d, const nmtgan_reward_plan* p, void* ws,
                   size_t ws_bytes, hipStream_t st) {
  NMTGAN_TRY(check_plan(g, d, p));
  NMTGAN_REQUIRE(p->x && p->x_pad && p->x_pad_to_T && p->y && p->dec_in && p->mask && p->grp_first && p->rewards,
                 "plan: null input");
  NMTGAN_REQUIRE(p->n_rows == 0 || (p->row_b && p->row_g && p->row_gp1 && p->row_lrow && p->row_uoff && p->uniforms),
                 "plan: null row tables");
  NMTGAN_REQUIRE(p->n_steps == 0 || p->step_off, "plan: null device step_off");
  Arena ar(ws, ws_bytes);
  RewardBufs b;
  NMTGAN_TRY(carve_reward(ar, g, d, p, b));
  NMTGAN_REQUIRE(ws != nullptr && !ar.overflow, "rollout_reward: workspace too small (%zu < %zu)", ws_bytes, ar.off);
  const int B = p->B, Ls = p->Ls, T = p->T, dm = g->d, V = g->vocab_dst, op = g->op_dtype;
  
  // 1. encoder once, cross K/V once per block (the reference re-encodes on every sess.run)
  NMTGAN_TRY(encode(g, p->x, p->x_pad, B, Ls, b.act, b.kbuf, b.vbuf, st));
  NMTGAN_TRY(cross_kv(g, b.act.sp(), b.act.xo, B * Ls, b.crossK, b.crossV, st));
  // 2. teacher-forced pass over y_sample: shared K/V of positions 0..T-1 and the prefix logits
  NMTGAN_TRY(prefill(g, p->dec_in, p->x_pad, B, Ls, T, b.act, b.shK, b.shV, b.crossK, b.crossV, st));
  NMTGAN_TRY(logits_gemm(g, b.act.sp(), b.act.xo, B * T, b.prefix_logits, st));
  if (b.group_R > 0) {                 // transposed copies of the shared values for the MFMA group kernel
    const int es = dtype_size(op);
    for (int l = 0; l < g->n_blocks; ++l) {
      if (b.shVt)
        NMTGAN_TRY(launch_transpose_v(at(b.shV, (int64_t)l * B * T * dm, es), at(b.shVt, (int64_t)l * B * dm * b.Tp, es), B,
                                      T, b.Tp, dm, es, st));
      NMTGAN_TRY(launch_transpose_v(at(b.crossV, (int64_t)l * B * Ls * dm, es),
                                    at(b.crossVt, (int64_t)l * B * dm * b.Lsp, es), B, Ls, b.Lsp, dm, es, st));
    }
  }
  // Two-stream schedule: everything the discriminator does goes to `aux`; it only needs finished rollouts.
  hipStream_t aux = p->aux_stream ? (hipStream_t)p->aux_stream : st;
  const bool ovl = aux != st;
  SyncEvents ev;
  static const int env_wgs = [] { const char* e = getenv("NMTGAN_AUX_WGS"); return e ? atoi(e) : 0; }();
  static const int env_min = [] { const char* e = getenv("NMTGAN_AUX_MIN_ROWS"); return e ? atoi(e) : 0; }();
  const int aux_wgs = env_wgs > 0 ? env_wgs : (p->aux_gemm_wgs > 0 ? p->aux_gemm_wgs : 224);
  const int aux_min = env_min > 0 ? env_min : (p->aux_min_rows > 0 ? p->aux_min_rows : 6144);
  struct CapGuard { ~CapGuard() { set_gemm_wg_cap(0); } } cap_guard;

  // discriminator work that does not depend on the rollouts: source tower once per sentence, target tower of
  // y_sample itself, and the window values of y_sample the rollouts re-use
  auto disc_static = [&](hipStream_t ds) -> int {
    for (int r0 = 0; r0 < B; r0 += b.disc_chunk) {
      const int rows = B - r0 < b.disc_chunk ? B - r0 : b.disc_chunk;
      NMTGAN_TRY(disc_features_chunk(d, d->src, p->x_pad_to_T + (int64_t)r0 * T, rows, b.feat_s + (int64_t)r0 * d->Fp,
                                     b.disc, ds));
    }
    for (int r0 = 0; r0 < B; r0 += b.disc_chunk) {
      const int rows = B - r0 < b.disc_chunk ? B - r0 : b.disc_chunk;
      NMTGAN_TRY(disc_features_chunk(d, d->tgt, p->y + (int64_t)r0 * T, rows, b.feat_t, b.disc, ds));
      NMTGAN_TRY(launch_disc_head(b.feat_t, b.feat_s + (int64_t)r0 * d->Fp, nullptr, rows, d->Fp, d->head_w, d->head_b,
                                  b.p_base + r0, nullptr, ds));
    }
    if (b.use_table) NMTGAN_TRY(build_conv_table(d, d->tgt, b.table, ds));
    if (b.reuse) {
      for (int r0 = 0; r0 < B; r0 += b.disc_chunk) {
        const int rows = B - r0 < b.disc_chunk ? B - r0 : b.disc_chunk;
        if (b.use_table) {
          ConvTable t = b.table;                  // the chunk's slice of the partial sums
          for (int i = 0; i < t.n_widths; ++i) t.S[i] += (int64_t)r0 * (T - t.w[i] + 1) * (t.w[i] - 1) * t.nfp[i];
          NMTGAN_TRY(launch_table_prefix(t, p->y + (int64_t)r0 * T, rows, b.win_vals, ds));
        } else
        NMTGAN_TRY(disc_window_values(d, d->tgt, p->y + (int64_t)r0 * T, rows, b.win_vals, b.disc, ds));
        NMTGAN_TRY(launch_prefix_max(b.win_vals, b.win_pm + (int64_t)r0 * T * d->Fp, rows, T, d->Fp, d->n_widths,
                                     d->widths, d->n_filters, ds));
      }
      if (b.bucket.delta > 0) NMTGAN_TRY(bucket_tables(d, d->tgt, p->y, B, b.bucket, ds));
    }
    return 0;
  };
  // target tower + head of the rollout rows [lo, hi)
  auto disc_rows = [&](int lo, int hi, hipStream_t ds) -> int {
    const int dl = b.reuse ? b.bucket.delta : 0;
    for (int r0 = lo, rows = 0; r0 < hi; r0 += rows) {
      rows = hi - r0 < b.disc_chunk ? hi - r0 : b.disc_chunk;
      if (dl > 0 && r0 + rows < hi) {
        // end the chunk where give_num / delta changes (the partial-window reuse works on runs of equal quotient,
        // and a run cut in two costs a second, small set of conv GEMMs), unless that would leave under half a chunk
        int cut = r0 + rows;
        const int q = p->row_g_host[cut] / dl;
        while (cut > r0 && p->row_g_host[cut - 1] / dl == q) --cut;
        if (cut - r0 >= b.disc_chunk / 2) rows = cut - r0;
      }
      PrefixReuse pr{b.win_pm, p->row_b + r0, p->row_g + r0, b.reuse ? p->row_g_host + r0 : nullptr};
      NMTGAN_TRY(disc_features_chunk(d, d->tgt, b.Y + (int64_t)r0 * T, rows, b.feat_t, b.disc, ds, b.reuse ? &pr : nullptr,
                                     b.reuse && b.bucket.delta > 0 ? &b.bucket : nullptr, b.use_table ? &b.table : nullptr));
      NMTGAN_TRY(launch_disc_head(b.feat_t, b.feat_s, p->row_b + r0, rows, d->Fp, d->head_w, d->head_b, b.p_rows + r0,
                                  nullptr, ds));
    }
    return 0;
  };
  if (ovl) {
    NMTGAN_TRY(ev.order(st, aux));          // the caller's uploads and the previous call's readers come first
    set_gemm_wg_cap(aux_wgs);
    NMTGAN_TRY(disc_static(aux));
    set_gemm_wg_cap(0);
  }
  // 3. rollouts
  if (p->n_rows > 0) {
    NMTGAN_TRY(launch_init_rollouts(p->y, p->row_b, p->row_g, p->n_rows, T, b.Y, st));
    // first sampled token of row (b,r,g): position g, logits shared by all rolls of (b,g)
    nmtgan_sample_args s{};
    s.logits = b.prefix_logits; s.ld = V; s.V = V; s.M = p->n_rows; s.row_map = p->row_lrow;
    s.u = p->uniforms; s.u_off = p->row_uoff; s.u_add = 0; s.out = b.tok_cur;
    NMTGAN_TRY(launch_sample_rows(&s, st));
    NMTGAN_TRY(launch_scatter_tokens(b.tok_cur, p->row_g, 0, p->n_rows, T, b.Y, st));
    // rows [live[s+1], live[s]) are complete once step s has scattered its tokens: hand them to aux
    int done_hi = p->n_rows;
    auto hand_over = [&](int step) -> int {
      if (!ovl) return 0;
      const int lo = step < p->n_steps ? p->live[step + 1] : 0;
      if (done_hi - lo < aux_min && step < p->n_steps) return 0;
      if (done_hi == lo) return 0;
      NMTGAN_TRY(ev.order(st, aux));
      set_gemm_wg_cap(step < p->n_steps ? aux_wgs : 0);
      const int rc = disc_rows(lo, done_hi, aux);
      set_gemm_wg_cap(0);
      done_hi = lo;
      return rc;
    };
    NMTGAN_TRY(hand_over(0));
    for (int step = 1; step <= p->n_steps; ++step) {
      const int M = p->live[step];
      nmtgan_embed_args e{};
      e.table = g->emb_dst; e.timing = g->timing; e.vocab = V; e.max_pos = g->max_pos;
      e.tok = b.Y; e.tok_ld = T; e.tok_col = p->row_g; e.tok_col_add = step - 1;     // input token y_{g+s-1}
      e.pos = p->row_g; e.pos_add = step;                                            // position g+s
      e.mult = 2.0f * g->emb_mult; e.M = M; e.d = dm; e.out_f32 = b.act.xf; e.out_op = b.act.xo; e.op_dtype = op;
      e.out_split = g->gemm_split ? b.act.img : nullptr;
      NMTGAN_TRY(launch_embed(&e, st));
      DecAttn c{};
      c.shK = b.shK; c.shV = b.shV; c.len_sh = T; c.sh_layer_stride = (int64_t)B * T * dm;
      c.seg1_blk = p->row_b; c.seg1_cnt = p->row_gp1;
      c.prK = b.prK; c.prV = b.prV; c.pr_layer_stride = b.pr_rows * dm; c.step_off = p->step_off; c.n_steps2 = step;
      c.wrK = b.prK; c.wrV = b.prV; c.wr_layer_stride = b.pr_rows * dm; c.wr_off = p->step_off_host[step - 1] * dm;
      c.crossK = b.crossK; c.crossV = b.crossV; c.Ls = Ls; c.cross_layer_stride = (int64_t)B * Ls * dm;
      c.cross_blk = p->row_b; c.x_pad = p->x_pad; c.step_mode = 1;
      c.group_R = b.group_R; c.shVt = b.shVt; c.sh_Lp = b.Tp; c.sht_layer_stride = (int64_t)B * dm * b.Tp;
      c.crossVt = b.crossVt; c.cross_Lp = b.Lsp; c.crosst_layer_stride = (int64_t)B * dm * b.Lsp;
      c.o_part = b.o_part; c.m_part = b.m_part; c.l_part = b.l_part;
      if (b.group_R > 0 && p->grp_table && p->grp_live && p->grp_slots > 0 && p->grp_slots * b.group_R <= 32) {
        c.slot_table = p->grp_table; c.slots = p->grp_slots; c.n_entries = p->grp_live[step];
      }
      NMTGAN_TRY(decoder_pass(g, M, b.act, c, st));
      if (b.fused && g->gemm_split) {    // the split operands through the fp16 form of the fused path: K = 3 d;
                                         // the last add_layernorm of the pass left x's image in act.img
        NMTGAN_TRY(launch_logits_sample(NMTGAN_F16, b.act.img, g->softmax_w, M, V, 3 * dm, p->uniforms, p->row_uoff, step,
                                        b.tok_cur, b.fused_ws, st));
      } else if (b.fused) {
        NMTGAN_TRY(launch_logits_sample(op, b.act.xo, g->softmax_w, M, V, dm, p->uniforms, p->row_uoff, step, b.tok_cur,
                                        b.fused_ws, st));
      } else for (int r0 = 0; r0 < M; r0 += (int)b.logits_rows) {
        const int rows = M - r0 < b.logits_rows ? M - r0 : (int)b.logits_rows;
        NMTGAN_TRY(logits_gemm(g, b.act.sp(), at(b.act.xo, (int64_t)r0 * dm, dtype_size(op)), rows, b.logits, st));
        nmtgan_sample_args q{};
        q.logits = b.logits; q.ld = V; q.V = V; q.M = rows;
        q.u = p->uniforms; q.u_off = p->row_uoff + r0; q.u_add = step; q.out = b.tok_cur + r0;
        NMTGAN_TRY(launch_sample_rows(&q, st));
      }
      NMTGAN_TRY(launch_scatter_tokens(b.tok_cur, p->row_g, step, M, T, b.Y, st));
      NMTGAN_TRY(hand_over(step));
    }
  }
  // 4. discriminator (single-stream schedule: after the rollouts)
  if (!ovl) {
    NMTGAN_TRY(disc_static(st));
    NMTGAN_TRY(disc_rows(0, p->n_rows, st));
  } else {
    NMTGAN_TRY(ev.order(aux, st));
  }
  // 5. per-prefix averaging
  return launch_reward_reduce(b.p_rows, b.p_base, p->grp_first, p->mask, B, T, p->rollnum, p->use_bias, p->bias_num,
                              p->rewards, st);
}

// ------------------------------------------------------------------------------------------ teacher-forced logits
struct TfBufs { Act act; void *kbuf, *vbuf, *crossK, *crossV, *shK, *shV; };
static void carve_tf(Arena& ar, const nmtgan_generator* g, int B, int Ls, int T, TfBufs& b) {
  const int es = dtype_size(g->op_dtype), dm = g->d, N = g->n_blocks;
  const int64_t enc_rows = (int64_t)B * Ls, pre_rows = (int64_t)B * T;
  carve_act(ar, enc_rows > pre_rows ? enc_rows : pre_rows, dm, es, b.act, g->gemm_split != 0);
  b.kbuf = ar.take(enc_rows * dm * es);
  b.vbuf = ar.take(enc_rows * dm * es);
  b.crossK = ar.take(N * enc_rows * dm * es);
  b.crossV = ar.take(N * enc_rows * dm * es);
  b.shK = ar.take(N * pre_rows * dm * es);
  b.shV = ar.take(N * pre_rows * dm * es);
}
int teacher_forced_workspace(const nmtgan_generator* g, int B, int Ls, int T, size_t* bytes) {
  NMTGAN_TRY(check_generator(g));
  NMTGAN_REQUIRE(B > 0 && Ls > 0 && T > 0, "teacher_forced: bad shape");
  Arena ar(nullptr, 0);
  TfBufs b;
  carve_tf(ar, g, B, Ls, T, b);
  *bytes = align_up(ar.off, 256);
  return 0;
}
int teacher_forced_logits(const nmtgan_generator* g, const int32_t* x, const uint8_t* x_pad, const int32_t* dec_in,
                          int B, int Ls, int T, float* logits, void* ws, size_t ws_bytes, hipStream_t st) {
  NMTGAN_TRY(check_generator(g));
  NMTGAN_REQUIRE(B > 0 && Ls > 0 && T > 0 && x && x_pad && dec_in && logits, "teacher_forced: bad arguments");
  Arena ar(ws, ws_bytes);
  TfBufs b;
  carve_tf(ar, g, B, Ls, T, b);
  NMTGAN_REQUIRE(ws != nullptr && !ar.overflow, "teacher_forced: workspace too small (%zu < %zu)", ws_bytes, ar.off);
    NMTGAN_TRY(encode(g, x, x_pad, B, Ls, b.act, b.kbuf, b.vbuf, st));
  NMTGAN_TRY(cross_kv(g, b.act.sp(), b.act.xo, B * Ls, b.crossK, b.crossV, st));
  NMTGAN_TRY(prefill(g, dec_in, x_pad, B, Ls, T, b.act, b.shK, b.shV, b.crossK, b.crossV, st));
  return logits_gemm(g, b.act.sp(), b.act.xo, B * T, logits, st);
}

// ------------------------------------------------------------------------------------------ greedy generation
__global__ void fill_greedy_tables(int64_t* step_off, int n_steps, int B, int32_t* bos, int32_t* zeros, int32_t* blk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_steps) step_off[i] = (int64_t)i * B;
  if (i < B) { zeros[i] = 0; blk[i] = i; }
  if (i == 0) bos[0] = 2;                                       // <S> (model.py:764-766)
}
struct GreedyBufs {
  Act act; void *kbuf, *vbuf, *crossK, *crossV, *prK, *prV; float* logits;
  int64_t* step_off; int32_t *bos, *zeros, *blk, *tok_cur;
  void* splitk; size_t splitk_bytes;
};
static void carve_greedy(Arena& ar, const nmtgan_generator* g, int B, int Ls, int T, GreedyBufs& b) {
  const int es = dtype_size(g->op_dtype), dm = g->d, N = g->n_blocks;
  const int64_t enc_rows = (int64_t)B * Ls;
  carve_act(ar, enc_rows > B ? enc_rows : B, dm, es, b.act, g->gemm_split != 0);
  b.kbuf = ar.take(enc_rows * dm * es);
  b.vbuf = ar.take(enc_rows * dm * es);
  b.crossK = ar.take(N * enc_rows * dm * es);
  b.crossV = ar.take(N * enc_rows * dm * es);
  b.prK = ar.take((size_t)N * T * B * dm * es);
  b.prV = ar.take((size_t)N * T * B * dm * es);
  b.logits = ar.get<float>((int64_t)B * g->vocab_dst);
  b.step_off = ar.get<int64_t>(T);
  b.bos = ar.get<int32_t>(4);
  b.zeros = ar.get<int32_t>(B);
  b.blk = ar.get<int32_t>(B);
  b.tok_cur = ar.get<int32_t>(B);
  // split-K partials of the decode-step GEMMs (B rows against K = d .. 4d): up to 32 slices of the widest output
  b.splitk_bytes = (size_t)32 * B * 4 * dm * sizeof(float);
  b.splitk = ar.take(b.splitk_bytes);
}
int generate_workspace(const nmtgan_generator* g, int B, int Ls, int T, size_t* bytes) {
  NMTGAN_TRY(check_generator(g));
  NMTGAN_REQUIRE(B > 0 && Ls > 0 && T > 0, "generate: bad shape");
  Arena ar(nullptr, 0);
  GreedyBufs b;
  carve_greedy(ar, g, B, Ls, T, b);
  *bytes = align_up(ar.off, 256);
  return 0;
}
int generate_greedy(const nmtgan_generator* g, const int32_t* x, const uint8_t* x_pad, int B, int Ls, int T,
                    int32_t* y, void* ws, size_t ws_bytes, hipStream_t st) {
  NMTGAN_TRY(check_generator(g));
  NMTGAN_REQUIRE(B > 0 && Ls > 0 && T > 0 && x && x_pad && y, "generate: bad arguments");
  NMTGAN_REQUIRE(T <= g->max_pos && Ls <= g->max_pos, "generate: length exceeds timing table");
  Arena ar(ws, ws_bytes);
  GreedyBufs b;
  carve_greedy(ar, g, B, Ls, T, b);
  NMTGAN_REQUIRE(ws != nullptr && !ar.overflow, "generate: workspace too small (%zu < %zu)", ws_bytes, ar.off);
  const int dm = g->d, V = g->vocab_dst, op = g->op_dtype;
    NMTGAN_TRY(encode(g, x, x_pad, B, Ls, b.act, b.kbuf, b.vbuf, st));
  NMTGAN_TRY(cross_kv(g, b.act.sp(), b.act.xo, B * Ls, b.crossK, b.crossV, st));
  const int nfill = T > B ? T : B;
  hipLaunchKernelGGL(fill_greedy_tables, dim3((nfill + 255) / 256), dim3(256), 0, st, b.step_off, T, B, b.bos, b.zeros,
                     b.blk);
  NMTGAN_CHECK_LAUNCH("fill_greedy_tables");
  NMTGAN_REQUIRE((uintptr_t)y % 16 == 0, "generate: y must be 16-byte aligned");
  NMTGAN_TRY(launch_zero(y, (size_t)B * T * sizeof(int32_t), st));
  SplitKScope split_k(b.splitk, b.splitk_bytes);      // single stream: the B-row step GEMMs are cut into k-slices
  for (int i = 0; i < T; ++i) {
    nmtgan_embed_args e{};
    e.table = g->emb_dst; e.timing = g->timing; e.vocab = V; e.max_pos = g->max_pos;
    if (i == 0) { e.tok = b.bos; e.tok_ld = 0; }
    else { e.tok = y + (i - 1); e.tok_ld = T; }
    e.pos_mod = 0; e.pos_add = i;
    e.mult = 2.0f * g->emb_mult; e.M = B; e.d = dm; e.out_f32 = b.act.xf; e.out_op = b.act.xo; e.op_dtype = op;
    e.out_split = g->gemm_split ? b.act.img : nullptr;
    NMTGAN_TRY(launch_embed(&e, st));
    DecAttn c{};
    c.prK = b.prK; c.prV = b.prV; c.pr_layer_stride = (int64_t)T * B * dm; c.step_off = b.step_off; c.n_steps2 = i + 1;
    c.wrK = b.prK; c.wrV = b.prV; c.wr_layer_stride = (int64_t)T * B * dm; c.wr_off = (int64_t)i * B * dm;
    c.crossK = b.crossK; c.crossV = b.crossV; c.Ls = Ls; c.cross_layer_stride = (int64_t)B * Ls * dm;
    c.cross_blk = b.blk; c.x_pad = x_pad; c.step_mode = 1;
    NMTGAN_TRY(decoder_pass(g, B, b.act, c, st));
    NMTGAN_TRY(logits_gemm(g, b.act.sp(), b.act.xo, B, b.logits, st));
    nmtgan_sample_args q{};
    q.logits = b.logits; q.ld = V; q.V = V; q.M = B; q.greedy = 1; q.out = b.tok_cur;
    NMTGAN_TRY(launch_sample_rows(&q, st));
    NMTGAN_TRY(launch_scatter_tokens(b.tok_cur, b.zeros, i, B, T, y, st));
  }
  return 0;
}

}  // namespace nmtgan
