// Internal launch functions (host side) of the gfx950 kernels; the extern "C" ABI in api.hip and the
// pipelines in pipeline.hip call these.
#pragma once
#include "common.h"

namespace nmtgan {

int launch_gemm(const nmtgan_gemm_args* a, hipStream_t st);
// n independent problems of one epilogue kind (CONVMAX - with or without the additive term - or LINEAR with one fp32 or
// op-dtype output) in one persistent launch per 24 problems; min_tiles > 0 leaves problems of at least that many 256x128 tiles to
// the single-problem kernels
int launch_gemm_group(const nmtgan_gemm_args* args, int n, hipStream_t st, int min_tiles = 0);
int launch_gemm_stats(int op_dtype, const void* W, int V, const void* X, int rows, int K, float2* stats,
                      hipStream_t st);
bool gemm_stats_supported(int op_dtype, int V, int K);
// cap on the grid of the persistent GEMM kernels launched by this thread (0 = one workgroup per CU);
// the two-stream rollout schedule uses it to leave CUs to the concurrently running decode steps
void set_gemm_wg_cap(int cap);
// split-K scratch of the calling thread for the small LINEAR GEMMs (nullptr = off, the default); a pipeline that
// issues all its GEMMs on ONE stream sets it around its launches
void set_gemm_splitk_scratch(void* buf, size_t bytes);
bool gemm_splitk_active();
// shapes the NMTGAN_EPI_ADDLN epilogue (projection + residual + LayerNorm in one kernel) takes
bool gemm_addln_supported(int op_dtype, int N, int K);
struct SplitKScope {
  SplitKScope(void* buf, size_t bytes) { set_gemm_splitk_scratch(buf, bytes); }
  ~SplitKScope() { set_gemm_splitk_scratch(nullptr, 0); }
};
// [M][K] fp32 (row stride lda) -> [M][3K] fp16 = [hi | lo | hi], hi = fp16(a), lo = fp16(a - hi): the A operand of the
// split-precision GEMM form (nmtgan_generator.gemm_split); K % 8 == 0, 16-byte aligned rows
int launch_split3(const float* A, int64_t lda, int M, int K, void* out, hipStream_t st);
int launch_logits_sample(int op_dtype, const void* x, const void* w, int M, int V, int d, const float* u,
                         const int64_t* u_off, int64_t u_add, int32_t* out, void* ws, hipStream_t st);
size_t logits_sample_workspace(int M, int V);
int launch_logits_sample_static(int op_dtype, const void* x, const void* w, int M, int V, int d, const float* u,
                                const int64_t* u_off, int64_t u_add, int32_t* out, hipStream_t st);
int launch_zero(void* p, size_t bytes, hipStream_t st);      // fill kernel (16-byte aligned buffer); no memset nodes
int launch_round_f16(float* x, int64_t n, hipStream_t st);   // x <- float(fp16(x)) in place (error-budget runs only)
int launch_embed(const nmtgan_embed_args* a, hipStream_t st);
int launch_add_layernorm(const nmtgan_addln_args* a, hipStream_t st);
int launch_attn_rows(const nmtgan_attn_args* a, hipStream_t st);
int launch_sample_rows(const nmtgan_sample_args* a, hipStream_t st);
int launch_attn_group(const nmtgan_attn_group_args* a, hipStream_t st);
int launch_transpose_v(const void* v, void* vt, int nblk, int L, int Lp, int d, int esize, hipStream_t st);

// rollout token bookkeeping (model.py:242-243, 250-252)
int launch_init_rollouts(const int32_t* y, const int32_t* row_b, const int32_t* row_g, int n_rows, int T,
                         int32_t* Y, hipStream_t st);
int launch_scatter_tokens(const int32_t* tok, const int32_t* row_g, int add, int n, int T, int32_t* Y,
                          hipStream_t st);
// discriminator small kernels
int launch_disc_embed(const void* table, int op_dtype, int vocab, const int32_t* ids, int64_t n_tok, int E,
                      void* out, hipStream_t st);
int launch_cast_rows(const float* in, void* out, int op_dtype, int64_t n, hipStream_t st);
int launch_prefix_max(const float* C, float* pm, int B, int T, int Fp, int n_widths, const int* widths,
                      const int* n_filters, hipStream_t st);
int launch_pool_init(const float* pm, const int32_t* row_b, const int32_t* row_g, int rows, int T, int Fp,
                     float* pooled, hipStream_t st);
int launch_add_inplace(float* dst, const float* src, int64_t n, hipStream_t st);
int launch_bucket_finish(const float* raw, const float* S, const int32_t* row_b, int rows, int nq, int N, int P, int pa,
                         const float* scale, const float* shift, float* pooled, int64_t ld_pool, hipStream_t st);
// table form of the conv + max of the rollouts (disc_table.hip): per width i a table tab[i][vocab][w][nfp] of
// per-(token, tap) filter responses in `dtype` and the partial sums S[i][B][T-w+1][w-1][nfp] (fp32) over y_sample
constexpr int TABLE_MAX_LEN = 512;
using ConvTable = nmtgan_conv_table;
// filter bank Wt[nf][w taps][row_bytes] -> out[w * nfp][row_bytes], row j * nfp + f = tap j of filter f (zero for f >= nf)
int launch_table_weights(const void* wt, int w, int nf, int nfp, int row_bytes, void* out, hipStream_t st);
// y[B][T] -> S and the window values win_vals[B*T][Fp] = relu(bn(conv)) of every window of y
int launch_table_prefix(const ConvTable& t, const int32_t* y, int B, float* win_vals, hipStream_t st);
// pooled[r][:] = max(pooled[r][:], max over the windows of row r that hold a sampled token)
int launch_table_conv(const ConvTable& t, const int32_t* ids, const int32_t* row_b, const int32_t* row_g, int rows,
                      float* pooled, hipStream_t st);
// the tables of the target tower of d (pipeline.hip; t describes them: geometry, entry type, buffers)
int conv_table_build(const nmtgan_discriminator* d, const ConvTable* t, hipStream_t st);
int launch_highway_gate(const float* H, const float* o, const float* t, float* out, int64_t n, hipStream_t st);
int launch_disc_head(const float* feat_t, const float* feat_s, const int32_t* src_row, int M, int Fp,
                     const float* head_w, const float* head_b, float* p_real, float* probs2, hipStream_t st);
int launch_reward_reduce(const float* p, const float* p_base, const int32_t* grp_first, const float* mask, int B,
                         int T, int rollnum, int use_bias, double bias_num, float* rewards, hipStream_t st);

}  // namespace nmtgan
