// extern "C" surface of libnmtgan_hip.so (declared in include/nmtgan.h).
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace nmtgan {
static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
namespace prof {
bool g_on = false;
thread_local int g_hint = -1;
struct Rec { int cls; hipEvent_t a, b; };
static std::vector<Rec> g_recs;
void begin(int cls, hipStream_t st) {
  Rec r{cls, nullptr, nullptr};
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
}
void end(hipStream_t st) {
  if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().b, st);
}
}  // namespace prof
int disc_workspace(const nmtgan_discriminator* d, int chunk, size_t* bytes);
int disc_features(const nmtgan_discriminator* d, int source, const int32_t* ids, int M, float* feat, int chunk,
                  void* ws, size_t ws_bytes, hipStream_t st);
int reward_workspace(const nmtgan_generator* g, const nmtgan_discriminator* d, const nmtgan_reward_plan* p,
                     size_t* bytes);
int rollout_reward(const nmtgan_generator* g, const nmtgan_discriminator* d, const nmtgan_reward_plan* p, void* ws,
                   size_t ws_bytes, hipStream_t st);
int teacher_forced_workspace(const nmtgan_generator* g, int B, int Ls, int T, size_t* bytes);
int teacher_forced_logits(const nmtgan_generator* g, const int32_t* x, const uint8_t* x_pad, const int32_t* dec_in,
                          int B, int Ls, int T, float* logits, void* ws, size_t ws_bytes, hipStream_t st);
int generate_workspace(const nmtgan_generator* g, int B, int Ls, int T, size_t* bytes);
int generate_greedy(const nmtgan_generator* g, const int32_t* x, const uint8_t* x_pad, int B, int Ls, int T,
                    int32_t* y, void* ws, size_t ws_bytes, hipStream_t st);
int pg_workspace(const nmtgan_generator* g, int B, int Ls, int T, size_t* bytes);
int pg_grads(const nmtgan_generator* g, const nmtgan_generator* gr, const int32_t* x, const uint8_t* x_pad, const int32_t* y,
             const int32_t* dec_in, const float* reward, int B, int Ls, int T, float* loss, float* p_target, void* ws,
             size_t ws_bytes, hipStream_t st);
int rmsprop_update(float* w, float* ms, float* mom, const float* grad, int64_t n, float lr, float decay, float momentum,
                   float eps, float grad_scale, float grad_clip, void* w_op, int op_dtype, hipStream_t st);
int adadelta_update(float* w, float* accum, float* accum_update, const float* grad, int64_t n, float lr, float rho,
                    float eps, float grad_scale, float grad_clip, void* w_op, int op_dtype, hipStream_t st);
int disc_train_workspace(const nmtgan_discriminator* d, int M, size_t* bytes);
int disc_train_grads(const nmtgan_discriminator* d, const nmtgan_disc_params* p, const nmtgan_disc_params* g, const int32_t* x,
                     const int32_t* xs, const float* labels, int M, float* loss, float* accuracy, void* ws, size_t ws_bytes,
                     hipStream_t st);
int disc_fold_bn(const float* b, const float* gamma, const float* beta, const float* mean, const float* var, int n,
                 float* scale, float* shift, hipStream_t st);
int clip_values(float* w, int64_t n, float lo, float hi, void* w_op, int op_dtype, hipStream_t st);
int sgd_adam_update(float* w, float* m, float* v, const float* grad, int64_t n, float lr, float b1, float b2, float eps,
                    int t, float grad_scale, float grad_clip, void* w_op, int op_dtype, hipStream_t st);
}  // namespace nmtgan

using namespace nmtgan;

extern "C" {

int nmtgan_version(void) { return NMTGAN_VERSION; }
const char* nmtgan_last_error(void) { return g_err.c_str(); }

int nmtgan_profile_classes(void) { return prof::N_CLS; }
int nmtgan_profile_start(void) {
  prof::g_recs.clear();
  prof::g_on = true;
  return 0;
}
int nmtgan_profile_stop(double* ms_per_class, int64_t* launches_per_class, int n_classes) {
  prof::g_on = false;
  if (hipDeviceSynchronize() != hipSuccess) { set_error("profile_stop: device synchronize failed"); return -3; }
  for (int i = 0; i < n_classes; ++i) { ms_per_class[i] = 0.0; launches_per_class[i] = 0; }
  for (auto& r : prof::g_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.cls >= 0 && r.cls < n_classes) {
      ms_per_class[r.cls] += ms;
      launches_per_class[r.cls] += 1;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  prof::g_recs.clear();
  return 0;
}

int nmtgan_gemm(const nmtgan_gemm_args* a, void* stream) { return launch_gemm(a, (hipStream_t)stream); }
int nmtgan_gemm_grouped(const nmtgan_gemm_args* a, int n, void* stream) {
  return launch_gemm_group(a, n, (hipStream_t)stream);
}
int nmtgan_conv_table_build(const nmtgan_discriminator* d, const nmtgan_conv_table* t, void* stream) {
  return conv_table_build(d, t, (hipStream_t)stream);
}
int nmtgan_conv_table_prefix(const nmtgan_conv_table* t, const int32_t* y, int B, float* win_vals, void* stream) {
  if (!t) { set_error("conv_table_prefix: null table"); return -2; }
  return launch_table_prefix(*t, y, B, win_vals, (hipStream_t)stream);
}
int nmtgan_conv_table_rows(const nmtgan_conv_table* t, const int32_t* ids, const int32_t* row_b, const int32_t* row_g,
                           int rows, float* pooled, void* stream) {
  if (!t) { set_error("conv_table_rows: null table"); return -2; }
  return launch_table_conv(*t, ids, row_b, row_g, rows, pooled, (hipStream_t)stream);
}
int nmtgan_embed(const nmtgan_embed_args* a, void* stream) { return launch_embed(a, (hipStream_t)stream); }
int nmtgan_add_layernorm(const nmtgan_addln_args* a, void* stream) {
  return launch_add_layernorm(a, (hipStream_t)stream);
}
int nmtgan_split3(const float* A, int64_t lda, int M, int K, void* out, void* stream) {
  return launch_split3(A, lda, M, K, out, (hipStream_t)stream);
}
int nmtgan_attn_rows(const nmtgan_attn_args* a, void* stream) { return launch_attn_rows(a, (hipStream_t)stream); }
int nmtgan_attn_group(const nmtgan_attn_group_args* a, void* stream) {
  return launch_attn_group(a, (hipStream_t)stream);
}
int nmtgan_transpose_v(const void* v, void* vt, int nblk, int len, int len_padded, int d, void* stream) {
  return launch_transpose_v(v, vt, nblk, len, len_padded, d, 2, (hipStream_t)stream);
}
int nmtgan_transpose_v_f32(const float* v, float* vt, int nblk, int len, int len_padded, int d, void* stream) {
  return launch_transpose_v(v, vt, nblk, len, len_padded, d, 4, (hipStream_t)stream);
}
size_t nmtgan_logits_sample_workspace_bytes(int M, int V) { return (M > 0 && V > 0) ? logits_sample_workspace(M, V) : 0; }
int nmtgan_logits_sample(int op_dtype, const void* x, const void* w, int M, int V, int d, const float* u,
                         const int64_t* u_off, int64_t u_add, int32_t* out, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (M > 0 && (!gemm_stats_supported(op_dtype, V, d) || workspace_bytes < logits_sample_workspace(M, V))) {
    set_error("logits_sample: unsupported shape (V %% 64, d*esize %% 128, >= 4 k-tiles) or workspace too small");
    return -2;
  }
  return launch_logits_sample(op_dtype, x, w, M, V, d, u, u_off, u_add, out, workspace, (hipStream_t)stream);
}
int nmtgan_logits_sample_static(int op_dtype, const void* x, const void* w, int M, int V, int d, const float* u,
                                const int64_t* u_off, int64_t u_add, int32_t* out, void* stream) {
  return launch_logits_sample_static(op_dtype, x, w, M, V, d, u, u_off, u_add, out, (hipStream_t)stream);
}
int nmtgan_sample_rows(const nmtgan_sample_args* a, void* stream) {
  return launch_sample_rows(a, (hipStream_t)stream);
}

size_t nmtgan_disc_workspace_bytes(const nmtgan_discriminator* d, int chunk_rows) {
  size_t b = 0;
  return disc_workspace(d, chunk_rows, &b) == 0 ? b : 0;
}
int nmtgan_disc_features(const nmtgan_discriminator* d, int source_tower, const int32_t* ids, int M, float* feat,
                         int chunk_rows, void* workspace, size_t workspace_bytes, void* stream) {
  return disc_features(d, source_tower, ids, M, feat, chunk_rows, workspace, workspace_bytes, (hipStream_t)stream);
}
int nmtgan_disc_head(const nmtgan_discriminator* d, const float* feat_t, const float* feat_s,
                     const int32_t* src_row, int M, float* p_real, float* probs2, void* stream) {
  if (!d) { set_error("disc_head: null discriminator"); return -2; }
  return launch_disc_head(feat_t, feat_s, src_row, M, d->Fp, d->head_w, d->head_b, p_real, probs2,
                          (hipStream_t)stream);
}
int nmtgan_reward_reduce(const float* p, const float* p_base, const int32_t* grp_first, const float* mask, int B,
                         int T, int rollnum, int use_bias, double bias_num, float* rewards, void* stream) {
  return launch_reward_reduce(p, p_base, grp_first, mask, B, T, rollnum, use_bias, bias_num, rewards,
                              (hipStream_t)stream);
}

size_t nmtgan_rollout_reward_workspace_bytes(const nmtgan_generator* g, const nmtgan_discriminator* d,
                                             const nmtgan_reward_plan* plan) {
  size_t b = 0;
  return reward_workspace(g, d, plan, &b) == 0 ? b : 0;
}
int nmtgan_rollout_reward(const nmtgan_generator* g, const nmtgan_discriminator* d, const nmtgan_reward_plan* plan,
                          void* workspace, size_t workspace_bytes, void* stream) {
  return rollout_reward(g, d, plan, workspace, workspace_bytes, (hipStream_t)stream);
}

size_t nmtgan_generate_workspace_bytes(const nmtgan_generator* g, int B, int Ls, int T) {
  size_t b = 0;
  return generate_workspace(g, B, Ls, T, &b) == 0 ? b : 0;
}
int nmtgan_generate_greedy(const nmtgan_generator* g, const int32_t* x, const uint8_t* x_pad, int B, int Ls, int T,
                           int32_t* y, void* workspace, size_t workspace_bytes, void* stream) {
  return generate_greedy(g, x, x_pad, B, Ls, T, y, workspace, workspace_bytes, (hipStream_t)stream);
}

size_t nmtgan_teacher_forced_workspace_bytes(const nmtgan_generator* g, int B, int Ls, int T) {
  size_t b = 0;
  return teacher_forced_workspace(g, B, Ls, T, &b) == 0 ? b : 0;
}
int nmtgan_teacher_forced_logits(const nmtgan_generator* g, const int32_t* x, const uint8_t* x_pad,
                                 const int32_t* dec_in, int B, int Ls, int T, float* logits, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  return teacher_forced_logits(g, x, x_pad, dec_in, B, Ls, T, logits, workspace, workspace_bytes,
                               (hipStream_t)stream);
}

size_t nmtgan_pg_workspace_bytes(const nmtgan_generator* g, int B, int Ls, int T) {
  size_t b = 0;
  if (!g) { set_error("pg_workspace: null generator"); return 0; }
  return pg_workspace(g, B, Ls, T, &b) == 0 ? b : 0;
}
int nmtgan_pg_grads(const nmtgan_generator* g, const nmtgan_generator* grads, const int32_t* x, const uint8_t* x_pad,
                    const int32_t* y, const int32_t* dec_in, const float* reward, int B, int Ls, int T, float* loss,
                    float* p_target, void* workspace, size_t workspace_bytes, void* stream) {
  return pg_grads(g, grads, x, x_pad, y, dec_in, reward, B, Ls, T, loss, p_target, workspace, workspace_bytes,
                  (hipStream_t)stream);
}
int nmtgan_rmsprop_update(float* w, float* ms, float* mom, const float* grad, int64_t n, float lr, float decay,
                          float momentum, float eps, float grad_scale, float grad_clip, void* w_op, int op_dtype,
                          void* stream) {
  return rmsprop_update(w, ms, mom, grad, n, lr, decay, momentum, eps, grad_scale, grad_clip, w_op, op_dtype,
                        (hipStream_t)stream);
}
int nmtgan_adadelta_update(float* w, float* accum, float* accum_update, const float* grad, int64_t n, float lr, float rho,
                           float eps, float grad_scale, float grad_clip, void* w_op, int op_dtype, void* stream) {
  return adadelta_update(w, accum, accum_update, grad, n, lr, rho, eps, grad_scale, grad_clip, w_op, op_dtype,
                         (hipStream_t)stream);
}
size_t nmtgan_disc_train_workspace_bytes(const nmtgan_discriminator* d, int M) {
  size_t b = 0;
  if (!d) { set_error("disc_train_workspace: null discriminator"); return 0; }
  return disc_train_workspace(d, M, &b) == 0 ? b : 0;
}
int nmtgan_disc_train_grads(const nmtgan_discriminator* d, const nmtgan_disc_params* p, const nmtgan_disc_params* grads,
                            const int32_t* x, const int32_t* xs, const float* labels, int M, float* loss, float* accuracy,
                            void* workspace, size_t workspace_bytes, void* stream) {
  return disc_train_grads(d, p, grads, x, xs, labels, M, loss, accuracy, workspace, workspace_bytes, (hipStream_t)stream);
}
int nmtgan_disc_fold_bn(const float* conv_b, const float* gamma, const float* beta, const float* moving_mean,
                        const float* moving_var, int n, float* scale, float* shift, void* stream) {
  return disc_fold_bn(conv_b, gamma, beta, moving_mean, moving_var, n, scale, shift, (hipStream_t)stream);
}
int nmtgan_clip_values(float* w, int64_t n, float lo, float hi, void* w_op, int op_dtype, void* stream) {
  return clip_values(w, n, lo, hi, w_op, op_dtype, (hipStream_t)stream);
}
int nmtgan_sgd_update(float* w, const float* grad, int64_t n, float lr, float grad_scale, float grad_clip, void* w_op,
                      int op_dtype, void* stream) {
  return sgd_adam_update(w, nullptr, nullptr, grad, n, lr, 0.f, 0.f, 0.f, 0, grad_scale, grad_clip, w_op, op_dtype,
                         (hipStream_t)stream);
}
int nmtgan_adam_update(float* w, float* m, float* v, const float* grad, int64_t n, float lr, float beta1, float beta2,
                       float eps, int t, float grad_scale, float grad_clip, void* w_op, int op_dtype, void* stream) {
  if (!m || !v) { set_error("adam_update: null slot"); return -2; }
  return sgd_adam_update(w, m, v, grad, n, lr, beta1, beta2, eps, t, grad_scale, grad_clip, w_op, op_dtype,
                         (hipStream_t)stream);
}

}  // extern "C"
