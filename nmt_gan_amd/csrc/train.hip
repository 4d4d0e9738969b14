// Policy-gradient step of the generator: the step gan_train.py:112 runs right after
// get_reward.  Reference: Model.generate_step_and_update (model.py:273-276) -> the loss half of build_generate
// (model.py:178-190): decoder(shift_right(Y)) -> gan_output (model.py:649-666)
//     g_loss = -sum_{b,t} softmax(logits)[b,t,Y[b,t]] * reward[b,t] / B
// -> optimizer.compute_gradients over every trainable variable -> RMSProp apply.
//
// Forward with every sub-layer input kept in the workspace, then a hand-written backward: every matrix
// product goes through the library's MFMA GEMM (C = A . Wt^T) on transposed copies
//     dX [M,K] = dY [M,N] . (W^T)^T          A = dY (op dtype),   Wt' = W^T [K][N]
//     dWt[N,K] = dY^T [N,Mp] . (X^T)^T       A = dY^T,            Wt' = X^T [K][Mp]   (Mp = M padded to 64, zeros)
// and the row kernels below do softmax-probability loss, LayerNorm, attention, ReLU and embedding backward.
// Gradients are fp32 in the layout of the packed weights (Wt[N][K]); residual dropout is neutralised as on the
// rollout path.  The batch is a few thousand rows (B*T), so these kernels are sized for
// correctness and determinism (fixed-order column sums; only the embedding scatter uses float atomics).
#include <algorithm>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace nmtgan {

// ------------------------------------------------------------------------------------------ small kernels
// out[c][r] = in[r][c] for r < R, 0 for R <= r < Rp  (cast TI -> TO on the way)
// group > 0: row r starts at (r / group) * group_stride + (r % group) * ld_in (the overlapping conv windows of a
// sentence image, as in nmtgan_gemm_args.a_group), else at r * ld_in
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const TI* in, int64_t ld_in, int R, int C, TO* out, int Rp,
                                                             int group, int64_t group_stride) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    const int64_t off = group > 0 ? (int64_t)(r / group) * group_stride + (int64_t)(r % group) * ld_in : (int64_t)r * ld_in;
    tile[ty + 4 * i][tx] = (r < R && c < C) ? OpTraits<TI>::load(in + off + c) : 0.f;
  }
  __syncthreads();
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < C && r < Rp) OpTraits<TO>::store(out + (int64_t)c * Rp + r, tile[tx][ty + 4 * i]);
  }
}
template <typename TI, typename TO>
static int transpose_t(const void* in, int64_t ld_in, int R, int C, void* out, int Rp, int group, int64_t group_stride,
                       hipStream_t st) {
  dim3 grid((Rp + 63) / 64, (C + 63) / 64), block(256);
  hipLaunchKernelGGL((transpose_cast_kernel<TI, TO>), grid, block, 0, st, (const TI*)in, ld_in, R, C, (TO*)out, Rp, group,
                     group_stride);
  NMTGAN_CHECK_LAUNCH("transpose_cast");
  return 0;
}
// in_f32 != 0: input is fp32, else op dtype; output always op dtype
static int launch_transpose(int op, int in_f32, const void* in, int64_t ld_in, int R, int C, void* out, int Rp,
                            hipStream_t st, int group = 0, int64_t group_stride = 0) {
  prof::Scope prof_scope(prof::OTHER, st);
  NMTGAN_REQUIRE(in && out && R > 0 && C > 0 && Rp >= R, "transpose: bad arguments");
  if (op == NMTGAN_F32) return transpose_t<float, float>(in, ld_in, R, C, out, Rp, group, group_stride, st);
  return in_f32 ? transpose_t<float, bf16_t>(in, ld_in, R, C, out, Rp, group, group_stride, st)
                : transpose_t<bf16_t, bf16_t>(in, ld_in, R, C, out, Rp, group, group_stride, st);
}

// out[c] = sum_r in[r][c] in a fixed order: grid.y row chunks of `rows_per_chunk` rows (4 row lanes per column,
// lane sums added 0..3) leave partials in part[chunk][C]; colsum_finish_kernel adds the chunks in order.
static const int COLSUM_CHUNKS = 64;
__global__ __launch_bounds__(256) void colsum_kernel(const float* in, int64_t ld, int R, int C, int rows_per_chunk, float* out) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float s = 0.f;
  if (c < C)
    for (int r = r0 + ty; r < r1; r += 4) s += in[(int64_t)r * ld + c];
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < C) out[(int64_t)blockIdx.y * C + c] = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
}
__global__ void colsum_finish_kernel(const float* part, int chunks, int C, float* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += part[(int64_t)k * C + c];
  out[c] = s;
}
// scratch: COLSUM_CHUNKS * C floats (may be nullptr when R is small enough for one chunk)
static int launch_colsum(const float* in, int64_t ld, int R, int C, float* out, float* scratch, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  NMTGAN_REQUIRE(in && out && R > 0 && C > 0, "colsum: bad arguments");
  int chunks = scratch ? std::min(COLSUM_CHUNKS, (R + 63) / 64) : 1;
  const int per = (R + chunks - 1) / chunks;
  chunks = (R + per - 1) / per;
  hipLaunchKernelGGL(colsum_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, st, in, ld, R, C, per, chunks > 1 ? scratch : out);
  NMTGAN_CHECK_LAUNCH("colsum");
  if (chunks > 1) {
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, chunks, C, out);
    NMTGAN_CHECK_LAUNCH("colsum_finish");
  }
  return 0;
}

// dh[i] = h[i] > 0 ? dh[i] : 0   (h = relu output in op dtype)
template <typename OpT>
__global__ void relu_bwd_kernel(float* dh, const OpT* h, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !(OpTraits<OpT>::load(h + i) > 0.f)) dh[i] = 0.f;
}
static int launch_relu_bwd(int op, float* dh, const void* h, int64_t n, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (op == NMTGAN_BF16) hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, grid, block, 0, st, dh, (const bf16_t*)h, n);
  else hipLaunchKernelGGL(relu_bwd_kernel<float>, grid, block, 0, st, dh, (const float*)h, n);
  NMTGAN_CHECK_LAUNCH("relu_bwd");
  return 0;
}

// Loss and d(loss)/d(logits) of gan_output (model.py:657-665), one workgroup per row r = (b, t):
//   p = softmax(l);  loss_r = -reward_r * p_y / B;  dl_j = -(reward_r / B) * p_y * ([j == y] - p_j)
template <typename OpT>
__global__ __launch_bounds__(256) void pg_dlogits_kernel(const float* logits, int V, const int32_t* y, const float* reward,
                                                         float inv_B, OpT* dl, float* loss_row, float* p_target) {
  __shared__ float red[4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* l = logits + (int64_t)r * V;
  OpT* o = dl + (int64_t)r * V;
  float m = -INFINITY;
  for (int j = tid; j < V; j += 256) m = fmaxf(m, l[j]);
  m = wave_max(m);
  if (lane == 0) red[w] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int j = tid; j < V; j += 256) s += __expf(l[j] - m);
  s = wave_sum(s);
  if (lane == 0) red[w] = s;
  __syncthreads();
  const float inv_l = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
  int tgt = y[r];
  tgt = min(max(tgt, 0), V - 1);
  const float p_y = __expf(l[tgt] - m) * inv_l;
  const float coef = -reward[r] * inv_B;
  if (tid == 0) {
    loss_row[r] = coef * p_y;
    if (p_target) p_target[r] = p_y;
  }
  const float cp = coef * p_y;
  for (int j = tid; j < V; j += 256) {
    const float p = __expf(l[j] - m) * inv_l;
    OpTraits<OpT>::store(o + j, cp * ((j == tgt ? 1.f : 0.f) - p));
  }
}

// out[0] = sum_i in[i] in a fixed order (one workgroup)
__global__ __launch_bounds__(256) void sum_kernel(const float* in, int n, float* out) {
  __shared__ double part[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)in[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) part[threadIdx.x] += part[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)part[0];
}

// LayerNorm backward of out = LN(x + y) * gamma + beta (model.py:737-738, common_layers.py:336-341):
// ds = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dout * gamma; one wave per row, a workgroup walks
// rows blockIdx*4 + wave, += 4*gridDim and leaves its partial (dgamma | dbeta) in part[blockIdx][2d].
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* x, const float* y, const float* gamma, const float* dout,
                                                     const uint8_t* zero_row, int M, int d, float* ds, float* part) {
  extern __shared__ float acc[];                  // [4 waves][2d]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = d / 64;                         // <= 16
  float dg[16], db[16];
  for (int k = 0; k < per; ++k) dg[k] = db[k] = 0.f;
  for (int r = blockIdx.x * 4 + w; r < M; r += 4 * gridDim.x) {
    const int64_t base = (int64_t)r * d;
    if (zero_row && zero_row[r]) {                // forward zeroed this row: no gradient flows
      for (int k = 0; k < per; ++k) ds[base + lane + 64 * k] = 0.f;
      continue;
    }
    float s[16], go[16];
    float sum = 0.f;
    for (int k = 0; k < per; ++k) {
      const int c = lane + 64 * k;
      s[k] = x[base + c] + (y ? y[base + c] : 0.f);
      sum += s[k];
    }
    const float mean = wave_sum(sum) / (float)d;
    float q = 0.f;
    for (int k = 0; k < per; ++k) { s[k] -= mean; q += s[k] * s[k]; }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-6f);
    float a = 0.f, b = 0.f;
    for (int k = 0; k < per; ++k) {
      const int c = lane + 64 * k;
      const float o = dout[base + c];
      s[k] *= rstd;                               // xhat
      go[k] = o * gamma[c];
      dg[k] += o * s[k];
      db[k] += o;
      a += go[k];
      b += go[k] * s[k];
    }
    a = wave_sum(a) / (float)d;
    b = wave_sum(b) / (float)d;
    for (int k = 0; k < per; ++k) ds[base + lane + 64 * k] = rstd * (go[k] - a - s[k] * b);
  }
  for (int k = 0; k < per; ++k) {
    acc[w * 2 * d + lane + 64 * k] = dg[k];
    acc[w * 2 * d + d + lane + 64 * k] = db[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * d; c += 256)
    part[(int64_t)blockIdx.x * 2 * d + c] = (acc[c] + acc[2 * d + c]) + (acc[4 * d + c] + acc[6 * d + c]);
}

// Attention backward (dot_product_attention, common_attention.py:294-331), one workgroup per (batch, head).
// q is the saved, already scaled query; scores of masked keys (padding bias / above the diagonal) have
// probability exactly 0 as in the forward kernels.  The head slices of q, k, v, dO ([T][dh] fp32, rows padded to
// dh+1 words so a column walk is bank-conflict free) and the two [Tq][Tk] matrices P and dS live in LDS.
template <typename OpT>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const OpT* q, const OpT* k, const OpT* v, const float* dO, int Tq, int Tk,
                                                       int d, int heads, int causal, const uint8_t* mask, float* dq,
                                                       int64_t ld_dq, float* dk, int64_t ld_dk, float* dv, int64_t ld_dv,
                                                       float dq_scale) {
  extern __shared__ float lds[];
  const int b = blockIdx.x / heads, h = blockIdx.x % heads, dh = d / heads, dp1 = dh + 1;
  float* P = lds;
  float* S = P + Tq * Tk;
  float* Q = S + Tq * Tk;
  float* O = Q + Tq * dp1;
  float* K = O + Tq * dp1;
  float* Vv = K + Tk * dp1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const OpT* qb = q + (int64_t)b * Tq * d + h * dh;
  const OpT* kb = k + (int64_t)b * Tk * d + h * dh;
  const OpT* vb = v + (int64_t)b * Tk * d + h * dh;
  const float* ob = dO + (int64_t)b * Tq * d + h * dh;
  for (int idx = tid; idx < Tq * dh; idx += 256) {
    const int i = idx / dh, c = idx % dh;
    Q[i * dp1 + c] = OpTraits<OpT>::load(qb + (int64_t)i * d + c);
    O[i * dp1 + c] = ob[(int64_t)i * d + c];
  }
  for (int idx = tid; idx < Tk * dh; idx += 256) {
    const int j = idx / dh, c = idx % dh;
    K[j * dp1 + c] = OpTraits<OpT>::load(kb + (int64_t)j * d + c);
    Vv[j * dp1 + c] = OpTraits<OpT>::load(vb + (int64_t)j * d + c);
  }
  __syncthreads();
  for (int idx = tid; idx < Tq * Tk; idx += 256) {
    const int i = idx / Tk, j = idx % Tk;
    const bool valid = !(causal && j > i) && !(mask && mask[(int64_t)b * Tk + j]);
    float s = -INFINITY, dp = 0.f;
    if (valid) {
      s = 0.f;
      for (int c = 0; c < dh; ++c) {
        s += Q[i * dp1 + c] * K[j * dp1 + c];
        dp += O[i * dp1 + c] * Vv[j * dp1 + c];
      }
    }
    P[idx] = s;
    S[idx] = dp;
  }
  __syncthreads();
  for (int i = w; i < Tq; i += 4) {               // softmax row i, then dS = P * (dP - sum_j P dP)
    float m = -INFINITY;
    for (int j = lane; j < Tk; j += 64) m = fmaxf(m, P[i * Tk + j]);
    m = wave_max(m);
    float l = 0.f;
    for (int j = lane; j < Tk; j += 64) {
      const float e = (P[i * Tk + j] == -INFINITY) ? 0.f : __expf(P[i * Tk + j] - m);
      P[i * Tk + j] = e;
      l += e;
    }
    l = wave_sum(l);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float dsum = 0.f;
    for (int j = lane; j < Tk; j += 64) {
      const float p = P[i * Tk + j] * inv;
      P[i * Tk + j] = p;
      dsum += p * S[i * Tk + j];
    }
    dsum = wave_sum(dsum);
    for (int j = lane; j < Tk; j += 64) S[i * Tk + j] = P[i * Tk + j] * (S[i * Tk + j] - dsum);
  }
  __syncthreads();
  for (int idx = tid; idx < Tk * dh; idx += 256) {
    const int j = idx / dh, c = idx % dh;
    float av = 0.f, ak = 0.f;
    for (int i = 0; i < Tq; ++i) {
      av += P[i * Tk + j] * O[i * dp1 + c];
      ak += S[i * Tk + j] * Q[i * dp1 + c];
    }
    dv[((int64_t)b * Tk + j) * ld_dv + h * dh + c] = av;
    dk[((int64_t)b * Tk + j) * ld_dk + h * dh + c] = ak;
  }
  for (int idx = tid; idx < Tq * dh; idx += 256) {
    const int i = idx / dh, c = idx % dh;
    float aq = 0.f;
    for (int j = 0; j < Tk; ++j) aq += S[i * Tk + j] * K[j * dp1 + c];
    dq[((int64_t)b * Tq + i) * ld_dq + h * dh + c] = aq * dq_scale;
  }
}
static size_t attn_bwd_lds(int Tq, int Tk, int dh) {
  return ((size_t)2 * Tq * Tk + (size_t)2 * (Tq + Tk) * (dh + 1)) * sizeof(float);
}
static const size_t ATTN_BWD_LDS_MAX = 160 * 1024;
template <typename OpT>
static int attn_bwd_t(const void* q, const void* k, const void* v, const float* dO, int B, int Tq, int Tk, int d, int heads,
                      int causal, const uint8_t* mask, float* dq, int64_t ld_dq, float* dk, int64_t ld_dk, float* dv,
                      int64_t ld_dv, float dq_scale, size_t lds, hipStream_t st) {
  static thread_local int raised_for_device = -1;        // opt in to > 64 KiB of dynamic LDS once per device
  int dev = -1;
  NMTGAN_REQUIRE(hipGetDevice(&dev) == hipSuccess, "attn_bwd: hipGetDevice failed");
  if (raised_for_device != dev) {
    NMTGAN_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<OpT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATTN_BWD_LDS_MAX) == hipSuccess,
                   "attn_bwd: cannot raise the dynamic LDS limit");
    raised_for_device = dev;
  }
  hipLaunchKernelGGL(attn_bwd_kernel<OpT>, dim3(B * heads), dim3(256), lds, st, (const OpT*)q, (const OpT*)k, (const OpT*)v, dO,
                     Tq, Tk, d, heads, causal, mask, dq, ld_dq, dk, ld_dk, dv, ld_dv, dq_scale);
  NMTGAN_CHECK_LAUNCH("attn_bwd");
  return 0;
}
static int launch_attn_bwd(int op, const void* q, const void* k, const void* v, const float* dO, int B, int Tq, int Tk,
                           int d, int heads, int causal, const uint8_t* mask, float* dq, int64_t ld_dq, float* dk,
                           int64_t ld_dk, float* dv, int64_t ld_dv, float dq_scale, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  const size_t lds = attn_bwd_lds(Tq, Tk, d / heads);
  NMTGAN_REQUIRE(lds <= ATTN_BWD_LDS_MAX, "attn_bwd: Tq=%d Tk=%d head width %d need %zu bytes of LDS (> 160 KiB)", Tq, Tk,
                 d / heads, lds);
  if (op == NMTGAN_BF16)
    return attn_bwd_t<bf16_t>(q, k, v, dO, B, Tq, Tk, d, heads, causal, mask, dq, ld_dq, dk, ld_dk, dv, ld_dv, dq_scale, lds, st);
  return attn_bwd_t<float>(q, k, v, dO, B, Tq, Tk, d, heads, causal, mask, dq, ld_dq, dk, ld_dk, dv, ld_dv, dq_scale, lds, st);
}

// dtable[tok[r]] += mult * dx[r]   (tf.gather backward; rows of one token collide -> float atomics)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* dx, const int32_t* tok, int M, int d, int vocab, float mult,
                                                        float* dtable) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const int t = min(max(tok[r], 0), vocab - 1);   // same clamp as the forward gather
  for (int c = lane; c < d; c += 64) atomicAdd(dtable + (int64_t)t * d + c, mult * dx[(int64_t)r * d + c]);
}

// tf.train.RMSPropOptimizer dense update (model.py:119; decay 0.9, momentum 0, epsilon 1e-10 by default):
//   ms += (g*g - ms) * (1 - decay);  mom = mom * momentum + lr * g / sqrt(ms + eps);  w -= mom
// g = grad * grad_scale (tower average, model.py:689-722); w_op (optional) = the refreshed MFMA-operand copy.
template <typename OpT>
__global__ void rmsprop_kernel(float* w, float* ms, float* mom, const float* grad, int64_t n, float lr, float decay,
                               float momentum, float eps, float grad_scale, float grad_clip, OpT* w_op) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float g = grad[i] * grad_scale;
  if (grad_clip > 0.f) g = fminf(fmaxf(g, -grad_clip), grad_clip);
  const float m = ms[i] + (g * g - ms[i]) * (1.0f - decay);
  ms[i] = m;
  float step = lr * g / sqrtf(m + eps);
  if (mom) { step += mom[i] * momentum; mom[i] = step; }
  const float nw = w[i] - step;
  w[i] = nw;
  if (w_op) OpTraits<OpT>::store(w_op + i, nw);
}

int rmsprop_update(float* w, float* ms, float* mom, const float* grad, int64_t n, float lr, float decay, float momentum,
                   float eps, float grad_scale, float grad_clip, void* w_op, int op_dtype, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  if (n == 0) return 0;
  NMTGAN_REQUIRE(w && ms && grad && n > 0, "rmsprop_update: null pointer");
  NMTGAN_REQUIRE(lr >= 0.f && decay >= 0.f && decay <= 1.f && eps >= 0.f, "rmsprop_update: bad hyper-parameter");
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (w_op && op_dtype == NMTGAN_BF16)
    hipLaunchKernelGGL(rmsprop_kernel<bf16_t>, grid, block, 0, st, w, ms, mom, grad, n, lr, decay, momentum, eps, grad_scale,
                       grad_clip, (bf16_t*)w_op);
  else
    hipLaunchKernelGGL(rmsprop_kernel<float>, grid, block, 0, st, w, ms, mom, grad, n, lr, decay, momentum, eps, grad_scale,
                       grad_clip, (float*)(op_dtype == NMTGAN_F32 ? w_op : nullptr));
  NMTGAN_CHECK_LAUNCH("rmsprop");
  return 0;
}

// tf.train.AdadeltaOptimizer() (model.py:115-116: learning_rate 0.001, rho 0.95, epsilon 1e-8, both slots 0), dense:
//   accum = rho accum + (1 - rho) g^2;  upd = sqrt(accum_update + eps) * rsqrt(accum + eps) * g;
//   accum_update = rho accum_update + (1 - rho) upd^2;  w -= lr upd
template <typename OpT>
__global__ void adadelta_kernel(float* w, float* accum, float* accum_update, const float* grad, int64_t n, float lr,
                                float rho, float eps, float grad_scale, float grad_clip, OpT* w_op) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float g = grad[i] * grad_scale;
  if (grad_clip > 0.f) g = fminf(fmaxf(g, -grad_clip), grad_clip);
  const float a = accum[i] * rho + g * g * (1.0f - rho);
  accum[i] = a;
  const float upd = sqrtf(accum_update[i] + eps) * rsqrtf(a + eps) * g;
  accum_update[i] = accum_update[i] * rho + upd * upd * (1.0f - rho);
  const float nw = w[i] - lr * upd;
  w[i] = nw;
  if (w_op) OpTraits<OpT>::store(w_op + i, nw);
}

int adadelta_update(float* w, float* accum, float* accum_update, const float* grad, int64_t n, float lr, float rho,
                    float eps, float grad_scale, float grad_clip, void* w_op, int op_dtype, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  if (n == 0) return 0;
  NMTGAN_REQUIRE(w && accum && accum_update && grad && n > 0, "adadelta_update: null pointer");
  NMTGAN_REQUIRE(lr >= 0.f && rho >= 0.f && rho <= 1.f && eps > 0.f, "adadelta_update: bad hyper-parameter");
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (w_op && op_dtype == NMTGAN_BF16)
    hipLaunchKernelGGL(adadelta_kernel<bf16_t>, grid, block, 0, st, w, accum, accum_update, grad, n, lr, rho, eps, grad_scale,
                       grad_clip, (bf16_t*)w_op);
  else
    hipLaunchKernelGGL(adadelta_kernel<float>, grid, block, 0, st, w, accum, accum_update, grad, n, lr, rho, eps, grad_scale,
                       grad_clip, (float*)(op_dtype == NMTGAN_F32 ? w_op : nullptr));
  NMTGAN_CHECK_LAUNCH("adadelta");
  return 0;
}

// The discriminator's two other optimizers (cnn_discriminator.py:153-161), dense, same conventions as above:
//   tf.train.GradientDescentOptimizer(0.0001):  w -= lr g
//   tf.train.AdamOptimizer() (lr 0.001, beta1 0.9, beta2 0.999, epsilon 1e-8; t = 1, 2, ..):
//     m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  w -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)
template <typename OpT>
__global__ void sgd_adam_kernel(float* w, float* m, float* v, const float* grad, int64_t n, float lr_t, float b1, float b2,
                                float eps, float grad_scale, float grad_clip, OpT* w_op) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float g = grad[i] * grad_scale;
  if (grad_clip > 0.f) g = fminf(fmaxf(g, -grad_clip), grad_clip);
  float step = lr_t * g;                                   // plain gradient descent when there are no slots
  if (m) {
    const float mi = b1 * m[i] + (1.0f - b1) * g, vi = b2 * v[i] + (1.0f - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    step = lr_t * mi / (sqrtf(vi) + eps);
  }
  const float nw = w[i] - step;
  w[i] = nw;
  if (w_op) OpTraits<OpT>::store(w_op + i, nw);
}
int sgd_adam_update(float* w, float* m, float* v, const float* grad, int64_t n, float lr, float b1, float b2, float eps,
                    int t, float grad_scale, float grad_clip, void* w_op, int op_dtype, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  if (n == 0) return 0;
  NMTGAN_REQUIRE(w && grad && n > 0 && ((m == nullptr) == (v == nullptr)), "sgd/adam update: null pointer");
  NMTGAN_REQUIRE(lr >= 0.f && (!m || (t >= 1 && b1 >= 0.f && b1 < 1.f && b2 >= 0.f && b2 < 1.f && eps > 0.f)),
                 "sgd/adam update: bad hyper-parameter");
  const float lr_t = m ? (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t))) : lr;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (w_op && op_dtype == NMTGAN_BF16)
    hipLaunchKernelGGL(sgd_adam_kernel<bf16_t>, grid, block, 0, st, w, m, v, grad, n, lr_t, b1, b2, eps, grad_scale, grad_clip,
                       (bf16_t*)w_op);
  else
    hipLaunchKernelGGL(sgd_adam_kernel<float>, grid, block, 0, st, w, m, v, grad, n, lr_t, b1, b2, eps, grad_scale, grad_clip,
                       (float*)(op_dtype == NMTGAN_F32 ? w_op : nullptr));
  NMTGAN_CHECK_LAUNCH("sgd_adam");
  return 0;
}

// ------------------------------------------------------------------------------------------ host helpers
static int gemm_f32out(int op, const void* A, int64_t lda, int M, int K, int N, const void* Wt, const float* bias,
                       float* out, int64_t ldo, hipStream_t st) {
  nmtgan_gemm_args a{};
  a.op_dtype = op; a.epilogue = NMTGAN_EPI_LINEAR;
  a.A = A; a.Wt = Wt; a.M = M; a.N = N; a.K = K; a.lda = lda;
  a.bias = bias; a.n_seg = 1; a.seg_width = N; a.out[0] = out; a.ldo[0] = ldo; a.out_f32[0] = 1; a.seg_scale[0] = 1.f;
  return launch_gemm(&a, st);
}
static int dense_seg(int op, const void* A, int64_t lda, int M, int K, int N, const nmtgan_dense& w, int n_seg, int seg_width,
                     void* const* outs, const int* out_f32, const float* scales, int relu, hipStream_t st) {
  nmtgan_gemm_args a{};
  a.op_dtype = op; a.epilogue = NMTGAN_EPI_LINEAR;
  a.A = A; a.Wt = w.w; a.M = M; a.N = N; a.K = K; a.lda = lda; a.bias = w.b; a.relu = relu;
  a.n_seg = n_seg; a.seg_width = seg_width;
  for (int s = 0; s < n_seg; ++s) {
    a.out[s] = outs[s]; a.ldo[s] = seg_width; a.out_f32[s] = out_f32[s]; a.seg_scale[s] = scales ? scales[s] : 1.f;
  }
  return launch_gemm(&a, st);
}

// out[r][c] = in[r][c] for c < C, 0 for C <= c < Cp (op dtype): gives a GEMM operand whose row length meets the
// 16-byte rule when C does not (discriminator filter counts 100 + 10 w)
template <typename OpT>
__global__ void cast_pad_kernel(const float* in, int R, int C, int Cp, OpT* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)R * Cp) return;
  const int c = (int)(i % Cp);
  const int64_t r = i / Cp;
  OpTraits<OpT>::store(out + i, c < C ? in[r * C + c] : 0.f);
}
static int launch_cast_pad(int op, const float* in, int R, int C, int Cp, void* out, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  const int64_t n = (int64_t)R * Cp;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (op == NMTGAN_BF16) hipLaunchKernelGGL(cast_pad_kernel<bf16_t>, grid, block, 0, st, in, R, C, Cp, (bf16_t*)out);
  else hipLaunchKernelGGL(cast_pad_kernel<float>, grid, block, 0, st, in, R, C, Cp, (float*)out);
  NMTGAN_CHECK_LAUNCH("cast_pad");
  return 0;
}

struct BwdScratch { void* dYo; void* dYT; void* XT; void* WT; float* colsum; void* splitk; size_t splitk_bytes; };
static const size_t SPLITK_BYTES = (size_t)64 << 20;      // split-K partials of the few-tile dW products

// Backward of Y = X . W + b with Wt[N][K] packed: gw.w <- dWt [N][K] fp32, gw.b <- colsum(dY), dX (optional) [M][K].
// dY is fp32 [M][N] (dY_op == nullptr) or already in op dtype (dY_f32 == nullptr; no bias then).
static int dense_bwd(int op, const float* dY_f32, const void* dY_op, int M, int N, int K, const void* X, int x_f32,
                     const nmtgan_dense& w, const nmtgan_dense& gw, float* dX, const BwdScratch& s, hipStream_t st,
                     int64_t x_ld = 0, int x_group = 0, int64_t x_group_stride = 0) {
  const int Mp = (M + 63) / 64 * 64;
  if (gw.b) {
    NMTGAN_REQUIRE(dY_f32 != nullptr, "dense_bwd: bias gradient needs the fp32 dY");
    NMTGAN_TRY(launch_colsum(dY_f32, N, M, N, (float*)gw.b, s.colsum, st));
  }
  const int Np = (N + 7) / 8 * 8;                 // operand row length of the dX product (16-byte rule)
  const void* dYo = dY_op;
  if (dY_f32) {
    NMTGAN_TRY(launch_transpose(op, 1, dY_f32, N, M, N, s.dYT, Mp, st));
    if (dX && Np != N) { NMTGAN_TRY(launch_cast_pad(op, dY_f32, M, N, Np, s.dYo, st)); dYo = s.dYo; }
    else if (op == NMTGAN_F32) dYo = dY_f32;
    else if (dX) { NMTGAN_TRY(launch_cast_rows(dY_f32, s.dYo, op, (int64_t)M * N, st)); dYo = s.dYo; }
  } else {
    NMTGAN_REQUIRE(Np == N || !dX, "dense_bwd: op-dtype dY needs N %% 8 == 0");
    NMTGAN_TRY(launch_transpose(op, 0, dY_op, N, M, N, s.dYT, Mp, st));
  }
  NMTGAN_TRY(launch_transpose(op, x_f32, X, x_ld > 0 ? x_ld : K, M, K, s.XT, Mp, st, x_group, x_group_stride));
  NMTGAN_TRY(gemm_f32out(op, s.dYT, Mp, N, Mp, K, s.XT, nullptr, (float*)gw.w, K, st));          // dWt = dY^T . X
  if (dX) {
    NMTGAN_TRY(launch_transpose(op, 0, w.w, K, N, K, s.WT, Np, st));                              // W^T as [K][Np]
    NMTGAN_TRY(gemm_f32out(op, dYo, Np, M, Np, K, s.WT, nullptr, dX, K, st));                     // dX = dY . W
  }
  return 0;
}

static int ln_forward(const nmtgan_generator* g, const float* x, const float* y, const nmtgan_ln& ln, int M, float* out,
                      void* out_op, const uint8_t* zero_row, hipStream_t st) {
  nmtgan_addln_args a{};
  a.x = x; a.y = y; a.gamma = ln.g; a.beta = ln.b; a.M = M; a.d = g->d;
  a.out_f32 = out; a.out_op = out_op; a.op_dtype = g->op_dtype; a.zero_row = zero_row;
  return launch_add_layernorm(&a, st);
}
static int ln_backward(const nmtgan_generator* g, const float* x, const float* y, const nmtgan_ln& ln, const nmtgan_ln& gln,
                       const float* dout, const uint8_t* zero_row, int M, float* ds, float* part, int part_blocks,
                       hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  const int d = g->d;
  const int blocks = std::min(part_blocks, (M + 3) / 4);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), (size_t)8 * d * sizeof(float), st, x, y, ln.g, dout, zero_row, M,
                     d, ds, part);
  NMTGAN_CHECK_LAUNCH("ln_bwd");
  NMTGAN_TRY(launch_colsum(part, 2 * d, blocks, d, (float*)gln.g, nullptr, st));
  NMTGAN_TRY(launch_colsum(part + d, 2 * d, blocks, d, (float*)gln.b, nullptr, st));
  return 0;
}

// ------------------------------------------------------------------------------------------ the step
static const int LN_PART_BLOCKS = 256;

struct PgBufs {
  // encoder
  std::vector<float*> Xe;                 // [2N+1] residual stream after embedding / every LayerNorm, [Me][d]
  std::vector<void*> qe, ke, ve, aoe, he; // per block, op dtype
  std::vector<float*> y0e, y1e;           // sub-layer outputs before the residual add
  void* enc_op;                           // encoder output, op dtype
  // decoder
  std::vector<float*> Xd;                 // [3N+1]
  std::vector<void*> qs, ks, vs, aos, qc, ck, cv, aoc, hd;
  std::vector<float*> ys, yc, yf;
  void* xo;                               // op copy of the current residual stream (forward GEMM operand)
  float* logits; void* dlo; float* loss_rows;
  // backward flow
  float *dx, *ds, *dwide, *dao, *dq, *dkv, *denc, *dtmp, *ln_part;
  BwdScratch s;
};

static void carve_pg(Arena& ar, const nmtgan_generator* g, int B, int Ls, int T, PgBufs& b) {
  const int d = g->d, N = g->n_blocks, es = dtype_size(g->op_dtype), V = g->vocab_dst;
  const int64_t Me = (int64_t)B * Ls, Md = (int64_t)B * T, Mx = std::max(Me, Md), Mp = (Mx + 63) / 64 * 64;
  b.Xe.resize(2 * N + 1); b.Xd.resize(3 * N + 1);
  for (auto& p : b.Xe) p = ar.get<float>(Me * d);
  for (auto& p : b.Xd) p = ar.get<float>(Md * d);
  auto ops = [&](std::vector<void*>& v, int64_t rows, int width) {
    v.resize(N);
    for (auto& p : v) p = ar.take(rows * width * es);
  };
  auto f32s = [&](std::vector<float*>& v, int64_t rows) {
    v.resize(N);
    for (auto& p : v) p = ar.get<float>(rows * d);
  };
  ops(b.qe, Me, d); ops(b.ke, Me, d); ops(b.ve, Me, d); ops(b.aoe, Me, d); ops(b.he, Me, 4 * d);
  f32s(b.y0e, Me); f32s(b.y1e, Me);
  b.enc_op = ar.take(Me * d * es);
  ops(b.qs, Md, d); ops(b.ks, Md, d); ops(b.vs, Md, d); ops(b.aos, Md, d); ops(b.qc, Md, d);
  ops(b.ck, Me, d); ops(b.cv, Me, d); ops(b.aoc, Md, d); ops(b.hd, Md, 4 * d);
  f32s(b.ys, Md); f32s(b.yc, Md); f32s(b.yf, Md);
  b.xo = ar.take(Mx * d * es);
  b.logits = ar.get<float>(Md * V);
  b.dlo = ar.take(Md * V * es);
  b.loss_rows = ar.get<float>(Md);
  b.dx = ar.get<float>(Mx * d); b.ds = ar.get<float>(Mx * d); b.dwide = ar.get<float>(Mx * 4 * d);
  b.dao = ar.get<float>(Mx * d); b.dq = ar.get<float>(Md * d); b.dkv = ar.get<float>(Me * 2 * d);
  b.denc = ar.get<float>(Me * d); b.dtmp = ar.get<float>(Mx * d);
  b.ln_part = ar.get<float>((int64_t)LN_PART_BLOCKS * 2 * d);
  const int64_t wideN = std::max<int64_t>(4 * d, V);
  b.s.dYo = ar.take(Mx * 4 * d * es);
  b.s.dYT = ar.take(wideN * Mp * es);
  b.s.XT = ar.take((int64_t)4 * d * Mp * es);
  b.s.WT = ar.take(std::max<int64_t>((int64_t)4 * d * d, (int64_t)d * V) * es);
  b.s.colsum = ar.get<float>((int64_t)COLSUM_CHUNKS * 4 * d);
  b.s.splitk_bytes = SPLITK_BYTES;
  b.s.splitk = ar.take(SPLITK_BYTES);
}

static int check_pg(const nmtgan_generator* g, const nmtgan_generator* gr, int B, int Ls, int T) {
  NMTGAN_REQUIRE(g != nullptr, "pg_grads: null model");
  NMTGAN_REQUIRE(g->op_dtype == NMTGAN_BF16 || g->op_dtype == NMTGAN_F32, "pg_grads: bad op_dtype");
  NMTGAN_REQUIRE(g->d > 0 && g->d % 64 == 0 && g->d <= 1024, "pg_grads: hidden_units must be a multiple of 64, <= 1024");
  NMTGAN_REQUIRE(g->heads > 0 && 64 % g->heads == 0 && g->d % g->heads == 0, "pg_grads: num_heads must divide 64");
  NMTGAN_REQUIRE(g->n_blocks > 0 && g->n_blocks <= NMTGAN_MAX_BLOCKS, "pg_grads: bad num_blocks");
  NMTGAN_REQUIRE(g->vocab_dst % 8 == 0, "pg_grads: dst_vocab_size must be a multiple of 8");
  NMTGAN_REQUIRE(B > 0 && Ls > 0 && T > 0 && Ls <= g->max_pos && T <= g->max_pos, "pg_grads: bad B/Ls/T");
  {
    const int L = std::max(T, Ls);
    NMTGAN_REQUIRE(attn_bwd_lds(L, L, g->d / g->heads) <= ATTN_BWD_LDS_MAX,
                   "pg_grads: max(Ls, T) = %d with head width %d exceeds the LDS budget of the attention backward", L,
                   g->d / g->heads);
  }
  if (!gr) return 0;                       // workspace query: shapes only
  NMTGAN_REQUIRE(gr->emb_src && gr->emb_dst && gr->softmax_w, "pg_grads: null gradient table");
  // shared_embedding (model.py:777-818: one table for source, target and softmax): the three gradient pointers
  // name ONE buffer and the contributions are accumulated; anything in between is a caller error
  const bool shared = (const void*)gr->emb_dst == gr->softmax_w;
  NMTGAN_REQUIRE(shared ? gr->emb_src == gr->emb_dst && g->emb_src == g->emb_dst && g->vocab_src == g->vocab_dst
                        : gr->emb_src != gr->emb_dst && (const void*)gr->emb_src != gr->softmax_w,
                 "pg_grads: gradient tables must be all distinct or all the same buffer (shared_embedding)");
  return 0;
}

int pg_workspace(const nmtgan_generator* g, int B, int Ls, int T, size_t* bytes) {
  NMTGAN_TRY(check_pg(g, nullptr, B, Ls, T));
  Arena ar(nullptr, 0);
  PgBufs b;
  carve_pg(ar, g, B, Ls, T, b);
  *bytes = align_up(ar.off, 256) + 256;
  return 0;
}

int pg_grads(const nmtgan_generator* g, const nmtgan_generator* gr, const int32_t* x, const uint8_t* x_pad, const int32_t* y,
             const int32_t* dec_in, const float* reward, int B, int Ls, int T, float* loss, float* p_target, void* ws,
             size_t ws_bytes, hipStream_t st) {
  NMTGAN_REQUIRE(gr != nullptr, "pg_grads: null gradient struct");
  NMTGAN_TRY(check_pg(g, gr, B, Ls, T));
  NMTGAN_REQUIRE(x && x_pad && y && dec_in && reward && loss && ws, "pg_grads: null pointer");
  Arena ar(ws, ws_bytes);
  PgBufs b;
  carve_pg(ar, g, B, Ls, T, b);
  NMTGAN_REQUIRE(!ar.overflow, "pg_grads: workspace too small (%zu bytes given)", ws_bytes);
  const int op = g->op_dtype, d = g->d, N = g->n_blocks, H = g->heads, V = g->vocab_dst;
  const int Me = B * Ls, Md = B * T;
  const float qscale = 1.0f / sqrtf((float)(d / H));
  const int f0[3] = {0, 0, 0};
  SplitKScope split_k(b.s.splitk, b.s.splitk_bytes);     // everything below runs on ONE stream

  // ---------------- forward: encoder (model.py:486-543)
  {
    nmtgan_embed_args e{};
    e.table = g->emb_src; e.timing = g->timing; e.vocab = g->vocab_src; e.max_pos = g->max_pos;
    e.tok = x; e.tok_ld = 1; e.pos_mod = Ls; e.mult = g->emb_mult; e.M = Me; e.d = d;
    e.out_f32 = b.Xe[0]; e.out_op = b.xo; e.op_dtype = op;
    NMTGAN_TRY(launch_embed(&e, st));
  }
  for (int l = 0; l < N; ++l) {
    const nmtgan_enc_block& w = g->enc[l];
    void* outs[3] = {b.qe[l], b.ke[l], b.ve[l]};
    const float sc[3] = {qscale, 1.f, 1.f};
    NMTGAN_TRY(dense_seg(op, b.xo, d, Me, d, 3 * d, w.qkv, 3, d, outs, f0, sc, 0, st));
    nmtgan_attn_args t{};
    t.q = b.qe[l]; t.q_op_dtype = 1; t.M = Me; t.d = d; t.heads = H; t.kv_dtype = op;
    t.k1 = b.ke[l]; t.v1 = b.ve[l]; t.len1 = Ls; t.rows_per_blk = Ls; t.causal = 0; t.mask1 = x_pad; t.out = b.aoe[l];
    NMTGAN_TRY(launch_attn_rows(&t, st));
    NMTGAN_TRY(gemm_f32out(op, b.aoe[l], d, Me, d, d, w.o.w, w.o.b, b.y0e[l], d, st));
    NMTGAN_TRY(ln_forward(g, b.Xe[2 * l], b.y0e[l], w.ln0, Me, b.Xe[2 * l + 1], b.xo, nullptr, st));
    void* ho[1] = {b.he[l]};
    NMTGAN_TRY(dense_seg(op, b.xo, d, Me, d, 4 * d, w.ffn1, 1, 4 * d, ho, f0, nullptr, 1, st));
    NMTGAN_TRY(gemm_f32out(op, b.he[l], 4 * d, Me, 4 * d, d, w.ffn2.w, w.ffn2.b, b.y1e[l], d, st));
    NMTGAN_TRY(ln_forward(g, b.Xe[2 * l + 1], b.y1e[l], w.ln1, Me, b.Xe[2 * l + 2], l == N - 1 ? b.enc_op : b.xo,
                          l == N - 1 ? x_pad : nullptr, st));                                     // model.py:542
  }
  for (int l = 0; l < N; ++l) {                    // cross-attention K/V of every decoder block
    void* outs[2] = {b.ck[l], b.cv[l]};
    NMTGAN_TRY(dense_seg(op, b.enc_op, d, Me, d, 2 * d, g->dec[l].ckv, 2, d, outs, f0, nullptr, 0, st));
  }
  // ---------------- forward: decoder over shift_right(Y) (model.py:545-618) + top (model.py:804-820)
  {
    nmtgan_embed_args e{};
    e.table = g->emb_dst; e.timing = g->timing; e.vocab = g->vocab_dst; e.max_pos = g->max_pos;
    e.tok = dec_in; e.tok_ld = 1; e.pos_mod = T; e.mult = 2.0f * g->emb_mult; e.M = Md; e.d = d;      // quirk B.2
    e.out_f32 = b.Xd[0]; e.out_op = b.xo; e.op_dtype = op;
    NMTGAN_TRY(launch_embed(&e, st));
  }
  for (int l = 0; l < N; ++l) {
    const nmtgan_dec_block& w = g->dec[l];
    void* outs[3] = {b.qs[l], b.ks[l], b.vs[l]};
    const float sc[3] = {qscale, 1.f, 1.f};
    NMTGAN_TRY(dense_seg(op, b.xo, d, Md, d, 3 * d, w.qkv, 3, d, outs, f0, sc, 0, st));
    nmtgan_attn_args t{};
    t.q = b.qs[l]; t.q_op_dtype = 1; t.M = Md; t.d = d; t.heads = H; t.kv_dtype = op;
    t.k1 = b.ks[l]; t.v1 = b.vs[l]; t.len1 = T; t.rows_per_blk = T; t.causal = 1; t.out = b.aos[l];
    NMTGAN_TRY(launch_attn_rows(&t, st));
    NMTGAN_TRY(gemm_f32out(op, b.aos[l], d, Md, d, d, w.o.w, w.o.b, b.ys[l], d, st));
    NMTGAN_TRY(ln_forward(g, b.Xd[3 * l], b.ys[l], w.ln0, Md, b.Xd[3 * l + 1], b.xo, nullptr, st));
    void* qo[1] = {b.qc[l]};
    const float qs1[1] = {qscale};
    NMTGAN_TRY(dense_seg(op, b.xo, d, Md, d, d, w.cq, 1, d, qo, f0, qs1, 0, st));
    nmtgan_attn_args c{};
    c.q = b.qc[l]; c.q_op_dtype = 1; c.M = Md; c.d = d; c.heads = H; c.kv_dtype = op;
    c.k1 = b.ck[l]; c.v1 = b.cv[l]; c.len1 = Ls; c.rows_per_blk = T; c.causal = 0; c.mask1 = x_pad; c.out = b.aoc[l];
    NMTGAN_TRY(launch_attn_rows(&c, st));
    NMTGAN_TRY(gemm_f32out(op, b.aoc[l], d, Md, d, d, w.co.w, w.co.b, b.yc[l], d, st));
    NMTGAN_TRY(ln_forward(g, b.Xd[3 * l + 1], b.yc[l], w.ln1, Md, b.Xd[3 * l + 2], b.xo, nullptr, st));
    void* ho[1] = {b.hd[l]};
    NMTGAN_TRY(dense_seg(op, b.xo, d, Md, d, 4 * d, w.ffn1, 1, 4 * d, ho, f0, nullptr, 1, st));
    NMTGAN_TRY(gemm_f32out(op, b.hd[l], 4 * d, Md, 4 * d, d, w.ffn2.w, w.ffn2.b, b.yf[l], d, st));
    NMTGAN_TRY(ln_forward(g, b.Xd[3 * l + 2], b.yf[l], w.ln2, Md, b.Xd[3 * l + 3], b.xo, nullptr, st));
  }
  NMTGAN_TRY(gemm_f32out(op, b.xo, d, Md, d, V, g->softmax_w, nullptr, b.logits, V, st));

  // ---------------- loss and d(logits) (model.py:657-665)
  {
    prof::Scope prof_scope(prof::OTHER, st);
    if (op == NMTGAN_BF16)
      hipLaunchKernelGGL(pg_dlogits_kernel<bf16_t>, dim3(Md), dim3(256), 0, st, b.logits, V, y, reward, 1.0f / (float)B,
                         (bf16_t*)b.dlo, b.loss_rows, p_target);
    else
      hipLaunchKernelGGL(pg_dlogits_kernel<float>, dim3(Md), dim3(256), 0, st, b.logits, V, y, reward, 1.0f / (float)B,
                         (float*)b.dlo, b.loss_rows, p_target);
    NMTGAN_CHECK_LAUNCH("pg_dlogits");
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, b.loss_rows, Md, loss);
    NMTGAN_CHECK_LAUNCH("loss_sum");
  }

  // ---------------- backward: top
  {
    nmtgan_dense w{g->softmax_w, nullptr}, gw{gr->softmax_w, nullptr};
    NMTGAN_TRY(dense_bwd(op, nullptr, b.dlo, Md, V, d, b.Xd[3 * N], 1, w, gw, b.dx, b.s, st));
  }
  NMTGAN_TRY(launch_zero(b.denc, (size_t)Me * d * sizeof(float), st));
  // ---------------- backward: decoder blocks
  for (int l = N - 1; l >= 0; --l) {
    const nmtgan_dec_block& w = g->dec[l];
    const nmtgan_dec_block& gw = gr->dec[l];
    // FFN sub-layer: X[3l+3] = LN(X[3l+2] + ffn(X[3l+2]))
    NMTGAN_TRY(ln_backward(g, b.Xd[3 * l + 2], b.yf[l], w.ln2, gw.ln2, b.dx, nullptr, Md, b.ds, b.ln_part, LN_PART_BLOCKS, st));
    NMTGAN_TRY(dense_bwd(op, b.ds, nullptr, Md, d, 4 * d, b.hd[l], 0, w.ffn2, gw.ffn2, b.dwide, b.s, st));
    NMTGAN_TRY(launch_relu_bwd(op, b.dwide, b.hd[l], (int64_t)Md * 4 * d, st));
    NMTGAN_TRY(dense_bwd(op, b.dwide, nullptr, Md, 4 * d, d, b.Xd[3 * l + 2], 1, w.ffn1, gw.ffn1, b.dtmp, b.s, st));
    NMTGAN_TRY(launch_add_inplace(b.ds, b.dtmp, (int64_t)Md * d, st));                      // residual + FFN input
    // cross-attention sub-layer: X[3l+2] = LN(X[3l+1] + cross(X[3l+1], enc))
    NMTGAN_TRY(ln_backward(g, b.Xd[3 * l + 1], b.yc[l], w.ln1, gw.ln1, b.ds, nullptr, Md, b.dx, b.ln_part, LN_PART_BLOCKS, st));
    NMTGAN_TRY(dense_bwd(op, b.dx, nullptr, Md, d, d, b.aoc[l], 0, w.co, gw.co, b.dao, b.s, st));
    NMTGAN_TRY(launch_attn_bwd(op, b.qc[l], b.ck[l], b.cv[l], b.dao, B, T, Ls, d, H, 0, x_pad, b.dq, d, b.dkv, 2 * d,
                               b.dkv + d, 2 * d, qscale, st));
    NMTGAN_TRY(dense_bwd(op, b.dq, nullptr, Md, d, d, b.Xd[3 * l + 1], 1, w.cq, gw.cq, b.dtmp, b.s, st));
    NMTGAN_TRY(launch_add_inplace(b.dx, b.dtmp, (int64_t)Md * d, st));
    NMTGAN_TRY(dense_bwd(op, b.dkv, nullptr, Me, 2 * d, d, b.enc_op, 0, w.ckv, gw.ckv, b.dtmp, b.s, st));
    NMTGAN_TRY(launch_add_inplace(b.denc, b.dtmp, (int64_t)Me * d, st));
    // self-attention sub-layer: X[3l+1] = LN(X[3l] + self(X[3l]))
    NMTGAN_TRY(ln_backward(g, b.Xd[3 * l], b.ys[l], w.ln0, gw.ln0, b.dx, nullptr, Md, b.ds, b.ln_part, LN_PART_BLOCKS, st));
    NMTGAN_TRY(dense_bwd(op, b.ds, nullptr, Md, d, d, b.aos[l], 0, w.o, gw.o, b.dao, b.s, st));
    NMTGAN_TRY(launch_attn_bwd(op, b.qs[l], b.ks[l], b.vs[l], b.dao, B, T, T, d, H, 1, nullptr, b.dwide, 3 * d,
                               b.dwide + d, 3 * d, b.dwide + 2 * d, 3 * d, qscale, st));
    NMTGAN_TRY(dense_bwd(op, b.dwide, nullptr, Md, 3 * d, d, b.Xd[3 * l], 1, w.qkv, gw.qkv, b.dx, b.s, st));
    NMTGAN_TRY(launch_add_inplace(b.dx, b.ds, (int64_t)Md * d, st));
  }
  // decoder embedding: h = 2 * sqrt(d) * E_dst[shift_right(Y)] + timing (model.py:565)
  {
    prof::Scope prof_scope(prof::OTHER, st);
    if ((const void*)gr->emb_dst != gr->softmax_w)     // shared table: add to the softmax gradient written above
      NMTGAN_TRY(launch_zero((void*)gr->emb_dst, (size_t)g->vocab_dst * d * sizeof(float), st));
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((Md + 3) / 4), dim3(256), 0, st, b.dx, dec_in, Md, d, g->vocab_dst,
                       2.0f * g->emb_mult, (float*)gr->emb_dst);
    NMTGAN_CHECK_LAUNCH("embed_bwd");
  }
  // ---------------- backward: encoder blocks (gradient arrives through the cross-attention K/V)
  const float* dcur = b.denc;
  for (int l = N - 1; l >= 0; --l) {
    const nmtgan_enc_block& w = g->enc[l];
    const nmtgan_enc_block& gw = gr->enc[l];
    NMTGAN_TRY(ln_backward(g, b.Xe[2 * l + 1], b.y1e[l], w.ln1, gw.ln1, dcur, l == N - 1 ? x_pad : nullptr, Me, b.ds, b.ln_part,
                           LN_PART_BLOCKS, st));
    NMTGAN_TRY(dense_bwd(op, b.ds, nullptr, Me, d, 4 * d, b.he[l], 0, w.ffn2, gw.ffn2, b.dwide, b.s, st));
    NMTGAN_TRY(launch_relu_bwd(op, b.dwide, b.he[l], (int64_t)Me * 4 * d, st));
    NMTGAN_TRY(dense_bwd(op, b.dwide, nullptr, Me, 4 * d, d, b.Xe[2 * l + 1], 1, w.ffn1, gw.ffn1, b.dtmp, b.s, st));
    NMTGAN_TRY(launch_add_inplace(b.ds, b.dtmp, (int64_t)Me * d, st));
    NMTGAN_TRY(ln_backward(g, b.Xe[2 * l], b.y0e[l], w.ln0, gw.ln0, b.ds, nullptr, Me, b.dx, b.ln_part, LN_PART_BLOCKS, st));
    NMTGAN_TRY(dense_bwd(op, b.dx, nullptr, Me, d, d, b.aoe[l], 0, w.o, gw.o, b.dao, b.s, st));
    NMTGAN_TRY(launch_attn_bwd(op, b.qe[l], b.ke[l], b.ve[l], b.dao, B, Ls, Ls, d, H, 0, x_pad, b.dwide, 3 * d,
                               b.dwide + d, 3 * d, b.dwide + 2 * d, 3 * d, qscale, st));
    NMTGAN_TRY(dense_bwd(op, b.dwide, nullptr, Me, 3 * d, d, b.Xe[2 * l], 1, w.qkv, gw.qkv, b.denc, b.s, st));
    NMTGAN_TRY(launch_add_inplace(b.denc, b.dx, (int64_t)Me * d, st));
    dcur = b.denc;
  }
  {
    prof::Scope prof_scope(prof::OTHER, st);
    if (gr->emb_src != gr->emb_dst)
      NMTGAN_TRY(launch_zero((void*)gr->emb_src, (size_t)g->vocab_src * d * sizeof(float), st));
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((Me + 3) / 4), dim3(256), 0, st, b.denc, x, Me, d, g->vocab_src, g->emb_mult,
                       (float*)gr->emb_src);
    NMTGAN_CHECK_LAUNCH("embed_bwd");
  }
  return 0;
}


// ==========================================================================================================
// Discriminator training step: one iteration of DisCNN.train (cnn_discriminator.py:664-691)
// = build_model (248-345) in training mode for one tower + the gradients of sum(losses).
//   forward  embed -> per width: conv + bias (MFMA GEMM over the overlapping windows, raw fp32) -> BatchNorm with
//            BATCH moments over (sentence, window) per filter -> ReLU -> max over time (+ argmax) -> highway ->
//            [target | source] . W + b -> softmax cross-entropy; moving statistics moved in place (decay 0.9)
//   backward head -> highway -> per width: max/ReLU/BatchNorm backward (dense: the batch moments couple every window)
//            -> conv weights (dc^T . im2col, transposed copies) and sentence image (dc . W^T, overlap-add) -> tables
// ==========================================================================================================

// out[c] = scale * sum_r (in[r][c] - (center ? center[c] : 0))^2, two-level like colsum_kernel
__global__ __launch_bounds__(256) void colsumsq_kernel(const float* in, int64_t ld, int R, int C, int rows_per_chunk,
                                                       const float* center, float* out) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float s = 0.f;
  if (c < C) {
    const float mu = center ? center[c] : 0.f;
    for (int r = r0 + ty; r < r1; r += 4) { const float v = in[(int64_t)r * ld + c] - mu; s += v * v; }
  }
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < C) out[(int64_t)blockIdx.y * C + c] = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
}
__global__ void scale_kernel(float* v, int n, float a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] *= a;
}
// batch mean / biased variance per column of raw[R][C] (tf.nn.moments); scratch: 2 * COLSUM_CHUNKS * C floats
static int launch_col_moments(const float* raw, int R, int C, float* mean, float* var, float* scratch, hipStream_t st) {
  NMTGAN_TRY(launch_colsum(raw, C, R, C, mean, scratch, st));
  prof::Scope prof_scope(prof::OTHER, st);
  hipLaunchKernelGGL(scale_kernel, dim3((C + 255) / 256), dim3(256), 0, st, mean, C, 1.0f / (float)R);
  int chunks = std::min(COLSUM_CHUNKS, (R + 63) / 64);
  const int per = (R + chunks - 1) / chunks;
  chunks = (R + per - 1) / per;
  hipLaunchKernelGGL(colsumsq_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, st, raw, (int64_t)C, R, C, per, mean,
                     chunks > 1 ? scratch : var);
  if (chunks > 1) hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, chunks, C, var);
  hipLaunchKernelGGL(scale_kernel, dim3((C + 255) / 256), dim3(256), 0, st, var, C, 1.0f / (float)R);
  NMTGAN_CHECK_LAUNCH("col_moments");
  return 0;
}

// pooled[m][f] = max_p relu((raw[m*P+p][f] - mean) * rstd * gamma + beta), arg[m][f] = first p reaching it
// (contrib batch_norm is_training=True eps 1e-3, relu, max_pool; cnn_discriminator.py:22-33, 105, 282-283);
// also moves the moving statistics (updates_collections=None): moving = moving*decay + batch*(1-decay)
__global__ void bn_relu_max_kernel(const float* raw, const float* mean, const float* var, const float* gamma,
                                   const float* beta, int M, int P, int n, float* pooled, int64_t ld_pool, int32_t* arg,
                                   int64_t ld_arg, float* moving_mean, float* moving_var, float decay) {
  const int f = blockIdx.y * blockDim.x + threadIdx.x;      // sentences on grid.x (no 65535 limit), filter tiles on grid.y
  const int m = blockIdx.x;
  if (f >= n) return;
  const float mu = mean[f], rs = rsqrtf(var[f] + 1e-3f), g = gamma[f], b = beta[f];
  float best = 0.f;
  int bi = 0;
  for (int p = 0; p < P; ++p) {
    const float v = fmaxf((raw[((int64_t)m * P + p) * n + f] - mu) * rs * g + b, 0.f);
    if (v > best) { best = v; bi = p; }
  }
  pooled[(int64_t)m * ld_pool + f] = best;
  arg[(int64_t)m * ld_arg + f] = bi;
  if (m == 0 && moving_mean) {
    moving_mean[f] = moving_mean[f] * decay + mu * (1.0f - decay);
    moving_var[f] = moving_var[f] * decay + var[f] * (1.0f - decay);
  }
}

// per filter: dbeta = sum_m dn, dgamma = sum_m dn * xhat at the arg-max window, dn = dH[m][f] where the pooled
// value is positive (ReLU active), fixed order over m
__global__ void bn_bwd_stats_kernel(const float* dH, int64_t ld_dh, const float* pooled, int64_t ld_pool, const int32_t* arg,
                                    int64_t ld_arg, const float* raw, const float* mean, const float* var, int M, int P, int n,
                                    float* dgamma, float* dbeta) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const float mu = mean[f], rs = rsqrtf(var[f] + 1e-3f);
  float dg = 0.f, db = 0.f;
  for (int m = 0; m < M; ++m) {
    if (!(pooled[(int64_t)m * ld_pool + f] > 0.f)) continue;
    const float d = dH[(int64_t)m * ld_dh + f];
    const int p = arg[(int64_t)m * ld_arg + f];
    dg += d * (raw[((int64_t)m * P + p) * n + f] - mu) * rs;
    db += d;
  }
  dgamma[f] = dg;
  dbeta[f] = db;
}
// dc[r][f] = gamma * rstd / R * (R * dn[r][f] - dbeta - xhat[r][f] * dgamma)
__global__ void bn_bwd_dense_kernel(const float* dH, int64_t ld_dh, const float* pooled, int64_t ld_pool, const int32_t* arg,
                                    int64_t ld_arg, const float* raw, const float* mean, const float* var, const float* gamma,
                                    const float* dgamma, const float* dbeta, int M, int P, int n, float* dc) {
  const int f = blockIdx.y * blockDim.x + threadIdx.x;      // windows on grid.x, filter tiles on grid.y
  const int r = blockIdx.x;
  if (f >= n) return;
  const int m = r / P, p = r % P;
  const float mu = mean[f], rs = rsqrtf(var[f] + 1e-3f);
  const float R = (float)M * (float)P;
  float dn = 0.f;
  if (pooled[(int64_t)m * ld_pool + f] > 0.f && arg[(int64_t)m * ld_arg + f] == p) dn = dH[(int64_t)m * ld_dh + f];
  const float xh = (raw[(int64_t)r * n + f] - mu) * rs;
  dc[(int64_t)r * n + f] = gamma[f] * rs / R * (R * dn - dbeta[f] - xh * dgamma[f]);
}

// de[m][t][c] += sum_j dcol[(m, t-j)][j*E + c] over the windows p = t - j in [0, P) that cover token t
__global__ void col2im_add_kernel(const float* dcol, int M, int T, int E, int w, int P, float* de) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * T * E) return;
  const int c = (int)(i % E), t = (int)((i / E) % T), m = (int)(i / ((int64_t)E * T));
  float s = 0.f;
  for (int j = 0; j < w; ++j) {
    const int p = t - j;
    if (p >= 0 && p < P) s += dcol[((int64_t)m * P + p) * ((int64_t)w * E) + (int64_t)j * E + c];
  }
  de[i] += s;
}

// highway backward (cnn_discriminator.py:50-59): out = t*o + (1-t)*H, o = relu(o_pre), t = sigmoid(t_pre)
__global__ void highway_bwd_kernel(const float* dout, const float* H, const float* o_pre, const float* t_pre, int64_t n,
                                   float* d_o_pre, float* d_t_pre, float* dH) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float t = 1.f / (1.f + expf(-t_pre[i])), o = fmaxf(o_pre[i], 0.f), d = dout[i];
  d_o_pre[i] = o_pre[i] > 0.f ? d * t : 0.f;
  d_t_pre[i] = d * (o - H[i]) * t * (1.f - t);
  dH[i] = d * (1.f - t);
}

// scores = [ft | fs] . W + b, softmax cross-entropy against labels[m][2] (cnn_discriminator.py:318-326):
// loss_row = -sum_c y_c log p_c, ds_c = p_c * sum(y) - y_c (gradient of the SUM of the losses), correct = argmax match
__global__ __launch_bounds__(256) void disc_head_train_kernel(const float* ft, const float* fs, int M, int Fp, const float* head_w,
                                                              const float* head_b, const float* labels, float* ds,
                                                              float* loss_row, float* correct) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  float s0 = 0.f, s1 = 0.f;
  for (int k = lane; k < Fp; k += 64) {
    const float a = ft[(int64_t)r * Fp + k], b = fs[(int64_t)r * Fp + k];
    s0 += a * head_w[k] + b * head_w[Fp + k];
    s1 += a * head_w[2 * Fp + k] + b * head_w[3 * Fp + k];
  }
  s0 = wave_sum(s0) + head_b[0];
  s1 = wave_sum(s1) + head_b[1];
  if (lane == 0) {
    const float mx = fmaxf(s0, s1), lse = mx + logf(expf(s0 - mx) + expf(s1 - mx));
    const float y0 = labels[2 * r], y1 = labels[2 * r + 1];
    const float p0 = expf(s0 - lse), p1 = expf(s1 - lse);
    loss_row[r] = -(y0 * (s0 - lse) + y1 * (s1 - lse));
    ds[2 * r] = p0 * (y0 + y1) - y0;
    ds[2 * r + 1] = p1 * (y0 + y1) - y1;
    correct[r] = ((s1 > s0) == (y1 > y0)) ? 1.f : 0.f;            // tf.argmax: first index on ties
  }
}
// dW[c][k] = sum_m ds[m][c] * featcat[m][k] (fixed order), dfeat[m][k] = sum_c ds[m][c] * W[c][k]; k in [0, 2Fp)
__global__ void disc_head_bwd_kernel(const float* ft, const float* fs, const float* ds, const float* head_w, int M, int Fp,
                                     float* dW, float* dft, float* dfs) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= 2 * Fp) return;
  const float* feat = k < Fp ? ft + k : fs + (k - Fp);
  float* dfeat = k < Fp ? dft + k : dfs + (k - Fp);
  const float w0 = head_w[k], w1 = head_w[2 * Fp + k];
  float a0 = 0.f, a1 = 0.f;
  for (int m = 0; m < M; ++m) {
    const float v = feat[(int64_t)m * Fp], d0 = ds[2 * m], d1 = ds[2 * m + 1];
    a0 += d0 * v;
    a1 += d1 * v;
    dfeat[(int64_t)m * Fp] = d0 * w0 + d1 * w1;
  }
  dW[k] = a0;
  dW[2 * Fp + k] = a1;
}
// db[c] = sum_m ds[m][c]; loss = mean(loss_row); accuracy = mean(correct)   (one workgroup, fixed order)
__global__ __launch_bounds__(64) void disc_head_reduce_kernel(const float* ds, const float* loss_row, const float* correct, int M,
                                                              float* db, float* loss, float* acc) {
  if (threadIdx.x != 0) return;
  double b0 = 0, b1 = 0, l = 0, a = 0;
  for (int m = 0; m < M; ++m) { b0 += ds[2 * m]; b1 += ds[2 * m + 1]; l += loss_row[m]; a += correct[m]; }
  db[0] = (float)b0; db[1] = (float)b1;
  loss[0] = (float)(l / M);
  acc[0] = (float)(a / M);
}

// inference constants of a tower from its trainable + moving statistics (what DisCNN.load_weights folds on the host):
// scale = gamma * rsqrt(var + 1e-3), shift = (b - mean) * scale + beta
__global__ void bn_fold_kernel(const float* b, const float* gamma, const float* beta, const float* mean, const float* var,
                               int n, float* scale, float* shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float sc = gamma[i] * rsqrtf(var[i] + 1e-3f);
  scale[i] = sc;
  shift[i] = (b[i] - mean[i]) * sc + beta[i];
}
int disc_fold_bn(const float* b, const float* gamma, const float* beta, const float* mean, const float* var, int n,
                 float* scale, float* shift, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  NMTGAN_REQUIRE(b && gamma && beta && mean && var && scale && shift && n > 0, "disc_fold_bn: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, st, b, gamma, beta, mean, var, n, scale, shift);
  NMTGAN_CHECK_LAUNCH("bn_fold");
  return 0;
}
// w = clamp(w, lo, hi) (+ refreshed operand copy): the one-off clip_ops of DisCNN.train (cnn_discriminator.py:583-591, 690)
template <typename OpT>
__global__ void clip_values_kernel(float* w, int64_t n, float lo, float hi, OpT* w_op) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = fminf(fmaxf(w[i], lo), hi);
  w[i] = v;
  if (w_op) OpTraits<OpT>::store(w_op + i, v);
}
int clip_values(float* w, int64_t n, float lo, float hi, void* w_op, int op_dtype, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  if (n == 0) return 0;
  NMTGAN_REQUIRE(w && n > 0 && lo <= hi, "clip_values: bad arguments");
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (w_op && op_dtype == NMTGAN_BF16) hipLaunchKernelGGL(clip_values_kernel<bf16_t>, grid, block, 0, st, w, n, lo, hi, (bf16_t*)w_op);
  else if (w_op && op_dtype == NMTGAN_F16) hipLaunchKernelGGL(clip_values_kernel<f16_t>, grid, block, 0, st, w, n, lo, hi, (f16_t*)w_op);
  else hipLaunchKernelGGL(clip_values_kernel<float>, grid, block, 0, st, w, n, lo, hi, (float*)(op_dtype == NMTGAN_F32 ? w_op : nullptr));
  NMTGAN_CHECK_LAUNCH("clip_values");
  return 0;
}

struct DtBufs {
  void* e[2]; float* raw[2]; float* mean[2]; float* var[2]; float* pooled[2]; int32_t* arg[2]; void* pooled_op;
  float* o_pre[2]; float* t_pre[2]; float* feat[2];
  float *ds, *loss_row, *correct, *dfeat[2], *d_o, *d_t, *dH, *dtmp, *dgb, *dc, *dcol, *de;
  BwdScratch s;
};
static int64_t raw_elems(const nmtgan_discriminator* d, int M) {
  int64_t n = 0;
  for (int i = 0; i < d->n_widths; ++i) n += (int64_t)M * (d->max_len - d->widths[i] + 1) * d->n_filters[i];
  return n;
}
static void carve_dt(Arena& ar, const nmtgan_discriminator* d, int M, DtBufs& b) {
  const int T = d->max_len, E = d->dim_word, Fp = d->Fp, es = dtype_size(d->op_dtype);
  int64_t max_rn = 0, max_rk = 0, max_kn = 0, max_n = 0, max_k = 0, max_rp = 0;
  for (int i = 0; i < d->n_widths; ++i) {
    const int64_t R = (int64_t)M * (T - d->widths[i] + 1), Rp = (R + 63) / 64 * 64, n = d->n_filters[i], k = (int64_t)d->widths[i] * E;
    const int64_t np8 = (n + 7) / 8 * 8;
    max_rn = std::max(max_rn, R * np8); max_rk = std::max(max_rk, Rp * k); max_kn = std::max(max_kn, k * np8);
    max_n = std::max(max_n, n); max_k = std::max(max_k, k); max_rp = std::max(max_rp, Rp * n);
  }
  const int64_t Mp = (M + 63) / 64 * 64;
  for (int t = 0; t < 2; ++t) {
    b.e[t] = ar.take((int64_t)M * T * E * es);
    b.raw[t] = ar.get<float>(raw_elems(d, M));
    b.mean[t] = ar.get<float>(Fp); b.var[t] = ar.get<float>(Fp);
    b.pooled[t] = ar.get<float>((int64_t)M * Fp); b.arg[t] = ar.get<int32_t>((int64_t)M * Fp);
    b.o_pre[t] = ar.get<float>((int64_t)M * Fp); b.t_pre[t] = ar.get<float>((int64_t)M * Fp);
    b.feat[t] = ar.get<float>((int64_t)M * Fp); b.dfeat[t] = ar.get<float>((int64_t)M * Fp);
  }
  b.pooled_op = ar.take((int64_t)M * Fp * es);
  b.ds = ar.get<float>(2 * M); b.loss_row = ar.get<float>(M); b.correct = ar.get<float>(M);
  b.d_o = ar.get<float>((int64_t)M * Fp); b.d_t = ar.get<float>((int64_t)M * Fp); b.dH = ar.get<float>((int64_t)M * Fp);
  b.dtmp = ar.get<float>((int64_t)M * Fp); b.dgb = ar.get<float>(2 * Fp);
  b.dc = ar.get<float>(max_rn); b.dcol = ar.get<float>(max_rk); b.de = ar.get<float>((int64_t)M * T * E);
  b.s.dYo = ar.take(std::max(max_rn, (int64_t)M * Fp) * es);
  b.s.dYT = ar.take(std::max(max_rp, (int64_t)Fp * Mp) * es);
  b.s.XT = ar.take(std::max(max_rk, (int64_t)Fp * Mp) * es);
  b.s.WT = ar.take(std::max(max_kn, (int64_t)Fp * Fp) * es);
  b.s.colsum = ar.get<float>((int64_t)2 * COLSUM_CHUNKS * std::max<int64_t>(max_n, Fp));
  b.s.splitk_bytes = SPLITK_BYTES;
  b.s.splitk = ar.take(SPLITK_BYTES);
}
static int check_dt(const nmtgan_discriminator* d, const nmtgan_disc_params* p, const nmtgan_disc_params* g, int M) {
  NMTGAN_REQUIRE(d && p && g, "disc_train: null argument");
  NMTGAN_REQUIRE(d->op_dtype == NMTGAN_BF16 || d->op_dtype == NMTGAN_F32, "disc_train: bad op_dtype");
  NMTGAN_REQUIRE(!d->gemm_split, "disc_train: the split-precision form is a scoring form (train on a bf16 / fp32 copy)");
  NMTGAN_REQUIRE(d->n_widths > 0 && d->n_widths <= NMTGAN_MAX_WIDTHS && d->max_len > 0 && d->dim_word > 0 && d->dim_word % 8 == 0,
                 "disc_train: bad geometry (dim_word must be a multiple of 8)");
  NMTGAN_REQUIRE(d->Fp % 64 == 0 && d->F <= d->Fp && M > 0, "disc_train: bad F/Fp/M");
  int F = 0;
  for (int i = 0; i < d->n_widths; ++i) {
    NMTGAN_REQUIRE(d->widths[i] >= 1 && d->widths[i] <= d->max_len && d->n_filters[i] > 0,
                   "disc_train: width %d / %d filters", d->widths[i], d->n_filters[i]);
    F += d->n_filters[i];
  }
  NMTGAN_REQUIRE(F == d->F, "disc_train: F does not match the filter counts");
  return 0;
}
int disc_train_workspace(const nmtgan_discriminator* d, int M, size_t* bytes) {
  nmtgan_disc_params dummy{};
  NMTGAN_TRY(check_dt(d, &dummy, &dummy, M));
  Arena ar(nullptr, 0);
  DtBufs b;
  carve_dt(ar, d, M, b);
  *bytes = align_up(ar.off, 256) + 256;
  return 0;
}

int disc_train_grads(const nmtgan_discriminator* d, const nmtgan_disc_params* p, const nmtgan_disc_params* g, const int32_t* x,
                     const int32_t* xs, const float* labels, int M, float* loss, float* accuracy, void* ws, size_t ws_bytes,
                     hipStream_t st) {
  NMTGAN_TRY(check_dt(d, p, g, M));
  NMTGAN_REQUIRE(x && xs && labels && loss && accuracy && ws, "disc_train: null pointer");
  Arena ar(ws, ws_bytes);
  DtBufs b;
  carve_dt(ar, d, M, b);
  NMTGAN_REQUIRE(!ar.overflow, "disc_train: workspace too small (%zu bytes given)", ws_bytes);
  const int T = d->max_len, E = d->dim_word, Fp = d->Fp, op = d->op_dtype;
  SplitKScope split_k(b.s.splitk, b.s.splitk_bytes);     // everything below runs on ONE stream
  const nmtgan_disc_tower* tws[2] = {&d->tgt, &d->src};
  const nmtgan_disc_tower_params* tps[2] = {&p->tgt, &p->src};
  const nmtgan_disc_tower_params* tgs[2] = {&g->tgt, &g->src};
  const int32_t* ids[2] = {x, xs};
  // ---------------- forward
  for (int t = 0; t < 2; ++t) {
    const nmtgan_disc_tower& tw = *tws[t];
    const nmtgan_disc_tower_params& tp = *tps[t];
    NMTGAN_REQUIRE(tp.conv_b && tp.gamma && tp.beta && tp.moving_mean && tp.moving_var, "disc_train: null tower parameter");
    NMTGAN_TRY(launch_disc_embed(tw.emb, op, t ? d->vocab_s : d->vocab, ids[t], (int64_t)M * T, E, b.e[t], st));
    NMTGAN_TRY(launch_zero(b.pooled[t], (size_t)M * Fp * sizeof(float), st));
    NMTGAN_TRY(launch_zero(b.arg[t], (size_t)M * Fp * sizeof(int32_t), st));
    int col = 0;
    float* raw = b.raw[t];
    for (int i = 0; i < d->n_widths; ++i) {
      const int w = d->widths[i], P = T - w + 1, n = d->n_filters[i], R = M * P;
      nmtgan_gemm_args a{};
      a.op_dtype = op; a.epilogue = NMTGAN_EPI_LINEAR;
      a.A = b.e[t]; a.Wt = tw.conv_w[i]; a.M = R; a.N = n; a.K = w * E;
      a.lda = E; a.a_group = P; a.a_group_stride = (int64_t)T * E;
      a.bias = tp.conv_b + col; a.n_seg = 1; a.seg_width = n; a.out[0] = raw; a.ldo[0] = n; a.out_f32[0] = 1; a.seg_scale[0] = 1.f;
      NMTGAN_TRY(launch_gemm(&a, st));
      NMTGAN_TRY(launch_col_moments(raw, R, n, b.mean[t] + col, b.var[t] + col, b.s.colsum, st));
      {
        prof::Scope prof_scope(prof::OTHER, st);
        hipLaunchKernelGGL(bn_relu_max_kernel, dim3(M, (n + 63) / 64), dim3(64), 0, st, raw, b.mean[t] + col, b.var[t] + col,
                           tp.gamma + col, tp.beta + col, M, P, n, b.pooled[t] + col, (int64_t)Fp, b.arg[t] + col, (int64_t)Fp,
                           tp.moving_mean + col, tp.moving_var + col, 0.9f);
        NMTGAN_CHECK_LAUNCH("bn_relu_max");
      }
      raw += (int64_t)R * n;
      col += n;
    }
    NMTGAN_TRY(launch_cast_rows(b.pooled[t], b.pooled_op, op, (int64_t)M * Fp, st));
    NMTGAN_TRY(gemm_f32out(op, b.pooled_op, Fp, M, Fp, Fp, tw.hw_out, nullptr, b.o_pre[t], Fp, st));
    NMTGAN_TRY(gemm_f32out(op, b.pooled_op, Fp, M, Fp, Fp, tw.hw_tr, nullptr, b.t_pre[t], Fp, st));
    NMTGAN_TRY(launch_highway_gate(b.pooled[t], b.o_pre[t], b.t_pre[t], b.feat[t], (int64_t)M * Fp, st));
  }
  {
    prof::Scope prof_scope(prof::OTHER, st);
    hipLaunchKernelGGL(disc_head_train_kernel, dim3((M + 3) / 4), dim3(256), 0, st, b.feat[0], b.feat[1], M, Fp, d->head_w,
                       d->head_b, labels, b.ds, b.loss_row, b.correct);
    hipLaunchKernelGGL(disc_head_reduce_kernel, dim3(1), dim3(64), 0, st, b.ds, b.loss_row, b.correct, M, g->head_b, loss,
                       accuracy);
    hipLaunchKernelGGL(disc_head_bwd_kernel, dim3((2 * Fp + 255) / 256), dim3(256), 0, st, b.feat[0], b.feat[1], b.ds,
                       d->head_w, M, Fp, g->head_w, b.dfeat[0], b.dfeat[1]);
    NMTGAN_CHECK_LAUNCH("disc_head_train");
  }
  // ---------------- backward, tower by tower
  for (int t = 0; t < 2; ++t) {
    const nmtgan_disc_tower& tw = *tws[t];
    const nmtgan_disc_tower_params& tp = *tps[t];
    const nmtgan_disc_tower_params& tg = *tgs[t];
    NMTGAN_REQUIRE(tg.emb && tg.conv_b && tg.gamma && tg.beta && tg.hw_out && tg.hw_tr, "disc_train: null gradient buffer");
    {
      prof::Scope prof_scope(prof::OTHER, st);
      const int64_t n = (int64_t)M * Fp;
      hipLaunchKernelGGL(highway_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, b.dfeat[t], b.pooled[t],
                         b.o_pre[t], b.t_pre[t], n, b.d_o, b.d_t, b.dH);
      NMTGAN_CHECK_LAUNCH("highway_bwd");
    }
    nmtgan_dense wo{tw.hw_out, nullptr}, wt{tw.hw_tr, nullptr}, go{tg.hw_out, nullptr}, gt{tg.hw_tr, nullptr};
    NMTGAN_TRY(dense_bwd(op, b.d_o, nullptr, M, Fp, Fp, b.pooled[t], 1, wo, go, b.dtmp, b.s, st));
    NMTGAN_TRY(launch_add_inplace(b.dH, b.dtmp, (int64_t)M * Fp, st));
    NMTGAN_TRY(dense_bwd(op, b.d_t, nullptr, M, Fp, Fp, b.pooled[t], 1, wt, gt, b.dtmp, b.s, st));
    NMTGAN_TRY(launch_add_inplace(b.dH, b.dtmp, (int64_t)M * Fp, st));
    NMTGAN_TRY(launch_zero(b.de, (size_t)M * T * E * sizeof(float), st));
    int col = 0;
    const float* raw = b.raw[t];
    for (int i = 0; i < d->n_widths; ++i) {
      const int w = d->widths[i], P = T - w + 1, n = d->n_filters[i], R = M * P, K = w * E;
      NMTGAN_REQUIRE(tg.conv_w[i] != nullptr, "disc_train: null conv gradient buffer");
      {
        prof::Scope prof_scope(prof::OTHER, st);
        hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3((n + 63) / 64), dim3(64), 0, st, b.dH + col, (int64_t)Fp, b.pooled[t] + col,
                           (int64_t)Fp, b.arg[t] + col, (int64_t)Fp, raw, b.mean[t] + col, b.var[t] + col, M, P, n,
                           tg.gamma + col, tg.beta + col);
        hipLaunchKernelGGL(bn_bwd_dense_kernel, dim3(R, (n + 63) / 64), dim3(64), 0, st, b.dH + col, (int64_t)Fp,
                           b.pooled[t] + col, (int64_t)Fp, b.arg[t] + col, (int64_t)Fp, raw, b.mean[t] + col, b.var[t] + col,
                           tp.gamma + col, tg.gamma + col, tg.beta + col, M, P, n, b.dc);
        NMTGAN_CHECK_LAUNCH("bn_bwd");
      }
      nmtgan_dense cw{tw.conv_w[i], nullptr}, cg{tg.conv_w[i], tg.conv_b + col};
      NMTGAN_TRY(dense_bwd(op, b.dc, nullptr, R, n, K, b.e[t], 0, cw, cg, b.dcol, b.s, st, E, P, (int64_t)T * E));
      {
        prof::Scope prof_scope(prof::OTHER, st);
        const int64_t tot = (int64_t)M * T * E;
        hipLaunchKernelGGL(col2im_add_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, b.dcol, M, T, E, w, P, b.de);
        NMTGAN_CHECK_LAUNCH("col2im_add");
      }
      raw += (int64_t)R * n;
      col += n;
    }
    {
      prof::Scope prof_scope(prof::OTHER, st);
      const int vocab = t ? d->vocab_s : d->vocab;
      NMTGAN_TRY(launch_zero(tg.emb, (size_t)vocab * E * sizeof(float), st));
      hipLaunchKernelGGL(embed_bwd_kernel, dim3((M * T + 3) / 4), dim3(256), 0, st, b.de, ids[t], M * T, E, vocab, 1.0f, tg.emb);
      NMTGAN_CHECK_LAUNCH("disc_embed_bwd");
    }
  }
  return 0;
}

}  // namespace nmtgan
