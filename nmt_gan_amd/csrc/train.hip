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

static inline char* at(const void* base, int64_t elems, int es) { return (char*)base + elems * es; }

// ------------------------------------------------------------------------------------------ small kernels
// out[c][r] = in[r][c] for r < R, 0 for R <= r < Rp  (cast TI -> TO on the way)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const TI* in, int64_t ld_in, int R, int C, TO* out, int Rp) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    tile[ty + 4 * i][tx] = (r < R && c < C) ? OpTraits<TI>::load(in + (int64_t)r * ld_in + c) : 0.f;
  }
  __syncthreads();
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < C && r < Rp) OpTraits<TO>::store(out + (int64_t)c * Rp + r, tile[tx][ty + 4 * i]);
  }
}
template <typename TI, typename TO>
static int transpose_t(const void* in, int64_t ld_in, int R, int C, void* out, int Rp, hipStream_t st) {
  dim3 grid((Rp + 63) / 64, (C + 63) / 64), block(256);
  hipLaunchKernelGGL((transpose_cast_kernel<TI, TO>), grid, block, 0, st, (const TI*)in, ld_in, R, C, (TO*)out, Rp);
  NMTGAN_CHECK_LAUNCH("transpose_cast");
  return 0;
}
// in_f32 != 0: input is fp32, else op dtype; output always op dtype
static int launch_transpose(int op, int in_f32, const void* in, int64_t ld_in, int R, int C, void* out, int Rp,
                            hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  NMTGAN_REQUIRE(in && out && R > 0 && C > 0 && Rp >= R, "transpose: bad arguments");
  if (op == NMTGAN_F32) return transpose_t<float, float>(in, ld_in, R, C, out, Rp, st);
  return in_f32 ? transpose_t<float, bf16_t>(in, ld_in, R, C, out, Rp, st)
                : transpose_t<bf16_t, bf16_t>(in, ld_in, R, C, out, Rp, st);
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
                               float momentum, float eps, float grad_scale, OpT* w_op) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = grad[i] * grad_scale;
  const float m = ms[i] + (g * g - ms[i]) * (1.0f - decay);
  ms[i] = m;
  float step = lr * g / sqrtf(m + eps);
  if (mom) { step += mom[i] * momentum; mom[i] = step; }
  const float nw = w[i] - step;
  w[i] = nw;
  if (w_op) OpTraits<OpT>::store(w_op + i, nw);
}

int rmsprop_update(float* w, float* ms, float* mom, const float* grad, int64_t n, float lr, float decay, float momentum,
                   float eps, float grad_scale, void* w_op, int op_dtype, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  if (n == 0) return 0;
  NMTGAN_REQUIRE(w && ms && grad && n > 0, "rmsprop_update: null pointer");
  NMTGAN_REQUIRE(lr >= 0.f && decay >= 0.f && decay <= 1.f && eps >= 0.f, "rmsprop_update: bad hyper-parameter");
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (w_op && op_dtype == NMTGAN_BF16)
    hipLaunchKernelGGL(rmsprop_kernel<bf16_t>, grid, block, 0, st, w, ms, mom, grad, n, lr, decay, momentum, eps, grad_scale,
                       (bf16_t*)w_op);
  else
    hipLaunchKernelGGL(rmsprop_kernel<float>, grid, block, 0, st, w, ms, mom, grad, n, lr, decay, momentum, eps, grad_scale,
                       (float*)(op_dtype == NMTGAN_F32 ? w_op : nullptr));
  NMTGAN_CHECK_LAUNCH("rmsprop");
  return 0;
}

// tf.train.AdadeltaOptimizer() (model.py:115-116: learning_rate 0.001, rho 0.95, epsilon 1e-8, both slots 0), dense:
//   accum = rho accum + (1 - rho) g^2;  upd = sqrt(accum_update + eps) * rsqrt(accum + eps) * g;
//   accum_update = rho accum_update + (1 - rho) upd^2;  w -= lr upd
template <typename OpT>
__global__ void adadelta_kernel(float* w, float* accum, float* accum_update, const float* grad, int64_t n, float lr,
                                float rho, float eps, float grad_scale, OpT* w_op) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = grad[i] * grad_scale;
  const float a = accum[i] * rho + g * g * (1.0f - rho);
  accum[i] = a;
  const float upd = sqrtf(accum_update[i] + eps) * rsqrtf(a + eps) * g;
  accum_update[i] = accum_update[i] * rho + upd * upd * (1.0f - rho);
  const float nw = w[i] - lr * upd;
  w[i] = nw;
  if (w_op) OpTraits<OpT>::store(w_op + i, nw);
}

int adadelta_update(float* w, float* accum, float* accum_update, const float* grad, int64_t n, float lr, float rho,
                    float eps, float grad_scale, void* w_op, int op_dtype, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  if (n == 0) return 0;
  NMTGAN_REQUIRE(w && accum && accum_update && grad && n > 0, "adadelta_update: null pointer");
  NMTGAN_REQUIRE(lr >= 0.f && rho >= 0.f && rho <= 1.f && eps > 0.f, "adadelta_update: bad hyper-parameter");
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (w_op && op_dtype == NMTGAN_BF16)
    hipLaunchKernelGGL(adadelta_kernel<bf16_t>, grid, block, 0, st, w, accum, accum_update, grad, n, lr, rho, eps, grad_scale,
                       (bf16_t*)w_op);
  else
    hipLaunchKernelGGL(adadelta_kernel<float>, grid, block, 0, st, w, accum, accum_update, grad, n, lr, rho, eps, grad_scale,
                       (float*)(op_dtype == NMTGAN_F32 ? w_op : nullptr));
  NMTGAN_CHECK_LAUNCH("adadelta");
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

struct BwdScratch { void* dYo; void* dYT; void* XT; void* WT; float* colsum; };

// Backward of Y = X . W + b with Wt[N][K] packed: gw.w <- dWt [N][K] fp32, gw.b <- colsum(dY), dX (optional) [M][K].
// dY is fp32 [M][N] (dY_op == nullptr) or already in op dtype (dY_f32 == nullptr; no bias then).
static int dense_bwd(int op, const float* dY_f32, const void* dY_op, int M, int N, int K, const void* X, int x_f32,
                     const nmtgan_dense& w, const nmtgan_dense& gw, float* dX, const BwdScratch& s, hipStream_t st) {
  const int Mp = (M + 63) / 64 * 64;
  if (gw.b) {
    NMTGAN_REQUIRE(dY_f32 != nullptr, "dense_bwd: bias gradient needs the fp32 dY");
    NMTGAN_TRY(launch_colsum(dY_f32, N, M, N, (float*)gw.b, s.colsum, st));
  }
  const void* dYo = dY_op;
  if (dY_f32) {
    NMTGAN_TRY(launch_transpose(op, 1, dY_f32, N, M, N, s.dYT, Mp, st));
    if (op == NMTGAN_F32) dYo = dY_f32;
    else if (dX) { NMTGAN_TRY(launch_cast_rows(dY_f32, s.dYo, op, (int64_t)M * N, st)); dYo = s.dYo; }
  } else {
    NMTGAN_TRY(launch_transpose(op, 0, dY_op, N, M, N, s.dYT, Mp, st));
  }
  NMTGAN_TRY(launch_transpose(op, x_f32, X, K, M, K, s.XT, Mp, st));
  NMTGAN_TRY(gemm_f32out(op, s.dYT, Mp, N, Mp, K, s.XT, nullptr, (float*)gw.w, K, st));          // dWt = dY^T . X
  if (dX) {
    NMTGAN_TRY(launch_transpose(op, 0, w.w, K, N, K, s.WT, N, st));                               // W^T as [K][N]
    NMTGAN_TRY(gemm_f32out(op, dYo, N, M, N, K, s.WT, nullptr, dX, K, st));                       // dX = dY . W
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
}

static int check_pg(const nmtgan_generator* g, const nmtgan_generator* gr, int B, int Ls, int T) {
  NMTGAN_REQUIRE(g && gr, "pg_grads: null model");
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
  NMTGAN_REQUIRE((const void*)g->emb_dst != g->softmax_w && g->emb_src != g->emb_dst,
                 "pg_grads: shared_embedding is not supported by the update step");
  NMTGAN_REQUIRE(gr->emb_src && gr->emb_dst && gr->softmax_w, "pg_grads: null gradient table");
  return 0;
}

int pg_workspace(const nmtgan_generator* g, int B, int Ls, int T, size_t* bytes) {
  nmtgan_generator dummy = *g;
  NMTGAN_TRY(check_pg(g, &dummy, B, Ls, T));
  Arena ar(nullptr, 0);
  PgBufs b;
  carve_pg(ar, g, B, Ls, T, b);
  *bytes = align_up(ar.off, 256) + 256;
  return 0;
}

int pg_grads(const nmtgan_generator* g, const nmtgan_generator* gr, const int32_t* x, const uint8_t* x_pad, const int32_t* y,
             const int32_t* dec_in, const float* reward, int B, int Ls, int T, float* loss, float* p_target, void* ws,
             size_t ws_bytes, hipStream_t st) {
  NMTGAN_TRY(check_pg(g, gr, B, Ls, T));
  NMTGAN_REQUIRE(x && x_pad && y && dec_in && reward && loss && ws, "pg_grads: null pointer");
  Arena ar(ws, ws_bytes);
  PgBufs b;
  carve_pg(ar, g, B, Ls, T, b);
  NMTGAN_REQUIRE(!ar.overflow, "pg_grads: workspace too small (%zu bytes given)", ws_bytes);
  const int op = g->op_dtype, d = g->d, N = g->n_blocks, H = g->heads, V = g->vocab_dst, es = dtype_size(op);
  const int Me = B * Ls, Md = B * T;
  const float qscale = 1.0f / sqrtf((float)(d / H));
  const int f0[3] = {0, 0, 0};

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
  NMTGAN_REQUIRE(hipMemsetAsync(b.denc, 0, (size_t)Me * d * sizeof(float), st) == hipSuccess, "pg_grads: memset failed");
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
    NMTGAN_REQUIRE(hipMemsetAsync((void*)gr->emb_dst, 0, (size_t)g->vocab_dst * d * sizeof(float), st) == hipSuccess,
                   "pg_grads: memset failed");
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
    NMTGAN_REQUIRE(hipMemsetAsync((void*)gr->emb_src, 0, (size_t)g->vocab_src * d * sizeof(float), st) == hipSuccess,
                   "pg_grads: memset failed");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((Me + 3) / 4), dim3(256), 0, st, b.denc, x, Me, d, g->vocab_src, g->emb_mult,
                       (float*)gr->emb_src);
    NMTGAN_CHECK_LAUNCH("embed_bwd");
  }
  (void)es;
  return 0;
}

}  // namespace nmtgan
