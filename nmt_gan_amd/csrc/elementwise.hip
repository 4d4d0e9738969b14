// Memory-bound row kernels of the rollout-reward path (gfx950): embedding + timing signal,
// residual-add + LayerNorm, rollout token bookkeeping, discriminator gather / highway gate / head,
// per-prefix reward averaging.  One wavefront per row where a row reduction is needed; every
// global access is a contiguous 4-16 B per lane.
#include "common.h"
#include "kernels.h"

namespace nmtgan {

// ------------------------------------------------------------------ zero fill
// Every buffer the library clears goes through this kernel, never through hipMemsetAsync: a caller may capture the
// calls into a HIP graph, and memset nodes have replayed wrongly on this platform.
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4* p16, int64_t n16, unsigned char* tail, int ntail) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) p16[i] = make_uint4(0, 0, 0, 0);
  if (i < ntail) tail[i] = 0;
}
int launch_zero(void* p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return 0;
  NMTGAN_REQUIRE(p != nullptr && ((uintptr_t)p % 16) == 0, "zero fill: null or misaligned buffer");
  const int64_t n16 = (int64_t)(bytes / 16);
  const int ntail = (int)(bytes % 16);
  const int64_t n = n16 > ntail ? n16 : ntail;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (uint4*)p, n16,
                     (unsigned char*)p + n16 * 16, ntail);
  NMTGAN_CHECK_LAUNCH("zero_fill");
  return 0;
}

// ------------------------------------------------------------------ embedding (model.py:772-802, 506, 565)
template <typename OpT>
__global__ __launch_bounds__(256) void embed_kernel(nmtgan_embed_args a) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.M) return;
  int tok = a.tok[(int64_t)r * a.tok_ld + (a.tok_col ? a.tok_col[r] : 0) + a.tok_col_add];
  tok = min(max(tok, 0), a.vocab - 1);
  int pos = (a.pos ? a.pos[r] : (a.pos_mod > 0 ? r % a.pos_mod : 0)) + a.pos_add;
  pos = min(max(pos, 0), a.max_pos - 1);
  const float* e = a.table + (int64_t)tok * a.d;
  const float* s = a.timing + (int64_t)pos * a.d;
  float* of = a.out_f32 + (int64_t)r * a.d;
  OpT* oo = a.out_op && a.out_op != (void*)a.out_f32 ? reinterpret_cast<OpT*>(a.out_op) + (int64_t)r * a.d : nullptr;
  f16_t* os = a.out_split ? reinterpret_cast<f16_t*>(a.out_split) + (int64_t)r * 3 * a.d : nullptr;
  for (int i = lane; i < a.d; i += 64) {
    const float v = e[i] * a.mult + s[i];
    of[i] = v;
    if (oo) OpTraits<OpT>::store(oo + i, v);
    if (os) {
      const float one[1] = {v};
      store_split3<1>(os + i, a.d, one);
    }
  }
}

int launch_embed(const nmtgan_embed_args* a, hipStream_t st) {
  prof::Scope prof_scope(prof::EMBED, st);
  NMTGAN_REQUIRE(a != nullptr, "embed: null args");
  if (a->M == 0) return 0;
  NMTGAN_REQUIRE(a->M > 0 && a->d > 0 && a->vocab > 0 && a->max_pos > 0, "embed: bad shape");
  NMTGAN_REQUIRE(a->table && a->timing && a->tok && a->out_f32, "embed: null pointer");
  NMTGAN_REQUIRE((uintptr_t)a->out_split % 16 == 0, "embed: out_split must be 16-byte aligned");
  dim3 grid((a->M + 3) / 4), block(256);
  NMTGAN_REQUIRE(valid_op_dtype(a->op_dtype), "embed: bad op_dtype %d", a->op_dtype);
  NMTGAN_DISPATCH_OP(a->op_dtype, OpT, hipLaunchKernelGGL(embed_kernel<OpT>, grid, block, 0, st, *a));
  NMTGAN_CHECK_LAUNCH("embed");
  return 0;
}

// ------------------------------------------------------------------ residual + LayerNorm
// (model.py:725-739; common_layers.py:336-341: mean, biased variance of (x-mean), rsqrt(var+1e-6))
template <typename OpT, int EPL>
__global__ __launch_bounds__(256) void addln_kernel(nmtgan_addln_args a) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.M) return;
  const int64_t base = (int64_t)r * a.d + lane * EPL;
  float v[EPL];
  VecIO<float, EPL>::load(a.x + base, v);
  if (a.y) {
    float y[EPL];
    VecIO<float, EPL>::load(a.y + base, y);
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] += y[e];
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) s += v[e];
  const float mean = wave_sum(s) / (float)a.d;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) { v[e] -= mean; q += v[e] * v[e]; }
  const float var = wave_sum(q) / (float)a.d;
  const float rs = rsqrtf(var + 1e-6f);
  float g[EPL], b[EPL];
  VecIO<float, EPL>::load(a.gamma + lane * EPL, g);
  VecIO<float, EPL>::load(a.beta + lane * EPL, b);
  const bool zero = a.zero_row && a.zero_row[r];
#pragma unroll
  for (int e = 0; e < EPL; ++e) v[e] = zero ? 0.f : (v[e] * rs) * g[e] + b[e];
  if (a.out_f32) VecIO<float, EPL>::store(a.out_f32 + base, v);
  if (a.out_op && a.out_op != (void*)a.out_f32) VecIO<OpT, EPL>::store(reinterpret_cast<OpT*>(a.out_op) + base, v);
  if (a.out_split) store_split3<EPL>(reinterpret_cast<f16_t*>(a.out_split) + (int64_t)r * 3 * a.d + lane * EPL, a.d, v);
}

template <typename OpT>
static int launch_addln_t(const nmtgan_addln_args& a, hipStream_t st) {
  dim3 grid((a.M + 3) / 4), block(256);
  switch (a.d / 64) {
    case 1: hipLaunchKernelGGL((addln_kernel<OpT, 1>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((addln_kernel<OpT, 2>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((addln_kernel<OpT, 4>), grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL((addln_kernel<OpT, 8>), grid, block, 0, st, a); break;
    case 16: hipLaunchKernelGGL((addln_kernel<OpT, 16>), grid, block, 0, st, a); break;
    default: NMTGAN_FAIL(-2, "add_layernorm: d/64 must be 1,2,4,8 or 16 (d=%d)", a.d);
  }
  NMTGAN_CHECK_LAUNCH("add_layernorm");
  return 0;
}

int launch_add_layernorm(const nmtgan_addln_args* a, hipStream_t st) {
  prof::Scope prof_scope(prof::ADDLN, st);
  NMTGAN_REQUIRE(a != nullptr, "add_layernorm: null args");
  if (a->M == 0) return 0;
  NMTGAN_REQUIRE(a->M > 0 && a->d > 0 && a->d % 64 == 0, "add_layernorm: bad M=%d d=%d", a->M, a->d);
  NMTGAN_REQUIRE(a->x && a->gamma && a->beta && (a->out_f32 || a->out_op), "add_layernorm: null pointer");
  NMTGAN_REQUIRE(valid_op_dtype(a->op_dtype), "add_layernorm: bad op_dtype %d", a->op_dtype);
  NMTGAN_REQUIRE((uintptr_t)a->out_split % 16 == 0, "add_layernorm: out_split must be 16-byte aligned");
  NMTGAN_DISPATCH_OP(a->op_dtype, OpT, return launch_addln_t<OpT>(*a, st));
  return 0;
}

// ------------------------------------------------------------------ split-precision GEMM operand
__global__ __launch_bounds__(256) void split3_kernel(const float* A, int64_t lda, int M, int K, f16_t* out) {
  const int kc = K / 8;                            // 8-element chunks per row
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * kc) return;
  const int64_t m = i / kc;
  const int c = (int)(i % kc);
  float v[8];
  VecIO<float, 8>::load(A + m * lda + 8 * c, v);
  unsigned hi[4], lo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f16_t h0 = (f16_t)fminf(fmaxf(v[2 * e], -65504.f), 65504.f);       // clamped: see store_split3
    const f16_t h1 = (f16_t)fminf(fmaxf(v[2 * e + 1], -65504.f), 65504.f);
    const f16x2_t h = {h0, h1};
    const f16x2_t l = {(f16_t)(v[2 * e] - (float)h0), (f16_t)(v[2 * e + 1] - (float)h1)};
    hi[e] = __builtin_bit_cast(unsigned, h);
    lo[e] = __builtin_bit_cast(unsigned, l);
  }
  f16_t* o = out + m * 3 * K + 8 * c;
  const uint4 h4 = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  *reinterpret_cast<uint4*>(o) = h4;
  *reinterpret_cast<uint4*>(o + K) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  *reinterpret_cast<uint4*>(o + 2 * K) = h4;
}

int launch_split3(const float* A, int64_t lda, int M, int K, void* out, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  if (M == 0) return 0;
  NMTGAN_REQUIRE(A && out && M > 0 && K > 0 && K % 8 == 0 && lda % 4 == 0, "split3: bad arguments (K %% 8, lda %% 4)");
  NMTGAN_REQUIRE((uintptr_t)A % 16 == 0 && (uintptr_t)out % 16 == 0, "split3: operands must be 16-byte aligned");
  const int64_t n = (int64_t)M * (K / 8);
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A, lda, M, K, (f16_t*)out);
  NMTGAN_CHECK_LAUNCH("split3");
  return 0;
}

// x <- float(fp16(x)): what a value keeps when it is stored in the fp16 tiers' caches (error-budget runs of the
// split-precision tier, NMTGAN_SPLIT_ABLATE)
__global__ void round_f16_kernel(float* x, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = (float)(f16_t)x[i];
}
int launch_round_f16(float* x, int64_t n, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(round_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n);
  NMTGAN_CHECK_LAUNCH("round_f16");
  return 0;
}

// ------------------------------------------------------------------ rollout tokens
// Y[row][t] = y[b][t] for t < g else 0 (model.py:250-252: given_y = Y[:, :given_num], zero padded)
__global__ void init_rollouts_kernel(const int32_t* y, const int32_t* row_b, const int32_t* row_g, int n_rows, int T,
                                     int32_t* Y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_rows * T) return;
  const int row = (int)(i / T), t = (int)(i % T);
  Y[i] = t < row_g[row] ? y[(int64_t)row_b[row] * T + t] : 0;
}
int launch_init_rollouts(const int32_t* y, const int32_t* row_b, const int32_t* row_g, int n_rows, int T,
                         int32_t* Y, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  if (n_rows == 0) return 0;
  const int64_t n = (int64_t)n_rows * T;
  hipLaunchKernelGGL(init_rollouts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, row_b, row_g,
                     n_rows, T, Y);
  NMTGAN_CHECK_LAUNCH("init_rollouts");
  return 0;
}
// Y[row][row_g[row] + add] = tok[row]  (model.py:242: append the sampled token)
__global__ void scatter_tokens_kernel(const int32_t* tok, const int32_t* row_g, int add, int n, int T, int32_t* Y) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int t = row_g[r] + add;
  if (t >= 0 && t < T) Y[(int64_t)r * T + t] = tok[r];
}
int launch_scatter_tokens(const int32_t* tok, const int32_t* row_g, int add, int n, int T, int32_t* Y,
                          hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  if (n == 0) return 0;
  hipLaunchKernelGGL(scatter_tokens_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tok, row_g, add, n, T, Y);
  NMTGAN_CHECK_LAUNCH("scatter_tokens");
  return 0;
}

// ------------------------------------------------------------------ discriminator gather
// tf.nn.embedding_lookup (cnn_discriminator.py:454, 485): out[i] = table[ids[i]]; 16 B per lane.
__global__ __launch_bounds__(256) void disc_embed_kernel(const char* table, int vocab, const int32_t* ids,
                                                         int64_t n_tok, int row_bytes, char* out) {
  const int cpr = row_bytes >> 4;                       // 16-byte chunks per row
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tok * cpr) return;
  const int64_t t = i / cpr;
  const int c = (int)(i % cpr);
  const int id = min(max(ids[t], 0), vocab - 1);
  reinterpret_cast<uint4*>(out + t * row_bytes)[c] = reinterpret_cast<const uint4*>(table + (int64_t)id * row_bytes)[c];
}
int launch_disc_embed(const void* table, int op_dtype, int vocab, const int32_t* ids, int64_t n_tok, int E,
                      void* out, hipStream_t st) {
  prof::Scope prof_scope(prof::DISC_SMALL, st);
  if (n_tok == 0) return 0;
  const int row_bytes = E * dtype_size(op_dtype);
  NMTGAN_REQUIRE(row_bytes % 16 == 0, "disc_embed: dim_word*esize must be a multiple of 16 bytes");
  const int64_t n = n_tok * (row_bytes >> 4);
  hipLaunchKernelGGL(disc_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const char*)table,
                     vocab, ids, n_tok, row_bytes, (char*)out);
  NMTGAN_CHECK_LAUNCH("disc_embed");
  return 0;
}

template <typename OpT>
__global__ void cast_kernel(const float* in, OpT* out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  float v[4];
  VecIO<float, 4>::load(in + i, v);
  VecIO<OpT, 4>::store(out + i, v);
}
int launch_cast_rows(const float* in, void* out, int op_dtype, int64_t n, hipStream_t st) {
  prof::Scope prof_scope(prof::DISC_SMALL, st);
  if (n == 0) return 0;
  NMTGAN_REQUIRE(n % 4 == 0, "cast: n must be a multiple of 4");
  dim3 grid((unsigned)((n / 4 + 255) / 256)), block(256);
  NMTGAN_DISPATCH_OP(op_dtype, OpT, hipLaunchKernelGGL(cast_kernel<OpT>, grid, block, 0, st, in, (OpT*)out, n));
  NMTGAN_CHECK_LAUNCH("cast");
  return 0;
}

// Prefix-window reuse in the discriminator (the rollouts of (b, g) share tokens 0..g-1 with y_sample[b], so
// every conv window that ends inside the prefix has the same value as in y_sample):
// C[b*T + p][col] holds relu(bn(conv)) of window p of y_sample[b]; pm[b*T + g][col] = max over p <= g - w(col).
struct WidthTable { int n; int col_end[NMTGAN_MAX_WIDTHS]; int width[NMTGAN_MAX_WIDTHS]; };
__global__ void prefix_max_kernel(const float* C, float* pm, int B, int T, int Fp, WidthTable wt) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (col >= Fp) return;
  int w = T + 1;                                   // padding columns: never any window
  for (int i = 0; i < wt.n; ++i)
    if (col < wt.col_end[i]) { w = wt.width[i]; break; }
  float run = 0.f;                                 // ReLU floor: the neutral element of the max-pool here
  for (int g = 0; g < T; ++g) {
    if (g - w >= 0) run = fmaxf(run, C[((int64_t)b * T + g - w) * Fp + col]);
    pm[((int64_t)b * T + g) * Fp + col] = run;
  }
}
int launch_prefix_max(const float* C, float* pm, int B, int T, int Fp, int n_widths, const int* widths,
                      const int* n_filters, hipStream_t st) {
  prof::Scope prof_scope(prof::DISC_SMALL, st);
  NMTGAN_REQUIRE(C && pm && B > 0 && T > 0 && Fp > 0 && n_widths > 0 && n_widths <= NMTGAN_MAX_WIDTHS,
                 "prefix_max: bad arguments");
  WidthTable wt{};
  wt.n = n_widths;
  int col = 0;
  for (int i = 0; i < n_widths; ++i) { col += n_filters[i]; wt.col_end[i] = col; wt.width[i] = widths[i]; }
  hipLaunchKernelGGL(prefix_max_kernel, dim3((Fp + 255) / 256, B), dim3(256), 0, st, C, pm, B, T, Fp, wt);
  NMTGAN_CHECK_LAUNCH("prefix_max");
  return 0;
}

// pooled[i][:] = pm[row_b[i]*T + row_g[i]][:]
__global__ void pool_init_kernel(const float* pm, const int32_t* row_b, const int32_t* row_g, int T, int Fp4,
                                 int64_t n4, float4* pooled) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int r = (int)(i / Fp4), c = (int)(i % Fp4);
  pooled[i] = reinterpret_cast<const float4*>(pm)[((int64_t)row_b[r] * T + row_g[r]) * Fp4 + c];
}
// partial-window reuse of the discriminator conv (pipeline.hip): S += prev (prefix sums over tap blocks), and the
// finish of one bucket of windows: pooled[m][f] = max(pooled[m][f], max_q relu((raw[m*nq+q][f] + S[(b_m*P + pa+q)][f])
// * scale[f] + shift[f])) - the conv bias + BN + ReLU + max-over-time of cnn_discriminator.py:94-105, 469 applied to
// (taps computed from the rollout) + (taps taken from y_sample)
__global__ void add_inplace_kernel(float* dst, const float* src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
int launch_add_inplace(float* dst, const float* src, int64_t n, hipStream_t st) {
  prof::Scope prof_scope(prof::DISC_SMALL, st);
  if (n == 0) return 0;
  NMTGAN_REQUIRE(dst && src && n > 0, "add_inplace: bad arguments");
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, src, n);
  NMTGAN_CHECK_LAUNCH("add_inplace");
  return 0;
}
__global__ void bucket_finish_kernel(const float* raw, const float* S, const int32_t* row_b, int rows, int nq, int N,
                                     int P, int pa, const float* scale, const float* shift, float* pooled,
                                     int64_t ld_pool) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = blockIdx.y;
  if (f >= N || m >= rows) return;
  const float sc = scale[f], sh = shift[f];
  const float* r = raw + (int64_t)m * nq * N + f;
  const float* s = S + ((int64_t)row_b[m] * P + pa) * N + f;
  float mx = pooled[(int64_t)m * ld_pool + f];
  for (int q = 0; q < nq; ++q) mx = fmaxf(mx, (r[(int64_t)q * N] + s[(int64_t)q * N]) * sc + sh);   // relu: pooled >= 0
  pooled[(int64_t)m * ld_pool + f] = mx;
}
int launch_bucket_finish(const float* raw, const float* S, const int32_t* row_b, int rows, int nq, int N, int P, int pa,
                         const float* scale, const float* shift, float* pooled, int64_t ld_pool, hipStream_t st) {
  prof::Scope prof_scope(prof::DISC_SMALL, st);
  if (rows == 0) return 0;
  NMTGAN_REQUIRE(raw && S && row_b && scale && shift && pooled && nq > 0 && N > 0 && pa >= 0 && pa + nq <= P,
                 "bucket_finish: bad arguments");
  hipLaunchKernelGGL(bucket_finish_kernel, dim3((N + 127) / 128, rows), dim3(128), 0, st, raw, S, row_b, rows, nq, N, P,
                     pa, scale, shift, pooled, ld_pool);
  NMTGAN_CHECK_LAUNCH("bucket_finish");
  return 0;
}

int launch_pool_init(const float* pm, const int32_t* row_b, const int32_t* row_g, int rows, int T, int Fp,
                     float* pooled, hipStream_t st) {
  prof::Scope prof_scope(prof::DISC_SMALL, st);
  if (rows == 0) return 0;
  NMTGAN_REQUIRE(pm && row_b && row_g && pooled && Fp % 4 == 0, "pool_init: bad arguments");
  const int64_t n4 = (int64_t)rows * (Fp / 4);
  hipLaunchKernelGGL(pool_init_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, pm, row_b, row_g, T,
                     Fp / 4, n4, (float4*)pooled);
  NMTGAN_CHECK_LAUNCH("pool_init");
  return 0;
}

// highway (cnn_discriminator.py:50-59, layer_size 1, bias 0): out = sig(t)*relu(o) + (1-sig(t))*H
__global__ void highway_gate_kernel(const float* H, const float* o, const float* t, float* out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  float h[4], ov[4], tv[4], r[4];
  VecIO<float, 4>::load(H + i, h);
  VecIO<float, 4>::load(o + i, ov);
  VecIO<float, 4>::load(t + i, tv);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gate = 1.f / (1.f + expf(-tv[e]));
    r[e] = gate * fmaxf(ov[e], 0.f) + (1.f - gate) * h[e];
  }
  VecIO<float, 4>::store(out + i, r);
}
int launch_highway_gate(const float* H, const float* o, const float* t, float* out, int64_t n, hipStream_t st) {
  prof::Scope prof_scope(prof::DISC_SMALL, st);
  if (n == 0) return 0;
  NMTGAN_REQUIRE(n % 4 == 0, "highway_gate: n must be a multiple of 4");
  hipLaunchKernelGGL(highway_gate_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, H, o, t, out, n);
  NMTGAN_CHECK_LAUNCH("highway_gate");
  return 0;
}

// scores = [feat_t | feat_s] . W + b; softmax over 2 classes (cnn_discriminator.py:504-508)
__global__ __launch_bounds__(256) void disc_head_kernel(const float* feat_t, const float* feat_s,
                                                        const int32_t* src_row, int M, int Fp, const float* head_w,
                                                        const float* head_b, float* p_real, float* probs2) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const float* ft = feat_t + (int64_t)r * Fp;
  const float* fs = feat_s + (int64_t)(src_row ? src_row[r] : r) * Fp;
  float s0 = 0.f, s1 = 0.f;
  for (int i = lane; i < Fp; i += 64) {
    const float a = ft[i], b = fs[i];
    s0 = fmaf(a, head_w[i], s0);
    s0 = fmaf(b, head_w[Fp + i], s0);
    s1 = fmaf(a, head_w[2 * Fp + i], s1);
    s1 = fmaf(b, head_w[3 * Fp + i], s1);
  }
  s0 = wave_sum(s0) + head_b[0];
  s1 = wave_sum(s1) + head_b[1];
  if (lane == 0) {
    const float mx = fmaxf(s0, s1);
    const float e0 = expf(s0 - mx), e1 = expf(s1 - mx);
    const float inv = 1.f / (e0 + e1);
    if (p_real) p_real[r] = e1 * inv;
    if (probs2) { probs2[2 * r] = e0 * inv; probs2[2 * r + 1] = e1 * inv; }
  }
}
int launch_disc_head(const float* feat_t, const float* feat_s, const int32_t* src_row, int M, int Fp,
                     const float* head_w, const float* head_b, float* p_real, float* probs2, hipStream_t st) {
  prof::Scope prof_scope(prof::DISC_SMALL, st);
  if (M == 0) return 0;
  NMTGAN_REQUIRE(feat_t && feat_s && head_w && head_b && (p_real || probs2), "disc_head: null pointer");
  hipLaunchKernelGGL(disc_head_kernel, dim3((M + 3) / 4), dim3(256), 0, st, feat_t, feat_s, src_row, M, Fp, head_w,
                     head_b, p_real, probs2);
  NMTGAN_CHECK_LAUNCH("disc_head");
  return 0;
}

// ------------------------------------------------------------------ reward averaging (model.py:315-345)
__global__ void reward_reduce_kernel(const float* p, const float* p_base, const int32_t* grp_first,
                                     const float* mask, int B, int T, int rollnum, int use_bias, float bias_total,
                                     float* rewards) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  const int b = i / T, t = i % T;
  float q = 0.f;
  const int first = (t < T - 1) ? grp_first[b * (T - 1) + t] : -1;
  if (first >= 0) {
    for (int r = 0; r < rollnum; ++r) q += p[first + r];     // rewards[g-1] += ypred, roll order (model.py:319)
  } else {
    const float pb = p_base[b];
    for (int r = 0; r < rollnum; ++r) q += pb;               // same D input every roll (model.py:321-332; B.6)
  }
  if (use_bias) q -= bias_total;                             // model.py:340-342 (fp32 bias array)
  rewards[i] = (q * mask[i]) / (float)rollnum;               // model.py:344-345
}
int launch_reward_reduce(const float* p, const float* p_base, const int32_t* grp_first, const float* mask, int B,
                         int T, int rollnum, int use_bias, double bias_num, float* rewards, hipStream_t st) {
  prof::Scope prof_scope(prof::DISC_SMALL, st);
  NMTGAN_REQUIRE(B > 0 && T > 1 && rollnum > 0, "reward_reduce: bad shape");
  NMTGAN_REQUIRE(p_base && grp_first && mask && rewards, "reward_reduce: null pointer");
  hipLaunchKernelGGL(reward_reduce_kernel, dim3((B * T + 255) / 256), dim3(256), 0, st, p, p_base, grp_first, mask,
                     B, T, rollnum, use_bias, (float)(bias_num * (double)rollnum), rewards);
  NMTGAN_CHECK_LAUNCH("reward_reduce");
  return 0;
}

}  // namespace nmtgan
