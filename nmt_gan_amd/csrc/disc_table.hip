// Table form of the discriminator's conv + BatchNorm + ReLU + max-over-time (cnn_discriminator.py:88-109, 461-477)
// for the rollouts of one get_reward call (model.py:303-313).
//
// A filter of width w applied at window p of a sentence is sum_{j < w} E[tok[p+j]] . W_w[j] - a sum of w terms that
// each depend on ONE token and ONE tap.  One call scores 50,000 sentences over a vocabulary of 32,000, so every
// (token, tap) product is needed ~80 times: P_w[v][j][f] = E[v] . W_w[j][:, f] is computed once per call for the whole
// vocabulary (one pass of GEMMs over the embedding table, a few percent of the conv's flops) and a window becomes w
// gathered rows of P added in tap order - HBM-bound reads of contiguous n_f-wide rows instead of 2*w*E*n_f flops.
// A rollout (b, g) shares tokens 0..g-1 with y_sample[b]: windows that end inside the prefix come from the prefix-max
// table as before, and a window that straddles the give point starts from S_w[b][p][g-p] = the sum of its first g-p
// taps over y_sample (cumulative over taps, written once per call by table_prefix_kernel).  Which terms of a window
// come from S and which are gathered follows from the row's own give point alone, and the taps are added in the order
// 0, 1, .., w-1, so a row's features do not depend on the chunking, the schedule or the sharding.
#include "kernels.h"

namespace nmtgan {

struct TableParams {
  const void* tab[NMTGAN_MAX_WIDTHS];
  float* S[NMTGAN_MAX_WIDTHS];
  const float* scale[NMTGAN_MAX_WIDTHS];
  const float* shift[NMTGAN_MAX_WIDTHS];
  int w[NMTGAN_MAX_WIDTHS], nf[NMTGAN_MAX_WIDTHS], nfp[NMTGAN_MAX_WIDTHS], col[NMTGAN_MAX_WIDTHS];
  const int32_t* ids;           // [rows][T]
  const int32_t* row_b;         // conv: sentence of the row
  const int32_t* row_g;         // conv: give_num of the row
  float* out;                   // conv: pooled [rows][Fp]; prefix: window values [rows*T][Fp]
  int T, Fp, vocab;
};
static_assert(sizeof(TableParams) <= 4096, "the width table travels as kernel arguments");

// Taps of an fp32 table are added in fp64 (the kernels wait on memory, the adds are free): what is left of the
// reference's fp32 conv sum is then the rounding of the table entries themselves, not the order of 2..49 adds at
// |sum| ~ 100.  16-bit tables add in fp32, far below their entries' own rounding.
template <typename TT> struct Row { static constexpr int VN = 16 / sizeof(TT); using Acc = float; };
template <> struct Row<float> { static constexpr int VN = 4; using Acc = double; };

// a[0..VN) += the VN table entries at p
template <typename TT, int VN, typename Acc>
__device__ __forceinline__ void add_row(Acc (&a)[VN], const TT* p) {
  float v[VN];
  VecIO<TT, VN>::load(p, v);
#pragma unroll
  for (int k = 0; k < VN; ++k) a[k] += (Acc)v[k];
}

// y_sample: block (p, b, i) adds the taps of window p of sentence b for width i in order, writes the partial sums
// S[b][p][j-1] (first j taps, j = 1..w-1) and the window's value relu(bn(sum of all w taps))
template <typename TT>
__global__ __launch_bounds__(128) void table_prefix_kernel(TableParams p) {
  constexpr int VN = Row<TT>::VN;
  const int i = blockIdx.z, b = blockIdx.y, q = blockIdx.x;
  const int w = p.w[i], nfp = p.nfp[i], T = p.T, Pw = T - w + 1;
  if (q >= Pw) return;
  __shared__ int32_t tok[TABLE_MAX_LEN];
  for (int t = threadIdx.x; t < w; t += blockDim.x) tok[t] = min(max(p.ids[(int64_t)b * T + q + t], 0), p.vocab - 1);
  __syncthreads();
  const TT* tab = reinterpret_cast<const TT*>(p.tab[i]);
  float* S = p.S[i] + ((int64_t)b * Pw + q) * (w - 1) * nfp;
  using Acc = typename Row<TT>::Acc;
  for (int c = threadIdx.x * VN; c < nfp; c += blockDim.x * VN) {
    Acc a[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) a[k] = 0;
    for (int j = 0; j < w; ++j) {
      add_row<TT, VN, Acc>(a, tab + ((int64_t)tok[j] * w + j) * nfp + c);
      if (j + 1 < w) {
        float r[VN];
#pragma unroll
        for (int k = 0; k < VN; ++k) r[k] = (float)a[k];
        VecIO<float, VN>::store(S + (int64_t)j * nfp + c, r);
      }
    }
#pragma unroll
    for (int k = 0; k < VN; ++k)
      if (c + k < p.nf[i])
        p.out[((int64_t)b * T + q) * p.Fp + p.col[i] + c + k] = fmaxf((float)a[k] * p.scale[i][c + k] + p.shift[i][c + k], 0.f);
  }
}

// rollouts: block (row s, width i).  Windows q >= g - w + 1 hold a sampled token; the first g - q taps of window q
// (if any) come from S, the rest are gathered with the row's own tokens.  Threads = vector columns x window slots.
template <typename TT>
__global__ __launch_bounds__(256) void table_conv_kernel(TableParams p) {
  constexpr int VN = Row<TT>::VN;
  using Acc = typename Row<TT>::Acc;
  __shared__ int32_t tok[TABLE_MAX_LEN];
  __shared__ float red[256 * VN];
  const int i = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
  const int w = p.w[i], nfp = p.nfp[i], T = p.T, Pw = T - w + 1;
  const int g = p.row_g[s], b = p.row_b[s];
  for (int t = tid; t < T; t += 256) tok[t] = min(max(p.ids[(int64_t)s * T + t], 0), p.vocab - 1);
  __syncthreads();
  const int CV = nfp / VN;                       // vector columns of this width
  const int CP = CV < 256 ? CV : 256;            // ... handled per pass
  const int G = 256 / CP;                        // window slots
  const int cidx = tid % CP, slot = tid / CP;
  const TT* tab = reinterpret_cast<const TT*>(p.tab[i]);
  const float* S = p.S[i] + (int64_t)b * Pw * (w - 1) * nfp;
  const int q_lo = g - w + 1 > 0 ? g - w + 1 : 0;
  for (int cv = cidx; cv < CV; cv += CP) {
    const int c = cv * VN;
    float m[VN], sc[VN], sh[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) {
      m[k] = 0.f;                                // ReLU floor
      const int f = c + k < p.nf[i] ? c + k : p.nf[i] - 1;
      sc[k] = p.scale[i][f]; sh[k] = p.shift[i][f];
    }
    if (slot < G) {
      for (int q = q_lo + slot; q < Pw; q += G) {
        Acc a[VN];
        int j0 = 0;
        if (g > q) {
          j0 = g - q;                            // 1 .. w-1 taps of the prefix
          float r[VN];
          VecIO<float, VN>::load(S + ((int64_t)q * (w - 1) + (j0 - 1)) * nfp + c, r);
#pragma unroll
          for (int k = 0; k < VN; ++k) a[k] = (Acc)r[k];
        } else {
#pragma unroll
          for (int k = 0; k < VN; ++k) a[k] = 0;
        }
#pragma unroll 8
        for (int j = j0; j < w; ++j) add_row<TT, VN, Acc>(a, tab + ((int64_t)tok[q + j] * w + j) * nfp + c);
#pragma unroll
        for (int k = 0; k < VN; ++k) m[k] = fmaxf(m[k], (float)a[k] * sc[k] + sh[k]);
      }
    }
    if (G > 1) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < VN; ++k) red[tid * VN + k] = m[k];
      __syncthreads();
      if (slot == 0)
        for (int o = 1; o < G; ++o)
#pragma unroll
          for (int k = 0; k < VN; ++k) m[k] = fmaxf(m[k], red[(o * CP + cidx) * VN + k]);
    }
    if (slot == 0) {
      float* o = p.out + (int64_t)s * p.Fp + p.col[i] + c;
#pragma unroll
      for (int k = 0; k < VN; ++k)
        if (c + k < p.nf[i]) o[k] = fmaxf(o[k], m[k]);
    }
  }
}

// filter bank Wt[nf][w][row_bytes] (one operand row per tap) -> tap-major rows out[j * nfp + f][row_bytes], rows
// f >= nf zero: the W operand of ONE GEMM per width whose output row is a token's whole table entry [w][nfp]
__global__ __launch_bounds__(256) void table_weights_kernel(const char* wt, int w, int nf, int nfp, int row_bytes, char* out) {
  const int cpr = row_bytes >> 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)w * nfp * cpr) return;
  const int c = (int)(idx % cpr);
  const int64_t row = idx / cpr;
  const int j = (int)(row / nfp), f = (int)(row % nfp);
  uint4 v = make_uint4(0, 0, 0, 0);
  if (f < nf) v = reinterpret_cast<const uint4*>(wt + ((int64_t)f * w + j) * row_bytes)[c];
  reinterpret_cast<uint4*>(out + row * row_bytes)[c] = v;
}

static int fill_table_params(const ConvTable& t, TableParams& p) {
  NMTGAN_REQUIRE(t.n_widths > 0 && t.n_widths <= NMTGAN_MAX_WIDTHS && t.T > 0 && t.T <= TABLE_MAX_LEN && t.vocab > 0 &&
                     valid_op_dtype(t.dtype), "conv table: bad geometry");
  const int vn = 16 / dtype_size(t.dtype);
  int col = 0;
  for (int i = 0; i < t.n_widths; ++i) {
    NMTGAN_REQUIRE(t.tab[i] && t.S[i] && t.scale[i] && t.shift[i], "conv table: null table of width %d", t.w[i]);
    NMTGAN_REQUIRE(t.w[i] >= 1 && t.w[i] <= t.T && t.nf[i] > 0 && t.nfp[i] >= t.nf[i] && t.nfp[i] % vn == 0 &&
                       t.nfp[i] % 4 == 0, "conv table: bad width entry %d", i);
    NMTGAN_REQUIRE((uintptr_t)t.tab[i] % 16 == 0 && (uintptr_t)t.S[i] % 16 == 0, "conv table: tables must be 16-byte aligned");
    p.tab[i] = t.tab[i]; p.S[i] = t.S[i]; p.scale[i] = t.scale[i]; p.shift[i] = t.shift[i];
    p.w[i] = t.w[i]; p.nf[i] = t.nf[i]; p.nfp[i] = t.nfp[i]; p.col[i] = col;
    col += t.nf[i];
  }
  NMTGAN_REQUIRE(col <= t.Fp && t.Fp % 4 == 0, "conv table: feature row shorter than the filters");
  p.T = t.T; p.Fp = t.Fp; p.vocab = t.vocab;
  return 0;
}

int launch_table_weights(const void* wt, int w, int nf, int nfp, int row_bytes, void* out, hipStream_t st) {
  prof::Scope prof_scope(prof::DISC_SMALL, st);
  NMTGAN_REQUIRE(wt && out && w > 0 && nf > 0 && nfp >= nf && row_bytes > 0 && row_bytes % 16 == 0, "table_weights: bad arguments");
  const int64_t n = (int64_t)w * nfp * (row_bytes >> 4);
  hipLaunchKernelGGL(table_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const char*)wt, w, nf, nfp,
                     row_bytes, (char*)out);
  NMTGAN_CHECK_LAUNCH("table_weights");
  return 0;
}

int launch_table_prefix(const ConvTable& t, const int32_t* y, int B, float* win_vals, hipStream_t st) {
  prof::Scope prof_scope(prof::CONV_TABLE, st);
  if (B == 0) return 0;
  TableParams p{};
  NMTGAN_TRY(fill_table_params(t, p));
  NMTGAN_REQUIRE(y && win_vals && B > 0 && B <= 65535, "table_prefix: bad arguments");
  p.ids = y; p.out = win_vals;
  const dim3 grid(t.T, B, t.n_widths);
  NMTGAN_DISPATCH_OP(t.dtype, TT, hipLaunchKernelGGL(table_prefix_kernel<TT>, grid, dim3(128), 0, st, p));
  NMTGAN_CHECK_LAUNCH("table_prefix");
  return 0;
}

int launch_table_conv(const ConvTable& t, const int32_t* ids, const int32_t* row_b, const int32_t* row_g, int rows,
                      float* pooled, hipStream_t st) {
  prof::Scope prof_scope(prof::CONV_TABLE, st);
  if (rows == 0) return 0;
  TableParams p{};
  NMTGAN_TRY(fill_table_params(t, p));
  NMTGAN_REQUIRE(ids && row_b && row_g && pooled && rows > 0, "table_conv: bad arguments");
  p.ids = ids; p.row_b = row_b; p.row_g = row_g; p.out = pooled;
  const dim3 grid(rows, t.n_widths);
  NMTGAN_DISPATCH_OP(t.dtype, TT, hipLaunchKernelGGL(table_conv_kernel<TT>, grid, dim3(256), 0, st, p));
  NMTGAN_CHECK_LAUNCH("table_conv");
  return 0;
}

}  // namespace nmtgan
