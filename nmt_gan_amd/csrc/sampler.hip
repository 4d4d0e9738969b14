// Categorical sampler over the vocabulary with injected uniforms (gfx950).
//
// One 256-thread workgroup per output row, ONE pass over the logits row from HBM:
//   pass 1 (all 4 waves): coalesced float4 loads; every thread keeps an online running max and
//           NG fp64 partial sums of exp(l - max), one per contiguous 1/NG slice of the row
//           ("groups"); rescales happen only when a thread's max grows (fp64 factor).
//           Block reduction gives the row max, the NG group sums (fixed tree -> deterministic)
//           and the total.
//   pass 2 (wave 0): target = u * total; pick the group whose running prefix crosses the target,
//           then re-read only that group (1/NG of the row, L2-resident) in 256-element blocks:
//           fp64 inclusive wave scan, ballot for the first element whose CDF exceeds the target
//           (std::upper_bound).  A miss caused by the ~1e-7 inconsistency between the two passes
//           simply continues into the next elements; running off the end clamps to V-1 (quirk B.10).
// Algorithmic HBM bytes per row: V*4 + 8.
#include "common.h"
#include "kernels.h"

namespace nmtgan {

constexpr int NG = 8;

// exp(x) for x <= 0 as one v_exp_f32 (2^(x*log2 e)); relative error ~1e-6 for |x| < 20, far below the
// bf16/fp32 logit noise; the accurate expf costs ~15 VALU instructions per logit.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

struct SampleParams {
  const float* logits; int64_t ld; int V; int M;
  const int32_t* row_map; const float* u; const int64_t* u_off; int64_t u_add;
  int32_t* out;
};

__device__ __forceinline__ double scan_incl_d(double v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

__global__ __launch_bounds__(256) void sample_kernel(SampleParams p) {
  __shared__ double s_g[4][NG];
  __shared__ float s_m[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = blockIdx.x;
  const int64_t src = p.row_map ? p.row_map[r] : r;
  const float4* L4 = reinterpret_cast<const float4*>(p.logits + src * p.ld);
  const int V4 = p.V >> 2;
  const int n_iter = (V4 + 255) >> 8;
  const int ipg = (n_iter + NG - 1) / NG;

  float m = -INFINITY;
  double acc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) acc[g] = 0.0;

#pragma unroll
  for (int g = 0; g < NG; ++g) {
    for (int j0 = 0; j0 < ipg; j0 += 4) {
      float4 v[4];
      bool ok[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int idx = ((g * ipg + j0 + jj) << 8) + tid;
        ok[jj] = (j0 + jj < ipg) && idx < V4;
        if (ok[jj]) v[jj] = L4[idx];
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        if (!ok[jj]) continue;
        const float mx = fmaxf(fmaxf(v[jj].x, v[jj].y), fmaxf(v[jj].z, v[jj].w));
        if (mx == -INFINITY) continue;                  // all four are -inf: contribute exactly 0
        if (mx > m) {
          const double sc = (m == -INFINITY) ? 0.0 : exp((double)m - (double)mx);
#pragma unroll
          for (int k = 0; k < NG; ++k) acc[k] *= sc;
          m = mx;
        }
        // four terms summed in fp32 (<= 4 ulp on a partial), the running slice sum in fp64
        const float e = (fast_exp(v[jj].x - m) + fast_exp(v[jj].y - m)) + (fast_exp(v[jj].z - m) + fast_exp(v[jj].w - m));
        acc[g] += (double)e;
      }
    }
  }
  // ---- block max, rescale to it, group sums
  const float wm = wave_max(m);
  if (lane == 0) s_m[wave] = wm;
  __syncthreads();
  const float Mx = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
  const double f = (m == -INFINITY) ? 0.0 : exp((double)m - (double)Mx);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const double t = wave_sum_d(acc[g] * f);
    if (lane == 0) s_g[wave][g] = t;
  }
  __syncthreads();
  if (wave != 0) return;

  double G[NG];
  double total = 0.0;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    G[g] = ((s_g[0][g] + s_g[1][g]) + s_g[2][g]) + s_g[3][g];
    total += G[g];
  }
  const float u = p.u[(p.u_off ? p.u_off[r] : (int64_t)r) + p.u_add];
  const double target = (double)u * total;
  int gsel = NG - 1;
  double pre = 0.0;
  {
    bool found = false;
    double run = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (!found) {
        if (run + G[g] > target) { found = true; gsel = g; pre = run; }
        else run += G[g];
      }
    }
    if (!found) { gsel = NG; pre = run; }
  }
  int result = p.V - 1;
  for (int i4 = gsel * ipg * 256; i4 < V4; i4 += 64) {       // wave-uniform loop
    const int idx = i4 + lane;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (idx < V4) {
      const float4 v = L4[idx];
      c0 = (double)fast_exp(v.x - Mx);
      c1 = c0 + (double)fast_exp(v.y - Mx);
      c2 = c1 + (double)fast_exp(v.z - Mx);
      c3 = c2 + (double)fast_exp(v.w - Mx);
    }
    const double incl = scan_incl_d(c3, lane);
    const double base = pre + (incl - c3);
    int hit = -1;
    if (idx < V4) {
      if (base + c0 > target) hit = 0;
      else if (base + c1 > target) hit = 1;
      else if (base + c2 > target) hit = 2;
      else if (base + c3 > target) hit = 3;
    }
    const unsigned long long bal = __ballot(hit >= 0);
    if (bal) {
      const int first = __ffsll((long long)bal) - 1;
      const int h = __shfl(hit, first, 64);
      result = (i4 + first) * 4 + h;
      break;
    }
    pre += __shfl(incl, 63, 64);
  }
  if (lane == 0) p.out[r] = result;
}

// first argmax (np.argmax / tf.argmax tie rule) for greedy decoding (model.py:153-154)
__global__ __launch_bounds__(256) void argmax_kernel(SampleParams p) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = blockIdx.x;
  const int64_t src = p.row_map ? p.row_map[r] : r;
  const float* L = p.logits + src * p.ld;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = tid; i < p.V; i += 256) {
    const float v = L[i];
    if (v > bv) { bv = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) { bv = s_v[w]; bi = s_i[w]; }
    p.out[r] = bi == 0x7fffffff ? 0 : bi;
  }
}

int launch_sample_rows(const nmtgan_sample_args* a, hipStream_t st) {
  prof::Scope prof_scope(prof::SAMPLER_ROWS, st);
  NMTGAN_REQUIRE(a != nullptr, "sample: null args");
  if (a->M == 0) return 0;
  NMTGAN_REQUIRE(a->M > 0 && a->V > 0, "sample: bad M=%d V=%d", a->M, a->V);
  NMTGAN_REQUIRE(a->logits && a->out, "sample: null logits/out");
  SampleParams p{a->logits, a->ld, a->V, a->M, a->row_map, a->u, a->u_off, a->u_add, a->out};
  if (a->greedy) {
    hipLaunchKernelGGL(argmax_kernel, dim3(a->M), dim3(256), 0, st, p);
  } else {
    NMTGAN_REQUIRE(a->u != nullptr, "sample: null uniforms");
    NMTGAN_REQUIRE(a->V % 4 == 0 && a->ld % 4 == 0 && ((uintptr_t)a->logits % 16) == 0,
                   "sample: V and ld must be multiples of 4 and logits 16-byte aligned (V=%d)", a->V);
    hipLaunchKernelGGL(sample_kernel, dim3(a->M), dim3(256), 0, st, p);
  }
  NMTGAN_CHECK_LAUNCH("sample_rows");
  return 0;
}

}  // namespace nmtgan
