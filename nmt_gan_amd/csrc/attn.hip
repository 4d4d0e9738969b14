// Row attention for gfx950: one wavefront per query row, all heads at once.
//
// Lane l owns elements [l*EPL, (l+1)*EPL) of the d-wide row (EPL = d/64), so one wave-wide load
// reads one whole K (or V) vector - d*esize contiguous bytes, 16 B per lane at d=512 bf16 - and a
// head's dot product is a shuffle reduction over its 64/heads adjacent lanes.  Softmax is online
// (running max / sum in fp32).  Keys come from two segments: a block of keys shared by many rows
// (teacher-forced prefix of the sentence, or the encoder memory) and the row's private suffix in
// the step-major rollout cache.  The kernel is HBM/L2-bandwidth bound: per key it moves
// 2*d*esize bytes for ~5*EPL VALU instructions.
#include "common.h"
#include "kernels.h"

namespace nmtgan {

struct AttnParams {
  const float* q; int M, d, heads;
  const void* k1; const void* v1; int len1;
  const int32_t* seg1_blk; const int32_t* seg1_cnt; int rows_per_blk; int causal;
  const uint8_t* mask1;
  const void* k2; const void* v2; const int64_t* step_off; int n_steps2;
  void* out;
  const float* o_init; const float* m_init; const float* l_init;
};

template <typename KT, int EPL, int KB>
__global__ __launch_bounds__(256) void attn_rows_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int lph = 64 / p.heads;                       // lanes per head
  const int d = p.d;
  const int blk = p.seg1_blk ? p.seg1_blk[row] : (p.rows_per_blk > 0 ? row / p.rows_per_blk : 0);
  int cnt1 = 0;
  if (p.k1) cnt1 = p.seg1_cnt ? p.seg1_cnt[row] : (p.causal ? row % p.rows_per_blk + 1 : p.len1);
  const int n_total = cnt1 + p.n_steps2;
  const KT* k1 = reinterpret_cast<const KT*>(p.k1) + ((int64_t)blk * p.len1) * d + lane * EPL;
  const KT* v1 = reinterpret_cast<const KT*>(p.v1) + ((int64_t)blk * p.len1) * d + lane * EPL;
  const KT* k2 = reinterpret_cast<const KT*>(p.k2) + lane * EPL;
  const KT* v2 = reinterpret_cast<const KT*>(p.v2) + lane * EPL;
  const uint8_t* mk = p.mask1 ? p.mask1 + (int64_t)blk * p.len1 : nullptr;

  float q[EPL];
  VecIO<float, EPL>::load(p.q + (int64_t)row * d + lane * EPL, q);

  float m = -INFINITY, l = 0.f;
  float o[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] = 0.f;
  if (p.o_init) {                                      // continue from the group kernel's state
    const int head = (lane * EPL) / (d / p.heads);
    m = p.m_init[(int64_t)row * p.heads + head];
    l = p.l_init[(int64_t)row * p.heads + head];
    VecIO<float, EPL>::load(p.o_init + (int64_t)row * d + lane * EPL, o);
  }

  for (int jb = 0; jb < n_total; jb += KB) {
    float kf[KB][EPL], vf[KB][EPL];
    bool ok[KB];
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int j = jb + u;
      ok[u] = j < n_total;
      if (ok[u]) {
        const KT *kp, *vp;
        if (j < cnt1) {
          if (mk && mk[j]) ok[u] = false;
          kp = k1 + (int64_t)j * d;
          vp = v1 + (int64_t)j * d;
        } else {
          const int64_t off = (p.step_off[j - cnt1] + row) * (int64_t)d;
          kp = k2 + off;
          vp = v2 + off;
        }
        if (ok[u]) {
          VecIO<KT, EPL>::load(kp, kf[u]);
          VecIO<KT, EPL>::load(vp, vf[u]);
        }
      }
    }
    float s[KB];
    float mx = m;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      float part = 0.f;
      if (ok[u]) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) part = fmaf(q[e], kf[u][e], part);
      }
      for (int off = 1; off < lph; off <<= 1) part += __shfl_xor(part, off, 64);
      s[u] = ok[u] ? part : -INFINITY;
      mx = fmaxf(mx, s[u]);
    }
    if (mx == -INFINITY) continue;                    // nothing valid yet
    const float alpha = expf(m - mx);                 // m = -inf -> 0
    l *= alpha;
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] *= alpha;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      if (ok[u]) {
        const float w = expf(s[u] - mx);
        l += w;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = fmaf(w, vf[u][e], o[e]);
      }
    }
    m = mx;
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] *= inv;
  VecIO<KT, EPL>::store(reinterpret_cast<KT*>(p.out) + (int64_t)row * d + lane * EPL, o);
}

template <typename KT>
static int launch_attn_t(const AttnParams& p, hipStream_t st) {
  const int epl = p.d / 64;
  dim3 grid((p.M + 3) / 4), block(256);
  switch (epl) {
    case 1: hipLaunchKernelGGL((attn_rows_kernel<KT, 1, 4>), grid, block, 0, st, p); break;
    case 2: hipLaunchKernelGGL((attn_rows_kernel<KT, 2, 4>), grid, block, 0, st, p); break;
    case 4: hipLaunchKernelGGL((attn_rows_kernel<KT, 4, 4>), grid, block, 0, st, p); break;
    case 8: hipLaunchKernelGGL((attn_rows_kernel<KT, 8, 4>), grid, block, 0, st, p); break;
    case 16: hipLaunchKernelGGL((attn_rows_kernel<KT, 16, 2>), grid, block, 0, st, p); break;
    default: NMTGAN_FAIL(-2, "attn: d/64 must be 1,2,4,8 or 16 (d=%d)", p.d);
  }
  NMTGAN_CHECK_LAUNCH("attn_rows");
  return 0;
}

int launch_attn_rows(const nmtgan_attn_args* a, hipStream_t st) {
  prof::Scope prof_scope(prof::ATTN_PREFILL, st);
  NMTGAN_REQUIRE(a != nullptr, "attn: null args");
  if (a->M == 0) return 0;
  NMTGAN_REQUIRE(a->M > 0 && a->d > 0 && a->d % 64 == 0, "attn: bad M=%d d=%d", a->M, a->d);
  NMTGAN_REQUIRE(a->heads > 0 && 64 % a->heads == 0 && (a->d / a->heads) % (a->d / 64) == 0,
                 "attn: heads=%d must divide 64", a->heads);
  NMTGAN_REQUIRE(a->q && a->out, "attn: null q/out");
  NMTGAN_REQUIRE(a->k1 || a->n_steps2 > 0, "attn: no keys");
  NMTGAN_REQUIRE(!a->k1 || (a->v1 && a->len1 > 0), "attn: segment 1 incomplete");
  NMTGAN_REQUIRE(!a->k1 || a->seg1_blk || a->rows_per_blk > 0, "attn: segment 1 needs seg1_blk or rows_per_blk");
  NMTGAN_REQUIRE(!(a->causal && !a->seg1_cnt) || (a->rows_per_blk > 0 && a->rows_per_blk <= a->len1),
                 "attn: causal needs rows_per_blk <= len1");
  NMTGAN_REQUIRE(a->n_steps2 == 0 || (a->k2 && a->v2 && a->step_off), "attn: segment 2 incomplete");
  NMTGAN_REQUIRE((!a->o_init && !a->m_init && !a->l_init) || (a->o_init && a->m_init && a->l_init),
                 "attn: o_init/m_init/l_init must be given together");
  AttnParams p{a->q, a->M, a->d, a->heads, a->k1, a->v1, a->len1, a->seg1_blk, a->seg1_cnt, a->rows_per_blk,
               a->causal, a->mask1, a->k2, a->v2, a->step_off, a->n_steps2, a->out, a->o_init, a->m_init, a->l_init};
  if (a->kv_dtype == NMTGAN_BF16) return launch_attn_t<bf16_t>(p, st);
  if (a->kv_dtype == NMTGAN_F32) return launch_attn_t<float>(p, st);
  NMTGAN_FAIL(-2, "attn: bad kv_dtype %d", a->kv_dtype);
}

}  // namespace nmtgan
