// Row attention for gfx950: one wavefront per query row, all heads at once.
//
// Lane l owns elements [l*EPL, (l+1)*EPL) of the d-wide row (EPL = d/64), so one wave-wide load
// reads one whole K (or V) vector - d*esize contiguous bytes, 16 B per lane at d=512 bf16 - and a
// head's dot product is a shuffle reduction over its 64/heads adjacent lanes.  Softmax is online
// (running max / sum in fp32).  Keys come from two segments: a block of keys shared by many rows
// (teacher-forced prefix of the sentence, or the encoder memory) and the row's private suffix in
// the step-major rollout cache.  The kernel is HBM/L2-bandwidth bound: per key it moves
// 2*d*esize bytes for ~5*EPL VALU instructions.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace nmtgan {

struct AttnParams {
  const void* q; int M, d, heads;
  const void* k1; const void* v1; int len1;
  const int32_t* seg1_blk; const int32_t* seg1_cnt; int rows_per_blk; int causal;
  const uint8_t* mask1;
  const void* k2; const void* v2; const int64_t* step_off; int n_steps2;
  void* out;
  const float* o_init; const float* m_init; const float* l_init;
  int out_split;               // fp32 K/V only: out is the fp16 [M][3d] split-precision operand image of the result
};

template <typename KT, int EPL, int KB, typename QT>
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
  VecIO<QT, EPL>::load(reinterpret_cast<const QT*>(p.q) + (int64_t)row * d + lane * EPL, q);

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
  if constexpr (sizeof(KT) == 4) {
    if (p.out_split) {
      store_split3<EPL>(reinterpret_cast<f16_t*>(p.out) + (int64_t)row * 3 * d + lane * EPL, d, o);
      return;
    }
  }
  VecIO<KT, EPL>::store(reinterpret_cast<KT*>(p.out) + (int64_t)row * d + lane * EPL, o);
}

// ------------------------------------------------------------------------------------------ 16-bit fast path
// Same row-per-wave algorithm for bf16 / fp16 caches with d = 512 / 1024 (the rollout step's self-attention: the
// shared teacher-forced prefix of the row's sentence - L2-resident, and the four rows of a workgroup belong to one
// (sentence, prefix) group, so three of four reads hit L1 - followed by the row's private suffix, 278 GB of K/V per C2
// get_reward), tuned for bytes in flight (workgroups of 8 rows run the same, of 16 rows 10 % slower):
//  - K/V stay packed (16 B per lane and vector) and the NEXT batch of KB keys is requested before the
//    current one is consumed, so 2*KB*2 vector loads per wave are in flight across the softmax chain;
//  - head reductions use DPP (quad_perm / row_half_mirror / row_mirror) instead of ds_bpermute;
//  - scores are kept in log2 units (q pre-scaled by log2 e): every exponential is one v_exp_f32.
__device__ __forceinline__ float head_sum(float v, int lph) {   // sum over the lph adjacent lanes of a head
  if (lph > 1) v = dpp_add<0xB1>(v);               // quad_perm [1,0,3,2]
  if (lph > 2) v = dpp_add<0x4E>(v);               // quad_perm [2,3,0,1]
  if (lph > 4) v = dpp_add<0x141>(v);              // row_half_mirror
  if (lph > 8) v = dpp_add<0x140>(v);              // row_mirror
  if (lph > 16) v += __shfl_xor(v, 16, 64);
  if (lph > 32) v += __shfl_xor(v, 32, 64);
  return v;
}

template <typename KT, int EPL, int KB, typename QT>
__global__ __launch_bounds__(256) void attn_rows_16_kernel(AttnParams p) {
  constexpr int NV = EPL / 8;                      // 16-byte vectors per lane and key
  constexpr float LOG2E = 1.44269504088896341f;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int lph = 64 / p.heads;
  const int d = p.d;
  const int blk = __builtin_amdgcn_readfirstlane(p.seg1_blk ? p.seg1_blk[row] : (p.rows_per_blk > 0 ? row / p.rows_per_blk : 0));
  int cnt1 = 0;
  if (p.k1) cnt1 = p.seg1_cnt ? p.seg1_cnt[row] : (p.causal ? row % p.rows_per_blk + 1 : p.len1);
  cnt1 = __builtin_amdgcn_readfirstlane(cnt1);
  const int n_total = cnt1 + p.n_steps2;
  const KT* k1 = reinterpret_cast<const KT*>(p.k1) + ((int64_t)blk * p.len1) * d + lane * EPL;
  const KT* v1 = reinterpret_cast<const KT*>(p.v1) + ((int64_t)blk * p.len1) * d + lane * EPL;
  const KT* k2 = reinterpret_cast<const KT*>(p.k2) + (int64_t)row * d + lane * EPL;
  const KT* v2 = reinterpret_cast<const KT*>(p.v2) + (int64_t)row * d + lane * EPL;
  const uint8_t* mk = p.mask1 ? p.mask1 + (int64_t)blk * p.len1 : nullptr;

  // fp16 q against an fp16 cache: the products go through v_dot2_f32_f16 (two exact fp16 x fp16 products per
  // instruction, fp32 accumulation) on the packed registers as loaded, and the score is scaled to log2 units after the
  // head reduction; every other combination multiplies unpacked fp32 values with q pre-scaled
  constexpr bool DOT2 = __is_same(KT, f16_t) && __is_same(QT, f16_t);
  float q[EPL];
  uint4 qp[NV];
  if constexpr (DOT2) {
#pragma unroll
    for (int i = 0; i < NV; ++i) qp[i] = reinterpret_cast<const uint4*>(reinterpret_cast<const KT*>(p.q) + (int64_t)row * d + lane * EPL)[i];
  } else {
    VecIO<QT, EPL>::load(reinterpret_cast<const QT*>(p.q) + (int64_t)row * d + lane * EPL, q);
#pragma unroll
    for (int e = 0; e < EPL; ++e) q[e] *= LOG2E;
  }

  float m = -INFINITY, l = 0.f;                    // m in log2 units
  float o[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] = 0.f;
  if (p.o_init) {
    const int head = (lane * EPL) / (d / p.heads);
    m = p.m_init[(int64_t)row * p.heads + head] * LOG2E;
    l = p.l_init[(int64_t)row * p.heads + head];
    VecIO<float, EPL>::load(p.o_init + (int64_t)row * d + lane * EPL, o);
  }

  uint4 kc[KB][NV], vc[KB][NV], kn[KB][NV], vn[KB][NV];
  unsigned okc = 0, okn = 0;                       // bit u: key u of the batch is valid
  auto fetch = [&](int jb, uint4 (&kr)[KB][NV], uint4 (&vr)[KB][NV], unsigned& okm) {
    okm = 0;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int j = jb + u;                        // wave-uniform
      if (j < cnt1) {                              // uniform branches: the loads sit inside them on purpose
        if (!(mk && mk[j])) {                      // (a pointer select here is lowered to a scratch array)
          okm |= 1u << u;
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            kr[u][i] = reinterpret_cast<const uint4*>(k1 + (int64_t)j * d)[i];
            vr[u][i] = reinterpret_cast<const uint4*>(v1 + (int64_t)j * d)[i];
          }
        }
      } else if (j < n_total) {
        const int64_t off = p.step_off[j - cnt1] * (int64_t)d;
        okm |= 1u << u;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          kr[u][i] = reinterpret_cast<const uint4*>(k2 + off)[i];
          vr[u][i] = reinterpret_cast<const uint4*>(v2 + off)[i];
        }
      }
    }
  };
  auto unpack = [](const uint4 (&r)[NV], float (&f)[EPL]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) unpack8<KT>(r[i], f + 8 * i);
  };
  auto consume = [&](const uint4 (&kr)[KB][NV], const uint4 (&vr)[KB][NV], unsigned okm) {
    float s[KB];
    float mx = m;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      float part = 0.f;
      const bool ok = (okm >> u) & 1;
      if (ok) {
        if constexpr (DOT2) {
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            part = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, qp[i].x), __builtin_bit_cast(f16x2_t, kr[u][i].x), part, false);
            part = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, qp[i].y), __builtin_bit_cast(f16x2_t, kr[u][i].y), part, false);
            part = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, qp[i].z), __builtin_bit_cast(f16x2_t, kr[u][i].z), part, false);
            part = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, qp[i].w), __builtin_bit_cast(f16x2_t, kr[u][i].w), part, false);
          }
        } else {
          float kf[EPL];
          unpack(kr[u], kf);
#pragma unroll
          for (int e = 0; e < EPL; ++e) part = fmaf(q[e], kf[e], part);
        }
      }
      part = head_sum(part, lph);
      if constexpr (DOT2) part *= LOG2E;
      s[u] = ok ? part : -INFINITY;
      mx = fmaxf(mx, s[u]);
    }
    if (mx == -INFINITY) return;
    const float alpha = __builtin_amdgcn_exp2f(m - mx);
    l *= alpha;
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] *= alpha;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      if ((okm >> u) & 1) {
        const float w = __builtin_amdgcn_exp2f(s[u] - mx);
        l += w;
        float vf[EPL];
        unpack(vr[u], vf);
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = fmaf(w, vf[e], o[e]);
      }
    }
    m = mx;
  };

  fetch(0, kc, vc, okc);
  for (int jb = 0; jb < n_total; jb += 2 * KB) {
    fetch(jb + KB, kn, vn, okn);
    consume(kc, vc, okc);
    fetch(jb + 2 * KB, kc, vc, okc);
    consume(kn, vn, okn);
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] *= inv;
  if constexpr (sizeof(KT) == 4) {
    if (p.out_split) {
      store_split3<EPL>(reinterpret_cast<f16_t*>(p.out) + (int64_t)row * 3 * d + lane * EPL, d, o);
      return;
    }
  }
  VecIO<KT, EPL>::store(reinterpret_cast<KT*>(p.out) + (int64_t)row * d + lane * EPL, o);
}

// The same kernel for calls without a padding mask (the rollout steps' self-attention, 94 % of the launches), written so
// that the instruction stream of a wave is arithmetic and loads: the compiled loop of the kernel above spends 83 scalar
// instructions per key (each `if (valid)` on a wave-uniform flag becomes an EXEC-mask sequence and a branch, every key
// offset a 64-bit scalar multiply, the head reduction six run-time tests) beside 53 vector instructions; this one 22 and
// 34.  Each segment is cut into batches of KB keys on its own (first the prefix, then the suffix); a batch always loads
// KB keys (indices clamped to the segment's last key, so the tail re-reads a valid key) and an invalid slot only gets
// its score replaced by -inf, which makes its weight exactly 0.  The kernel above batches the concatenated sequence,
// so when the prefix length is no multiple of KB the two rescale the running sums at different keys and differ in the
// last place of the stored value (tests/test_gpu_kernels.py compares them at that level); which kernel a call takes
// depends on its mask argument and shape alone, never on the batch, so a row's bits are still a function of the row.
// (Worth 2 % of the kernel's time, no more: it waits on memory either way - two-key batches at seven waves per SIMD
// instead of four-key batches at four run the same.)
// d = 64 EPL and the lanes per head (LPH = 64 / heads) are compile-time values: key offsets are shifts instead of
// 64-bit scalar multiplies, and the head reduction has no run-time tests.
template <typename KT, int EPL, int KB, typename QT, int LPH>
__global__ __launch_bounds__(256) void attn_rows_16s_kernel(AttnParams p) {
  constexpr int NV = EPL / 8;
  constexpr float LOG2E = 1.44269504088896341f;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  constexpr int lph = LPH;
  constexpr int d = 64 * EPL;
  const int blk = __builtin_amdgcn_readfirstlane(p.seg1_blk ? p.seg1_blk[row] : (p.rows_per_blk > 0 ? row / p.rows_per_blk : 0));
  int cnt1 = 0;
  if (p.k1) cnt1 = p.seg1_cnt ? p.seg1_cnt[row] : (p.causal ? row % p.rows_per_blk + 1 : p.len1);
  cnt1 = __builtin_amdgcn_readfirstlane(cnt1);
  const int n2 = p.n_steps2;
  const int nb1 = (cnt1 + KB - 1) / KB, nb = nb1 + (n2 + KB - 1) / KB;       // batches of segment 1, of both
  const KT* k1 = reinterpret_cast<const KT*>(p.k1) + ((int64_t)blk * p.len1) * d + lane * EPL;
  const KT* v1 = reinterpret_cast<const KT*>(p.v1) + ((int64_t)blk * p.len1) * d + lane * EPL;
  const KT* k2 = reinterpret_cast<const KT*>(p.k2) + (int64_t)row * d + lane * EPL;
  const KT* v2 = reinterpret_cast<const KT*>(p.v2) + (int64_t)row * d + lane * EPL;

  constexpr bool DOT2 = __is_same(KT, f16_t) && __is_same(QT, f16_t);
  float q[EPL];
  uint4 qp[NV];
  if constexpr (DOT2) {
#pragma unroll
    for (int i = 0; i < NV; ++i) qp[i] = reinterpret_cast<const uint4*>(reinterpret_cast<const KT*>(p.q) + (int64_t)row * d + lane * EPL)[i];
  } else {
    VecIO<QT, EPL>::load(reinterpret_cast<const QT*>(p.q) + (int64_t)row * d + lane * EPL, q);
#pragma unroll
    for (int e = 0; e < EPL; ++e) q[e] *= LOG2E;
  }

  float m = -INFINITY, l = 0.f;                    // m in log2 units
  float o[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] = 0.f;
  if (p.o_init) {
    const int head = (lane * EPL) / (d / p.heads);
    m = p.m_init[(int64_t)row * p.heads + head] * LOG2E;
    l = p.l_init[(int64_t)row * p.heads + head];
    VecIO<float, EPL>::load(p.o_init + (int64_t)row * d + lane * EPL, o);
  }

  uint4 kc[KB][NV], vc[KB][NV], kn[KB][NV], vn[KB][NV];
  auto fetch = [&](int b, uint4 (&kr)[KB][NV], uint4 (&vr)[KB][NV]) {
    if (b < nb1) {
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int64_t off = (int64_t)min(b * KB + u, cnt1 - 1) * d;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          kr[u][i] = reinterpret_cast<const uint4*>(k1 + off)[i];
          vr[u][i] = reinterpret_cast<const uint4*>(v1 + off)[i];
        }
      }
    } else {
      int64_t so[KB];                              // the batch's step offsets first: KB scalar loads in flight together
#pragma unroll
      for (int u = 0; u < KB; ++u) so[u] = p.step_off[min((b - nb1) * KB + u, n2 - 1)];
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int64_t off = so[u] * (int64_t)d;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          kr[u][i] = reinterpret_cast<const uint4*>(k2 + off)[i];
          vr[u][i] = reinterpret_cast<const uint4*>(v2 + off)[i];
        }
      }
    }
  };
  auto consume = [&](int b, const uint4 (&kr)[KB][NV], const uint4 (&vr)[KB][NV]) {
    const int nvalid = b < nb1 ? cnt1 - b * KB : n2 - (b - nb1) * KB;     // >= 1; slots u >= nvalid repeat the last key
    float s[KB];
    float mx = m;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      float part = 0.f;
      if constexpr (DOT2) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          part = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, qp[i].x), __builtin_bit_cast(f16x2_t, kr[u][i].x), part, false);
          part = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, qp[i].y), __builtin_bit_cast(f16x2_t, kr[u][i].y), part, false);
          part = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, qp[i].z), __builtin_bit_cast(f16x2_t, kr[u][i].z), part, false);
          part = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, qp[i].w), __builtin_bit_cast(f16x2_t, kr[u][i].w), part, false);
        }
      } else {
        float kf[EPL];
#pragma unroll
        for (int i = 0; i < NV; ++i) unpack8<KT>(kr[u][i], kf + 8 * i);
#pragma unroll
        for (int e = 0; e < EPL; ++e) part = fmaf(q[e], kf[e], part);
      }
      part = head_sum(part, lph);
      if constexpr (DOT2) part *= LOG2E;
      s[u] = u < nvalid ? part : -INFINITY;
      mx = fmaxf(mx, s[u]);
    }
    const float alpha = __builtin_amdgcn_exp2f(m - mx);
    l *= alpha;
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] *= alpha;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const float w = __builtin_amdgcn_exp2f(s[u] - mx);           // exactly 0 for an invalid slot
      l += w;
      float vf[EPL];
#pragma unroll
      for (int i = 0; i < NV; ++i) unpack8<KT>(vr[u][i], vf + 8 * i);
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = fmaf(w, vf[e], o[e]);
    }
    m = mx;
  };

  if (nb > 0) fetch(0, kc, vc);
  for (int b = 0; b < nb; b += 2) {
    const bool second = b + 1 < nb;
    if (second) fetch(b + 1, kn, vn);
    consume(b, kc, vc);
    if (second) {
      if (b + 2 < nb) fetch(b + 2, kc, vc);
      consume(b + 1, kn, vn);
    }
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] *= inv;
  VecIO<KT, EPL>::store(reinterpret_cast<KT*>(p.out) + (int64_t)row * d + lane * EPL, o);
}

template <typename KT, typename QT>
static int launch_attn_t(const AttnParams& p, hipStream_t st) {
  const int epl = p.d / 64;
  dim3 grid((p.M + 3) / 4), block(256);
  if constexpr (sizeof(KT) == 2) {
    static const bool fast = [] { const char* e = getenv("NMTGAN_ATTN_FAST"); return !e || atoi(e) != 0; }();
    static const bool lean = [] { const char* e = getenv("NMTGAN_ATTN_LEAN"); return !e || atoi(e) != 0; }();
    const int lph = 64 / p.heads;
    if (fast && lean && !p.mask1 && epl == 8 && lph == 8) { hipLaunchKernelGGL((attn_rows_16s_kernel<KT, 8, 4, QT, 8>), grid, block, 0, st, p); NMTGAN_CHECK_LAUNCH("attn_rows"); return 0; }
    if (fast && lean && !p.mask1 && epl == 16 && lph == 4) { hipLaunchKernelGGL((attn_rows_16s_kernel<KT, 16, 2, QT, 4>), grid, block, 0, st, p); NMTGAN_CHECK_LAUNCH("attn_rows"); return 0; }
    if (fast && epl == 8) { hipLaunchKernelGGL((attn_rows_16_kernel<KT, 8, 4, QT>), grid, block, 0, st, p); NMTGAN_CHECK_LAUNCH("attn_rows"); return 0; }
    if (fast && epl == 16) { hipLaunchKernelGGL((attn_rows_16_kernel<KT, 16, 2, QT>), grid, block, 0, st, p); NMTGAN_CHECK_LAUNCH("attn_rows"); return 0; }
  }
  switch (epl) {
    case 1: hipLaunchKernelGGL((attn_rows_kernel<KT, 1, 4, QT>), grid, block, 0, st, p); break;
    case 2: hipLaunchKernelGGL((attn_rows_kernel<KT, 2, 4, QT>), grid, block, 0, st, p); break;
    case 4: hipLaunchKernelGGL((attn_rows_kernel<KT, 4, 4, QT>), grid, block, 0, st, p); break;
    case 8: hipLaunchKernelGGL((attn_rows_kernel<KT, 8, 4, QT>), grid, block, 0, st, p); break;
    case 16: hipLaunchKernelGGL((attn_rows_kernel<KT, 16, 2, QT>), grid, block, 0, st, p); break;
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
               a->causal, a->mask1, a->k2, a->v2, a->step_off, a->n_steps2, a->out, a->o_init, a->m_init, a->l_init,
               a->out_split};
  NMTGAN_REQUIRE(!a->out_split || (a->kv_dtype == NMTGAN_F32 && (uintptr_t)a->out % 16 == 0),
                 "attn: out_split needs kv_dtype NMTGAN_F32 and a 16-byte aligned image");
  if (a->kv_dtype == NMTGAN_BF16) return a->q_op_dtype ? launch_attn_t<bf16_t, bf16_t>(p, st) : launch_attn_t<bf16_t, float>(p, st);
  if (a->kv_dtype == NMTGAN_F16) return a->q_op_dtype ? launch_attn_t<f16_t, f16_t>(p, st) : launch_attn_t<f16_t, float>(p, st);
  if (a->kv_dtype == NMTGAN_F32) return launch_attn_t<float, float>(p, st);
  NMTGAN_FAIL(-2, "attn: bad kv_dtype %d", a->kv_dtype);
}

}  // namespace nmtgan
