// Group attention for gfx950 (16-bit tiers: bf16 or fp16 operands): the R rollouts of one (sentence, prefix) group share their
// keys - the teacher-forced prefix of the sentence, or the encoder memory - so one workgroup loads
// every K/V element ONCE for all R query rows and does the arithmetic on the matrix cores.
//
// Per head, per tile of 32 keys (v_mfma_f32_32x32x16_bf16 throughout):
//   S^T[key, row] = K[key, :] . Q[row, :]^T     A = K rows straight from global (16 B per lane),
//                                               B = Q rows (fp32 -> bf16 once per head)
//   online softmax along the keys: a query row is a lane column of the accumulator, its 32 keys sit
//   in the 16 registers of the two half-waves -> in-lane reductions + one cross-half exchange
//   O^T[dv, row] += V^T[dv, key] . P^T[key, row]  B = the S^T accumulator itself, converted to bf16
//                                               in place (guide section 3, "accumulator tile as the
//                                               next MFMA's operand"); A = V^T, which is why shared
//                                               V is kept TRANSPOSED in HBM: [block][d][Lp].
// Outputs either the normalised attention (cross-attention) or the un-normalised state (o, m, l)
// that attn_rows_kernel continues with over the row's private suffix keys (rollout self-attention).
// Replaces dot_product_attention (common_attention.py:294-331) for the shared-key part.
#include "common.h"
#include "kernels.h"

namespace nmtgan {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct GroupAttnParams {
  const void* q;               // [M][d] fp32 or the 16-bit op dtype (q_bf16), already scaled
  int M, d, heads, R;          // R rows per group (<= 32)
  const void* k;               // [nblk][L][d]       16-bit op dtype (bf16 or fp16)
  const void* vt;              // [nblk][d][Lp]
  int L, Lp;
  const int32_t* row_blk;      // [M] block of a row (read at the group's first row)
  const int32_t* row_cnt;      // [M] keys of a row, or NULL -> L
  const uint8_t* mask;         // [nblk][L] or NULL
  void* out;                   // [M][d] normalised output in the op dtype (final mode) or NULL
  float* o_part; float* m_part; float* l_part;   // partial mode: [M][d], [M][heads], [M][heads]
  // slot form: workgroup e serves `slots` groups of ONE block of keys, table row e =
  // [blk, 0, row0[slots], cnt[slots]]; a slot is live when 0 <= row0 < M (its R rows are row0 .. row0+R-1)
  const int32_t* slot_table; int slots;
  int out_split;               // fp32 kernel, final mode: out is the fp16 [M][3d] split-precision operand image
};


// One wave per (group, head); a workgroup = 4 heads of one group (grid.y = heads / 4).  The kernel is
// latency-bound (two 32-key tiles per head at T = Ls = 50, ~30 MFMAs), so every global load of a 64-key
// pair - q, both K tiles, both V^T tiles, the padding mask - is issued before the first MFMA.
template <typename KT, int DK, bool QB>
__global__ __launch_bounds__(256, (DK <= 64 ? 3 : 1)) void attn_group_kernel(GroupAttnParams p) {
  constexpr int KS = DK / 16;      // k-steps of the QK^T product
  constexpr int DT = DK / 32;      // 32-row tiles of O^T
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, h2 = lane >> 5;
  const int head = blockIdx.y * 4 + wave;
  if (head >= p.heads) return;
  int row, blk, cnt, cnt_lane;
  bool row_ok;
  if (p.slot_table) {
    // several groups of one sentence side by side in the 32 MFMA columns (R = 16: two groups per wave)
    const int32_t* tab = p.slot_table + (int64_t)blockIdx.x * (2 + 2 * p.slots);
    blk = tab[0];
    cnt = 0;
    for (int sl = 0; sl < p.slots; ++sl) {          // uniform: scalar loads
      const int r0 = tab[2 + sl];
      if (r0 >= 0 && r0 < p.M) cnt = max(cnt, p.row_cnt ? tab[2 + p.slots + sl] : p.L);
    }
    const int sl = n / p.R;
    const int r0 = sl < p.slots ? tab[2 + sl] : -1;
    row_ok = r0 >= 0 && r0 < p.M;
    row = row_ok ? r0 + n % p.R : 0;
    cnt_lane = row_ok ? (p.row_cnt ? tab[2 + p.slots + sl] : p.L) : 0;
  } else {
    const int row0 = blockIdx.x * p.R;
    row = row0 + n;
    row_ok = n < p.R && row < p.M;
    blk = p.row_blk[row0];
    cnt = p.row_cnt ? p.row_cnt[row0] : p.L;
    cnt_lane = cnt;
  }
  const int d = p.d;
  const uint8_t* mk = p.mask ? p.mask + (int64_t)blk * p.L : nullptr;

  uint4 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if constexpr (QB) {
      uint4 t = make_uint4(0, 0, 0, 0);
      if (row_ok) t = *reinterpret_cast<const uint4*>(reinterpret_cast<const KT*>(p.q) + (int64_t)row * d + head * DK + 16 * s + 8 * h2);
      qf[s] = t;
    } else {
      float t[8];
      if (row_ok) VecIO<float, 8>::load(reinterpret_cast<const float*>(p.q) + (int64_t)row * d + head * DK + 16 * s + 8 * h2, t);
      else {
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = 0.f;
      }
      qf[s] = pack8u<KT>(t);
    }
  }
  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
  float m = -INFINITY, l = 0.f;
  const KT* Kb = reinterpret_cast<const KT*>(p.k) + (int64_t)blk * p.L * d + head * DK + 8 * h2;
  const KT* Vb = reinterpret_cast<const KT*>(p.vt) + ((int64_t)blk * d + head * DK + n) * p.Lp + 4 * h2;

  auto load_k = [&](int k0, uint4 (&kr)[KS]) {
    const int keyc = min(k0 + n, p.L - 1);
#pragma unroll
    for (int s = 0; s < KS; ++s) kr[s] = *reinterpret_cast<const uint4*>(Kb + (int64_t)keyc * d + 16 * s);
  };
  auto load_v = [&](int k0, uint4 (&vr)[DT][2]) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        // element j of half h2 must be key 16s + 8(j>>2) + 4*h2 + (j&3) (operand k-order of the acc tile)
        const KT* vp = Vb + (int64_t)(32 * t) * p.Lp + k0 + 16 * s;
        const uint2 lo = *reinterpret_cast<const uint2*>(vp);
        const uint2 hi = *reinterpret_cast<const uint2*>(vp + 8);
        vr[t][s] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
  };
  // one 32-key tile: S^T = K.Q^T, online softmax along the keys, O^T += V^T.P^T; bad = padding bits of the tile
  auto tile = [&](int k0, const uint4 (&kr)[KS], const uint4 (&vr)[DT][2], unsigned bad) {
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      sacc = mfma16<KT>(kr[s], qf[s], sacc);
    float sv[16];
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int kt = (e & 3) + 8 * (e >> 2) + 4 * h2;
      const bool ok = k0 + kt < cnt_lane && !((bad >> kt) & 1u);
      sv[e] = ok ? sacc[e] : -INFINITY;
      mx = fmaxf(mx, sv[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    // rows whose keys so far are all masked keep m = -inf: their p and alpha evaluate to 0 via the guard
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __expf(m - m_use);
    float pv[16];
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      pv[e] = __expf(sv[e] - m_use);
      ls += pv[e];
    }
    ls += __shfl_xor(ls, 32, 64);
    l = l * alpha + ls;
    m = m_new;
    const uint4 pf[2] = {pack8u<KT>(pv), pack8u<KT>(pv + 8)};
#pragma unroll
    for (int t = 0; t < DT; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
#pragma unroll
      for (int s = 0; s < 2; ++s)
        o[t] = mfma16<KT>(vr[t][s], pf[s], o[t]);
    }
  };

  for (int k0 = 0; k0 < cnt; k0 += 64) {
    const bool two = k0 + 32 < cnt;               // wave-uniform
    uint4 ka[KS], kb[KS], va[DT][2], vb[DT][2];
    bool mb = false;
    if (mk) {
      const int key = k0 + lane;
      mb = key < p.L && mk[key] != 0;
    }
    load_k(k0, ka);
    if (two) load_k(k0 + 32, kb);
    load_v(k0, va);
    if (two) load_v(k0 + 32, vb);                 // two => k0 + 64 <= Lp (Lp is a multiple of 32, >= L >= cnt)
    const unsigned long long bad = __ballot(mb);
    tile(k0, ka, va, (unsigned)bad);
    if (two) tile(k0 + 32, kb, vb, (unsigned)(bad >> 32));
  }
  if (!row_ok) return;
  // O^T tile t: this lane = query row n, register e = dv 32t + (e&3) + 8(e>>2) + 4*h2
  if (p.out) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v4[4] = {o[t][4 * c] * inv, o[t][4 * c + 1] * inv, o[t][4 * c + 2] * inv, o[t][4 * c + 3] * inv};
        VecIO<KT, 4>::store(reinterpret_cast<KT*>(p.out) + (int64_t)row * d + head * DK + 32 * t + 8 * c + 4 * h2, v4);
      }
  } else {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v4[4] = {o[t][4 * c], o[t][4 * c + 1], o[t][4 * c + 2], o[t][4 * c + 3]};
        VecIO<float, 4>::store(p.o_part + (int64_t)row * d + head * DK + 32 * t + 8 * c + 4 * h2, v4);
      }
    if (h2 == 0) {
      p.m_part[(int64_t)row * p.heads + head] = m;
      p.l_part[(int64_t)row * p.heads + head] = l;
    }
  }
}

// fp32 form of the same kernel (fp32 K / V^T caches and q; the split-precision generator tier): v_mfma_f32_32x32x2f32
// throughout.  A 32x32x2 MFMA sums two k: one from each half-wave.  QK^T: half-wave h2 owns the head dims
// [h2 DK/2, (h2+1) DK/2) of BOTH operands (any split of k between the halves leaves the dot product unchanged), so a
// lane's K row slice and q slice are contiguous 16-byte loads and step s multiplies element s of the two slices.
// PV: step e multiplies the lane's own p register e - key (e&3) + 8 (e>>2) + 4 h2 of the tile, the C layout - by
// V^T[dv][that key], so the probabilities never leave their registers and stay fp32.
template <int DK>
__global__ __launch_bounds__(256) void attn_group_f32_kernel(GroupAttnParams p) {
  constexpr int HK = DK / 2;       // head dims per half-wave
  constexpr int DT = DK / 32;      // 32-row tiles of O^T
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, h2 = lane >> 5;
  const int head = blockIdx.y * 4 + wave;
  if (head >= p.heads) return;
  int row, blk, cnt, cnt_lane;
  bool row_ok;
  if (p.slot_table) {
    const int32_t* tab = p.slot_table + (int64_t)blockIdx.x * (2 + 2 * p.slots);
    blk = tab[0];
    cnt = 0;
    for (int sl = 0; sl < p.slots; ++sl) {
      const int r0 = tab[2 + sl];
      if (r0 >= 0 && r0 < p.M) cnt = max(cnt, p.row_cnt ? tab[2 + p.slots + sl] : p.L);
    }
    const int sl = n / p.R;
    const int r0 = sl < p.slots ? tab[2 + sl] : -1;
    row_ok = r0 >= 0 && r0 < p.M;
    row = row_ok ? r0 + n % p.R : 0;
    cnt_lane = row_ok ? (p.row_cnt ? tab[2 + p.slots + sl] : p.L) : 0;
  } else {
    const int row0 = blockIdx.x * p.R;
    row = row0 + n;
    row_ok = n < p.R && row < p.M;
    blk = p.row_blk[row0];
    cnt = p.row_cnt ? p.row_cnt[row0] : p.L;
    cnt_lane = cnt;
  }
  const int d = p.d;
  const uint8_t* mk = p.mask ? p.mask + (int64_t)blk * p.L : nullptr;

  float qf[HK];
  if (row_ok) VecIO<float, HK>::load(reinterpret_cast<const float*>(p.q) + (int64_t)row * d + head * DK + HK * h2, qf);
  else {
#pragma unroll
    for (int i = 0; i < HK; ++i) qf[i] = 0.f;
  }
  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
  float m = -INFINITY, l = 0.f;
  const float* Kb = reinterpret_cast<const float*>(p.k) + (int64_t)blk * p.L * d + head * DK + HK * h2;
  const float* Vb = reinterpret_cast<const float*>(p.vt) + ((int64_t)blk * d + head * DK + n) * p.Lp + 4 * h2;

  for (int k0 = 0; k0 < cnt; k0 += 32) {
    bool mb = false;
    if (mk) {
      const int key = k0 + n;
      mb = key < p.L && mk[key] != 0;
    }
    float kr[HK], vr[DT][16];
    VecIO<float, HK>::load(Kb + (int64_t)min(k0 + n, p.L - 1) * d, kr);
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {             // keys k0 + 8 gq + 4 h2 + (0..3) = registers 4 gq .. 4 gq + 3
        const float4 v = *reinterpret_cast<const float4*>(Vb + (int64_t)(32 * t) * p.Lp + k0 + 8 * gq);
        vr[t][4 * gq] = v.x; vr[t][4 * gq + 1] = v.y; vr[t][4 * gq + 2] = v.z; vr[t][4 * gq + 3] = v.w;
      }
    const unsigned bad = (unsigned)(__ballot(mb && h2 == 0));
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < HK; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[s], qf[s], sacc, 0, 0, 0);
    float sv[16];
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int kt = (e & 3) + 8 * (e >> 2) + 4 * h2;
      const bool ok = k0 + kt < cnt_lane && !((bad >> kt) & 1u);
      sv[e] = ok ? sacc[e] : -INFINITY;
      mx = fmaxf(mx, sv[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __expf(m - m_use);
    float pv[16];
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      pv[e] = __expf(sv[e] - m_use);
      ls += pv[e];
    }
    ls += __shfl_xor(ls, 32, 64);
    l = l * alpha + ls;
    m = m_new;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[t][e], pv[e], o[t], 0, 0, 0);
    }
  }
  if (!row_ok) return;
  // O^T tile t: this lane = query row n, register e = dv 32t + (e&3) + 8(e>>2) + 4*h2
  const float inv = p.out ? (l > 0.f ? 1.f / l : 0.f) : 1.f;
  float* dst = p.out ? reinterpret_cast<float*>(p.out) : p.o_part;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v4[4] = {o[t][4 * c] * inv, o[t][4 * c + 1] * inv, o[t][4 * c + 2] * inv, o[t][4 * c + 3] * inv};
      const int col = head * DK + 32 * t + 8 * c + 4 * h2;
      if (p.out && p.out_split) store_split3<4>(reinterpret_cast<f16_t*>(p.out) + (int64_t)row * 3 * d + col, d, v4);
      else VecIO<float, 4>::store(dst + (int64_t)row * d + col, v4);
    }
  if (!p.out && h2 == 0) {
    p.m_part[(int64_t)row * p.heads + head] = m;
    p.l_part[(int64_t)row * p.heads + head] = l;
  }
}

// V [nblk*L][d] -> V^T [nblk][d][Lp], zero padded to Lp (multiple of 32, >= L)
template <typename RawT>                           // raw elements: 16-bit (bf16 or fp16) or 32-bit (fp32)
__global__ void transpose_v_kernel(const RawT* v, RawT* vt, int nblk, int L, int Lp, int d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)nblk * d * Lp) return;
  const int t = (int)(i % Lp);
  const int c = (int)((i / Lp) % d);
  const int b = (int)(i / ((int64_t)Lp * d));
  vt[i] = t < L ? v[((int64_t)b * L + t) * d + c] : (RawT)0;
}

int launch_transpose_v(const void* v, void* vt, int nblk, int L, int Lp, int d, int esize, hipStream_t st) {
  prof::Scope prof_scope(prof::OTHER, st);
  NMTGAN_REQUIRE(v && vt && nblk > 0 && L > 0 && Lp >= L && Lp % 32 == 0 && d > 0 && (esize == 2 || esize == 4),
                 "transpose_v: bad arguments");
  const int64_t n = (int64_t)nblk * d * Lp;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (esize == 2)
    hipLaunchKernelGGL(transpose_v_kernel<unsigned short>, grid, dim3(256), 0, st, (const unsigned short*)v, (unsigned short*)vt,
                       nblk, L, Lp, d);
  else
    hipLaunchKernelGGL(transpose_v_kernel<unsigned>, grid, dim3(256), 0, st, (const unsigned*)v, (unsigned*)vt, nblk, L, Lp, d);
  NMTGAN_CHECK_LAUNCH("transpose_v");
  return 0;
}

int launch_attn_group(const nmtgan_attn_group_args* a, hipStream_t st) {
  prof::Scope prof_scope(prof::ATTN_PREFILL, st);
  NMTGAN_REQUIRE(a != nullptr, "attn_group: null args");
  if (a->M == 0) return 0;
  NMTGAN_REQUIRE(a->M > 0 && a->d > 0 && a->heads > 0 && a->d % a->heads == 0, "attn_group: bad M/d/heads");
  NMTGAN_REQUIRE(a->rows_per_group > 0 && a->rows_per_group <= 32 && a->M % a->rows_per_group == 0,
                 "attn_group: rows_per_group must be 1..32 and divide M");
  NMTGAN_REQUIRE(a->q && a->k && a->vt && (a->row_blk || a->slot_table), "attn_group: null pointer");
  NMTGAN_REQUIRE(!a->slot_table || (a->slots > 0 && a->slots * a->rows_per_group <= 32 && a->n_entries >= 0),
                 "attn_group: slot table needs 1 <= slots <= 32 / rows_per_group and n_entries >= 0");
  if (a->slot_table && a->n_entries == 0) return 0;
  NMTGAN_REQUIRE(a->len > 0 && a->len_padded >= a->len && a->len_padded % 32 == 0, "attn_group: bad key lengths");
  NMTGAN_REQUIRE((a->out != nullptr) != (a->o_part != nullptr), "attn_group: exactly one of out / o_part");
  NMTGAN_REQUIRE(a->out || (a->m_part && a->l_part), "attn_group: partial mode needs m_part and l_part");
  GroupAttnParams p{a->q, a->M, a->d, a->heads, a->rows_per_group, a->k, a->vt, a->len,
                    a->len_padded, a->row_blk, a->row_cnt, a->mask, a->out, a->o_part, a->m_part, a->l_part,
                    a->slot_table, a->slots, a->out_split};
  NMTGAN_REQUIRE(!a->out_split || (a->kv_dtype == NMTGAN_F32 && a->out), "attn_group: out_split needs NMTGAN_F32 and `out`");
  NMTGAN_REQUIRE(valid_op_dtype(a->kv_dtype), "attn_group: bad kv_dtype %d", a->kv_dtype);
  const bool half = a->kv_dtype == NMTGAN_F16;
  dim3 grid(a->slot_table ? a->n_entries : a->M / a->rows_per_group, (a->heads + 3) / 4), block(256);
  if (a->kv_dtype == NMTGAN_F32) {                 // fp32 caches, fp32 q (either flag value), fp32 output
    NMTGAN_REQUIRE((uintptr_t)a->q % 16 == 0 && (uintptr_t)a->k % 16 == 0 && (uintptr_t)a->vt % 16 == 0 && a->d % 4 == 0,
                   "attn_group: fp32 operands must be 16-byte aligned");
    switch (a->d / a->heads) {
      case 32: hipLaunchKernelGGL(attn_group_f32_kernel<32>, grid, block, 0, st, p); break;
      case 64: hipLaunchKernelGGL(attn_group_f32_kernel<64>, grid, block, 0, st, p); break;
      default: NMTGAN_FAIL(-2, "attn_group: fp32 head dim %d not in {32, 64}", a->d / a->heads);
    }
    NMTGAN_CHECK_LAUNCH("attn_group_f32");
    return 0;
  }
  const bool qb = a->q_bf16 != 0;
  NMTGAN_REQUIRE(!qb || ((uintptr_t)a->q % 16 == 0 && a->d % 8 == 0), "attn_group: bf16 q must be 16-byte aligned");
#define NMTGAN_GROUP(DK)                                                                              \
  if (half) {                                                                                         \
    if (qb) hipLaunchKernelGGL((attn_group_kernel<f16_t, DK, true>), grid, block, 0, st, p);          \
    else hipLaunchKernelGGL((attn_group_kernel<f16_t, DK, false>), grid, block, 0, st, p);            \
  } else if (qb) hipLaunchKernelGGL((attn_group_kernel<bf16_t, DK, true>), grid, block, 0, st, p);    \
  else hipLaunchKernelGGL((attn_group_kernel<bf16_t, DK, false>), grid, block, 0, st, p)
  switch (a->d / a->heads) {
    case 32: NMTGAN_GROUP(32); break;
    case 64: NMTGAN_GROUP(64); break;
    case 128: NMTGAN_GROUP(128); break;
    default: NMTGAN_FAIL(-2, "attn_group: head dim %d not in {32, 64, 128}", a->d / a->heads);
  }
#undef NMTGAN_GROUP
  NMTGAN_CHECK_LAUNCH("attn_group");
  return 0;
}

}  // namespace nmtgan
