// MFMA GEMM for gfx950: C[M,N] = A[M,K] . Wt[N,K]^T with fused epilogues.
//
// Both operands are K-contiguous, so A and Wt tiles share one loader and one LDS image:
// 128 rows x 128 BYTES per k-tile (64 bf16 or 32 fp32), 16-byte chunks XOR-swizzled by
// (row>>1)&7 so the ds_read_b128 fragment reads of a 32-row MFMA operand are conflict-free
// (bank = (addr/4)%64, 16-lane groups: cdna_hip_programming.md section 2).
// Block = 256 threads = 4 waves (2x2), block tile 128x128, wave tile 64x64 = 2x2 MFMA 32x32 tiles.
//   bf16: v_mfma_f32_32x32x16_bf16, one per (16-byte A chunk, 16-byte B chunk) pair
//   fp32: v_mfma_f32_32x32x2_f32, four per chunk pair (exact fp32, the 1e-5 parity tier)
// The k order inside a tile is whatever the chunk order gives; A and B use the same order, so the
// sum is over the same k set (order differs from the reference's, as any GEMM's does).
// Register-staged double buffering: global loads of tile t+1 are issued before the MFMAs of tile t
// and written to the other LDS buffer after them; one barrier per k-tile.
// Tile order: XCD-aware remap (blocks b and b+8 share an XCD / L2) + grouped-M order inside it so
// GROUP_M row panels of A stay L2-resident while the Wt tiles stream.
#include "common.h"
#include "kernels.h"

namespace nmtgan {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int BM = 128, BN = 128, BKB = 128;   // BKB: k-tile width in bytes
constexpr int GROUP_M = 16;
constexpr int TILE_BYTES = BM * BKB;            // 16 KiB per operand per buffer

struct GemmParams {
  const char* A; const char* W;
  int M, N, K;
  int64_t lda_b;            // bytes
  int a_group; int64_t a_group_stride_b;
  int64_t ldw_b;            // bytes per Wt row (= K*esize)
  int kbytes;               // K * esize
  int ntm, ntn;
  // linear epilogue
  const float* bias; int relu; int n_seg; int seg_width;
  void* out[3]; int64_t ldo[3]; int out_f32[3]; float seg_scale[3];
  // convmax epilogue
  const float* scale; const float* shift; float* pooled; int64_t ld_pool;
};

template <typename OpT>
__device__ __forceinline__ void mma_chunk(const uint4& a, const uint4& b, f32x16& c);
template <>
__device__ __forceinline__ void mma_chunk<bf16_t>(const uint4& a, const uint4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<float>(const uint4& a, const uint4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
}

template <typename OpT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 buffers][A 16K | B 16K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile index: bijective XCD remap, then grouped-M order
  const int nwg = p.ntm * p.ntn;
  int idx;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int group_size = GROUP_M * p.ntn;
  const int group_id = idx / group_size;
  const int first_m = group_id * GROUP_M;
  const int gsz = min(p.ntm - first_m, GROUP_M);
  const int tm = first_m + (idx % group_size) % gsz;
  const int tn = (idx % group_size) / gsz;
  const int bm0 = tm * BM, bn0 = tn * BN;

  // ---- loader mapping: thread -> (row r0 + 32 i, chunk c)
  const int lc = tid & 7, lr0 = tid >> 3;
  const int lds_wr = lr0 * BKB + ((lc ^ ((lr0 >> 1) & 7)) << 4);
  const char* a_ptr[4];
  const char* w_ptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = min(bm0 + lr0 + 32 * i, p.M - 1);
    int64_t off = (p.a_group > 0) ? (int64_t)(r / p.a_group) * p.a_group_stride_b + (int64_t)(r % p.a_group) * p.lda_b
                                  : (int64_t)r * p.lda_b;
    a_ptr[i] = p.A + off + lc * 16;
    int n = min(bn0 + lr0 + 32 * i, p.N - 1);
    w_ptr[i] = p.W + (int64_t)n * p.ldw_b + lc * 16;
  }
  const int nkt = (p.kbytes + BKB - 1) / BKB;

  uint4 sa[4], sb[4];
  auto gload = [&](int kt) {
    const int kb = kt * BKB + lc * 16;
    const bool ok = kb < p.kbytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sa[i] = ok ? *reinterpret_cast<const uint4*>(a_ptr[i] + (int64_t)kt * BKB) : make_uint4(0, 0, 0, 0);
      sb[i] = ok ? *reinterpret_cast<const uint4*>(w_ptr[i] + (int64_t)kt * BKB) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* base = smem + buf * 2 * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(base + lds_wr + i * 32 * BKB) = sa[i];
      *reinterpret_cast<uint4*>(base + TILE_BYTES + lds_wr + i * 32 * BKB) = sb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment read addressing: row = lane&31 (+32 per MFMA tile), chunk = 2q + (lane>>5)
  const int fr = lane & 31, fh = lane >> 5;
  const int fsw = (fr >> 1) & 7;
  const int a_rd = (wm * 64 + fr) * BKB;
  const int b_rd = TILE_BYTES + (wn * 64 + fr) * BKB;

  gload(0);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) gload(kt + 1);
    const unsigned char* base = smem + cur * 2 * TILE_BYTES;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pc = ((2 * q + fh) ^ fsw) << 4;
      uint4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const uint4*>(base + a_rd + i * 32 * BKB + pc);
        fb[i] = *reinterpret_cast<const uint4*>(base + b_rd + i * 32 * BKB + pc);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma_chunk<OpT>(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < nkt) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue.  32x32 C layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = bn0 + wn * 64 + j * 32 + fr;
    if (n >= p.N) continue;
    if constexpr (EPI == NMTGAN_EPI_LINEAR) {
      const float bias = p.bias ? p.bias[n] : 0.f;
      const int seg = (p.n_seg > 1) ? n / p.seg_width : 0;
      const int nn = (p.n_seg > 1) ? n - seg * p.seg_width : n;
      const float sc = p.seg_scale[seg];
      const int64_t ldo = p.ldo[seg];
      const bool of32 = p.out_f32[seg] != 0;
      float* of = reinterpret_cast<float*>(p.out[seg]);
      OpT* oo = reinterpret_cast<OpT*>(p.out[seg]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = bm0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
          if (m < p.M) {
            float v = (acc[i][j][e] + bias) * sc;
            if (p.relu) v = fmaxf(v, 0.f);
            if (of32) of[(int64_t)m * ldo + nn] = v;
            else OpTraits<OpT>::store(oo + (int64_t)m * ldo + nn, v);
          }
        }
      }
    } else {
      const float sc = p.scale[n], sh = p.shift[n];
      int cur_g = -1;
      float cur_v = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = bm0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
          if (m < p.M) {
            const float v = fmaxf(acc[i][j][e] * sc + sh, 0.f);
            const int g = m / p.a_group;
            if (g != cur_g) {
              if (cur_g >= 0 && cur_v > 0.f)
                atomicMax(reinterpret_cast<int*>(p.pooled + (int64_t)cur_g * p.ld_pool + n), __float_as_int(cur_v));
              cur_g = g;
              cur_v = v;
            } else {
              cur_v = fmaxf(cur_v, v);
            }
          }
        }
      }
      if (cur_g >= 0 && cur_v > 0.f)
        atomicMax(reinterpret_cast<int*>(p.pooled + (int64_t)cur_g * p.ld_pool + n), __float_as_int(cur_v));
    }
  }
}

template <typename OpT, int EPI>
static int launch_gemm_t(const GemmParams& p, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<OpT, EPI>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    attr_done = true;
  }
  const int nwg = p.ntm * p.ntn;
  hipLaunchKernelGGL((gemm_kernel<OpT, EPI>), dim3(nwg), dim3(256), 4 * TILE_BYTES, st, p);
  NMTGAN_CHECK_LAUNCH("gemm");
  return 0;
}

int launch_gemm(const nmtgan_gemm_args* a, hipStream_t st) {
  prof::Scope prof_scope(prof::GEMM_PROJ, st);
  NMTGAN_REQUIRE(a != nullptr, "gemm: null args");
  NMTGAN_REQUIRE(a->op_dtype == NMTGAN_BF16 || a->op_dtype == NMTGAN_F32, "gemm: bad op_dtype %d", a->op_dtype);
  NMTGAN_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, "gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
  if (a->M == 0) return 0;
  const int es = dtype_size(a->op_dtype);
  NMTGAN_REQUIRE(a->A && a->Wt, "gemm: null operand");
  NMTGAN_REQUIRE(((int64_t)a->K * es) % 16 == 0, "gemm: K*esize must be a multiple of 16 bytes (K=%d)", a->K);
  NMTGAN_REQUIRE((a->lda * es) % 16 == 0, "gemm: lda*esize must be a multiple of 16 bytes");
  NMTGAN_REQUIRE(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->Wt % 16) == 0, "gemm: operands must be 16-byte aligned");
  NMTGAN_REQUIRE(a->a_group <= 0 || (a->a_group_stride * es) % 16 == 0, "gemm: a_group_stride misaligned");
  GemmParams p{};
  p.A = (const char*)a->A; p.W = (const char*)a->Wt;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.lda_b = a->lda * es;
  p.a_group = a->a_group > 0 ? a->a_group : 0;
  p.a_group_stride_b = a->a_group_stride * es;
  p.ldw_b = (int64_t)a->K * es;
  p.kbytes = a->K * es;
  p.ntm = (a->M + BM - 1) / BM;
  p.ntn = (a->N + BN - 1) / BN;
  if (a->epilogue == NMTGAN_EPI_LINEAR) {
    NMTGAN_REQUIRE(a->n_seg >= 1 && a->n_seg <= 3, "gemm: n_seg must be 1..3");
    NMTGAN_REQUIRE(a->n_seg == 1 || (a->seg_width > 0 && a->seg_width * a->n_seg >= a->N),
                   "gemm: segments do not cover N");
    p.bias = a->bias; p.relu = a->relu; p.n_seg = a->n_seg; p.seg_width = a->seg_width > 0 ? a->seg_width : a->N;
    for (int s = 0; s < 3; ++s) {
      p.out[s] = a->out[s]; p.ldo[s] = a->ldo[s]; p.out_f32[s] = a->out_f32[s];
      p.seg_scale[s] = a->seg_scale[s];
      if (s < a->n_seg) NMTGAN_REQUIRE(a->out[s] != nullptr, "gemm: null output segment %d", s);
    }
    return a->op_dtype == NMTGAN_BF16 ? launch_gemm_t<bf16_t, NMTGAN_EPI_LINEAR>(p, st)
                                      : launch_gemm_t<float, NMTGAN_EPI_LINEAR>(p, st);
  } else if (a->epilogue == NMTGAN_EPI_CONVMAX) {
    NMTGAN_REQUIRE(a->a_group > 0, "gemm: CONVMAX needs a_group (windows per sentence)");
    NMTGAN_REQUIRE(a->scale && a->shift && a->pooled, "gemm: CONVMAX needs scale/shift/pooled");
    p.scale = a->scale; p.shift = a->shift; p.pooled = a->pooled; p.ld_pool = a->ld_pool;
    return a->op_dtype == NMTGAN_BF16 ? launch_gemm_t<bf16_t, NMTGAN_EPI_CONVMAX>(p, st)
                                      : launch_gemm_t<float, NMTGAN_EPI_CONVMAX>(p, st);
  }
  NMTGAN_FAIL(-2, "gemm: unknown epilogue %d", a->epilogue);
}

}  // namespace nmtgan
