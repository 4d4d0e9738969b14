// MFMA GEMM for gfx950: C[M,N] = A[M,K] . Wt[N,K]^T with fused epilogues.
//
// Both operands are K-contiguous, so A and Wt tiles share one loader and one LDS image:
// rows of 128 BYTES per k-tile (64 bf16 or 32 fp32), 16-byte chunks XOR-swizzled by (row>>1)&7 so
// the ds_read_b128 fragment reads of a 32-row MFMA operand are conflict-free (bank = (addr/4)%64,
// 16-lane groups: cdna_hip_programming.md section 2).
// Wave tile 64x64 = 4x4 MFMA 16x16 tiles (the 16x16 shapes hold a higher clock under load than the 32x32 ones
// at equal cycles per flop):
//   bf16/fp16: v_mfma_f32_16x16x32_{bf16,f16}: lane l supplies 8 consecutive k (one 16-byte LDS chunk) of operand row
//         l & 15; the four 16-lane groups cover 32 k, so a 128-byte k-tile is two such k-halves
//   fp32: v_mfma_f32_16x16x4_f32, four per chunk (exact fp32, the 1e-5 parity tier)
// C layout of a 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + reg.
// The k order inside a tile is whatever the chunk order gives; A and B use the same order, so the
// sum is over the same k set (order differs from the reference's, as any GEMM's does).
//
// Two main loops:
//   gemm_glds_kernel  (K*esize % 128 == 0, every production shape): tiles go HBM/L2 -> LDS directly
//       with global_load_lds_dwordx4 (no staging registers, no ds_write).  One wave instruction
//       writes 1 KiB = 8 rows linearly, so the swizzle is applied to the per-lane SOURCE address
//       (guide rule 21).  Double-buffered, one barrier per k-tile; block tile BM x BN in {128,256}x128.
//   gemm_kernel       (any K with K*esize % 16 == 0): register-staged 128x128 fallback that zero
//       fills the k tail.
// Tile order: XCD-aware remap (blocks b and b+8 share an XCD / L2) + grouped-M order inside it so
// GROUP_M row panels of A stay L2-resident while the Wt tiles stream.
#include <stdlib.h>

#include <type_traits>

#include <algorithm>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace nmtgan {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int BKB = 128;                        // k-tile width in bytes
constexpr int GROUP_M = 16;

struct GemmParams {
  const char* A; const char* W;
  int M, N, K;
  int64_t lda_b;            // bytes
  int a_group; int64_t a_group_stride_b;
  int64_t ldw_b;            // bytes per Wt row (= K*esize)
  int kbytes;               // K * esize
  // k-tile visiting order (same for both operands): k_inner tiles k_step bytes apart, then k_jump to the next
  // run.  Plain GEMM: k_inner = 1, k_jump = 128.  Conv windows (K = w taps x E channels): the taps of one
  // 128-byte channel block are visited back to back, so consecutive k-tiles of A are the same rows of the
  // sentence image shifted by one token and hit in L2 instead of being re-fetched w times through the fabric.
  int k_inner; int64_t k_step, k_jump;
  int ntm, ntn, group_m;
  // linear epilogue
  const float* bias; int relu; int n_seg; int seg_width; int wide_ok;
  void* out[3]; int64_t ldo[3]; int out_f32[3]; float seg_scale[3];
  // convmax epilogue
  const float* scale; const float* shift; float* pooled; int64_t ld_pool;
  // stats epilogue (fused sampling)
  float2* stats;
  // split-K (gemm_glds_kernel, LINEAR epilogue): grid.y = split_k slices of the k-tiles, each leaves its raw fp32
  // partial in split_part[z][M][N]; splitk_finish_kernel adds the slices in order and applies the epilogue
  int split_k; float* split_part;
  // add + LayerNorm epilogue (gemm_ln_kernel)
  float* ln_x; const float* ln_gamma; const float* ln_beta; void* ln_out;
  // CONVMAX with an additive term (EPI_CONVADD, grouped kernel): the part of a window's sum that comes from a table
  // instead of the product - add[(add_blk[r / a_group] * add_rows + add_row0 + r % a_group) * N + n] is added to the
  // accumulator before scale / shift / ReLU / max
  const float* add; const int32_t* add_blk; int add_rows; int add_row0;
};
constexpr int EPI_STATS = 100;                  // internal: never a value of nmtgan_gemm_args.epilogue
constexpr int EPI_CONVADD = 101;                // internal: CONVMAX + additive term, chosen when nmtgan_gemm_args.add is set

// acc[I0 + i][J0 + j] += A-fragment i x B-fragment j for one k-half of a k-tile (MT x NT 16x16 tiles).  Consecutive
// MFMAs go to different accumulators, so the dependent-accumulator latency of the fp32 shape is never exposed.
template <typename OpT, int MT, int NT, int I0, int J0, int AM, int AN>
__device__ __forceinline__ void mma_block(const uint4 (&fa)[MT], const uint4 (&fb)[NT], f32x4 (&acc)[AM][AN]) {
  if constexpr (sizeof(OpT) == 4) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const unsigned au = t == 0 ? fa[i].x : (t == 1 ? fa[i].y : (t == 2 ? fa[i].z : fa[i].w));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const unsigned bu = t == 0 ? fb[j].x : (t == 1 ? fb[j].y : (t == 2 ? fb[j].z : fb[j].w));
          acc[I0 + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(au), __uint_as_float(bu),
                                                                     acc[I0 + i][J0 + j], 0, 0, 0);
        }
      }
  } else if constexpr (__is_same(OpT, bf16_t)) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[I0 + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                      __builtin_bit_cast(bf16x8, fb[j]), acc[I0 + i][J0 + j], 0, 0, 0);
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[I0 + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[i]),
                                                                     __builtin_bit_cast(f16x8, fb[j]), acc[I0 + i][J0 + j], 0, 0, 0);
  }
}

// element i of a 3-entry kernel-argument array WITHOUT dynamic indexing: an indexed access makes hipcc fetch
// the argument with a vector load, and the vmcnt wait it places (inside the conditional store blocks) leaks
// into the k-loops as a vmcnt(0) that drains the LDS-DMA pipeline every iteration
template <typename T>
__device__ __forceinline__ T pick3(const T (&a)[3], int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : a[2]); }

// bijective XCD remap + grouped-M order -> (tile_m, tile_n)
__device__ __forceinline__ void tile_index(const GemmParams& p, int& tm, int& tn) {
  const int nwg = p.ntm * p.ntn;
  const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int group_size = GROUP_M * p.ntn;
  const int first_m = (idx / group_size) * GROUP_M;
  const int gsz = min(p.ntm - first_m, GROUP_M);
  tm = first_m + (idx % group_size) % gsz;
  tn = (idx % group_size) / gsz;
}

__device__ __forceinline__ int64_t a_row_offset(const GemmParams& p, int r) {
  return (p.a_group > 0) ? (int64_t)(r / p.a_group) * p.a_group_stride_b + (int64_t)(r % p.a_group) * p.lda_b
                         : (int64_t)r * p.lda_b;
}

// Epilogue of one wave's (16 MT) x 64 tile.  16x16 C layout: col = lane & 15, row = 4 * (lane >> 4) + reg
// ev: optional LDS copy of the wave's 64 per-column epilogue values ([64] bias, or [64] scale + [64] shift),
// staged by LDS-DMA so that no compiler-visible vector load sits next to the hand-counted DMA stream
template <typename OpT, int EPI, int MT = 4>
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x4 (&acc)[MT][4], int m0, int n0, int lane,
                                         const float* ev = nullptr) {
  const int fc = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + j * 16 + fc;
    if (n >= p.N) continue;
    if constexpr (EPI == NMTGAN_EPI_LINEAR) {
      const float bias = ev ? ev[j * 16 + fc] : (p.bias ? p.bias[n] : 0.f);
      const int seg = (p.n_seg > 1) ? n / p.seg_width : 0;
      const int nn = (p.n_seg > 1) ? n - seg * p.seg_width : n;
      const float sc = pick3(p.seg_scale, seg);
      const int64_t ldo = pick3(p.ldo, seg);
      const int omode = pick3(p.out_f32, seg);   // 0 op dtype, 1 fp32, 2 split-precision operand image
      const bool of32 = omode == 1;
      const int64_t wimg = (p.n_seg > 1) ? p.seg_width : p.N;
      float* of = reinterpret_cast<float*>(pick3(p.out, seg));
      OpT* oo = reinterpret_cast<OpT*>(pick3(p.out, seg));
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + i * 16 + 4 * fq + e;
          if (m < p.M) {
            float v = (acc[i][j][e] + bias) * sc;
            if (p.relu) v = fmaxf(v, 0.f);
            if (of32) of[(int64_t)m * ldo + nn] = v;
            else if (omode == 2) {
              const float one[1] = {v};
              store_split3<1>(reinterpret_cast<f16_t*>(oo) + (int64_t)m * ldo + nn, wimg, one);
            } else OpTraits<OpT>::store(oo + (int64_t)m * ldo + nn, v);
          }
        }
      }
    } else {
      const float sc = ev ? ev[j * 16 + fc] : p.scale[n], sh = ev ? ev[64 + j * 16 + fc] : p.shift[n];
      int cur_g = -1;
      float cur_v = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + i * 16 + 4 * fq + e;
          if (m < p.M) {
            const int g = m / p.a_group;
            float a = acc[i][j][e];
            if constexpr (EPI == EPI_CONVADD)
              if (p.add) a += p.add[((int64_t)p.add_blk[g] * p.add_rows + p.add_row0 + (m - g * p.a_group)) * p.N + n];
            const float v = fmaxf(a * sc + sh, 0.f);
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

// one k-tile of MFMAs for a wave's 64x64 tile: A rows at a_rd, W rows at b_rd inside `base` (a_rd / b_rd already hold
// the lane's row, lane & 15); fq = lane >> 4 picks the lane's 16-byte chunk inside a k-half, fsw the row's swizzle
template <typename OpT>
__device__ __forceinline__ void compute_tile(const unsigned char* base, int a_rd, int b_rd, int fq, int fsw,
                                             f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const int pc = ((4 * kh + fq) ^ fsw) << 4;
    uint4 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i] = *reinterpret_cast<const uint4*>(base + a_rd + i * 16 * BKB + pc);
      fb[i] = *reinterpret_cast<const uint4*>(base + b_rd + i * 16 * BKB + pc);
    }
    mma_block<OpT, 4, 4, 0, 0>(fa, fb, acc);
  }
}

// ------------------------------------------------------------------------------------------ glds kernel
template <typename OpT, int EPI, int BM, int BN>
__global__ __launch_bounds__(64 * (BM / 64) * (BN / 64)) void gemm_glds_kernel(GemmParams p) {
  constexpr int NW = (BM / 64) * (BN / 64);
  constexpr int WN = BN / 64;
  constexpr int STAGE = (BM + BN) * BKB;          // bytes per LDS stage
  constexpr int NINST = (BM + BN) / 8 / NW;       // glds instructions per wave per k-tile (8 rows each)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  int tm, tn;
  tile_index(p, tm, tn);
  const int bm0 = tm * BM, bn0 = tn * BN;

  // staging: instruction j of this wave covers tile rows 8*(wave + j*NW) .. +7 of the stacked [A; W] image
  const char* src[NINST];
#pragma unroll
  for (int j = 0; j < NINST; ++j) {
    const int rho = 8 * (wave + j * NW) + (lane >> 3);
    const int lc = (lane & 7) ^ ((rho >> 1) & 7);             // logical chunk this lane fetches
    if (rho < BM) src[j] = p.A + a_row_offset(p, min(bm0 + rho, p.M - 1)) + lc * 16;
    else src[j] = p.W + (int64_t)min(bn0 + rho - BM, p.N - 1) * p.ldw_b + lc * 16;
  }
  const int nkt = p.kbytes / BKB;
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int j = 0; j < NINST; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[j] + (kt % p.k_inner) * p.k_step + (int64_t)(kt / p.k_inner) * BKB),
                                       (lds_ptr_t)(smem + buf * STAGE + (wave + j * NW) * 1024), 16, 0, 0);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fc = lane & 15, fq = lane >> 4;
  const int fsw = (fc >> 1) & 7;
  const int a_rd = (wm * 64 + fc) * BKB;
  const int b_rd = (BM + wn * 64 + fc) * BKB;

  const int nsl = p.split_k > 1 ? p.split_k : 1;
  const int kt0 = (int)((int64_t)nkt * blockIdx.y / nsl), kt1 = (int)((int64_t)nkt * (blockIdx.y + 1) / nsl);
  stage(0, kt0);
  int cur = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    __syncthreads();                      // vmcnt(0) + barrier: tile kt landed, buffer cur^1 free
    if (kt + 1 < kt1) stage(cur ^ 1, kt + 1);
    compute_tile<OpT>(smem + cur * STAGE, a_rd, b_rd, fq, fsw, acc);
    cur ^= 1;
  }
  if constexpr (EPI == NMTGAN_EPI_LINEAR) {
    if (p.split_k > 1) {                  // raw partial of this k-slice
      float* part = p.split_part + (int64_t)blockIdx.y * p.M * p.N;
      const int m0 = bm0 + wm * 64, n0 = bn0 + wn * 64;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + j * 16 + fc;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int m = m0 + i * 16 + 4 * fq + e;
            if (m < p.M) part[(int64_t)m * p.N + n] = acc[i][j][e];
          }
      }
      return;
    }
  }
  epilogue<OpT, EPI>(p, acc, bm0 + wm * 64, bn0 + wn * 64, lane);
}

// out = epilogue(sum_z split_part[z]) with the LINEAR epilogue's semantics (bias, segment scale, ReLU, per-segment
// destination and dtype); slices added in z order
template <typename OpT>
__global__ void splitk_finish_kernel(GemmParams p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)p.M * p.N) return;
  const int n = (int)(i % p.N);
  const int64_t m = i / p.N;
  float v = 0.f;
  for (int z = 0; z < p.split_k; ++z) v += p.split_part[(int64_t)z * p.M * p.N + i];
  const int seg = (p.n_seg > 1) ? n / p.seg_width : 0;
  const int nn = (p.n_seg > 1) ? n - seg * p.seg_width : n;
  v = (v + (p.bias ? p.bias[n] : 0.f)) * pick3(p.seg_scale, seg);
  if (p.relu) v = fmaxf(v, 0.f);
  const int64_t ldo = pick3(p.ldo, seg);
  const int omode = pick3(p.out_f32, seg);
  if (omode == 1) reinterpret_cast<float*>(pick3(p.out, seg))[m * ldo + nn] = v;
  else if (omode == 2) {
    const float one[1] = {v};
    store_split3<1>(reinterpret_cast<f16_t*>(pick3(p.out, seg)) + m * ldo + nn, (p.n_seg > 1) ? p.seg_width : p.N, one);
  } else OpTraits<OpT>::store(reinterpret_cast<OpT*>(pick3(p.out, seg)) + m * ldo + nn, v);
}

// ------------------------------------------------------------------------------------------ pipelined kernel
// Persistent 256x128 tile, 8 waves (2 per SIMD), 3 LDS stages of 48 KiB, one block per CU.
// The (tile, k-tile) pairs of a block form ONE stream: loads run two k-tiles ahead of the MFMAs and
// keep flowing across the epilogue into the next tile, so neither the per-tile prologue latency nor
// the per-k-tile load latency is exposed.  Per iteration: counted s_waitcnt vmcnt (leaves the newest
// stage in flight) -> raw s_barrier -> issue stage it+2 -> 16 MFMAs per wave from stage it.
// The LDS-DMA loads are issued from inline asm on purpose: hipcc then keeps no vmcnt bookkeeping for
// them (it would otherwise drain them with vmcnt(0) in front of every ds_read it cannot prove
// disjoint), and the hand-placed counted waits below are the only ones (guide section 5.7).
constexpr int PBM = 256, PBN = 128, PNW = 8, PSTAGE = (PBM + PBN) * BKB, PNINST = (PBM + PBN) / 8 / PNW;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// LINEAR epilogue with 16-byte stores: each 16-row x 32-column piece of the wave tile (two 16x16 accumulator tiles)
// is transposed through a 2 KiB per-wave LDS scratch so that a store instruction writes whole 128-byte (fp32) /
// 64-byte (16-bit) row segments: 16 (fp32) or 8 (16-bit) store instructions per 64x64 instead of 64 scalar ones.
// Needs N % 64 == 0, segment widths % 64 == 0 and 16-byte aligned output rows (checked at launch).
template <typename OpT, int MT = 4>
__device__ __forceinline__ void epilogue_wide(const GemmParams& p, f32x4 (&acc)[MT][4], int m0, int n0, int lane,
                                              float* scr, bool f32_out, int seg, int nn0, const float* ev = nullptr,
                                              bool split_out = false) {
  using O16 = std::conditional_t<sizeof(OpT) == 2, OpT, bf16_t>;     // 16-bit output type (never taken for fp32)
  const int fc = lane & 15, fq = lane >> 4;
  const float sc = pick3(p.seg_scale, seg);
  const int64_t ldo = pick3(p.ldo, seg);
  char* outp = reinterpret_cast<char*>(pick3(p.out, seg));
  float bias[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias[j] = ev ? ev[j * 16 + fc] : (p.bias ? p.bias[n0 + j * 16 + fc] : 0.f);
  if (!f32_out && p.wide_ok == 2) {
    // 16-bit outputs: pieces of 8 rows x all 64 columns of the wave tile, so that a store instruction writes whole
    // 128-byte row segments (8 lanes x 16 bytes per row) instead of 64-byte halves.  v_permlane32_swap regroups the
    // values first: column tiles j and j + 2 trade their lane halves, after which EVERY lane holds values of the
    // piece's 8 rows (lanes 0-31 columns 0-31, lanes 32-63 columns 32-63; row 4 ((lane >> 4) & 1) + e) and all 64
    // lanes write.  Row r keeps column c at c ^ 16 ((r >> 2) & 1): the 64 lanes of a write hit 64 different banks.
    const int64_t wimg = (p.n_seg > 1) ? p.seg_width : p.N;
    const int g2 = lane >> 5, fq1 = fq & 1;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      unsigned ph[2][2][4];                        // [half h][jj][e]
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = (acc[i][jj][e] + bias[jj]) * sc, b = (acc[i][jj + 2][e] + bias[jj + 2]) * sc;
          if (p.relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
          const u32x2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
          ph[0][jj][e] = sw[0];
          ph[1][jj][e] = sw[1];
        }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            scr[(4 * fq1 + e) * 64 + ((32 * g2 + 16 * jj + fc) ^ (fq1 << 4))] = __uint_as_float(ph[h][jj][e]);
        const int r = lane >> 3;
        float x[8];
        VecIO<float, 8>::load(scr + r * 64 + ((8 * (lane & 7)) ^ (((r >> 2) & 1) << 4)), x);
        const int m = m0 + i * 16 + 8 * h + r;
        if (m < p.M) {
          const int64_t off = (int64_t)m * ldo + nn0 + 8 * (lane & 7);
          if (split_out) store_split3<8>(reinterpret_cast<f16_t*>(outp) + off, wimg, x);
          else VecIO<O16, 8>::store(reinterpret_cast<O16*>(outp) + off, x);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = (acc[i][2 * jp + jj][e] + bias[2 * jp + jj]) * sc;
          if (p.relu) v = fmaxf(v, 0.f);
          scr[(4 * fq + e) * 32 + 16 * jj + fc] = v;
        }
      const int mb = m0 + i * 16;
      if (f32_out) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int r = (lane >> 3) + 8 * t;
          const float4 x = *reinterpret_cast<const float4*>(scr + r * 32 + 4 * (lane & 7));
          if (mb + r < p.M)
            *reinterpret_cast<float4*>(outp + ((int64_t)(mb + r) * ldo + nn0 + jp * 32 + 4 * (lane & 7)) * 4) = x;
        }
      } else {
        const int r = lane >> 2;
        float x[8];
        VecIO<float, 8>::load(scr + r * 32 + 8 * (lane & 3), x);
        if (mb + r < p.M) {
          const int64_t off = (int64_t)(mb + r) * ldo + nn0 + jp * 32 + 8 * (lane & 3);
          if (split_out) store_split3<8>(reinterpret_cast<f16_t*>(outp) + off, (p.n_seg > 1) ? p.seg_width : p.N, x);
          else VecIO<O16, 8>::store(reinterpret_cast<O16*>(outp) + off, x);
        }
      }
    }
  }
}

// STATS epilogue (fused sampling, logits never reach HBM).  The GEMM runs TRANSPOSED: C rows = vocab
// entries (A = softmax table), C columns = decoder rows (Wt = activations), so a decoder row is a
// lane column of the accumulator and its 64 logits of this wave tile sit in 4x4 registers of each of the
// four 16-lane groups: tile max and sum of exp(l - max) are in-lane reductions plus two cross-group
// exchanges.  stats[vtile][row] = (max, sum) with vtile = vocab/64; launch_logits_sample turns them
// into a sampled token (sampler.hip).
template <int MT = 4>
__device__ __forceinline__ void epilogue_stats(const GemmParams& p, f32x4 (&acc)[MT][4], int m0, int n0, int lane) {
  const int fc = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int ip = 0; ip < MT / 4; ++ip) {
    if (m0 + 64 * ip >= p.M) continue;             // vocab size is a multiple of 64: whole tile in or out
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float mx = -INFINITY;
#pragma unroll
      for (int i = 4 * ip; i < 4 * ip + 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, acc[i][j][e]);
      mx = groups4_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int i = 4 * ip; i < 4 * ip + 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += __builtin_amdgcn_exp2f((acc[i][j][e] - mx) * 1.44269504088896341f);
      sum = groups4_sum(sum);
      const int row = n0 + j * 16 + fc;
      if (fq == 0 && row < p.N) p.stats[(int64_t)((m0 >> 6) + ip) * p.N + row] = make_float2(mx, sum);
    }
  }
}

template <typename OpT, int EPI>
__global__ __launch_bounds__(512) void gemm_pipe_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = p.ntm * p.ntn;
  const int G = gridDim.x;
  const int nkt = p.kbytes / BKB;
  const int my_tiles = (nwg - (int)blockIdx.x + G - 1) / G;   // nkt >= 4 (launch guarantees it)

  // tile of linear id L (XCD remap needs G % 8 == 0 or G == nwg; launch guarantees it)
  auto tile_of = [&](int L, int& tm, int& tn) {
    const int xcd = L & 7, q = nwg >> 3, r = nwg & 7;
    const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    const int group_size = p.group_m * p.ntn;
    const int first_m = (idx / group_size) * p.group_m;
    const int gsz = min(p.ntm - first_m, p.group_m);
    tm = first_m + (idx % group_size) % gsz;
    tn = (idx % group_size) / gsz;
  };

  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);
  const char* src[PNINST];
  int kin = 0;                                   // position inside the run of k_inner k-tiles
  auto set_src = [&](int L) {
    int tm, tn;
    tile_of(L, tm, tn);
    kin = 0;
#pragma unroll
    for (int j = 0; j < PNINST; ++j) {
      const int rho = 8 * (wave + j * PNW) + (lane >> 3);
      const int lc = (lane & 7) ^ ((rho >> 1) & 7);
      if (rho < PBM) src[j] = p.A + a_row_offset(p, min(tm * PBM + rho, p.M - 1)) + lc * 16;
      else src[j] = p.W + (int64_t)min(tn * PBN + rho - PBM, p.N - 1) * p.ldw_b + lc * 16;
    }
  };
  int is_buf = 0;                                // LDS stage the next issue() fills
  auto issue = [&]() {                           // one k-tile of the tile `src` points at; advances src
    kin += 1;
    const bool run_end = kin == p.k_inner;
    const int64_t adv = run_end ? p.k_jump : p.k_step;
    if (run_end) kin = 0;
#pragma unroll
    for (int j = 0; j < PNINST; ++j) {
      glds16(src[j], lds_base + is_buf * PSTAGE + (wave + j * PNW) * 1024);
      src[j] += adv;
    }
    is_buf = is_buf == 2 ? 0 : is_buf + 1;
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fc = lane & 15, fq = lane >> 4;
  const int fsw = (fc >> 1) & 7;
  const int a_rd = (wm * 64 + fc) * BKB;
  const int b_rd = (PBM + wn * 64 + fc) * BKB;

  float* scr = reinterpret_cast<float*>(smem + 3 * PSTAGE + wave * 2048);
  // vmcnt counts loads AND stores in issue order.  `pend` = epilogue stores this wave issued after its
  // newest LDS-DMA group; they are younger than the next two stages waited for, so those two waits
  // let them stay in flight and a tile's output drains under the next tile's MFMAs.
  int pend = 0, cbuf = 0;
#define NMTGAN_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  auto step = [&](bool do_issue) {               // after the wait: barrier, prefetch, 16 MFMAs per wave
    __builtin_amdgcn_s_barrier();
    if (do_issue) issue();                       // fills the stage read one iteration ago (all waves are past it)
    compute_tile<OpT>(smem + cbuf * PSTAGE, a_rd, b_rd, fq, fsw, acc);
    cbuf = cbuf == 2 ? 0 : cbuf + 1;
  };
  set_src((int)blockIdx.x);
  issue();
  issue();
  for (int t = 0; t < my_tiles; ++t) {
    const bool has_next = t + 1 < my_tiles;
    for (int kt = 0; kt < nkt - 2; ++kt) {       // k-tile kt landed, kt+1 in flight, prefetch kt+2
      if (kt < 2 && pend == 24) NMTGAN_WAIT(PNINST + 24);
      else if (kt < 2 && pend == 16) NMTGAN_WAIT(PNINST + 16);
      else if (kt < 2 && pend == 8) NMTGAN_WAIT(PNINST + 8);
      else NMTGAN_WAIT(PNINST);
      step(true);
    }
    pend = 0;
    if (has_next) set_src((int)blockIdx.x + (t + 1) * G);
    NMTGAN_WAIT(PNINST);                         // k-tile nkt-2: prefetch the next tile's k-tile 0
    step(has_next);
    if (has_next) NMTGAN_WAIT(PNINST);           // k-tile nkt-1: prefetch the next tile's k-tile 1
    else NMTGAN_WAIT(0);
    step(has_next);

    int tm, tn;
    tile_of((int)blockIdx.x + t * G, tm, tn);
    const int m0 = tm * PBM + wm * 64, n0 = tn * PBN + wn * 64;
    bool wide = false;
    if constexpr (EPI == NMTGAN_EPI_LINEAR) wide = p.wide_ok && n0 < p.N;
    if constexpr (EPI == EPI_STATS) {
      epilogue_stats(p, acc, m0, n0, lane);
    } else if (wide) {
      const int seg = (p.n_seg > 1) ? n0 / p.seg_width : 0;
      const int omode = pick3(p.out_f32, seg);
      const bool f32_out = omode == 1 || sizeof(OpT) == 4;
      epilogue_wide<OpT>(p, acc, m0, n0, lane, scr, f32_out, seg, n0 - seg * p.seg_width, nullptr, omode == 2);
      if (tm * PBM + PBM <= p.M) pend = f32_out ? 16 : (omode == 2 ? 24 : 8);      // every store instruction was issued
    } else {
      epilogue<OpT, EPI>(p, acc, m0, n0, lane);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#undef NMTGAN_WAIT
}

static thread_local int g_wg_cap = 0;
void set_gemm_wg_cap(int cap) { g_wg_cap = cap > 0 ? (cap + 7) / 8 * 8 : 0; }   // multiple of 8: one share per XCD
static inline int persistent_grid(int nwg) {
  const int cus = g_wg_cap > 0 ? g_wg_cap : 256;      // a cap above 256 = more workgroups than CUs: with one tile per
                                                      // workgroup a CU is released after every tile
  return nwg < cus ? nwg : cus;
}

template <typename OpT, int EPI>
static int launch_pipe(GemmParams p, hipStream_t st) {
  constexpr int smem = 3 * PSTAGE + PNW * 2048;  // 3 stages + per-wave epilogue scratch = 160 KiB
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_kernel<OpT, EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      NMTGAN_FAIL(-3, "gemm: cannot raise dynamic LDS limit");
    attr_done = true;
  }
  p.ntm = (p.M + PBM - 1) / PBM;
  p.ntn = (p.N + PBN - 1) / PBN;
  p.group_m = 2;                                 // tile order knob (L2 sharing of A panels vs Wt tiles); 1-4 measured best
  if (const char* e = getenv("NMTGAN_GEMM_GM")) p.group_m = atoi(e);
  if constexpr (EPI == NMTGAN_EPI_LINEAR) {
    const int es = sizeof(OpT);
    bool ok = p.N % 64 == 0 && (p.n_seg == 1 || p.seg_width % 64 == 0);
    for (int s = 0; s < p.n_seg; ++s) {
      const int oes = p.out_f32[s] == 1 ? 4 : (p.out_f32[s] == 2 ? 2 : es);
      ok = ok && ((uintptr_t)p.out[s] % 16 == 0) && ((p.ldo[s] * oes) % 16 == 0);
    }
    p.wide_ok = ok ? 2 : 0;
    if (const char* e = getenv("NMTGAN_GEMM_WIDE")) p.wide_ok = p.wide_ok ? atoi(e) : 0;
  }
  const int nwg = p.ntm * p.ntn;
  const int grid = persistent_grid(nwg);         // one block per CU; the tile walk is bijective for any grid
  int smem_use = smem;
  if (const char* e = getenv("NMTGAN_GEMM_SMEM")) smem_use = atoi(e);
  hipLaunchKernelGGL((gemm_pipe_kernel<OpT, EPI>), dim3(grid), dim3(512), smem_use, st, p);
  NMTGAN_CHECK_LAUNCH("gemm_pipe");
  return 0;
}

// ------------------------------------------------------------------------------------------ grouped kernel
// ONE persistent launch over the tiles of up to GROUP_MAX independent GEMM problems (the discriminator's per-width,
// per-row-range, per-tap-block convolutions, cnn_discriminator.py:461-477: most of them have fewer tiles than the chip
// has CUs, or a ragged last round of tiles, and launched one by one they leave CUs idle).  Same 256x128 tile, LDS
// image, three-stage LDS-DMA stream and counted waits as gemm_pipe_kernel; the stream of (tile, k-tile) pairs of a
// block now runs across problem boundaries: tile L of the launch belongs to the problem whose tile range holds L
// (problems ordered longest k-loop first by the host, so the long tiles start first), and everything a tile needs -
// operand pointers, shape, k-tile count and visiting order, epilogue vectors - comes from the problem's entry of the
// kernel-argument table.  Per output element the k order is the one of the single-problem kernels, so the bits are
// the same whichever kernel computes a problem.
constexpr int GROUP_MAX = 24;
struct GroupProb {
  const char* A; const char* W;
  void* out;                                    // CONVMAX / CONVADD: pooled features; LINEAR: output (fp32 or op dtype)
  const float* v0; const float* v1;             // CONVMAX / CONVADD: scale, shift; LINEAR: bias (may be null), unused
  const float* add; const int32_t* add_blk;     // CONVADD
  int64_t a_group_stride_b, ldw_b, ld_out, k_step, k_jump;
  int M, N, nkt, a_group, lda_b, k_inner, ntn, ntm, add_rows, add_row0, relu, omode;
};
struct GroupParams {
  int n_prob, n_tiles, group_m;
  int tile_end[GROUP_MAX];                      // cumulative tile counts (ascending)
  GroupProb prob[GROUP_MAX];
};

static_assert(sizeof(GroupParams) <= 4096, "the problem table travels as kernel arguments (4 KB)");

template <typename OpT, int EPI>
__global__ __launch_bounds__(512) void gemm_group_kernel(GroupParams gp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int G = gridDim.x;
  const int my_tiles = (gp.n_tiles - (int)blockIdx.x + G - 1) / G;

  // problem and tile coordinates of launch tile L (column tiles fastest: neighbouring workgroups share an A panel)
  auto locate = [&](int L, int& q, int& tm, int& tn) {
    int qq = 0;
    for (int i = 0; i + 1 < gp.n_prob; ++i) qq += L >= gp.tile_end[i] ? 1 : 0;
    qq = __builtin_amdgcn_readfirstlane(qq);
    const int first = qq > 0 ? gp.tile_end[qq - 1] : 0;
    const int ntn = gp.prob[qq].ntn;
    q = qq;
    // XCD-aware order inside the problem: workgroup b runs on XCD b & 7 and the grid is a multiple of 8 (or one tile
    // per workgroup), so the problem's tiles with equal (L - first) & 7 share an XCD; they get one contiguous range of
    // the problem's tiles, column tiles fastest - the column tiles of an A panel (up to 12.8 MB of window rows at
    // w = 49) run side by side on ONE L2 instead of being fetched by five of them
    const int n = gp.tile_end[qq] - first, i = L - first;
    const int o = i & 7, qd = n >> 3, r = n & 7;
    const int idx = (o < r ? o * (qd + 1) : r * (qd + 1) + (o - r) * qd) + (i >> 3);
    // ... and inside the range, groups of group_m row tiles x all column tiles, row tile fastest: the ~32 tiles an
    // XCD works on at a time are group_m A panels x ntn W tiles, so every W tile streamed from memory serves group_m
    // workgroups (a problem's filter bank, up to 30 MB, does not stay in a 4 MB L2 between tiles)
    const int ntm = gp.prob[qq].ntm, gm = gp.group_m;
    const int gsize = gm * ntn;
    const int first_m = (idx / gsize) * gm;
    const int gsz = min(ntm - first_m, gm);
    const int rem = idx - (idx / gsize) * gsize;
    tm = first_m + rem % gsz;
    tn = rem / gsz;
  };

  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);
  const char* src[PNINST];
  int kin = 0, s_inner = 1;                      // k-tile visiting order of the problem the DMA stream is in
  int64_t s_step = BKB, s_jump = BKB;
  auto set_src = [&](int L) {
    int q, tm, tn;
    locate(L, q, tm, tn);
    const GroupProb& pr = gp.prob[q];
    kin = 0;
    s_inner = pr.k_inner; s_step = pr.k_step; s_jump = pr.k_jump;
    const int a_group = pr.a_group, M = pr.M, N = pr.N;
    const int64_t lda_b = pr.lda_b, gs_b = pr.a_group_stride_b, ldw_b = pr.ldw_b;
    const char* A = pr.A;
    const char* W = pr.W;
#pragma unroll
    for (int j = 0; j < PNINST; ++j) {
      const int rho = 8 * (wave + j * PNW) + (lane >> 3);
      const int lc = (lane & 7) ^ ((rho >> 1) & 7);
      if (rho < PBM) {
        const int r = min(tm * PBM + rho, M - 1);
        src[j] = A + (a_group > 0 ? (int64_t)(r / a_group) * gs_b + (int64_t)(r % a_group) * lda_b : (int64_t)r * lda_b) + lc * 16;
      } else {
        src[j] = W + (int64_t)min(tn * PBN + rho - PBM, N - 1) * ldw_b + lc * 16;
      }
    }
  };
  int is_buf = 0;
  auto issue = [&]() {
    kin += 1;
    const bool run_end = kin == s_inner;
    const int64_t adv = run_end ? s_jump : s_step;
    if (run_end) kin = 0;
#pragma unroll
    for (int j = 0; j < PNINST; ++j) {
      glds16(src[j], lds_base + is_buf * PSTAGE + (wave + j * PNW) * 1024);
      src[j] += adv;
    }
    is_buf = is_buf == 2 ? 0 : is_buf + 1;
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fc = lane & 15, fq = lane >> 4;
  const int fsw = (fc >> 1) & 7;
  const int a_rd = (wm * 64 + fc) * BKB;
  const int b_rd = (PBM + wn * 64 + fc) * BKB;
  int cbuf = 0;
#define NMTGAN_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  auto step = [&](bool do_issue) {
    __builtin_amdgcn_s_barrier();
    if (do_issue) issue();
    compute_tile<OpT>(smem + cbuf * PSTAGE, a_rd, b_rd, fq, fsw, acc);
    cbuf = cbuf == 2 ? 0 : cbuf + 1;
  };
  if (my_tiles <= 0) return;
  set_src((int)blockIdx.x);
  issue();
  issue();
  for (int t = 0; t < my_tiles; ++t) {
    const bool has_next = t + 1 < my_tiles;
    int q, tm, tn;
    locate((int)blockIdx.x + t * G, q, tm, tn);
    const int nkt = gp.prob[q].nkt;              // >= 2 (the host checks >= 4)
    for (int kt = 0; kt < nkt - 2; ++kt) {       // k-tile kt landed, kt+1 in flight, prefetch kt+2
      NMTGAN_WAIT(PNINST);
      step(true);
    }
    if (has_next) set_src((int)blockIdx.x + (t + 1) * G);
    NMTGAN_WAIT(PNINST);                         // k-tile nkt-2: prefetch the next tile's k-tile 0
    step(has_next);
    if (has_next) NMTGAN_WAIT(PNINST);           // k-tile nkt-1: prefetch the next tile's k-tile 1
    else NMTGAN_WAIT(0);
    step(has_next);

    const GroupProb& pr = gp.prob[q];
    GemmParams v{};                              // the epilogue's view of this tile's problem
    v.M = pr.M; v.N = pr.N; v.a_group = pr.a_group;
    if constexpr (EPI == NMTGAN_EPI_LINEAR) {
      v.bias = pr.v0; v.relu = pr.relu; v.n_seg = 1; v.seg_width = pr.N;
      v.out[0] = pr.out; v.ldo[0] = pr.ld_out; v.out_f32[0] = pr.omode; v.seg_scale[0] = 1.f;
    } else {
      v.scale = pr.v0; v.shift = pr.v1; v.pooled = reinterpret_cast<float*>(pr.out); v.ld_pool = pr.ld_out;
      v.add = pr.add; v.add_blk = pr.add_blk; v.add_rows = pr.add_rows; v.add_row0 = pr.add_row0;
    }
    epilogue<OpT, EPI>(v, acc, tm * PBM + wm * 64, tn * PBN + wn * 64, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#undef NMTGAN_WAIT
}

template <typename OpT, int EPI>
static int launch_group(const GroupParams& gp, hipStream_t st) {
  constexpr int smem = 3 * PSTAGE;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_group_kernel<OpT, EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      NMTGAN_FAIL(-3, "gemm: cannot raise dynamic LDS limit");
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_group_kernel<OpT, EPI>), dim3(persistent_grid(gp.n_tiles)), dim3(512), smem, st, gp);
  NMTGAN_CHECK_LAUNCH("gemm_group");
  return 0;
}

// ------------------------------------------------------------------------------------------ 256x256 kernel
// Persistent 256x256 tile, 8 waves as 2 (M) x 4 (N), wave tile 128x64 = 8x4 MFMA 16x16 tiles: per k-tile a wave
// reads 24 KiB of LDS for 64 MFMAs (the 64x64 wave tile of gemm_pipe_kernel reads 16 KiB for 32, i.e. the
// LDS read pipe is as busy as the matrix cores), and the block fetches 64 KiB per 8.4 MFLOP (48 per 4.2).
// The k-tile is cut into four 16 KiB staging units - A0/A1 = the first/second 64 rows of every wave's
// 128, B0/B1 = the first/second 32 columns of every wave's 64 - held twice (k parity): 128 KiB.
// A k-step runs four phases, one C quadrant each:
//     P1 (A0,B0)   P2 (A0,B1)   P3 (A1,B1)   P4 (A1,B0)
// phase = [ds_read the operands this quadrant still lacks; issue one unit of LDS-DMA; counted vmcnt]
//         s_barrier [16 MFMAs at raised priority] s_barrier
// and the two wave groups (waves 0-3 / 4-7 = one wave per SIMD each) run ONE barrier apart, so on every
// SIMD one wave feeds the matrix core while the other does its LDS reads and DMA issue.
// Staging order A0(s+2)@P3 B0(s+2)@P4 B1(s+1)@P1 A1(s+1)@P2 keeps every unit >= 2 phases clear of its last
// read (WAR) and gives it >= 5 phases to land; vmcnt(8) = the four youngest units may still be in flight,
// placed in the phase BEFORE the one that reads the awaited unit (the other group reads it one barrier
// later; guide section 5, "Read a staged buffer one phase AFTER the wait that retires it").
constexpr int QBM = 256, QBN = 256, QUNIT = 128 * BKB, QSTAGE = 4 * QUNIT;

// LDS-DMA without the m0 save/restore of glds16: m0 is declared clobbered instead (gfx950 DS instructions do
// not read m0, so the compiler keeps nothing live in it around these)
__device__ __forceinline__ void glds16_m0(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}

// same with a uniform 64-bit base in scalar registers and a 32-bit per-lane byte offset
__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory", "m0");
}

template <typename OpT, int EPI>
__global__ __launch_bounds__(512) void gemm_p8_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nwg = p.ntm * p.ntn;
  const int G = gridDim.x;
  const int nkt = p.kbytes / BKB;                // >= 4 (launch guarantees it)
  const int my_tiles = (nwg - (int)blockIdx.x + G - 1) / G;

  auto tile_of = [&](int L, int& tm, int& tn) {
    const int xcd = L & 7, q = nwg >> 3, r = nwg & 7;
    const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    const int group_size = p.group_m * p.ntn;
    const int first_m = (idx / group_size) * p.group_m;
    const int gsz = min(p.ntm - first_m, p.group_m);
    tm = first_m + (idx % group_size) % gsz;
    tn = (idx % group_size) / gsz;
  };
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);

  // ---- staging: unit u (0 A0, 1 A1, 2 B0, 3 B1), two DMA instructions per wave and unit
  using U0 = std::integral_constant<int, 0>;
  using U1 = std::integral_constant<int, 1>;
  using U2 = std::integral_constant<int, 2>;
  using U3 = std::integral_constant<int, 3>;
  const char* src[4][2];
  int kin[4];                                    // position of unit u inside its run of k_inner k-tiles
  auto point = [&](auto U, int ord) {            // unit U's sources at k-tile 0 of the block's tile `ord`
    constexpr int u = decltype(U)::value;
    kin[u] = 0;
    int tm, tn;
    tile_of((int)blockIdx.x + ord * G, tm, tn);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rho = 8 * (wave + 8 * j) + (lane >> 3);
      const int lc = (lane & 7) ^ ((rho >> 1) & 7);
      if constexpr (u < 2) {
        const int row = tm * QBM + (rho >> 6) * 128 + u * 64 + (rho & 63);
        src[u][j] = p.A + a_row_offset(p, min(row, p.M - 1)) + lc * 16;
      } else {
        const int col = tn * QBN + (rho >> 5) * 64 + (u - 2) * 32 + (rho & 31);
        src[u][j] = p.W + (int64_t)min(col, p.N - 1) * p.ldw_b + lc * 16;
      }
    }
  };
  auto issue = [&](auto U, int parity) {         // next k-tile of unit U into the stage of that parity
    constexpr int u = decltype(U)::value;
    const unsigned dst = lds_base + parity * QSTAGE + u * QUNIT + wave * 1024;
    glds16_m0(src[u][0], dst);
    glds16_m0(src[u][1], dst + 8 * 1024);
    kin[u] += 1;
    const bool run_end = kin[u] == p.k_inner;
    const int64_t adv = run_end ? p.k_jump : p.k_step;
    if (run_end) kin[u] = 0;
    src[u][0] += adv;
    src[u][1] += adv;
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fc = lane & 15, fq = lane >> 4;
  const int fsw = (fc >> 1) & 7;
  int pc[2];                                     // the lane's chunk in the two k-halves of a k-tile
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) pc[kh] = ((4 * kh + fq) ^ fsw) << 4;
  const int a_rd = (wr * 64 + fc) * BKB;         // + h * QUNIT + i * 16 * BKB
  const int b_rd = 2 * QUNIT + (wc * 32 + fc) * BKB;   // + h * QUNIT + j * 16 * BKB
  float* scr = reinterpret_cast<float*>(smem + 2 * QSTAGE + wave * 2048);
  // per-column epilogue values of the wave's 64 columns, LDS-DMA'd at the start of the tile's last k-step:
  // they ride the same in-order vmcnt stream as the tile DMAs and the epilogue needs no vector load (which
  // the compiler would wait vmcnt(0) for, draining the prefetch of the next tile)
  constexpr int QEV = 2 * QSTAGE + 8 * 2048;
  const float* ev = reinterpret_cast<const float*>(smem + QEV + wave * 512);
  const bool use_ev = (EPI == NMTGAN_EPI_LINEAR && p.bias != nullptr) || EPI == NMTGAN_EPI_CONVMAX;
  auto stage_ev = [&](int ord) {
    int tm, tn;
    tile_of((int)blockIdx.x + ord * G, tm, tn);
    const int n = min(tn * QBN + wc * 64 + lane, p.N - 1);
    const unsigned dst = lds_base + QEV + wave * 512;
    if constexpr (EPI == NMTGAN_EPI_LINEAR) {
      if (p.bias) glds4(p.bias + n, dst);
    } else if constexpr (EPI == NMTGAN_EPI_CONVMAX) {
      glds4(p.scale + n, dst);
      glds4(p.shift + n, dst + 256);
    }
  };

  uint4 fa[2][4], fb0[2][2], fb1[2][2];          // [k-half][16-row block]
#define NMTGAN_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define NMTGAN_BAR()                       \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_s_barrier();          \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
  auto read_a = [&](const unsigned char* base, int h) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        fa[kh][i] = *reinterpret_cast<const uint4*>(base + h * QUNIT + a_rd + i * 16 * BKB + pc[kh]);
  };
  auto read_b = [&](const unsigned char* base, int h, uint4 (&fb)[2][2]) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fb[kh][j] = *reinterpret_cast<const uint4*>(base + h * QUNIT + b_rd + j * 16 * BKB + pc[kh]);
  };
  auto mfma8 = [&](auto H, auto HB, const uint4 (&fb)[2][2]) {      // one C quadrant: 16 MFMAs of 16x16x32
    constexpr int h = decltype(H)::value, hb = decltype(HB)::value;
    NMTGAN_BAR();
    __builtin_amdgcn_s_setprio(1);
    mma_block<OpT, 4, 2, 4 * h, 2 * hb>(fa[0], fb[0], acc);
    mma_block<OpT, 4, 2, 4 * h, 2 * hb>(fa[1], fb[1], acc);
    __builtin_amdgcn_s_setprio(0);
    NMTGAN_BAR();
  };

  int par = 0;                                   // LDS stage of the current k-step
  int t = 0;
  // one k-step.  MODE 0: its stagings stay inside the tile; 1: k-tile nkt-2 (A0, B0 go to the next tile);
  // 2: k-tile nkt-1 (all four go to the next tile).  Without a next tile the cross-tile stagings are
  // skipped and the waits shrink to the units that are still in flight.
  auto kstep = [&](auto MODE, bool has_next) {
    constexpr int mode = decltype(MODE)::value;
    const unsigned char* base = smem + par * QSTAGE;
    if constexpr (mode == 2) stage_ev(t);
    // P1: quadrant (A0, B0); stage B1 of the next k-step
    read_b(base, 0, fb0);
    read_a(base, 0);
    if constexpr (mode == 2) {
      if (has_next) { point(U3{}, t + 1); issue(U3{}, par ^ 1); }
      if (has_next) NMTGAN_WAIT(8); else NMTGAN_WAIT(2);
    } else {
      issue(U3{}, par ^ 1);
      NMTGAN_WAIT(8);                            // B1 of this k-step landed
    }
    mfma8(U0{}, U0{}, fb0);
    // P2: quadrant (A0, B1); stage A1 of the next k-step
    read_b(base, 1, fb1);
    if constexpr (mode == 2) {
      if (has_next) { point(U1{}, t + 1); issue(U1{}, par ^ 1); }
      if (has_next) NMTGAN_WAIT(8); else NMTGAN_WAIT(0);
    } else {
      issue(U1{}, par ^ 1);
      NMTGAN_WAIT(8);                            // A1 of this k-step landed
    }
    mfma8(U0{}, U1{}, fb1);
    // P3: quadrant (A1, B1); stage A0 two k-steps ahead
    read_a(base, 1);
    if constexpr (mode == 0) {
      issue(U0{}, par);
    } else if (has_next) {
      if constexpr (mode == 1) point(U0{}, t + 1);
      issue(U0{}, par);
    }
    mfma8(U1{}, U1{}, fb1);
    // P4: quadrant (A1, B0); stage B0 two k-steps ahead
    if constexpr (mode == 0) {
      issue(U2{}, par);
      NMTGAN_WAIT(8);                            // A0, B0 of the next k-step landed
    } else {
      if (has_next) {
        if constexpr (mode == 1) point(U2{}, t + 1);
        issue(U2{}, par);
        NMTGAN_WAIT(8);
      } else if constexpr (mode == 1) {
        NMTGAN_WAIT(4);
      }
    }
    mfma8(U1{}, U0{}, fb0);
    par ^= 1;
  };
  using M0 = std::integral_constant<int, 0>;
  using M1 = std::integral_constant<int, 1>;
  using M2 = std::integral_constant<int, 2>;

  // prologue: A0 B0 B1 A1 of k-step 0, A0 B0 of k-step 1
  point(U0{}, 0); point(U1{}, 0); point(U2{}, 0); point(U3{}, 0);
  issue(U0{}, 0); issue(U2{}, 0); issue(U3{}, 0); issue(U1{}, 0); issue(U0{}, 1); issue(U2{}, 1);
  NMTGAN_WAIT(8);
  NMTGAN_BAR();
  if (wr == 1) NMTGAN_BAR();                     // second wave group runs one barrier behind
  for (; t < my_tiles; ++t) {
    const bool has_next = t + 1 < my_tiles;
    for (int kt = 0; kt < nkt - 2; ++kt) kstep(M0{}, true);
    kstep(M1{}, has_next);
    kstep(M2{}, has_next);
    int tm, tn;
    tile_of((int)blockIdx.x + t * G, tm, tn);
    const int m0 = tm * QBM + wr * 128, n0 = tn * QBN + wc * 64;
    bool wide = false;
    if constexpr (EPI == NMTGAN_EPI_LINEAR) wide = p.wide_ok && n0 < p.N;
    if constexpr (EPI == EPI_STATS) {
      epilogue_stats<8>(p, acc, m0, n0, lane);
    } else if (wide) {
      const int seg = (p.n_seg > 1) ? n0 / p.seg_width : 0;
      const int omode = pick3(p.out_f32, seg);
      const bool f32_out = omode == 1 || sizeof(OpT) == 4;
      epilogue_wide<OpT, 8>(p, acc, m0, n0, lane, scr, f32_out, seg, n0 - seg * p.seg_width, use_ev ? ev : nullptr,
                            omode == 2);
    } else {
      epilogue<OpT, EPI, 8>(p, acc, m0, n0, lane, use_ev ? ev : nullptr);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (wr == 0) NMTGAN_BAR();                     // balance the barrier count of the two groups
#undef NMTGAN_WAIT
#undef NMTGAN_BAR
}

template <typename OpT, int EPI>
static int launch_p8(GemmParams p, hipStream_t st) {
  constexpr int smem = 2 * QSTAGE + 8 * 2048 + 8 * 512;   // 128 KiB of staging + per-wave epilogue scratch + vectors
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_kernel<OpT, EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      NMTGAN_FAIL(-3, "gemm: cannot raise dynamic LDS limit");
    attr_done = true;
  }
  p.ntm = (p.M + QBM - 1) / QBM;
  p.ntn = (p.N + QBN - 1) / QBN;
  // tile groups of 2 row tiles (4 are as fast); the logits GEMM (rows = vocabulary) takes 8: an XCD then keeps 2 MiB
  // of the table in its L2 and sweeps the activations twice instead of eight times - same time, a third of the
  // fetches past the L2
  p.group_m = EPI == EPI_STATS ? 8 : 2;
  if (const char* e = getenv("NMTGAN_GEMM_GM")) p.group_m = atoi(e);
  if constexpr (EPI == NMTGAN_EPI_LINEAR) {
    const int es = sizeof(OpT);
    bool ok = p.N % 64 == 0 && (p.n_seg == 1 || p.seg_width % 64 == 0);
    for (int s = 0; s < p.n_seg; ++s) {
      const int oes = p.out_f32[s] == 1 ? 4 : (p.out_f32[s] == 2 ? 2 : es);
      ok = ok && ((uintptr_t)p.out[s] % 16 == 0) && ((p.ldo[s] * oes) % 16 == 0);
    }
    p.wide_ok = ok ? 2 : 0;
    if (const char* e = getenv("NMTGAN_GEMM_WIDE")) p.wide_ok = p.wide_ok ? atoi(e) : 0;
  }
  const int nwg = p.ntm * p.ntn;
  const int grid = persistent_grid(nwg);
  hipLaunchKernelGGL((gemm_p8_kernel<OpT, EPI>), dim3(grid), dim3(512), smem, st, p);
  NMTGAN_CHECK_LAUNCH("gemm_p8");
  return 0;
}

// ------------------------------------------------------------------------------------------ full-row kernel
// x <- LayerNorm(x + A . Wt^T + bias) * gamma + beta for N = 512 and 16-bit operands: the residual add and the
// normalisation of model.py:725-739 in the epilogue of the projection that feeds them, so the projection's fp32
// output never goes to HBM and comes back (12 bytes per element of x instead of 20 for GEMM + add_layernorm).
// A block tile spans whole rows: BM x 512, 8 waves side by side, wave w owns columns 64 w .. 64 w + 63 of all BM
// rows (MT = BM / 16 row blocks x 4 column blocks of 16x16).  The column -> wave -> lane mapping and the order
// of every sum do not depend on BM or on the tile a row falls in: a row's result is a function of that row alone.
// k-tiles are 64 BYTES wide (one 16x16x32 MFMA k-step) so that three stages of the 576 / 640-row [A; W] image
// fit.  A DMA instruction writes 16 rows linearly; row r keeps logical 16-byte chunk c at physical chunk
// c ^ g((r >> 2) & 3) with g = (0, 3, 2, 1).  A ds_read_b128 is served in four groups of 16 lanes that are NOT
// contiguous - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 - so a group holds rows 0-3 and 12-15 at
// one k-chunk and rows 4-11 at the next; r & 3 picks the 64-byte quarter of the 256-byte bank line, and g makes the
// four rows that share a quarter land on four different 16-byte slots in every group.
// Same stream structure as gemm_pipe_kernel: persistent, loads two k-tiles ahead, counted vmcnt waits.
constexpr int LBN = 512, LKB = 64, LSCR = 16 * 36 * 4;

template <typename F, int... Hs>
__device__ __forceinline__ void for_each_const(F&& f, std::integer_sequence<int, Hs...>) {
  (f(std::integral_constant<int, Hs>{}), ...);
}

// NW waves side by side: wave w owns columns WC w .. WC w + WC - 1 (WC = 512 / NW) of all BM rows; instantiated as
// <64, 8> (two waves per SIMD, 64 x 64 wave tiles, 16 MFMAs per wave and k-step).
template <typename OpT, int BM, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_ln_kernel(GemmParams p) {
  constexpr int WC = LBN / NW, MT = BM / 16, NT = WC / 16, NJP = NT / 2, ROWS = BM + LBN;
  // a stage holds NUNIT DMA units of 16 rows (1 KiB each); wave w issues units w, w + NW, ...: NI of them, or NI - 1
  // when the unit count is not a multiple of the waves (`full` is then false for the last waves)
  constexpr int NST = 4, NUNIT = ROWS / 16, NI = (NUNIT + NW - 1) / NW, LSTAGE = NUNIT * 1024;
  constexpr int NA = (BM / 16 + NW - 1) / NW;    // units j < NA of a wave can be A units
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int nkt = p.kbytes / LKB;                // >= 4 and even (launch guarantees it)
  const int my_tiles = (p.ntm - (int)blockIdx.x + G - 1) / G;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);
  const bool full = (NUNIT % NW == 0) || wave + NW * (NI - 1) < NUNIT;

  // DMA sources = uniform base (A tile or W, advanced along k by scalar adds) + a 32-bit per-lane offset: unit
  // u = wave + NW j of the stacked [A; W] image holds rows 16 u .. 16 u + 15.  The offsets of A units change with
  // the tile (rows past M re-read row M - 1), the W offsets never do.
  // Everything derived from the lane id (DMA offsets, fragment read offsets) is recomputed at the start of every tile
  // from a value the optimiser cannot see through: the values then do not live across the epilogue, where the
  // register allocator would spill them and reload them inside the k-loop.
  int lc0 = 0, a_rd = 0, b_rd = 0, lrow = 0;
  unsigned voff[NI];
  const char* a_base = p.A;
  auto set_src = [&](int tile) {
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if (wave + NW * j < BM / 16) {
        const int r = min(tile * BM + 16 * (wave + NW * j) + lrow, p.M - 1) - tile * BM;
        voff[j] = (unsigned)(r * (int)p.lda_b + lc0 * 16);
      }
    a_base = p.A + (int64_t)tile * BM * p.lda_b;
  };
  auto lane_state = [&](int tile) {
    int l = lane;
    asm volatile("" : "+v"(l));
    lc0 = (l & 3) ^ ((0 - (l >> 4)) & 3);                          // row swizzle g((row >> 2) & 3): units start at multiples of 16 rows
    lrow = l >> 2;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int wrow = 16 * (wave + NW * j) + lrow - BM;           // W row of this lane in unit j (padding units clamp)
      voff[j] = (unsigned)(min(max(wrow, 0), LBN - 1) * (int)p.ldw_b + lc0 * 16);
    }
    const int pc = ((l >> 4) ^ ((0 - ((l & 15) >> 2)) & 3)) << 4;
    a_rd = (l & 15) * LKB + pc;
    b_rd = (BM + wave * WC + (l & 15)) * LKB + pc;
    set_src(tile);
  };
  // The (tile, k-step) pairs of the block form ONE stream of `total` k-steps; stage q of the stream lives in LDS buffer
  // q % NST.  issue_next() sends the DMA of the next stage that has not been requested yet.
  const int total = my_tiles * nkt;
  int issued = 0, iss_k = 0, iss_t = 0, is_buf = 0;
  auto issue_next = [&]() {
    if (iss_k == nkt) {
      iss_k = 0;
      iss_t += 1;
      set_src((int)blockIdx.x + iss_t * G);
    }
    const char* wk = p.W + iss_k * LKB;
    const char* ak = a_base + iss_k * LKB;
#pragma unroll
    for (int j = 0; j < NI; ++j)
      if (j < NI - 1 || full)
        glds16_s(voff[j], (j < NA && wave + NW * j < BM / 16) ? ak : wk,
                 __builtin_amdgcn_readfirstlane(lds_base + is_buf * LSTAGE + (wave + NW * j) * 1024));
    is_buf = (is_buf + 1) & (NST - 1);
    iss_k += 1;
    issued += 1;
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  static_assert(NW * LSCR + (2 * BM * NW + 2 * BM) * (int)sizeof(float) <= LSTAGE, "epilogue scratch must fit in one stage buffer");

#define NMTGAN_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  // all but this wave's k newest stages have landed
#define NMTGAN_WAIT_STAGES(k) do { if (full) NMTGAN_WAIT((k) * NI); else NMTGAN_WAIT((k) * (NI - 1)); } while (0)
  // Fragments are double-buffered in registers: k-step q multiplies the fragments read during k-step q - 1 while the
  // fragments of q + 1 are read from the stage that has just landed, so a wave's LDS reads run under its own MFMAs.
  // Per k-step: wait (stage q + 1 landed; q + 2 and q + 3 may be in flight) -> this wave's reads of stage q are complete
  // -> barrier (every wave is past its reads of stage q, whose buffer the DMA of stage q + 4 refills) -> issue q + 4
  // -> read q + 1 -> the MFMAs of q.  The LAST k-step of a tile issues nothing: the buffer of its own stage stays free
  // through the epilogue, which uses it as transpose scratch and statistics space; the next k-step issues two stages.
  uint4 fa[2][MT], fb[2][NT];
  auto read_frags = [&](int buf, uint4 (&a)[MT], uint4 (&b)[NT]) {
    const unsigned char* base = smem + buf * LSTAGE;
#pragma unroll
    for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const uint4*>(base + a_rd + i * 16 * LKB);
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const uint4*>(base + b_rd + j * 16 * LKB);
  };

  const int ncol = WC * wave;
  int rbuf = 0;                                    // LDS buffer of the next stage whose fragments have not been read
  auto epilogue_ln = [&](int t, int free_buf) {
    float* scr = reinterpret_cast<float*>(smem + free_buf * LSTAGE + wave * LSCR);
    float* stat0 = reinterpret_cast<float*>(smem + free_buf * LSTAGE + NW * LSCR);   // [BM][NW] partial row sums per wave
    float* stat1 = stat0 + BM * NW;
    float* red0 = stat1 + BM * NW;                 // [BM] row means, [BM] reciprocal deviations
    float* red1 = red0 + BM;
    // lane-derived indices are recomputed here from a value the optimiser cannot see through: otherwise it hoists the
    // ~50 loop-invariant offsets of the epilogue out of the k-step loop and spills them
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int fc = lane_e & 15, fq = lane_e >> 4, c4 = 4 * (lane_e & 7), rl = lane_e >> 3;
    // The accumulators stay where the MFMAs left them (16x16 C layout: column = lane & 15, row = 4 * (lane >> 4) + e)
    // from the residual add to the normalised value; only the global loads and stores use the row-major view of a
    // 16 x 32 piece (row rl + 8 t, columns c4 .. c4 + 3: 16 bytes per lane, whole 128-byte row segments per
    // instruction), and each piece changes view through the wave's LDS scratch.
    // addresses = one uniform tile base + a 32-bit lane offset (a 64-bit address per access would cost two registers each)
    const int m0 = ((int)blockIdx.x + t * G) * BM;
    const int rows_here = min(BM, p.M - m0);
    float* xt = p.ln_x + (int64_t)m0 * LBN;
    OpT* ot = p.ln_out ? reinterpret_cast<OpT*>(p.ln_out) + (int64_t)m0 * LBN : nullptr;
    const int col = ncol + c4;
    float bias[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bias[j] = p.bias ? p.bias[ncol + 16 * j + fc] : 0.f;
    // GI row blocks at a time (16 residual loads of 16 bytes per lane in flight), then each 16 x 32 piece changes view
    constexpr int GI = (MT % (8 / NJP) == 0) ? 8 / NJP : 2;
    auto add_residual = [&](auto H) {
      constexpr int h = decltype(H)::value;
      f32x4 res[GI][NJP][2];
#pragma unroll
      for (int i = 0; i < GI; ++i)
#pragma unroll
        for (int jp = 0; jp < NJP; ++jp)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {           // rows past M re-read the last row: no predicate, no branch
            const unsigned off = (unsigned)(min(16 * (GI * h + i) + rl + 8 * tt, rows_here - 1) * LBN + col + 32 * jp);
            res[i][jp][tt] = *reinterpret_cast<const f32x4*>(xt + off);
          }
#pragma unroll
      for (int i = 0; i < GI; ++i)
#pragma unroll
        for (int jp = 0; jp < NJP; ++jp) {
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) *reinterpret_cast<f32x4*>(scr + (rl + 8 * tt) * 36 + c4) = res[i][jp][tt];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[GI * h + i][2 * jp + jj][e] =
                  scr[(4 * fq + e) * 36 + 16 * jj + fc] + (acc[GI * h + i][2 * jp + jj][e] + bias[2 * jp + jj]);
          __builtin_amdgcn_sched_barrier(0);       // piece by piece: the scheduler otherwise gathers every piece's
        }                                          // LDS reads in front of the adds and spills them
    };
    for_each_const(add_residual, std::make_integer_sequence<int, MT / GI>{});
    // sum over a row: the lane's NT columns, the 16 lanes of the row's group, then the NW waves in wave order by ONE
    // thread per row, which leaves the row's mean / reciprocal deviation in red[row] for every wave to pick up
    auto row_total = [&](float s, float* stat, int row) {
      s = dpp_add<0xB1>(s);
      s = dpp_add<0x4E>(s);
      s = dpp_add<0x141>(s);
      s = dpp_add<0x140>(s);
      if (fc == 0) stat[row * NW + wave] = s;
    };
    auto row_sum = [&](const float* stat, int row) {
      const float4 a = *reinterpret_cast<const float4*>(stat + row * NW);
      float s = (a.x + a.y) + (a.z + a.w);
      if constexpr (NW == 8) {
        const float4 b = *reinterpret_cast<const float4*>(stat + row * NW + 4);
        s += (b.x + b.y) + (b.z + b.w);
      }
      return s;
    };
    auto lane_sum = [&](const float (&v)[NT]) {
      float s = (v[0] + v[1]) + (v[2] + v[3]);
      if constexpr (NT == 8) s += (v[4] + v[5]) + (v[6] + v[7]);
      return s;
    };
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) v[j] = acc[i][j][e];
        row_total(lane_sum(v), stat0, 16 * i + 4 * fq + e);
      }
    __syncthreads();
    if (tid < BM) red0[tid] = row_sum(stat0, tid) * (1.0f / LBN);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float mean = red0[16 * i + 4 * fq + e];
        float v[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j][e] -= mean;
          v[j] = acc[i][j][e] * acc[i][j][e];
        }
        row_total(lane_sum(v), stat1, 16 * i + 4 * fq + e);
      }
    __syncthreads();
    if (tid < BM) red1[tid] = rsqrtf(row_sum(stat1, tid) * (1.0f / LBN) + 1e-6f);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float rs = red1[16 * i + 4 * fq + e];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j][e] *= rs;
      }
    f32x4 gam[NJP], bet[NJP];
#pragma unroll
    for (int jp = 0; jp < NJP; ++jp) {
      gam[jp] = *reinterpret_cast<const f32x4*>(p.ln_gamma + ncol + 32 * jp + c4);
      bet[jp] = *reinterpret_cast<const f32x4*>(p.ln_beta + ncol + 32 * jp + c4);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int jp = 0; jp < NJP; ++jp) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int e = 0; e < 4; ++e) scr[(4 * fq + e) * 36 + 16 * jj + fc] = acc[i][2 * jp + jj][e];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int r = 16 * i + rl + 8 * tt;
          const f32x4 v = *reinterpret_cast<const f32x4*>(scr + (rl + 8 * tt) * 36 + c4);
          if (r < rows_here) {
            const f32x4 o = v * gam[jp] + bet[jp];
            const unsigned off = (unsigned)(r * LBN + col + 32 * jp);
            *reinterpret_cast<f32x4*>(xt + off) = o;
            if (ot)
              *reinterpret_cast<uint2*>(ot + off) = make_uint2(Pack16<OpT>::pack2(o[0], o[1]), Pack16<OpT>::pack2(o[2], o[3]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto kstep = [&](int q, const uint4 (&ca)[MT], const uint4 (&cb)[NT], uint4 (&na)[MT], uint4 (&nb)[NT], bool last) {
    if (q + 1 < total) {
      const int ahead = issued - (q + 2);          // stages requested beyond q + 1
      if (ahead >= 2) NMTGAN_WAIT_STAGES(2);
      else if (ahead == 1) NMTGAN_WAIT_STAGES(1);
      else NMTGAN_WAIT(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!last) {
      if (issued < total && issued < q + 1 + NST) issue_next();
      if (issued < total && issued < q + 1 + NST) issue_next();    // catches up after a tile's last k-step
      read_frags(rbuf, na, nb);
      rbuf = (rbuf + 1) & (NST - 1);
    }
    mma_block<OpT, MT, NT, 0, 0>(ca, cb, acc);
  };
  lane_state((int)blockIdx.x);
#pragma unroll
  for (int i = 0; i < NST; ++i) issue_next();    // total >= nkt >= 4
  NMTGAN_WAIT_STAGES(NST - 1);
  __builtin_amdgcn_s_barrier();
  int q = 0;
  for (int t = 0; t < my_tiles; ++t) {
    lane_state((int)blockIdx.x + iss_t * G);     // unconditional: the old values are dead here (the DMA stream is 0 or 1 tiles ahead)
    // the tile's first fragments are read here and not under the previous tile's last MFMAs: nothing but the
    // accumulators stays live across the epilogue
    read_frags(rbuf, fa[0], fb[0]);
    rbuf = (rbuf + 1) & (NST - 1);
#pragma unroll 1
    for (int kp = 0; kp < nkt / 2; ++kp) {
      kstep(q, fa[0], fb[0], fa[1], fb[1], false);
      kstep(q + 1, fa[1], fb[1], fa[0], fb[0], kp + 1 == nkt / 2);
      q += 2;
    }
    epilogue_ln(t, (q - 1) & (NST - 1));
  }
#undef NMTGAN_WAIT_STAGES
#undef NMTGAN_WAIT
}

// ------------------------------------------------------------------------------------------ generic kernel
constexpr int GBM = 128, GBN = 128, GTILE = GBM * BKB;

template <typename OpT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 buffers][A 16K | B 16K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  tile_index(p, tm, tn);
  const int bm0 = tm * GBM, bn0 = tn * GBN;

  const int lc = tid & 7, lr0 = tid >> 3;
  const int lds_wr = lr0 * BKB + ((lc ^ ((lr0 >> 1) & 7)) << 4);
  const char* a_ptr[4];
  const char* w_ptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_ptr[i] = p.A + a_row_offset(p, min(bm0 + lr0 + 32 * i, p.M - 1)) + lc * 16;
    w_ptr[i] = p.W + (int64_t)min(bn0 + lr0 + 32 * i, p.N - 1) * p.ldw_b + lc * 16;
  }
  const int nkt = (p.kbytes + BKB - 1) / BKB;

  uint4 sa[4], sb[4];
  auto gload = [&](int kt) {
    const bool ok = kt * BKB + lc * 16 < p.kbytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sa[i] = ok ? *reinterpret_cast<const uint4*>(a_ptr[i] + (int64_t)kt * BKB) : make_uint4(0, 0, 0, 0);
      sb[i] = ok ? *reinterpret_cast<const uint4*>(w_ptr[i] + (int64_t)kt * BKB) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* base = smem + buf * 2 * GTILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(base + lds_wr + i * 32 * BKB) = sa[i];
      *reinterpret_cast<uint4*>(base + GTILE + lds_wr + i * 32 * BKB) = sb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fc = lane & 15, fq = lane >> 4;
  const int fsw = (fc >> 1) & 7;
  const int a_rd = (wm * 64 + fc) * BKB;
  const int b_rd = GTILE + (wn * 64 + fc) * BKB;

  gload(0);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) gload(kt + 1);
    compute_tile<OpT>(smem + cur * 2 * GTILE, a_rd, b_rd, fq, fsw, acc);
    if (kt + 1 < nkt) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  epilogue<OpT, EPI>(p, acc, bm0 + wm * 64, bn0 + wn * 64, lane);
}

// ------------------------------------------------------------------------------------------ launch
template <typename OpT, int EPI, int BM, int BN>
static int launch_glds(GemmParams p, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * BKB;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<OpT, EPI, BM, BN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      NMTGAN_FAIL(-3, "gemm: cannot raise dynamic LDS limit");
    attr_done = true;
  }
  p.ntm = (p.M + BM - 1) / BM;
  p.ntn = (p.N + BN - 1) / BN;
  const int slices = (EPI == NMTGAN_EPI_LINEAR && p.split_k > 1) ? p.split_k : 1;
  if (slices == 1) p.split_k = 0;
  hipLaunchKernelGGL((gemm_glds_kernel<OpT, EPI, BM, BN>), dim3(p.ntm * p.ntn, slices), dim3(64 * (BM / 64) * (BN / 64)),
                     smem, st, p);
  NMTGAN_CHECK_LAUNCH("gemm_glds");
  if constexpr (EPI == NMTGAN_EPI_LINEAR) {
    if (slices > 1) {
      const int64_t n = (int64_t)p.M * p.N;
      hipLaunchKernelGGL(splitk_finish_kernel<OpT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
      NMTGAN_CHECK_LAUNCH("gemm_splitk_finish");
    }
  }
  return 0;
}

// Split-K scratch of the calling thread (set by single-stream pipelines around their GEMM calls; never by the
// two-stream rollout schedule, whose streams would share it).  With it, a LINEAR GEMM that would occupy few of the
// 256 CUs with a long serial k-loop (M = 64 decode rows; the dW products of the update steps) is cut into k-slices.
static thread_local float* g_splitk_buf = nullptr;
static thread_local size_t g_splitk_bytes = 0;
void set_gemm_splitk_scratch(void* buf, size_t bytes) { g_splitk_buf = (float*)buf; g_splitk_bytes = bytes; }
bool gemm_splitk_active() { return g_splitk_buf != nullptr; }
bool gemm_addln_supported(int op_dtype, int N, int K) {
  const int es = dtype_size(op_dtype);
  return es == 2 && N == LBN && (K * es) % (2 * LKB) == 0 && (K * es) / LKB >= 4;   // an even number of 64-byte k-steps
}
static int choose_split_k(const GemmParams& p, int nkt) {
  if (!g_splitk_buf) return 1;
  const int64_t tiles = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
  if (tiles > 64 || nkt < 8) return 1;
  int64_t s = std::min<int64_t>(std::min<int64_t>(nkt / 2, (256 + tiles - 1) / tiles), 32);
  const int64_t per = (int64_t)p.M * p.N * sizeof(float);
  if (per > 0) s = std::min<int64_t>(s, (int64_t)(g_splitk_bytes / per));
  return s >= 2 ? (int)s : 1;
}

template <typename OpT, int EPI>
static int launch_generic(GemmParams p, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<OpT, EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * GTILE) != hipSuccess)
      NMTGAN_FAIL(-3, "gemm: cannot raise dynamic LDS limit");
    attr_done = true;
  }
  p.ntm = (p.M + GBM - 1) / GBM;
  p.ntn = (p.N + GBN - 1) / GBN;
  hipLaunchKernelGGL((gemm_kernel<OpT, EPI>), dim3(p.ntm * p.ntn), dim3(256), 4 * GTILE, st, p);
  NMTGAN_CHECK_LAUNCH("gemm");
  return 0;
}

template <typename OpT, int BM, int NW>
static int launch_ln(GemmParams p, hipStream_t st) {
  // four stages of (BM + 512) rows x 64 bytes; the epilogue's scratch and statistics live in the stage buffer it keeps free
  constexpr int smem = 4 * (BM + LBN) * LKB;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln_kernel<OpT, BM, NW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      NMTGAN_FAIL(-3, "gemm: cannot raise dynamic LDS limit");
    attr_done = true;
  }
  p.ntm = (p.M + BM - 1) / BM;
  p.ntn = 1;
  hipLaunchKernelGGL((gemm_ln_kernel<OpT, BM, NW>), dim3(persistent_grid(p.ntm)), dim3(64 * NW), smem, st, p);
  NMTGAN_CHECK_LAUNCH("gemm_ln");
  return 0;
}
// Eight waves on 64-row tiles (96-row tiles from 16 k rows on: 239 VGPRs, four stages = 152 KiB of LDS).  Measured alternatives (fp16, M = 49152): the same kernel as <128, 4> - four waves, one
// per SIMD, 128 x 128 wave tiles, half the W re-fetch per flop - runs K = 512 / 2048 in 135 / 254 us against 83 / 196:
// with 32 KiB of W per 64-byte k-step only three stages fit in LDS, and two stages in flight per CU do not cover the
// L2 -> LDS latency (a lone tile takes 44 us at K = 512); <128, 8> spills inside the k-loop.  K = 2048 therefore stays
// on the 256x256 kernel + add_layernorm (176 us).
template <typename OpT>
static int dispatch_ln(const GemmParams& p, hipStream_t st) {
  if constexpr (sizeof(OpT) == 2) {
    // 96-row tiles re-fetch the 512 W rows a third less often per row of A (the kernel runs on the L2 -> LDS fill path);
    // they pay from about 16 k rows on (C2 sweep of the switch point: 24576 / 16384 / 12288 / 8192 rows -> projections
    // 59.0 / 58.1 / 58.3 / 58.4 ms against 59.7 with 64-row tiles only); below that the shorter tile finishes sooner.
    // NMTGAN_LN_BM=64|96 forces one form, NMTGAN_LN_BM96_ROWS moves the switch (a row's bits are the same in both).
    static const int bm = [] { const char* e = getenv("NMTGAN_LN_BM"); return e ? atoi(e) : 0; }();
    static const int rows96 = [] { const char* e = getenv("NMTGAN_LN_BM96_ROWS"); return e ? atoi(e) : 16384; }();
    if (bm == 96 || (bm == 0 && p.M >= rows96)) return launch_ln<OpT, 96, 8>(p, st);
    return launch_ln<OpT, 64, 8>(p, st);
  }
  NMTGAN_FAIL(-2, "gemm: the ADDLN epilogue needs a 16-bit op dtype");
}

static int tile_override() {        // NMTGAN_GEMM_TILE=128|256|0(generic) for experiments; unset = heuristic
  static int v = -2;
  if (v == -2) {
    const char* e = getenv("NMTGAN_GEMM_TILE");
    v = e ? atoi(e) : -1;
  }
  return v;
}

// Kernel choice.  Both persistent kernels are power-limited on MI355X (the clock drops to ~1.5 GHz under
// MFMA + LDS + DMA load), so the 256x256 kernel - fewer LDS and L2 bytes per flop - wins whenever its
// larger tiles are not wasted on padding and the k-loop is long enough to amortise its epilogue.
// Measured padded rates (bf16, random operands): ~1250 TFLOP/s for 256x256, ~1080 for 256x128.
static int p8_mode() {              // NMTGAN_GEMM_P8=0 never, 1 heuristic (default), 2 always when legal, 3 the tile-count
                                    // thresholds also for the LINEAR epilogue (the rule before the rounds model)
  static const int v = [] { const char* e = getenv("NMTGAN_GEMM_P8"); return e ? atoi(e) : 1; }();
  return v;
}
static bool prefer_p8(int64_t M, int N, int nkt) {
  const int mode = p8_mode();
  if (mode == 0 || nkt < 4) return false;
  if (mode == 2) return true;
  const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
  if (tiles < 192) return false;
  if (nkt < 16 && N < 2048) return false;        // K = 512 projections: the 256x128 kernel's store overlap wins
  const double t_p8 = (double)((N + 255) / 256 * 256) / 1250.0, t_pipe = (double)((N + 127) / 128 * 128) / 1080.0;
  return t_p8 < t_pipe;
}
// LINEAR epilogue, 16-bit operands: the two persistent kernels priced as (rounds of tiles over the CUs) x (time of one
// tile = fixed part + part per 128-byte k-tile), microseconds fitted to sweeps over the row counts of the rollout steps
// (tools/sweep_p8.py; 256x128: 5.7 + 0.854 nkt, 256x256: 13.3 + 1.21 nkt).  The 256x256 tile does twice the work in 1.4x
// the k-loop time but pays 2.3x the fixed part, and which kernel fills its last round better changes every few
// thousand rows; the model picks the faster kernel on every sampled point of qkv / cq / ffn1 and all but the largest of ffn2.
static bool prefer_p8_linear(int64_t M, int N, int nkt) {
  const int mode = p8_mode();
  if (mode == 0 || nkt < 4) return false;
  if (mode == 2) return true;
  const int64_t cus = persistent_grid(1 << 30);
  const int64_t rt = (M + 255) / 256;
  const int64_t rounds_pipe = (rt * ((N + 127) / 128) + cus - 1) / cus, rounds_p8 = (rt * ((N + 255) / 256) + cus - 1) / cus;
  return (double)rounds_p8 * (13.3 + 1.21 * nkt) < (double)rounds_pipe * (5.7 + 0.854 * nkt);
}
// LINEAR epilogue over plain rows, 16-bit operands: a problem that fills the CUs r times with 256x256 tiles and then
// leaves a last round mostly empty (N = 512, M = 49152: 384 tiles on 256 CUs) is cut by rows - the first m1 rows are
// r FULL rounds of the 256x256 kernel, the remainder goes back through the kernel choice on its own (usually one round
// of 256x128 tiles).  Priced with the same rounds model as prefer_p8_linear plus a charge for the extra launch; returns
// m1 or 0.  An output element's k order is the same in both kernels, so the cut changes no bits.
// NMTGAN_GEMM_ROWSPLIT=0 switches it off.
static int64_t p8_row_split(int64_t M, int N, int nkt) {
  static const bool on = [] { const char* e = getenv("NMTGAN_GEMM_ROWSPLIT"); return !e || atoi(e) != 0; }();
  if (!on || p8_mode() != 1 || nkt < 4) return 0;
  const int64_t cus = persistent_grid(1 << 30);
  const int64_t ntn8 = (N + 255) / 256, ntnp = (N + 127) / 128;
  const int64_t rows_round = (cus / ntn8) * 256;
  if (rows_round <= 0 || M <= rows_round) return 0;
  const double t8 = 13.3 + 1.21 * nkt, tp = 5.7 + 0.854 * nkt, launch = 6.0;
  auto whole = [&](int64_t m) {
    const int64_t rt = (m + 255) / 256;
    return std::min((double)((rt * ntn8 + cus - 1) / cus) * t8, (double)((rt * ntnp + cus - 1) / cus) * tp);
  };
  double best = whole(M) - 1.0;                  // a cut has to win by more than a microsecond
  int64_t m1 = 0;
  for (int64_t r = 1; r * rows_round < M; ++r) {
    const double c = (double)r * t8 + whole(M - r * rows_round) + launch;
    if (c < best) { best = c; m1 = r * rows_round; }
  }
  return m1;
}

template <typename OpT, int EPI>
static int dispatch(const GemmParams& p, hipStream_t st) {
  const int ov = tile_override();
  if (p.kbytes % BKB != 0 || ov == 0) return launch_generic<OpT, EPI>(p, st);
  // persistent pipelined kernels when the k-loop is long enough for their two-deep prefetch and there are at least
  // NMTGAN_PIPE_MIN_TILES (16) tiles; else the simple double-buffered 128x128 kernel.  The threshold was 192 (3/4 of
  // the CUs): the late rollout steps (1-12 k rows) then ran their K = 512 / 2048 GEMMs on the double-buffered
  // kernel, whose every k-tile waits out a full load latency (25-60 us per call against 12-20 on the pipelined one)
  const int nkt = p.kbytes / BKB;
  const int64_t tiles256 = (int64_t)((p.M + 255) / 256) * ((p.N + 127) / 128);
  const bool pipe_ok = nkt >= 4;
  if constexpr (EPI == NMTGAN_EPI_CONVMAX) {
    // filter counts are not multiples of 256: full 256-column panels on the 256x256 kernel, a remainder of
    // <= 128 columns on the 256x128 kernel
    const int rem = p.N % 256;
    if (pipe_ok && p8_mode() == 1 && p.N > 256 && rem > 0 && rem <= 128 && nkt >= 16 &&
        ((p.M + 255) / 256) * (int64_t)(p.N / 256) >= 192) {
      GemmParams a = p, b = p;
      a.N = p.N - rem;
      b.N = rem;
      b.W = p.W + (int64_t)a.N * p.ldw_b;
      b.scale = p.scale + a.N; b.shift = p.shift + a.N; b.pooled = p.pooled + a.N;
      NMTGAN_TRY((launch_p8<OpT, EPI>(a, st)));
      return launch_pipe<OpT, EPI>(b, st);
    }
  }
  const bool modelled = EPI == NMTGAN_EPI_LINEAR && sizeof(OpT) == 2 && tiles256 >= 16 && p8_mode() != 3;
  if constexpr (EPI == NMTGAN_EPI_LINEAR) {
    if (modelled && pipe_ok && ov < 0 && p.a_group == 0) {
      const int64_t m1 = p8_row_split(p.M, p.N, nkt);
      if (m1 > 0) {
        GemmParams a = p, b = p;
        a.M = (int)m1;
        b.M = p.M - (int)m1;
        b.A = p.A + m1 * p.lda_b;
        for (int sgm = 0; sgm < 3; ++sgm) {
          if (!p.out[sgm]) continue;
          const int64_t oes = p.out_f32[sgm] == 1 ? 4 : (p.out_f32[sgm] == 2 ? 2 : (int64_t)sizeof(OpT));
          b.out[sgm] = (char*)p.out[sgm] + m1 * p.ldo[sgm] * oes;
        }
        NMTGAN_TRY((launch_p8<OpT, EPI>(a, st)));
        return dispatch<OpT, EPI>(b, st);
      }
    }
  }
  if (pipe_ok && (modelled ? prefer_p8_linear(p.M, p.N, nkt) : prefer_p8(p.M, p.N, nkt))) return launch_p8<OpT, EPI>(p, st);
  static const int pipe_min_tiles = [] { const char* e = getenv("NMTGAN_PIPE_MIN_TILES"); return e && atoi(e) > 0 ? atoi(e) : 16; }();
  if (pipe_ok && (ov == 3 || (ov < 0 && tiles256 >= pipe_min_tiles))) return launch_pipe<OpT, EPI>(p, st);
  if (ov == 256) return launch_glds<OpT, EPI, 256, 128>(p, st);
  if constexpr (EPI == NMTGAN_EPI_LINEAR) {
    GemmParams q = p;
    q.split_k = choose_split_k(p, nkt);
    q.split_part = g_splitk_buf;
    return launch_glds<OpT, EPI, 128, 128>(q, st);
  }
  return launch_glds<OpT, EPI, 128, 128>(p, st);
}

// logits^T = W[V,K] . X[rows,K]^T reduced on the fly to per-(64-vocab tile, row) softmax statistics
int launch_gemm_stats(int op_dtype, const void* W, int V, const void* X, int rows, int K, float2* stats,
                      hipStream_t st) {
  prof::Scope prof_scope(prof::GEMM_LOGITS, st);
  const int es = dtype_size(op_dtype);
  NMTGAN_REQUIRE(W && X && stats && V > 0 && rows > 0 && K > 0, "gemm_stats: bad arguments");
  NMTGAN_REQUIRE(V % 64 == 0 && (K * es) % BKB == 0 && (K * es) / BKB >= 4, "gemm_stats: unsupported shape");
  GemmParams p{};
  p.A = (const char*)W; p.W = (const char*)X;
  p.M = V; p.N = rows; p.K = K;
  p.lda_b = (int64_t)K * es; p.ldw_b = (int64_t)K * es; p.kbytes = K * es;
  p.k_inner = 1; p.k_step = BKB; p.k_jump = BKB;
  p.stats = stats;
  static const bool stats_p8 = [] { const char* e = getenv("NMTGAN_STATS_P8"); return !e || atoi(e) != 0; }();
  if (stats_p8 && prefer_p8(V, rows, p.kbytes / BKB))
    NMTGAN_DISPATCH_OP(op_dtype, OpT, return launch_p8<OpT, EPI_STATS>(p, st));
  NMTGAN_DISPATCH_OP(op_dtype, OpT, return launch_pipe<OpT, EPI_STATS>(p, st));
  return 0;
}
bool gemm_stats_supported(int op_dtype, int V, int K) {
  const int es = dtype_size(op_dtype);
  return V % 64 == 0 && V <= 65536 && (K * es) % BKB == 0 && (K * es) / BKB >= 4;   // <= 1024 vocabulary tiles
}

static int launch_gemm_impl(const nmtgan_gemm_args* a, hipStream_t st);
// NMTGAN_TRACE_GEMM=1: time every GEMM call with a pair of events and a synchronise, print shape and time
// (diagnosis only - it serialises the stream)
int launch_gemm(const nmtgan_gemm_args* a, hipStream_t st) {
  static const bool trace = [] { const char* e = getenv("NMTGAN_TRACE_GEMM"); return e && atoi(e) != 0; }();
  if (!trace || !a) return launch_gemm_impl(a, st);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, st);
  const int rc = launch_gemm_impl(a, st);
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  fprintf(stderr, "GEMM epi=%d M=%d N=%d K=%d a_group=%d ldw=%lld us=%.1f tflops=%.1f\n", a->epilogue, a->M, a->N, a->K,
          a->a_group, (long long)a->ldw, ms * 1e3, 2.0 * a->M * a->N * a->K / (ms * 1e-3) / 1e12);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}
// nmtgan_gemm_args -> GemmParams (shape and alignment checks; the epilogue-specific fields of the three public
// epilogues); shared by the single-problem dispatch and the grouped launch
static int fill_params(const nmtgan_gemm_args* a, GemmParams& p) {
  NMTGAN_REQUIRE(a != nullptr, "gemm: null args");
  NMTGAN_REQUIRE(valid_op_dtype(a->op_dtype), "gemm: bad op_dtype %d", a->op_dtype);
  NMTGAN_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, "gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
  const int es = dtype_size(a->op_dtype);
  NMTGAN_REQUIRE(a->A && a->Wt, "gemm: null operand");
  NMTGAN_REQUIRE(((int64_t)a->K * es) % 16 == 0, "gemm: K*esize must be a multiple of 16 bytes (K=%d)", a->K);
  NMTGAN_REQUIRE((a->lda * es) % 16 == 0, "gemm: lda*esize must be a multiple of 16 bytes");
  NMTGAN_REQUIRE(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->Wt % 16) == 0, "gemm: operands must be 16-byte aligned");
  NMTGAN_REQUIRE(a->a_group <= 0 || (a->a_group_stride * es) % 16 == 0, "gemm: a_group_stride misaligned");
  p = GemmParams{};
  p.A = (const char*)a->A; p.W = (const char*)a->Wt;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.lda_b = a->lda * es;
  p.a_group = a->a_group > 0 ? a->a_group : 0;
  p.a_group_stride_b = a->a_group_stride * es;
  NMTGAN_REQUIRE(a->ldw == 0 || (a->ldw >= a->K && (a->ldw * es) % 16 == 0), "gemm: ldw must be 0 or >= K, 16-byte rows");
  p.ldw_b = (a->ldw > 0 ? a->ldw : (int64_t)a->K) * es;
  p.kbytes = a->K * es;
  p.k_inner = 1; p.k_step = BKB; p.k_jump = BKB;
  static const bool tap_order = [] { const char* e = getenv("NMTGAN_CONV_TAPORDER"); return !e || atoi(e) != 0; }();
  if (tap_order && a->a_group > 0 && a->lda > 0 && a->lda < a->K && a->K % a->lda == 0 && (a->lda * es) % BKB == 0) {
    p.k_inner = (int)(a->K / a->lda);            // taps
    p.k_step = a->lda * es;                      // next tap, same channel block
    p.k_jump = BKB - (int64_t)(p.k_inner - 1) * p.k_step;   // back to tap 0, next channel block
  }
  if (a->epilogue == NMTGAN_EPI_LINEAR) {
    NMTGAN_REQUIRE(a->n_seg >= 1 && a->n_seg <= 3, "gemm: n_seg must be 1..3");
    NMTGAN_REQUIRE(a->n_seg == 1 || (a->seg_width > 0 && a->seg_width * a->n_seg >= a->N),
                   "gemm: segments do not cover N");
    p.bias = a->bias; p.relu = a->relu; p.n_seg = a->n_seg; p.seg_width = a->seg_width > 0 ? a->seg_width : a->N;
    for (int s = 0; s < 3; ++s) {
      p.out[s] = a->out[s]; p.ldo[s] = a->ldo[s]; p.out_f32[s] = a->out_f32[s];
      p.seg_scale[s] = a->seg_scale[s];
      if (s < a->n_seg) NMTGAN_REQUIRE(a->out[s] != nullptr, "gemm: null output segment %d", s);
      if (s < a->n_seg && a->out_f32[s] == 2)
        NMTGAN_REQUIRE(es == 2 && p.seg_width % 64 == 0 && a->ldo[s] >= 3 * (int64_t)p.seg_width && a->ldo[s] % 8 == 0 &&
                           (uintptr_t)a->out[s] % 16 == 0,
                       "gemm: a split-image segment needs a 16-bit op dtype, width %% 64 == 0 and 16-byte rows of >= 3 widths");
    }
  } else if (a->epilogue == NMTGAN_EPI_CONVMAX) {
    NMTGAN_REQUIRE(a->a_group > 0, "gemm: CONVMAX needs a_group (windows per sentence)");
    NMTGAN_REQUIRE(a->scale && a->shift && a->pooled, "gemm: CONVMAX needs scale/shift/pooled");
    p.scale = a->scale; p.shift = a->shift; p.pooled = a->pooled; p.ld_pool = a->ld_pool;
    if (a->add) {
      NMTGAN_REQUIRE(a->add_blk && a->add_rows > 0 && a->add_row0 >= 0 && a->add_row0 + a->a_group <= a->add_rows,
                     "gemm: CONVMAX additive term needs add_blk and 0 <= add_row0, add_row0 + a_group <= add_rows");
      p.add = a->add; p.add_blk = a->add_blk; p.add_rows = a->add_rows; p.add_row0 = a->add_row0;
    }
  } else if (a->epilogue == NMTGAN_EPI_ADDLN) {
    NMTGAN_REQUIRE(gemm_addln_supported(a->op_dtype, a->N, a->K),
                   "gemm: ADDLN needs a 16-bit op dtype, N = 512 and K*esize a multiple of 128 bytes, >= 256 (N=%d K=%d)", a->N, a->K);
    NMTGAN_REQUIRE(a->a_group <= 0, "gemm: ADDLN takes plain rows (a_group = 0)");
    NMTGAN_REQUIRE(a->ln_x && a->ln_gamma && a->ln_beta, "gemm: ADDLN needs ln_x / ln_gamma / ln_beta");
    NMTGAN_REQUIRE((uintptr_t)a->ln_x % 16 == 0 && (uintptr_t)a->ln_gamma % 16 == 0 && (uintptr_t)a->ln_beta % 16 == 0 &&
                       (uintptr_t)a->ln_out % 8 == 0, "gemm: ADDLN pointers misaligned");
    p.bias = a->bias; p.ln_x = a->ln_x; p.ln_gamma = a->ln_gamma; p.ln_beta = a->ln_beta; p.ln_out = a->ln_out;
  } else {
    NMTGAN_FAIL(-2, "gemm: unknown epilogue %d", a->epilogue);
  }
  return 0;
}

static int launch_gemm_impl(const nmtgan_gemm_args* a, hipStream_t st) {
  if (a && a->epilogue == NMTGAN_EPI_CONVMAX && a->add) return launch_gemm_group(a, 1, st);   // the additive form lives in the grouped kernel
  prof::Scope prof_scope(prof::GEMM_PROJ, st);
  GemmParams p;
  NMTGAN_TRY(fill_params(a, p));
  if (a->M == 0) return 0;
  if (a->epilogue == NMTGAN_EPI_LINEAR) {
    NMTGAN_DISPATCH_OP(a->op_dtype, OpT, return dispatch<OpT, NMTGAN_EPI_LINEAR>(p, st));
  } else if (a->epilogue == NMTGAN_EPI_CONVMAX) {
    NMTGAN_DISPATCH_OP(a->op_dtype, OpT, return dispatch<OpT, NMTGAN_EPI_CONVMAX>(p, st));
  }
  NMTGAN_DISPATCH_OP(a->op_dtype, OpT, return dispatch_ln<OpT>(p, st));
  return 0;
}

// Several independent GEMMs of one epilogue kind (CONVMAX, with or without the additive term, or LINEAR with a single
// output segment, fp32 or op dtype) in as few persistent launches as their count needs (GROUP_MAX problems each), longest k-loop
// first.  A problem the grouped kernel does not take (k extent not a multiple of 128 bytes, fewer than 4 k-tiles)
// is launched on its own.  min_tiles > 0: problems with at least that many 256x128 tiles are launched on their own too
// (the 256x256 kernel serves them better), unless they carry the additive term.
int launch_gemm_group(const nmtgan_gemm_args* args, int n, hipStream_t st, int min_tiles) {
  NMTGAN_REQUIRE(n >= 0 && (n == 0 || args != nullptr), "gemm_grouped: bad arguments");
  if (n == 0) return 0;
  const int epi = args[0].epilogue, op = args[0].op_dtype;
  NMTGAN_REQUIRE(epi == NMTGAN_EPI_LINEAR || epi == NMTGAN_EPI_CONVMAX, "gemm_grouped: LINEAR or CONVMAX problems only");
  std::vector<GemmParams> ps;
  ps.reserve(n);
  for (int i = 0; i < n; ++i) {
    const nmtgan_gemm_args& a = args[i];
    NMTGAN_REQUIRE(a.epilogue == epi && a.op_dtype == op, "gemm_grouped: problems must share op dtype and epilogue");
    GemmParams p;
    NMTGAN_TRY(fill_params(&a, p));
    if (a.M == 0) continue;
    const bool fits = p.kbytes % BKB == 0 && p.kbytes / BKB >= 4;
    const bool lin_ok = epi != NMTGAN_EPI_LINEAR || (a.n_seg == 1 && a.out_f32[0] != 2 && a.seg_scale[0] == 1.f);
    const int64_t tiles = (int64_t)((p.M + PBM - 1) / PBM) * ((p.N + PBN - 1) / PBN);
    if (!fits || !lin_ok || (min_tiles > 0 && tiles >= min_tiles && !p.add)) {
      NMTGAN_REQUIRE(!p.add, "gemm: the additive CONVMAX form needs K*esize %% 128 == 0 and >= 4 k-tiles");
      NMTGAN_TRY(launch_gemm_impl(&a, st));
      continue;
    }
    ps.push_back(p);
  }
  std::stable_sort(ps.begin(), ps.end(), [](const GemmParams& x, const GemmParams& y) { return x.kbytes > y.kbytes; });
  for (size_t i0 = 0; i0 < ps.size(); i0 += GROUP_MAX) {
    GroupParams gp{};
    gp.n_prob = (int)std::min<size_t>(GROUP_MAX, ps.size() - i0);
    int64_t tiles = 0;
    for (int i = 0; i < gp.n_prob; ++i) {
      const GemmParams& p = ps[i0 + i];
      GroupProb& q = gp.prob[i];
      q.A = p.A; q.W = p.W;
      q.a_group_stride_b = p.a_group_stride_b; q.ldw_b = p.ldw_b; q.k_step = p.k_step; q.k_jump = p.k_jump;
      q.M = p.M; q.N = p.N; q.nkt = p.kbytes / BKB; q.a_group = p.a_group; q.lda_b = (int)p.lda_b; q.k_inner = p.k_inner;
      q.ntn = (p.N + PBN - 1) / PBN;
      q.ntm = (p.M + PBM - 1) / PBM;
      NMTGAN_REQUIRE(p.lda_b < (int64_t)1 << 31, "gemm_grouped: lda too large");
      if (epi == NMTGAN_EPI_LINEAR) {
        q.out = p.out[0]; q.ld_out = p.ldo[0]; q.v0 = p.bias; q.relu = p.relu; q.omode = p.out_f32[0];
      } else {
        q.out = p.pooled; q.ld_out = p.ld_pool; q.v0 = p.scale; q.v1 = p.shift;
        q.add = p.add; q.add_blk = p.add_blk; q.add_rows = p.add_rows; q.add_row0 = p.add_row0;
      }
      tiles += (int64_t)((p.M + PBM - 1) / PBM) * q.ntn;
      NMTGAN_REQUIRE(tiles < (int64_t)1 << 30, "gemm_grouped: too many tiles");
      gp.tile_end[i] = (int)tiles;
    }
    gp.n_tiles = (int)tiles;
    static const int group_m = [] { const char* e = getenv("NMTGAN_GROUP_GM"); return e && atoi(e) > 0 ? atoi(e) : 8; }();
    gp.group_m = group_m;
    prof::Scope prof_scope(prof::GEMM_PROJ, st);
    if (epi == NMTGAN_EPI_LINEAR) {
      NMTGAN_DISPATCH_OP(op, OpT, NMTGAN_TRY((launch_group<OpT, NMTGAN_EPI_LINEAR>(gp, st))));
    } else {
      NMTGAN_DISPATCH_OP(op, OpT, NMTGAN_TRY((launch_group<OpT, EPI_CONVADD>(gp, st))));
    }
  }
  return 0;
}

}  // namespace nmtgan
