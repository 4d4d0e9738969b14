// Fused logits + categorical sampling for gfx950: the rollout step's [rows, V] logits never reach HBM.
// Replaces top() -> softmax -> log -> tf.multinomial of the rollout loop (model.py:226-241) with the
// same CDF semantics as sampler.hip (fp64 running sum of exp(l - max l) in index order, first index
// whose CDF exceeds u * total, clamp to V-1):
//   1. launch_gemm_stats (gemm.hip, transposed logits GEMM): stats[vtile][row] = (max, sum exp(l - max))
//      over each 64-entry vocabulary tile - 8 bytes instead of 256 per (tile, row).
//   2. select_kernel: 16 lanes per row read the row's tile stats ONCE into registers, combine them into
//      the tile-granular fp64 CDF and find the tile where u * total is crossed; counts rows per tile.
//   3. worklist_kernel + scatter_kernel: counting sort of the rows by selected tile (no host sync: the
//      work list has a static upper bound of n_vtile + rows/32 entries).
//   4. recompute_kernel: one wave per (tile, <= 32 rows) recomputes exactly those 64 x 32 logits on the
//      matrix cores (operands straight from global: the softmax table tile is shared by the 32 rows)
//      and finishes the upper_bound inside the tile.
// A miss caused by the ~1e-6 mismatch between the two evaluations of a logit takes the tile's last entry.
// HBM bytes per sampled token: ~2 x 8 x V/64 (stats) + 2 KB, against 2 x 4 x V for materialised logits.
#include <mutex>

#include "common.h"
#include "kernels.h"

// Scratch of nmtgan_logits_sample_static: the workspace of the fused logits + sampling path (per-tile softmax
// statistics, selection arrays, counters, work list) for callers that may not allocate.  Every GPU the library runs
// on holds its own copy of this array until the process ends; the name below is only the host-side handle of the
// symbol - kernels get the per-device address hipGetSymbolAddress returns.
static __device__ __align__(256) unsigned char scratch[NMTGAN_LOGITS_SAMPLE_STATIC_BYTES];

namespace nmtgan {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

struct FusedBufs {
  float2* stats;           // [n_vtile][M]
  double* sel_pre;         // [M] CDF mass before the selected tile
  double* sel_target;      // [M] u * total
  float* sel_max;          // [M] row max
  int32_t* sel_tile;       // [M]
  int32_t* count;          // [1024] rows per tile
  int32_t* tile_off;       // [1024]
  int32_t* cursor;         // [1024]
  int32_t* n_work;         // [1]
  int2* work;              // [n_vtile + M/32 + 1] (tile, chunk)
  int32_t* order;          // [M] rows sorted by tile
};

static size_t carve_fused(char* base, int M, int V, FusedBufs& b) {
  const size_t nvt = V / 64;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
  b.stats = (float2*)take(nvt * M * sizeof(float2));
  b.sel_pre = (double*)take((size_t)M * sizeof(double));
  b.sel_target = (double*)take((size_t)M * sizeof(double));
  b.sel_max = (float*)take((size_t)M * sizeof(float));
  b.sel_tile = (int32_t*)take((size_t)M * sizeof(int32_t));
  b.count = (int32_t*)take(1024 * sizeof(int32_t));
  b.tile_off = (int32_t*)take(1024 * sizeof(int32_t));
  b.cursor = (int32_t*)take(1024 * sizeof(int32_t));
  b.n_work = (int32_t*)take(sizeof(int32_t));
  b.work = (int2*)take((nvt + M / 32 + 1) * sizeof(int2));
  b.order = (int32_t*)take((size_t)M * sizeof(int32_t));
  return off;
}

struct SelectParams {
  FusedBufs b; int n_vtile; int tpl; int M;
  const float* u; const int64_t* u_off; int64_t u_add;
};

// A workgroup = 16 rows x 16 tile groups: wave w, lane (ql = lane >> 4, rr = lane & 15) owns tiles [q tpl, (q + 1) tpl),
// q = 4 w + ql, of row 16 blockIdx.x + rr.  stats is [tile][row], so one load instruction of a wave reads four tiles x
// 16 consecutive rows = four whole 128-byte lines.  Row max, fp64 mass before each group and the total go across the
// four waves through LDS; sums are taken in tile order inside a lane, in group order across lanes.  At 49 k rows the
// kernel takes 132 us: 85 for the 200 MB of statistics (500 strided streams of one line per row block), 44 for the
// fp64 arithmetic and the uniform gather, 13 for the per-tile row counters.
template <int TPL>
__global__ __launch_bounds__(256) void select_kernel(SelectParams p) {
  __shared__ float s_max[4][16];
  __shared__ double s_tot[4][16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ql = lane >> 4, rr = lane & 15;
  const int q = 4 * w + ql;
  const int r = blockIdx.x * 16 + rr;
  const bool ok = r < p.M;
  const int rc = ok ? r : p.M - 1;
  const int t0 = q * p.tpl;
  float2 sv[TPL];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < TPL; ++i) {
    const int t = t0 + i;
    const bool valid = i < p.tpl && t < p.n_vtile;
    sv[i] = valid ? p.b.stats[(int64_t)t * p.M + rc] : make_float2(-INFINITY, 0.f);
    mx = fmaxf(mx, sv[i].x);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (ql == 0) s_max[w][rr] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_max[0][rr], s_max[1][rr]), fmaxf(s_max[2][rr], s_max[3][rr]));
  double sum = 0.0;
#pragma unroll
  for (int i = 0; i < TPL; ++i) sum += (double)sv[i].y * (double)fexp(sv[i].x - mx);
  double incl = sum;                               // inclusive over the wave's four groups of this row
  {
    const double a = __shfl_up(incl, 16, 64);
    if (lane >= 16) incl += a;
    const double b = __shfl_up(incl, 32, 64);
    if (lane >= 32) incl += b;
  }
  double excl = __shfl_up(incl, 16, 64);           // mass of the wave's earlier groups
  if (ql == 0) excl = 0.0;
  if (ql == 3) s_tot[w][rr] = incl;
  __syncthreads();
  const double w0 = s_tot[0][rr], w1 = s_tot[1][rr], w2 = s_tot[2][rr], w3 = s_tot[3][rr];
  const double before = w == 0 ? 0.0 : (w == 1 ? w0 : (w == 2 ? w0 + w1 : (w0 + w1) + w2));   // mass of earlier waves
  const double total = ((w0 + w1) + w2) + w3;
  incl += before;
  excl += before;                                  // mass before this lane's first tile
  const double target = (double)p.u[(p.u_off ? p.u_off[rc] : (int64_t)rc) + p.u_add] * total;
  // the first group (in q order) whose inclusive mass exceeds the target; none: the group that holds the last tile
  const bool c0 = w0 > target, c1 = w0 + w1 > target, c2 = (w0 + w1) + w2 > target, c3 = total > target;
  const int wsel = c0 ? 0 : (c1 ? 1 : (c2 ? 2 : (c3 ? 3 : -1)));
  const int last = p.n_vtile - 1;
  bool mine;
  if (wsel < 0) {
    mine = t0 <= last && last < t0 + p.tpl;
  } else {
    const unsigned long long bal = __ballot(incl > target) & (0x0001000100010001ull << rr);
    mine = w == wsel && bal != 0 && lane == __ffsll((long long)bal) - 1;
  }
  if (!mine || !ok) return;
  double run = excl, pre = excl, c_last = 0.0;
  int sel = -1;
#pragma unroll
  for (int i = 0; i < TPL; ++i) {
    const double c = (double)sv[i].y * (double)fexp(sv[i].x - mx);
    if (sel < 0 && i < p.tpl && t0 + i < p.n_vtile) {
      if (run + c > target) { sel = t0 + i; pre = run; }
      else { run += c; c_last = c; }
    }
  }
  // no tile of the lane crosses: u * total is not below the final CDF value (the lane holds the last tile), or the
  // running sum rounds differently from the scanned one at the group's end - the lane's last tile either way
  if (sel < 0) { sel = min(t0 + p.tpl, p.n_vtile) - 1; pre = run - c_last; }
  p.b.sel_tile[r] = sel;
  p.b.sel_max[r] = mx;
  p.b.sel_pre[r] = pre;
  p.b.sel_target[r] = target;
  atomicAdd(&p.b.count[sel], 1);
}

// one 1024-thread block: exclusive scans of rows per tile and of 32-row chunks per tile
__global__ __launch_bounds__(1024) void worklist_kernel(FusedBufs b, int n_vtile) {
  __shared__ int s_c[1024], s_k[1024];
  const int t = threadIdx.x;
  const int c = t < n_vtile ? b.count[t] : 0;
  const int k = (c + 31) >> 5;
  s_c[t] = c; s_k[t] = k;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int ac = t >= o ? s_c[t - o] : 0, ak = t >= o ? s_k[t - o] : 0;
    __syncthreads();
    s_c[t] += ac; s_k[t] += ak;
    __syncthreads();
  }
  if (t < n_vtile) {
    b.tile_off[t] = s_c[t] - c;
    b.cursor[t] = 0;
    const int w0 = s_k[t] - k;
    for (int j = 0; j < k; ++j) b.work[w0 + j] = make_int2(t, j);
  }
  if (t == 1023) *b.n_work = s_k[t];
}

__global__ __launch_bounds__(256) void scatter_kernel(FusedBufs b, int M) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= M) return;
  const int t = b.sel_tile[r];
  b.order[b.tile_off[t] + atomicAdd(&b.cursor[t], 1)] = r;
}

struct RecomputeParams {
  FusedBufs b; const void* x; const void* w; int d; int V; int32_t* out;
};

template <typename OpT>
__global__ __launch_bounds__(256) void recompute_kernel(RecomputeParams p) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= *p.b.n_work) return;
  const int2 wk = p.b.work[wid];
  const int tile = wk.x;
  const int nrow = min(32, p.b.count[tile] - 32 * wk.y);
  const int n = lane & 31, h2 = lane >> 5;
  const int row = p.b.order[p.b.tile_off[tile] + 32 * wk.y + min(n, nrow - 1)];
  const int d = p.d;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  if constexpr (sizeof(OpT) == 2) {
    // S^T[vocab, row]: A = table rows (lane = vocab entry n of the 32-tile), B = hidden rows; k = 8*h2 + j
    const OpT* wa = reinterpret_cast<const OpT*>(p.w) + ((int64_t)tile * 64 + n) * d + 8 * h2;
    const OpT* xb = reinterpret_cast<const OpT*>(p.x) + (int64_t)row * d + 8 * h2;
    for (int k1 = 0; k1 < d; k1 += 64)            // d * 2 bytes is a multiple of 128
#pragma unroll
    for (int k0 = k1; k0 < k1 + 64; k0 += 16) {
      const uint4 a0 = *reinterpret_cast<const uint4*>(wa + k0);
      const uint4 a1 = *reinterpret_cast<const uint4*>(wa + (int64_t)32 * d + k0);
      const uint4 bb = *reinterpret_cast<const uint4*>(xb + k0);
      acc[0] = mfma16<OpT>(a0, bb, acc[0]);
      acc[1] = mfma16<OpT>(a1, bb, acc[1]);
    }
  } else {
    // fp32 MFMA takes k = h2 per instruction; a 16-byte load per lane feeds 4 of them (k permuted the
    // same way for both operands, which leaves every dot product unchanged)
    const float* wa = reinterpret_cast<const float*>(p.w) + ((int64_t)tile * 64 + n) * d + 4 * h2;
    const float* xb = reinterpret_cast<const float*>(p.x) + (int64_t)row * d + 4 * h2;
    for (int k1 = 0; k1 < d; k1 += 32)            // d * 4 bytes is a multiple of 128
#pragma unroll
    for (int k0 = k1; k0 < k1 + 32; k0 += 8) {
      const float4 a0 = *reinterpret_cast<const float4*>(wa + k0);
      const float4 a1 = *reinterpret_cast<const float4*>(wa + (int64_t)32 * d + k0);
      const float4 bb = *reinterpret_cast<const float4*>(xb + k0);
      const float a0v[4] = {a0.x, a0.y, a0.z, a0.w}, a1v[4] = {a1.x, a1.y, a1.z, a1.w}, bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[i], bv[i], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[i], bv[i], acc[1], 0, 0, 0);
      }
    }
  }
  // this lane: query row n; acc[t][e] = logit of tile entry 32t + 8(e>>2) + 4*h2 + (e&3)
  const float mx = p.b.sel_max[row];
  const double target = p.b.sel_target[row];
  float ev[2][16];
  double qs[2][4], pq[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ev[t][4 * g + j] = fexp(acc[t][4 * g + j] - mx);
        s += (double)ev[t][4 * g + j];
      }
      qs[t][g] = s;
      pq[t][g] = __shfl_xor(s, 32, 64);
    }
  // both half-waves walk the 16 quads of 4 consecutive entries in vocabulary order
  double run = p.b.sel_pre[row];
  bool found = false;
  int token = -1;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const double s = hh == h2 ? qs[t][g] : pq[t][g];
        if (!found) {
          if (run + s > target) {
            found = true;
            if (hh == h2) {
              double r2 = run;
              int j3 = 3;
              bool f2 = false;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                r2 += (double)ev[t][4 * g + j];
                if (!f2 && r2 > target) { f2 = true; j3 = j; }
              }
              token = 32 * t + 8 * g + 4 * hh + j3;
            }
          } else {
            run += s;
          }
        }
      }
  if (!found && h2 == 1) token = 63;
  if (token >= 0 && n < nrow) p.out[row] = min(tile * 64 + token, p.V - 1);
}

// Address of `scratch` on the current device, looked up once per device (with that device current).  The lookup
// may load the code object, which is not allowed while a stream is being captured into a HIP graph: a first call on
// a device that arrives under capture is refused with a message (one warm call outside the capture resolves it).
static int static_scratch_ptr(void** out, hipStream_t st) {
  static std::mutex mu;
  static void* ptrs[64];
  static bool have[64];
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) NMTGAN_FAIL(-3, "logits_sample_static: no current device");
  // every GPU holds its own copy of the array: the stream must belong to the device whose address is handed out
  hipDevice_t sdev = -1;
  if (st != nullptr && hipStreamGetDevice(st, &sdev) == hipSuccess)
    NMTGAN_REQUIRE((int)sdev == dev, "logits_sample_static: stream belongs to device %d, the current device is %d",
                   (int)sdev, dev);
  std::lock_guard<std::mutex> lock(mu);
  if (!have[dev]) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st != nullptr && hipStreamIsCapturing(st, &cs) == hipSuccess)
      NMTGAN_REQUIRE(cs == hipStreamCaptureStatusNone,
                     "logits_sample_static: first call on device %d arrived under stream capture; call it once "
                     "outside the capture first (the scratch address is resolved on the first call)", dev);
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr)
      NMTGAN_FAIL(-3, "logits_sample_static: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
    ptrs[dev] = p;
    have[dev] = true;
  }
  *out = ptrs[dev];
  return 0;
}

// Same kernels and results as launch_logits_sample; the workspace is the per-device static array above, so the
// call allocates nothing.  Calls that overlap on one GPU share the array and race.
int launch_logits_sample_static(int op_dtype, const void* x, const void* w, int M, int V, int d, const float* u,
                                const int64_t* u_off, int64_t u_add, int32_t* out, hipStream_t st) {
  if (M == 0) return 0;
  NMTGAN_REQUIRE(M > 0 && V > 0 && d > 0, "logits_sample_static: bad shape M=%d V=%d d=%d", M, V, d);
  NMTGAN_REQUIRE(valid_op_dtype(op_dtype) && gemm_stats_supported(op_dtype, V, d),
                 "logits_sample_static: unsupported shape V=%d d=%d (V %% 64, d*esize %% 128, >= 4 k-tiles)", V, d);
  const size_t need = logits_sample_workspace(M, V);
  NMTGAN_REQUIRE(need <= (size_t)NMTGAN_LOGITS_SAMPLE_STATIC_BYTES,
                 "logits_sample_static: M=%d rows x V=%d needs %zu bytes of scratch, the static buffer holds %zu "
                 "(use nmtgan_logits_sample with a caller workspace, or fewer rows per call)",
                 M, V, need, (size_t)NMTGAN_LOGITS_SAMPLE_STATIC_BYTES);
  void* ws = nullptr;
  NMTGAN_TRY(static_scratch_ptr(&ws, st));
  return launch_logits_sample(op_dtype, x, w, M, V, d, u, u_off, u_add, out, ws, st);
}

size_t logits_sample_workspace(int M, int V) {
  FusedBufs b;
  return carve_fused(nullptr, M, V, b);
}

int launch_logits_sample(int op_dtype, const void* x, const void* w, int M, int V, int d, const float* u,
                         const int64_t* u_off, int64_t u_add, int32_t* out, void* ws, hipStream_t st) {
  if (M == 0) return 0;
  NMTGAN_REQUIRE(x && w && u && out && ws && M > 0, "logits_sample: bad arguments");
  NMTGAN_REQUIRE(gemm_stats_supported(op_dtype, V, d), "logits_sample: unsupported shape V=%d d=%d", V, d);
  const int nvt = V / 64;
  FusedBufs b;
  carve_fused((char*)ws, M, V, b);
  NMTGAN_TRY(launch_gemm_stats(op_dtype, w, V, x, M, d, b.stats, st));
  prof::Scope prof_scope(prof::SAMPLER, st);
  NMTGAN_TRY(launch_zero(b.count, 1024 * sizeof(int32_t), st));
  const int tpl = (nvt + 15) / 16;
  SelectParams sp{b, nvt, tpl, M, u, u_off, u_add};
  const dim3 sgrid((M + 15) / 16);
  if (tpl <= 8) hipLaunchKernelGGL(select_kernel<8>, sgrid, dim3(256), 0, st, sp);
  else if (tpl <= 32) hipLaunchKernelGGL(select_kernel<32>, sgrid, dim3(256), 0, st, sp);
  else hipLaunchKernelGGL(select_kernel<64>, sgrid, dim3(256), 0, st, sp);
  NMTGAN_CHECK_LAUNCH("sample_select");
  hipLaunchKernelGGL(worklist_kernel, dim3(1), dim3(1024), 0, st, b, nvt);
  NMTGAN_CHECK_LAUNCH("sample_worklist");
  hipLaunchKernelGGL(scatter_kernel, dim3((M + 255) / 256), dim3(256), 0, st, b, M);
  NMTGAN_CHECK_LAUNCH("sample_scatter");
  RecomputeParams rp{b, x, w, d, V, out};
  const dim3 rgrid((nvt + M / 32 + 1 + 3) / 4);
  NMTGAN_DISPATCH_OP(op_dtype, OpT, hipLaunchKernelGGL(recompute_kernel<OpT>, rgrid, dim3(256), 0, st, rp));
  NMTGAN_CHECK_LAUNCH("sample_recompute");
  return 0;
}

}  // namespace nmtgan
