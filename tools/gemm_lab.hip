// Standalone ablation lab for the persistent pipelined GEMM main loop (not part of the product).
// hipcc --offload-arch=gfx950 -O3 tools/gemm_lab.hip -o tools/gemm_lab && ./tools/gemm_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int BKB = 128, GROUP_M = 16;
constexpr int PBM = 256, PBN = 128, PNW = 8, PSTAGE = (PBM + PBN) * BKB, PNINST = (PBM + PBN) / 8 / PNW;
struct P { const char* A; const char* W; float* C; int M, N, K; int64_t lda_b, ldw_b; int kbytes, ntm, ntn, gm; };
enum { A_NOREAD = 1, A_NOGLDS = 2, A_NOBARRIER = 4, A_NOEPI = 8, A_NOMFMA = 16 };

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int ABL>
__global__ __launch_bounds__(512) void k(P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = p.ntm * p.ntn, G = gridDim.x, nkt = p.kbytes / BKB;
  const int my_tiles = (nwg - (int)blockIdx.x + G - 1) / G;
  const int total = my_tiles * nkt;
  auto tile_of = [&](int L, int& tm, int& tn) {
    const int xcd = L & 7, q = nwg >> 3, r = nwg & 7;
    const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    const int gs = p.gm * p.ntn, first_m = (idx / gs) * p.gm, gsz = min(p.ntm - first_m, p.gm);
    tm = first_m + (idx % gs) % gsz; tn = (idx % gs) / gsz;
  };
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);
  const char* src[PNINST];
  auto set_src = [&](int L) {
    int tm, tn; tile_of(L, tm, tn);
#pragma unroll
    for (int j = 0; j < PNINST; ++j) {
      const int rho = 8 * (wave + j * PNW) + (lane >> 3);
      const int lc = (lane & 7) ^ ((rho >> 1) & 7);
      if (rho < PBM) src[j] = p.A + (int64_t)min(tm * PBM + rho, p.M - 1) * p.lda_b + lc * 16;
      else src[j] = p.W + (int64_t)min(tn * PBN + rho - PBM, p.N - 1) * p.ldw_b + lc * 16;
    }
  };
  int is_kt = 0, is_tile = 0, is_buf = 0;
  auto issue = [&]() {
    if (is_kt == 0) set_src((int)blockIdx.x + is_tile * G);
    if (!(ABL & A_NOGLDS)) {
#pragma unroll
      for (int j = 0; j < PNINST; ++j) glds16(src[j] + (int64_t)is_kt * BKB, lds_base + is_buf * PSTAGE + (wave + j * PNW) * 1024);
    }
    if (++is_kt == nkt) { is_kt = 0; ++is_tile; }
    is_buf = is_buf == 2 ? 0 : is_buf + 1;
  };
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
  const int a_rd = (wm * 64 + fr) * BKB, b_rd = (PBM + wn * 64 + fr) * BKB;
  issue();
  if (total > 1) issue();
  int ck = 0, ctile = 0, cbuf = 0;
  uint4 cf = make_uint4(lane, lane * 3, lane * 7, 0x3f803f80);
  for (int it = 0; it < total; ++it) {
    if (!(ABL & A_NOBARRIER)) {
      if (it + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PNINST) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (it + 2 < total) issue();
    const unsigned char* base = smem + cbuf * PSTAGE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pc = ((2 * q + fh) ^ fsw) << 4;
      uint4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (ABL & A_NOREAD) { fa[i] = cf; fb[i] = cf; asm volatile("" : "+v"(fa[i].x), "+v"(fb[i].y)); }
        else {
          fa[i] = *reinterpret_cast<const uint4*>(base + a_rd + i * 32 * BKB + pc);
          fb[i] = *reinterpret_cast<const uint4*>(base + b_rd + i * 32 * BKB + pc);
        }
      }
      if (!(ABL & A_NOMFMA)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc[i][0][q] += __uint_as_float(fa[i].x); acc[0][i][q + 4] += __uint_as_float(fb[i].y); }
      }
    }
    cbuf = cbuf == 2 ? 0 : cbuf + 1;
    if (++ck == nkt) {
      int tm, tn; tile_of((int)blockIdx.x + ctile * G, tm, tn);
      const int m0 = tm * PBM + wm * 64, n0 = tn * PBN + wn * 64;
      if (ABL & A_NOEPI) {
        float s = 0; for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        if (s == 12345.678f) p.C[(int64_t)m0 * p.N + n0 + lane] = s;
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + j * 32 + fr;
          if (n < p.N)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const int m = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (m < p.M) p.C[(int64_t)m * p.N + n] = acc[i][j][e];
              }
        }
      }
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      ck = 0; ++ctile;
    }
  }
}
template <int ABL> float run(P p, int iters) {
  hipFuncSetAttribute((const void*)&k<ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * PSTAGE);
  const int nwg = p.ntm * p.ntn, grid = nwg < 256 ? nwg : 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<ABL>, dim3(grid), dim3(512), 3 * PSTAGE, 0, p);
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k<ABL>, dim3(grid), dim3(512), 3 * PSTAGE, 0, p);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / iters;
}
int main() {
  struct S { const char* n; int M, N, K; } shapes[] = {{"ffn1 K512", 49152, 2048, 512}, {"proj K512", 49152, 512, 512},
      {"logits", 16384, 32000, 512}, {"bigK", 106496, 384, 12800}, {"ffn2 K2048", 49152, 512, 2048}};
  for (auto& s : shapes) {
    const int pad = getenv("LAB_PAD") ? atoi(getenv("LAB_PAD")) : 0;
    P p{}; p.M = s.M; p.N = s.N; p.K = s.K; p.lda_b = p.ldw_b = (int64_t)s.K * 2 + pad; p.kbytes = s.K * 2;
    p.ntm = (s.M + PBM - 1) / PBM; p.ntn = (s.N + PBN - 1) / PBN;
    std::vector<uint16_t> h((size_t)s.M * (s.K + pad / 2));
    for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
    void *A, *W, *C; hipMalloc(&A, h.size() * 2); hipMalloc(&W, (size_t)s.N * (s.K * 2 + pad)); hipMalloc(&C, (size_t)s.M * s.N * 4);
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)s.N * (s.K * 2 + pad) < h.size() * 2 ? (size_t)s.N * (s.K * 2 + pad) : h.size() * 2, hipMemcpyHostToDevice);
    p.A = (char*)A; p.W = (char*)W; p.C = (float*)C;
    double fl = 2.0 * s.M * s.N * s.K;
    float t;
#define R(ABL, name) t = run<ABL>(p, 10); printf("%-12s gm=%-2d %-20s %8.3f ms %8.1f TF\n", s.n, p.gm, name, t, fl / t / 1e9);
    p.gm = 4; R(0, "full") R(A_NOMFMA, "no mfma") R(A_NOMFMA | A_NOREAD, "glds only") R(A_NOGLDS, "no glds")
    hipFree(A); hipFree(W); hipFree(C);
  }
  return 0;
}
