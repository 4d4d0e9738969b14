// Standalone timing lab for the memory pattern of the rollout steps' private-suffix attention (not part of the product):
// row r of a step reads key j = 0..s-1 as two 1 KB pieces, K[j][r] and V[j][r], from step-major caches.  Two forms of the
// same reads with trivial arithmetic: (a) one short-lived wave per row, batches of four keys double-buffered in
// registers (the shipped kernel's structure); (b) persistent waves that walk many rows as ONE stream of batches, so the
// loads of a row's first batch are in flight while the previous row's last batch is consumed.
// hipcc --offload-arch=gfx950 -O3 tools/attn_lab.hip -o tools/attn_lab && ./tools/attn_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int KB = 4, D = 512;     // keys per batch, fp16 elements per key vector (1 KB)

struct P { const char* K; const char* V; char* out; int M, s; int64_t slab; int64_t rstride; };   // key j of row r at r * rstride + j * slab

__device__ __forceinline__ void eat(uint4& acc, const uint4& k, const uint4& v) {
  acc.x += k.x ^ v.y; acc.y += k.y ^ v.z; acc.z += k.z ^ v.w; acc.w += k.w ^ v.x;
}

__global__ __launch_bounds__(256) void per_row(P p) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const char* k = p.K + (int64_t)row * p.rstride + lane * 16;
  const char* v = p.V + (int64_t)row * p.rstride + lane * 16;
  uint4 kc[KB], vc[KB], kn[KB], vn[KB], acc = make_uint4(0, 0, 0, 0);
  const int nb = (p.s + KB - 1) / KB;
  auto fetch = [&](int b, uint4 (&kr)[KB], uint4 (&vr)[KB]) {
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int64_t off = (int64_t)min(b * KB + u, p.s - 1) * p.slab;
      kr[u] = *reinterpret_cast<const uint4*>(k + off);
      vr[u] = *reinterpret_cast<const uint4*>(v + off);
    }
  };
  auto consume = [&](const uint4 (&kr)[KB], const uint4 (&vr)[KB]) {
#pragma unroll
    for (int u = 0; u < KB; ++u) eat(acc, kr[u], vr[u]);
  };
  fetch(0, kc, vc);
  for (int b = 0; b < nb; b += 2) {
    const bool second = b + 1 < nb;
    if (second) fetch(b + 1, kn, vn);
    consume(kc, vc);
    if (second) {
      if (b + 2 < nb) fetch(b + 2, kc, vc);
      consume(kn, vn);
    }
  }
  *reinterpret_cast<uint4*>(p.out + (int64_t)row * 1024 + lane * 16) = acc;
}

// persistent: wave w of W walks rows w, w + W, ...; batch t of the wave = (row index t / nb, batch t % nb)
__global__ __launch_bounds__(256) void persistent(P p) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), W = gridDim.x * 4;
  const int nb = (p.s + KB - 1) / KB;
  const int my_rows = (p.M - w + W - 1) / W;
  if (my_rows <= 0) return;
  const int total = my_rows * nb;
  uint4 kc[KB], vc[KB], kn[KB], vn[KB], acc = make_uint4(0, 0, 0, 0);
  int f_row = w, f_b = 0;                       // next batch to fetch
  auto fetch = [&](uint4 (&kr)[KB], uint4 (&vr)[KB]) {
    const char* k = p.K + (int64_t)f_row * p.rstride + lane * 16;
    const char* v = p.V + (int64_t)f_row * p.rstride + lane * 16;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int64_t off = (int64_t)min(f_b * KB + u, p.s - 1) * p.slab;
      kr[u] = *reinterpret_cast<const uint4*>(k + off);
      vr[u] = *reinterpret_cast<const uint4*>(v + off);
    }
    if (++f_b == nb) { f_b = 0; f_row += W; }
  };
  int c_row = w, c_b = 0;                       // batch being consumed
  auto consume = [&](const uint4 (&kr)[KB], const uint4 (&vr)[KB]) {
#pragma unroll
    for (int u = 0; u < KB; ++u) eat(acc, kr[u], vr[u]);
    if (++c_b == nb) {
      *reinterpret_cast<uint4*>(p.out + (int64_t)c_row * 1024 + lane * 16) = acc;
      acc = make_uint4(0, 0, 0, 0);
      c_b = 0; c_row += W;
    }
  };
  fetch(kc, vc);
  for (int t = 0; t < total; t += 2) {
    const bool second = t + 1 < total;
    if (second) fetch(kn, vn);
    consume(kc, vc);
    if (second) {
      if (t + 2 < total) fetch(kc, vc);
      consume(kn, vn);
    }
  }
}

int main() {
  const int Mmax = 50176, Smax = 48;
  char *K, *V, *out, *KV;
  const size_t plane = (size_t)Smax * Mmax * 1024;            // bytes of one of K / V
  CK(hipMalloc(&K, plane));
  CK(hipMalloc(&V, plane));
  CK(hipMalloc(&KV, 2 * plane));                              // layout 2: K and V of a key side by side
  CK(hipMemset(KV, 3, 2 * plane));
  CK(hipMalloc(&out, (size_t)Mmax * 1024));
  CK(hipMemset(K, 1, (size_t)Smax * Mmax * 1024));
  CK(hipMemset(V, 2, (size_t)Smax * Mmax * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int shapes[][2] = {{47104, 3}, {41984, 8}, {33792, 16}, {25600, 24}, {17408, 32}, {9216, 40}, {3072, 46}};
  for (auto& sh : shapes) {
    const double gb = 2.0 * sh[0] * sh[1] * 1024 / 1e9;
    for (int layout = 0; layout < 3; ++layout) {
    // 0: step-major [key][row] (shipped), 1: row-major [row][key] (a row's keys contiguous), 2: row-major with K and V of
    // a key side by side in ONE array (2 KB per key)
    P p{K, V, out, sh[0], sh[1], (int64_t)sh[0] * 1024, 1024};
    if (layout == 1) { p.slab = 1024; p.rstride = (int64_t)Smax * 1024; }
    size_t cap_k = plane, cap_v = plane;                      // bytes addressable from p.K / p.V
    if (layout == 2) { p.slab = 2048; p.rstride = (int64_t)Smax * 2048; p.K = KV; p.V = KV + 1024; cap_k = 2 * plane; cap_v = 2 * plane - 1024; }
    const size_t reach = (size_t)(p.M - 1) * p.rstride + (size_t)(p.s - 1) * p.slab + 1024;   // one past the last byte read
    if (reach > cap_k || reach > cap_v || p.s > Smax || p.M > Mmax) { printf("shape does not fit the buffers\n"); return 1; }
    for (int form = 0; form < (layout ? 1 : 3); ++form) {
      float best = 1e9;
      for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0));
        if (form == 0) per_row<<<(p.M + 3) / 4, 256>>>(p);
        else persistent<<<256 * (form == 1 ? 4 : 8), 256>>>(p);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      printf("rows %5d keys %2d  layout %d  %-22s %7.1f us  %5.2f TB/s\n", sh[0], sh[1], layout,
             form == 0 ? "wave per row" : (form == 1 ? "persistent 4 wg/CU" : "persistent 8 wg/CU"), best * 1e3, gb / best);
    }
    }
    fflush(stdout);
  }
  return 0;
}
