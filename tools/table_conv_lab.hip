// Standalone timing lab for the table form of the discriminator's conv + max (not part of the product):
// per width w a table P[v][tap][f] = E[v] . W_w[tap][:, f] turns the conv of a window into a sum of w gathered
// rows; this program times that gather at the benchmark geometry with random tables and tokens.
// hipcc --offload-arch=gfx950 -O3 tools/table_conv_lab.hip -o tools/table_conv_lab && ./tools/table_conv_lab [half] [noS]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int MAXW = 32;
struct Params {
  const void* tab[MAXW];      // [V][w][nfp]
  const float* S[MAXW];       // [B][Pw][w-1][nfp]: sum of the first j taps of window p of y_sample, j = 1..w-1
  int w[MAXW], nf[MAXW], nfp[MAXW], col[MAXW];
  const int32_t* ids;         // [rows][T]
  const int32_t* row_b; const int32_t* row_g;
  float* pooled;              // [rows][Fp]
  int rows, T, Fp, useS;
};

template <typename TT> struct Vec;
template <> struct Vec<float> { static constexpr int N = 4; using L = float4; };
template <> struct Vec<__half> { static constexpr int N = 8; using L = uint4; };
template <typename TT> __device__ __forceinline__ void accum(float* a, const TT* p);
template <> __device__ __forceinline__ void accum<float>(float* a, const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
}
template <> __device__ __forceinline__ void accum<__half>(float* a, const __half* p) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  const __half2* h = reinterpret_cast<const __half2*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) { const float2 f = __half22float2(h[i]); a[2 * i] += f.x; a[2 * i + 1] += f.y; }
}

template <typename TT>
__global__ __launch_bounds__(256) void table_conv_kernel(Params p) {
  constexpr int VN = Vec<TT>::N;
  __shared__ int32_t tok[128];
  __shared__ float red[256 * VN];
  const int i = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
  const int w = p.w[i], nfp = p.nfp[i], T = p.T;
  const int g = p.row_g[s], b = p.row_b[s];
  for (int t = tid; t < T; t += 256) tok[t] = p.ids[(int64_t)s * T + t];
  __syncthreads();
  const int CV = nfp / VN;                       // vector columns
  const int CP = CV < 256 ? CV : 256;
  const int G = 256 / CP;
  const int cidx = tid % CP, slot = tid / CP;
  const TT* tab = (const TT*)p.tab[i];
  const int Pw = T - w + 1;
  const int p_lo = g - w + 1 > 0 ? g - w + 1 : 0;
  for (int c = cidx; c < CV; c += CP) {
    float m[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) m[k] = 0.f;
    if (slot < G) {
      for (int q = p_lo + slot; q < Pw; q += G) {
        float a[VN];
#pragma unroll
        for (int k = 0; k < VN; ++k) a[k] = 0.f;
        int j0 = 0;
        if (p.useS && g > q) {
          j0 = g - q;
          const float* sp = p.S[i] + ((((int64_t)b * Pw + q) * (w - 1)) + (j0 - 1)) * nfp + c * VN;
#pragma unroll
          for (int k = 0; k < VN; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(sp + k);
            a[k] = v.x; a[k + 1] = v.y; a[k + 2] = v.z; a[k + 3] = v.w;
          }
        }
#pragma unroll 8
        for (int j = j0; j < w; ++j)
          accum<TT>(a, tab + ((int64_t)tok[q + j] * w + j) * nfp + c * VN);
#pragma unroll
        for (int k = 0; k < VN; ++k) m[k] = fmaxf(m[k], a[k] * 0.01f + 0.001f);
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
#pragma unroll
      for (int k = 0; k < VN; ++k) {
        const int f = c * VN + k;
        if (f < p.nf[i]) {
          float* o = p.pooled + (int64_t)s * p.Fp + p.col[i] + f;
          *o = fmaxf(*o, m[k]);
        }
      }
    }
  }
}

__global__ void fill_f32(float* p, int64_t n, unsigned seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xffff) * (1.f / 65536.f) - 0.5f;
  }
}
__global__ void fill_f16(__half* p, int64_t n, unsigned seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = __float2half((float)(h & 0xffff) * (1.f / 65536.f) - 0.5f);
  }
}

int main(int argc, char** argv) {
  bool half = false, useS = true;
  int pad = 32;
  for (int a = 1; a < argc; ++a) {
    if (!strcmp(argv[a], "half")) half = true;
    if (!strcmp(argv[a], "noS")) useS = false;
    if (!strncmp(argv[a], "pad=", 4)) pad = atoi(argv[a] + 4);
  }
  const int T = 50, V = 32000, B = 64, R = 16;
  const int es = half ? 2 : 4, padn = half ? 2 * pad : pad;
  Params p{};
  int nw = 0, F = 0;
  int64_t gathered = 0, sread = 0;
  std::vector<int32_t> rb, rg;
  for (int b = 0; b < B; ++b) for (int g = 1; g < T; ++g) for (int r = 0; r < R; ++r) { rb.push_back(b); rg.push_back(g); }
  const int rows = (int)rb.size();
  for (int w = 1; w < T; w += 4, ++nw) {
    p.w[nw] = w; p.nf[nw] = 100 + 10 * w; p.nfp[nw] = (p.nf[nw] + padn - 1) / padn * padn; p.col[nw] = F; F += p.nf[nw];
    const int64_t n = (int64_t)V * w * p.nfp[nw];
    void* t;
    CK(hipMalloc(&t, n * es));
    if (half) fill_f16<<<4096, 256>>>((__half*)t, n, nw); else fill_f32<<<4096, 256>>>((float*)t, n, nw);
    p.tab[nw] = t;
    const int Pw = T - w + 1;
    const int64_t ns = (int64_t)B * Pw * (w > 1 ? w - 1 : 1) * p.nfp[nw];
    float* s;
    CK(hipMalloc(&s, ns * 4));
    fill_f32<<<4096, 256>>>(s, ns, 100 + nw);
    p.S[nw] = s;
    for (int g = 1; g < T; ++g)
      for (int q = g - w + 1 > 0 ? g - w + 1 : 0; q < Pw; ++q) {
        const int j0 = useS && g > q ? g - q : 0;
        gathered += (int64_t)(w - j0) * p.nfp[nw] * es * B * R;
        if (j0) sread += (int64_t)p.nfp[nw] * 4 * B * R;
      }
  }
  const int Fp = (F + 63) / 64 * 64;
  std::vector<int32_t> ids((size_t)rows * T);
  unsigned h = 12345;
  for (auto& x : ids) { h = h * 1664525u + 1013904223u; x = (h >> 8) % V; }
  int32_t *d_ids, *d_rb, *d_rg;
  CK(hipMalloc(&d_ids, ids.size() * 4)); CK(hipMalloc(&d_rb, rows * 4)); CK(hipMalloc(&d_rg, rows * 4));
  CK(hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rb, rb.data(), rows * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rg, rg.data(), rows * 4, hipMemcpyHostToDevice));
  float* pooled;
  CK(hipMalloc(&pooled, (size_t)rows * Fp * 4));
  CK(hipMemset(pooled, 0, (size_t)rows * Fp * 4));
  p.ids = d_ids; p.row_b = d_rb; p.row_g = d_rg; p.pooled = pooled; p.rows = rows; p.T = T; p.Fp = Fp; p.useS = useS;
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // widths in descending cost order would go first; here blockIdx.y = width index as is
  for (int it = 0; it < 4; ++it) {
    CK(hipEventRecord(e0));
    if (half) table_conv_kernel<__half><<<dim3(rows, nw), 256>>>(p); else table_conv_kernel<float><<<dim3(rows, nw), 256>>>(p);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s %s pad %d: rows %d  %.2f ms  gathered %.1f GB (+S %.1f GB)  %.2f TB/s\n", half ? "fp16" : "fp32", useS ? "S" : "noS",
           pad, rows, ms, gathered / 1e9, sread / 1e9, (gathered + sread) / 1e9 / ms);
    fflush(stdout);
  }
  std::vector<float> out(1024);
  CK(hipMemcpy(out.data(), pooled, 4096, hipMemcpyDeviceToHost));
  double cs = 0; for (float x : out) cs += x;
  printf("checksum %.6f\n", cs);
  return 0;
}
