// Internal helpers shared by the gfx950 kernels of libnmtgan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/nmtgan.h"

namespace nmtgan {

// thread-local error text behind nmtgan_last_error()
void set_error(const char* fmt, ...);
#define NMTGAN_FAIL(code, ...)        \
  do {                                \
    ::nmtgan::set_error(__VA_ARGS__); \
    return (code);                    \
  } while (0)
#define NMTGAN_REQUIRE(cond, ...) \
  do {                            \
    if (!(cond)) NMTGAN_FAIL(-2, __VA_ARGS__); \
  } while (0)
#define NMTGAN_CHECK_LAUNCH(name)                                          \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) NMTGAN_FAIL(-3, "%s launch failed: %s", name, hipGetErrorString(e__)); \
  } while (0)
#define NMTGAN_TRY(expr)     \
  do {                       \
    int rc__ = (expr);       \
    if (rc__ != 0) return rc__; \
  } while (0)

// Optional per-launch HIP-event timing (bench.py's roofline leg; off in the product path).
namespace prof {
enum Cls {
  OTHER = 0, GEMM_PROJ, GEMM_FFN, GEMM_LOGITS, GEMM_CONV, GEMM_HIGHWAY, ATTN_SELF, ATTN_CROSS, ATTN_PREFILL,
  SAMPLER, ADDLN, EMBED, DISC_SMALL, ATTN_SELF_PRIVATE, SAMPLER_ROWS, N_CLS
};
extern bool g_on;
extern thread_local int g_hint;          // class of the next launches, set by the pipelines
void begin(int cls, hipStream_t st);
void end(hipStream_t st);
struct Scope {
  hipStream_t st; bool active;
  Scope(int default_cls, hipStream_t s) : st(s), active(g_on) {
    if (active) begin(g_hint >= 0 ? g_hint : default_cls, st);
  }
  ~Scope() { if (active) end(st); }
};
struct Hint {
  int prev;
  explicit Hint(int c) : prev(g_hint) { g_hint = c; }
  ~Hint() { g_hint = prev; }
};
}  // namespace prof

typedef unsigned short bf16_t;  // raw bfloat16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even f32 -> bf16; NaN stays NaN (hipcc lowers the cast to v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}

template <typename T> struct OpTraits;
template <> struct OpTraits<float> {
  static constexpr int kDtype = NMTGAN_F32;
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct OpTraits<bf16_t> {
  static constexpr int kDtype = NMTGAN_BF16;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// Load N consecutive elements (N in {1,2,4,8,16}) as floats with the widest aligned access.
template <typename T, int N> struct VecIO;
template <int N> struct VecIO<float, N> {
  __device__ static __forceinline__ void load(const float* p, float (&o)[N]) {
    if constexpr (N >= 4) {
#pragma unroll
      for (int i = 0; i < N / 4; ++i) {
        float4 v = reinterpret_cast<const float4*>(p)[i];
        o[4 * i] = v.x; o[4 * i + 1] = v.y; o[4 * i + 2] = v.z; o[4 * i + 3] = v.w;
      }
    } else if constexpr (N == 2) {
      float2 v = *reinterpret_cast<const float2*>(p);
      o[0] = v.x; o[1] = v.y;
    } else {
      o[0] = p[0];
    }
  }
  __device__ static __forceinline__ void store(float* p, const float (&o)[N]) {
    if constexpr (N >= 4) {
#pragma unroll
      for (int i = 0; i < N / 4; ++i)
        reinterpret_cast<float4*>(p)[i] = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    } else if constexpr (N == 2) {
      *reinterpret_cast<float2*>(p) = make_float2(o[0], o[1]);
    } else {
      p[0] = o[0];
    }
  }
};
template <int N> struct VecIO<bf16_t, N> {
  __device__ static __forceinline__ void load(const bf16_t* p, float (&o)[N]) {
    if constexpr (N >= 8) {
#pragma unroll
      for (int i = 0; i < N / 8; ++i) {
        uint4 v = reinterpret_cast<const uint4*>(p)[i];
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[8 * i + 2 * j] = __uint_as_float(w[j] << 16);
          o[8 * i + 2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
      }
    } else if constexpr (N == 4) {
      uint2 v = *reinterpret_cast<const uint2*>(p);
      o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
      o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    } else if constexpr (N == 2) {
      unsigned v = *reinterpret_cast<const unsigned*>(p);
      o[0] = __uint_as_float(v << 16); o[1] = __uint_as_float(v & 0xffff0000u);
    } else {
      o[0] = bf16_to_f32(p[0]);
    }
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float (&o)[N]) {
    if constexpr (N >= 8) {
#pragma unroll
      for (int i = 0; i < N / 8; ++i) {
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          w[j] = (unsigned)f32_to_bf16(o[8 * i + 2 * j]) | ((unsigned)f32_to_bf16(o[8 * i + 2 * j + 1]) << 16);
        reinterpret_cast<uint4*>(p)[i] = make_uint4(w[0], w[1], w[2], w[3]);
      }
    } else if constexpr (N == 4) {
      unsigned a = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16);
      unsigned b = (unsigned)f32_to_bf16(o[2]) | ((unsigned)f32_to_bf16(o[3]) << 16);
      *reinterpret_cast<uint2*>(p) = make_uint2(a, b);
    } else if constexpr (N == 2) {
      *reinterpret_cast<unsigned*>(p) = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16);
    } else {
      p[0] = f32_to_bf16(o[0]);
    }
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int dtype_size(int dt) { return dt == NMTGAN_F32 ? 4 : 2; }

// Bump allocator over the caller's workspace (dry run when base == nullptr).
struct Arena {
  char* base;
  size_t cap;
  size_t off = 0;
  bool overflow = false;
  Arena(void* b, size_t c) : base((char*)b), cap(c) {}
  void* take(size_t bytes) {
    size_t o = align_up(off, 256);
    off = o + bytes;
    if (base == nullptr) return nullptr;
    if (off > cap) { overflow = true; return nullptr; }
    return base + o;
  }
  template <typename T> T* get(size_t n) { return (T*)take(n * sizeof(T)); }
};

}  // namespace nmtgan
