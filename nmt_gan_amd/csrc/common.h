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
  SAMPLER, ADDLN, EMBED, DISC_SMALL, ATTN_SELF_PRIVATE, SAMPLER_ROWS, CONV_TABLE, N_CLS
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

// IEEE half as the second 16-bit operand type: 11 significand bits against bf16's 8, so an operand carries an
// 8x smaller rounding error at the same MFMA rate; range 6e-8 .. 65504 (stores saturate instead of producing inf).
typedef _Float16 f16_t;
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16_t f32_to_f16(float f) {
  return (f16_t)__builtin_amdgcn_fmed3f(f, -65504.f, 65504.f);        // v_cvt_f16_f32 rounds to nearest even
}
__device__ __forceinline__ unsigned pack_f16x2(float a, float b) {
  f16x2_t h = {f32_to_f16(a), f32_to_f16(b)};
  return __builtin_bit_cast(unsigned, h);
}

// 8 packed 16-bit elements (one 16-byte register quad) <-> 8 floats
template <typename T> struct Pack16;
template <> struct Pack16<bf16_t> {
  __device__ static __forceinline__ void unpack2(unsigned w, float& lo, float& hi) {
    lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);
  }
  __device__ static __forceinline__ unsigned pack2(float lo, float hi) {
    return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
  }
};
template <> struct Pack16<f16_t> {
  __device__ static __forceinline__ void unpack2(unsigned w, float& lo, float& hi) {
    const f16x2_t h = __builtin_bit_cast(f16x2_t, w);
    lo = (float)h[0]; hi = (float)h[1];
  }
  __device__ static __forceinline__ unsigned pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
};
template <typename T>
__device__ __forceinline__ void unpack8(const uint4& r, float* f) {
  Pack16<T>::unpack2(r.x, f[0], f[1]); Pack16<T>::unpack2(r.y, f[2], f[3]);
  Pack16<T>::unpack2(r.z, f[4], f[5]); Pack16<T>::unpack2(r.w, f[6], f[7]);
}
template <typename T>
__device__ __forceinline__ uint4 pack8u(const float* f) {
  return make_uint4(Pack16<T>::pack2(f[0], f[1]), Pack16<T>::pack2(f[2], f[3]), Pack16<T>::pack2(f[4], f[5]),
                    Pack16<T>::pack2(f[6], f[7]));
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

template <> struct OpTraits<f16_t> {
  static constexpr int kDtype = NMTGAN_F16;
  __device__ static __forceinline__ float load(const f16_t* p) { return (float)*p; }
  __device__ static __forceinline__ void store(f16_t* p, float v) { *p = f32_to_f16(v); }
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
template <typename T, int N> struct VecIO16 {
  __device__ static __forceinline__ void load(const T* p, float (&o)[N]) {
    if constexpr (N >= 8) {
#pragma unroll
      for (int i = 0; i < N / 8; ++i) {
        const uint4 v = reinterpret_cast<const uint4*>(p)[i];
        unpack8<T>(v, o + 8 * i);
      }
    } else if constexpr (N == 4) {
      const uint2 v = *reinterpret_cast<const uint2*>(p);
      Pack16<T>::unpack2(v.x, o[0], o[1]);
      Pack16<T>::unpack2(v.y, o[2], o[3]);
    } else if constexpr (N == 2) {
      Pack16<T>::unpack2(*reinterpret_cast<const unsigned*>(p), o[0], o[1]);
    } else {
      o[0] = OpTraits<T>::load(p);
    }
  }
  __device__ static __forceinline__ void store(T* p, const float (&o)[N]) {
    if constexpr (N >= 8) {
#pragma unroll
      for (int i = 0; i < N / 8; ++i) reinterpret_cast<uint4*>(p)[i] = pack8u<T>(o + 8 * i);
    } else if constexpr (N == 4) {
      *reinterpret_cast<uint2*>(p) = make_uint2(Pack16<T>::pack2(o[0], o[1]), Pack16<T>::pack2(o[2], o[3]));
    } else if constexpr (N == 2) {
      *reinterpret_cast<unsigned*>(p) = Pack16<T>::pack2(o[0], o[1]);
    } else {
      OpTraits<T>::store(p, o[0]);
    }
  }
};
template <int N> struct VecIO<bf16_t, N> : VecIO16<bf16_t, N> {};
template <int N> struct VecIO<f16_t, N> : VecIO16<f16_t, N> {};

// N consecutive fp32 values as a split-precision GEMM operand: img points at the element's column inside an fp16
// [rows][3K] image row; hi = fp16(v) goes to columns k and 2K + k, lo = fp16(v - hi) to K + k (what split3_kernel
// writes, so a producer can emit the image itself)
template <int N>
__device__ __forceinline__ void store_split3(f16_t* img, int64_t K, const float (&v)[N]) {
  float hi[N], lo[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    // hi is clamped to the fp16 range, so a value up to twice that range still splits exactly instead of turning
    // into inf - inf
    hi[e] = (float)(f16_t)fminf(fmaxf(v[e], -65504.f), 65504.f);
    lo[e] = v[e] - hi[e];
  }
  VecIO<f16_t, N>::store(img, hi);
  VecIO<f16_t, N>::store(img + K, lo);
  VecIO<f16_t, N>::store(img + 2 * K, hi);
}

// 32x32x16 MFMA on 16-bit operands held as raw 16-byte register quads
using f32x16_t = __attribute__((ext_vector_type(16))) float;
template <typename T>
__device__ __forceinline__ f32x16_t mfma16(const uint4& a, const uint4& b, f32x16_t c) {
  if constexpr (sizeof(T) == 2 && !__is_same(T, bf16_t)) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
  } else {
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
  }
}

// Run `...` with OpT bound to the element type of op dtype `dt` (bf16 / fp16 / fp32)
#define NMTGAN_DISPATCH_OP(dt, OpT, ...)                                            \
  do {                                                                              \
    if ((dt) == NMTGAN_BF16) { using OpT = ::nmtgan::bf16_t; __VA_ARGS__; }         \
    else if ((dt) == NMTGAN_F16) { using OpT = ::nmtgan::f16_t; __VA_ARGS__; }      \
    else { using OpT = float; __VA_ARGS__; }                                        \
  } while (0)
static inline bool valid_op_dtype(int dt) { return dt == NMTGAN_BF16 || dt == NMTGAN_F16 || dt == NMTGAN_F32; }

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
// Reductions ACROSS the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48), every lane gets the result:
// two VALU lane swaps (v_permlane16_swap / v_permlane32_swap of gfx950) instead of LDS-routed shuffles.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float groups4_max(float v) {
  u32x2_t a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  u32x2_t b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float groups4_sum(float v) {
  u32x2_t a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  u32x2_t b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// v + (v of the lane a DPP control selects): 0xB1 = quad_perm [1,0,3,2] (lane ^ 1), 0x4E = quad_perm [2,3,0,1]
// (lane ^ 2), 0x141 = row_half_mirror (7 - lane inside 8 lanes), 0x140 = row_mirror (15 - lane inside 16 lanes);
// applied in that order they sum over 2, 4, 8, 16 adjacent lanes without touching LDS
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
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
