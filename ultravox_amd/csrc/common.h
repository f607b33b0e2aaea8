// Shared device/host helpers for libuvx (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef unsigned short bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;

#define UVX_WAVE 64

// ---- status codes (see include/uvx.h) ----
#define UVX_OK 0
#define UVX_ERR_INVALID (-1)
#define UVX_ERR_SHAPE (-2)
#define UVX_ERR_WORKSPACE (-3)
#define UVX_ERR_UNSUPPORTED (-4)
#define UVX_ERR_RUNTIME (-5)

extern "C" __attribute__((visibility("hidden"))) void uvx_set_error(const char* fmt, ...);

#define UVX_CHECK(cond, code, ...)        \
  do {                                    \
    if (!(cond)) {                        \
      uvx_set_error(__VA_ARGS__);         \
      return (code);                      \
    }                                     \
  } while (0)

#define UVX_HIP(expr)                                                          \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      uvx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                    __FILE__, __LINE__);                                       \
      return (int)_e;                                                          \
    }                                                                          \
  } while (0)

#define UVX_LAUNCH_CHECK()                                                     \
  do {                                                                         \
    hipError_t _e = hipGetLastError();                                         \
    if (_e != hipSuccess) {                                                    \
      uvx_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), \
                    __FILE__, __LINE__);                                       \
      return (int)_e;                                                          \
    }                                                                          \
  } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, same as torch) ----
__host__ __device__ __forceinline__ float bf2f(bf16_t v) {
  return __builtin_bit_cast(float, (uint32_t)v << 16);
}
__host__ __device__ __forceinline__ bf16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  // native conversion: hipcc lowers this to v_cvt_pk_bf16_f32 (round-to-nearest-even), pairing neighbours
  return __builtin_bit_cast(unsigned short, (__bf16)f);
#endif
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// Generic element load/store used by dtype-templated kernels (T = bf16_t or float).
template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
// round-trip through the storage type (emulates a cast to the activation dtype)
template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }

// 8-element vector access: 16 B for bf16, 2x16 B for f32.
template <typename T> struct Vec8 { float v[8]; };
template <typename T> __device__ __forceinline__ void ld8(const T* p, float* o);
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float* o) {
  u16x8_t r = *reinterpret_cast<const u16x8_t*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = bf2f(r[i]);
}
template <> __device__ __forceinline__ void ld8<float>(const float* p, float* o) {
  float4 a = reinterpret_cast<const float4*>(p)[0];
  float4 b = reinterpret_cast<const float4*>(p)[1];
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
  o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float* o);
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float* o) {
  u16x8_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = f2bf(o[i]);
  *reinterpret_cast<u16x8_t*>(p) = r;
}
template <> __device__ __forceinline__ void st8<float>(float* p, const float* o) {
  reinterpret_cast<float4*>(p)[0] = make_float4(o[0], o[1], o[2], o[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(o[4], o[5], o[6], o[7]);
}

// ---- wave64 / block reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// Block-wide sum; `red` is LDS scratch of >= 16 floats. All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// GELU for the bf16 path: Abramowitz-Stegun 7.1.26 erf (|error| <= 1.5e-7, far below the 2^-9 relative
// rounding of the bf16 store that follows) with hardware rcp / exp2 — ~4x fewer VALU ops than erff();
// the f32 parity path keeps the exact erff().
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  const float pe = poly * e;                       // = erfc(|x|/sqrt2): no cancellation in the negative tail
  return 0.5f * x * (x < 0.f ? pe : 2.0f - pe);
}
// d gelu(x) / dx = Phi(x) + x phi(x) with the same erfc approximation (and the same exp2) as gelu_fast: the bf16 path's GELU backward,
// as a kernel of its own (gelu_bwd_k) and inside the GEMM epilogue (GemmDesc::act == 3) - one arithmetic, bit-identical either way.
__device__ __forceinline__ float gelu_fast_grad(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);      // exp(-x^2 / 2)
  const float hp = 0.5f * poly * e;                                           // erfc(|x| / sqrt 2) / 2
  return (x < 0.f ? hp : 1.0f - hp) + x * (0.3989422804014327f * e);
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE property of a kernel: one of these per call site remembers which
// devices have had it set (bit d of the mask; ADVICE r4: a process-wide `static bool` would leave a second device of the same process
// at the 64 KB default).  The bit is published by done() AFTER the attribute call succeeded: a second host thread either
// sees the bit - and the attribute is in place - or sets the attribute itself (idempotent); a failed call leaves the bit clear and is
// retried by the next launch.  Use through UVX_SET_ATTR_ONCE.
#include <atomic>
struct PerDeviceOnce {
  std::atomic<unsigned long long> mask{0};
  static unsigned long long bit() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return 0;      // unknown device: set the attribute every time, never publish
    return 1ull << (d & 63);
  }
  bool need() const { const unsigned long long b = bit(); return b == 0 || !(mask.load(std::memory_order_acquire) & b); }
  void done() { mask.fetch_or(bit(), std::memory_order_release); }
};
#define UVX_SET_ATTR_ONCE(once, fn, bytes)                                                                          \
  do {                                                                                                              \
    if ((once).need()) {                                                                                            \
      UVX_HIP(hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));    \
      (once).done();                                                                                                \
    }                                                                                                               \
  } while (0)
