// Shared device/host helpers for libbevbert_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define BEVBERT_API extern "C" __attribute__((visibility("default")))

// dtype enum of the C ABI (include/bevbert_hip.h)
enum { BB_F32 = 0, BB_BF16 = 1, BB_F16 = 2 };

// error codes
enum { BB_OK = 0, BB_EINVAL = -1, BB_ELAUNCH = -2, BB_EUNSUPPORTED = -3 };

void bb_set_error(const char* fmt, ...);

#define BB_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      bb_set_error(__VA_ARGS__);         \
      return BB_EINVAL;                  \
    }                                    \
  } while (0)

#define BB_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      bb_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return BB_ELAUNCH;                                                   \
    }                                                                      \
  } while (0)

typedef unsigned short bf16_raw;

// ---- scalar conversions (round-to-nearest-even, NaN preserved) ----------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_raw v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_raw)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_raw)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

template <typename T> struct io;  // load/store `float` through storage type T
template <> struct io<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct io<bf16_raw> {
  static __device__ __forceinline__ float ld(const bf16_raw* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_raw* p, float v) { *p = f32_to_bf16(v); }
};
template <> struct io<_Float16> {
  static __device__ __forceinline__ float ld(const _Float16* p) { return (float)*p; }
  static __device__ __forceinline__ void st(_Float16* p, float v) { *p = (_Float16)v; }
};

// 4-wide vector load/store of T as float4 (16 B for f32, 8 B for 16-bit types)
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<bf16_raw>(const bf16_raw* p) {
  uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
template <> __device__ __forceinline__ float4 ld4<_Float16>(const _Float16* p) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  h4 h = *reinterpret_cast<const h4*>(p);
  return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
template <typename T> __device__ __forceinline__ void st4(T* p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_raw>(bf16_raw* p, float4 v) {
  uint2 u;
  u.x = pack_bf16x2(v.x, v.y);
  u.y = pack_bf16x2(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = u;
}
template <> __device__ __forceinline__ void st4<_Float16>(_Float16* p, float4 v) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  h4 h;
  h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
  *reinterpret_cast<h4*>(p) = h;
}

// ---- wave64 reductions ----------------------------------------------------------------------
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

// ---- counter-based dropout RNG --------------------------------------------------------------
// keep(element) is a pure function of (seed, element index), so a backward kernel regenerates the
// forward's mask without storing it.  One 32-bit mix per element ("lowbias32"-style finaliser over
// seed-keyed 64-bit counter folded to 32 bits); statistical quality is ample for dropout.
__device__ __forceinline__ uint32_t bb_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t bb_rand32(uint64_t seed, uint64_t idx) {
  uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
  uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
  return bb_hash32(lo ^ bb_hash32(hi ^ s1 ^ 0x9e3779b9u) ^ s0);
}
// threshold = round(p * 2^32) clamped; keep iff rand >= threshold
__host__ __device__ __forceinline__ uint32_t bb_drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t <= 0.0) return 0u;
  if (t >= 4294967295.0) return 4294967295u;
  return (uint32_t)(t + 0.5);
}
__device__ __forceinline__ bool bb_keep(uint64_t seed, uint64_t idx, uint32_t thr) { return bb_rand32(seed, idx) >= thr; }

// erf-GELU (reference: pretrain_src/model/vilmodel.py:31-37) and its derivative
__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
