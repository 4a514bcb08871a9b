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
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even): let the compiler pick it
typedef __bf16 bb_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) { return __builtin_bit_cast(bf16_raw, (__bf16)f); }
// one v_cvt_pk_bf16_f32 for the pair.  As a VECTOR conversion: with two scalar casts the compiler, depending on what
// produced the operands, converts each value alone and merges the halves (v_cvt x 2 + v_and / v_perm), or moves a select in
// front of the operands behind the conversion (round 5: 2.5 instead of 1.5 instructions per attention score element)
typedef float bb_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const bb_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bb_bf16x2));
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
// keep(element) is a pure function of (seed, offset, element index), so a backward kernel regenerates the forward's
// mask instead of storing it.  A launch folds (seed, offset) into one 32-bit site key on the host; on the device one
// 32-bit mix ("lowbias32") serves TWO neighbouring elements (index pair 2j, 2j+1 -> low / high 16 bits), compared
// against a 16-bit threshold (p = 0.1 -> 6554/65536).  Element indices are < 2^32 per dropout site.
__host__ __device__ __forceinline__ uint32_t bb_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t bb_site_key(uint64_t seed, uint64_t offset) {
  uint32_t k = bb_hash32((uint32_t)seed ^ 0x9e3779b9u);
  k = bb_hash32(k ^ (uint32_t)(seed >> 32));
  k = bb_hash32(k ^ (uint32_t)offset);
  return bb_hash32(k ^ (uint32_t)(offset >> 32));
}
// threshold = round(p * 2^16); keep iff the element's 16 random bits >= threshold
__host__ __device__ __forceinline__ uint32_t bb_drop_threshold(float p) {
  const float t = p * 65536.0f + 0.5f;
  if (t <= 0.f) return 0u;
  return t >= 65535.f ? 65535u : (uint32_t)t;
}
// Per-step salt.  A dropout site's key is folded on the HOST from (seed, offset); both are launch arguments and would
// be frozen into a captured hipGraph.  bevbert_set_step_salt() registers ONE device word that the host rewrites before
// every training step (a 4-byte fill on the stream); every dropout kernel then uses hash(key ^ *salt) instead of key,
// so a replayed graph draws fresh masks and forward / backward of one step still agree.  No salt registered: key as is.
const uint32_t* bb_step_salt();
__device__ __forceinline__ uint32_t bb_salted(uint32_t key, const uint32_t* __restrict__ salt) {
  return salt ? bb_hash32(key ^ *salt) : key;
}
__device__ __forceinline__ uint32_t bb_pair_bits(uint32_t key, uint32_t pair_idx) { return bb_hash32(pair_idx ^ key); }
__device__ __forceinline__ bool bb_keep_lo(uint32_t bits, uint32_t thr) { return (bits & 0xffffu) >= thr; }
__device__ __forceinline__ bool bb_keep_hi(uint32_t bits, uint32_t thr) { return (bits >> 16) >= thr; }
__device__ __forceinline__ bool bb_keep(uint32_t key, uint32_t idx, uint32_t thr) {
  const uint32_t bits = bb_pair_bits(key, idx >> 1);
  return ((idx & 1u) ? (bits >> 16) : (bits & 0xffffu)) >= thr;
}

// erf-GELU in a dozen instructions for the bf16 kernels (libm's erff costs ~45 VALU instructions per element with
// its range branches, which made the GELU kernels VALU-bound: 108 us for 28224 x 3072 where HBM needs 43).
// Abramowitz & Stegun 7.1.26: erfc(z) = (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z >= 0, |error| <= 1.5e-7
// -- absolute on the side where the cdf is near 1, RELATIVE-ly good on the negative side (no cancellation: cdf = erfc/2).
// Returns the normal cdf Phi(x) and E = exp(-x^2 / 2) (the pdf's exponential, shared with the derivative).
__device__ __forceinline__ float gelu_cdf_fast(float x, float& E) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  E = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);     // exp(-x^2/2) = 2^(-x^2 log2(e) / 2)
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float half_erfc = 0.5f * poly * t * E;
  return x >= 0.f ? 1.0f - half_erfc : half_erfc;
}
__device__ __forceinline__ float gelu_erf_fast(float x) {
  float E;
  return x * gelu_cdf_fast(x, E);
}
__device__ __forceinline__ float gelu_erf_grad_fast(float x) {
  float E;
  const float cdf = gelu_cdf_fast(x, E);
  return fmaf(x * 0.39894228040143267794f, E, cdf);
}
// erf-GELU (reference: pretrain_src/model/vilmodel.py:31-37) and its derivative
__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// Two elements per lane and instruction: the polynomial and the products compile to packed fp32 (v_pk_fma_f32 /
// v_pk_mul_f32); the reciprocal and the exponential stay scalar (quarter-rate transcendental unit).
typedef float bb_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bb_f32x2 gelu_cdf_fast2(bb_f32x2 x, bb_f32x2& E) {
  const bb_f32x2 z = (bb_f32x2){fabsf(x[0]), fabsf(x[1])} * 0.70710678118654752440f;
  const bb_f32x2 den = z * 0.3275911f + 1.0f;
  const bb_f32x2 t = (bb_f32x2){__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  const bb_f32x2 e2 = x * x * -0.72134752044448170368f;
  E = (bb_f32x2){__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
  bb_f32x2 poly = t * 1.061405429f + -1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t + -0.284496736f;
  poly = poly * t + 0.254829592f;
  const bb_f32x2 he = poly * t * E * 0.5f;
  return (bb_f32x2){x[0] >= 0.f ? 1.0f - he[0] : he[0], x[1] >= 0.f ? 1.0f - he[1] : he[1]};
}
// y = gelu(a + b) and dy * gelu'(a + b) on four elements; fp32 tensors keep libm's erff (the exact-arithmetic path of the
// parity tests), bf16 tensors take the fast form, whose error (4e-7) is four orders below the bf16 rounding of the result
template <typename T> __device__ __forceinline__ float4 gelu4_of(float4 a, float4 b) {
  bb_f32x2 E;
  const bb_f32x2 x0 = (bb_f32x2){a.x + b.x, a.y + b.y}, x1 = (bb_f32x2){a.z + b.z, a.w + b.w};
  const bb_f32x2 y0 = x0 * gelu_cdf_fast2(x0, E), y1 = x1 * gelu_cdf_fast2(x1, E);
  return make_float4(y0[0], y0[1], y1[0], y1[1]);
}
template <> __device__ __forceinline__ float4 gelu4_of<float>(float4 a, float4 b) {
  return make_float4(gelu_erf(a.x + b.x), gelu_erf(a.y + b.y), gelu_erf(a.z + b.z), gelu_erf(a.w + b.w));
}
template <typename T> __device__ __forceinline__ float4 gelu_grad4_of(float4 d, float4 a, float4 b) {
  bb_f32x2 E0, E1;
  const bb_f32x2 x0 = (bb_f32x2){a.x + b.x, a.y + b.y}, x1 = (bb_f32x2){a.z + b.z, a.w + b.w};
  const bb_f32x2 c0 = gelu_cdf_fast2(x0, E0), c1 = gelu_cdf_fast2(x1, E1);
  const bb_f32x2 g0 = x0 * 0.39894228040143267794f * E0 + c0, g1 = x1 * 0.39894228040143267794f * E1 + c1;
  return make_float4(d.x * g0[0], d.y * g0[1], d.z * g1[0], d.w * g1[1]);
}
template <> __device__ __forceinline__ float4 gelu_grad4_of<float>(float4 d, float4 a, float4 b) {
  return make_float4(d.x * gelu_erf_grad(a.x + b.x), d.y * gelu_erf_grad(a.y + b.y), d.z * gelu_erf_grad(a.z + b.z),
                     d.w * gelu_erf_grad(a.w + b.w));
}
