// K2 (exact path): wave-per-row attention in fp32 arithmetic, no matrix cores.
//
// Used (a) for the fp32 parity mode (reference default is fp32: configs/r2r_pretrain.json "fp16": false),
// where scores must not be rounded to bf16, and (b) as an independent on-GPU cross-check of the MFMA
// kernels in attn_mfma.hip.  Storage type T may be f32 or bf16; all arithmetic is fp32.
#include "attn_common.h"

#define LK_MAX 1024

template <typename T>
__device__ __forceinline__ float dot64_row(const T* __restrict__ row, const float* __restrict__ s_vec) {
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < ATTN_D; d += 4) {
    const float4 kv = ld4<T>(row + d);
    acc += kv.x * s_vec[d] + kv.y * s_vec[d + 1] + kv.z * s_vec[d + 2] + kv.w * s_vec[d + 3];
  }
  return acc;
}

// raw score (before softmax) of (q-row held in s_q, key)
template <typename T>
__device__ __forceinline__ float score(const AttnArgs& a, const T* kbase, const float* s_q, int b, int qi, int key) {
  float s = dot64_row<T>(kbase + (size_t)key * a.ldk, s_q) * a.scale;
  if (a.key_mask) s += a.key_mask[(size_t)b * a.Lk + key];
  if (a.bias) s += a.bias[((size_t)b * a.Lq + qi) * a.Lk + key];
  return s;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_simple_fwd_kernel(AttnArgs a) {
  if (a.drop_p > 0.f) a.drop_key = bb_salted(a.drop_key, a.salt);
  __shared__ float s_p[4][LK_MAX];
  __shared__ float s_q[4][ATTN_D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int qi = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
  if (qi >= a.Lq) return;  // whole wave exits; no block-wide barriers below
  const T* qrow = (const T*)a.q + (size_t)b * a.bsq + (size_t)qi * a.ldq + h * ATTN_D;
  const T* kbase = (const T*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const T* vbase = (const T*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  s_q[w][lane] = io<T>::ld(qrow + lane);
  __builtin_amdgcn_wave_barrier();
  float mx = -INFINITY;
  for (int key = lane; key < a.Lk; key += 64) {
    const float s = score<T>(a, kbase, s_q[w], b, qi, key);
    s_p[w][key] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  const float m_use = (mx == -INFINITY) ? 0.f : mx;
  float l = 0.f;
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  for (int key = lane; key < a.Lk; key += 64) {
    float p = __expf(s_p[w][key] - m_use);
    l += p;
    if (a.drop_p > 0.f) p = bb_keep(a.drop_key, attn_elem(a, b, h, qi, key), a.drop_thr) ? p * keep_scale : 0.f;
    s_p[w][key] = p;
  }
  l = wave_sum(l);
  __builtin_amdgcn_wave_barrier();
  float o = 0.f;
  for (int key = 0; key < a.Lk; ++key) o += s_p[w][key] * io<T>::ld(vbase + (size_t)key * a.ldv + lane);
  io<T>::st((T*)a.o + (size_t)b * a.bso + (size_t)qi * a.ldo + h * ATTN_D + lane, o / l);
  if (a.lse && lane == 0) a.lse[((size_t)b * a.nh + h) * a.Lq + qi] = m_use + __logf(l);
}

// delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d]
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnArgs a, float* __restrict__ delta) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int qi = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
  if (qi >= a.Lq) return;
  const size_t off = (size_t)b * a.bso + (size_t)qi * a.ldo + h * ATTN_D + lane;
  const float v = io<T>::ld((const T*)a.o + off) * io<T>::ld((const T*)a.dout + off);
  const float s = wave_sum(v);
  if (lane == 0) delta[((size_t)b * a.nh + h) * a.Lq + qi] = s;
}

// dQ (and dbias): one wave per query row
template <typename T>
__global__ __launch_bounds__(256) void attn_simple_dq_kernel(AttnArgs a) {
  if (a.drop_p > 0.f) a.drop_key = bb_salted(a.drop_key, a.salt);
  __shared__ float s_ds[4][LK_MAX];
  __shared__ float s_q[4][ATTN_D];
  __shared__ float s_do[4][ATTN_D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int qi = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
  if (qi >= a.Lq) return;
  const T* qrow = (const T*)a.q + (size_t)b * a.bsq + (size_t)qi * a.ldq + h * ATTN_D;
  const T* kbase = (const T*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const T* vbase = (const T*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const T* dorow = (const T*)a.dout + (size_t)b * a.bso + (size_t)qi * a.ldo + h * ATTN_D;
  s_q[w][lane] = io<T>::ld(qrow + lane);
  s_do[w][lane] = io<T>::ld(dorow + lane);
  __builtin_amdgcn_wave_barrier();
  const size_t ridx = ((size_t)b * a.nh + h) * a.Lq + qi;
  const float lse = a.lse[ridx], delta = a.delta[ridx];
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  for (int key = lane; key < a.Lk; key += 64) {
    const float s = score<T>(a, kbase, s_q[w], b, qi, key);
    const float p = __expf(s - lse);
    float dp = dot64_row<T>(vbase + (size_t)key * a.ldv, s_do[w]);
    if (a.drop_p > 0.f) dp = bb_keep(a.drop_key, attn_elem(a, b, h, qi, key), a.drop_thr) ? dp * keep_scale : 0.f;
    const float ds = p * (dp - delta);
    s_ds[w][key] = ds;
    if (a.dbias) a.dbias[(((size_t)b * a.nh + h) * a.Lq + qi) * a.Lk + key] = ds;
  }
  __builtin_amdgcn_wave_barrier();
  float acc = 0.f;
  for (int key = 0; key < a.Lk; ++key) acc += s_ds[w][key] * io<T>::ld(kbase + (size_t)key * a.ldk + lane);
  io<T>::st((T*)a.dq + (size_t)b * a.bsq + (size_t)qi * a.ldq + h * ATTN_D + lane, acc * a.scale);
}

// dK, dV: one wave per key row, lanes over the head dim
template <typename T>
__global__ __launch_bounds__(256) void attn_simple_dkv_kernel(AttnArgs a) {
  if (a.drop_p > 0.f) a.drop_key = bb_salted(a.drop_key, a.salt);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int key = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
  if (key >= a.Lk) return;
  const float kd = io<T>::ld((const T*)a.k + (size_t)b * a.bsk + (size_t)key * a.ldk + h * ATTN_D + lane);
  const float vd = io<T>::ld((const T*)a.v + (size_t)b * a.bsv + (size_t)key * a.ldv + h * ATTN_D + lane);
  const float km = a.key_mask ? a.key_mask[(size_t)b * a.Lk + key] : 0.f;
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  float dk = 0.f, dv = 0.f;
  for (int qi = 0; qi < a.Lq; ++qi) {
    const float qd = io<T>::ld((const T*)a.q + (size_t)b * a.bsq + (size_t)qi * a.ldq + h * ATTN_D + lane);
    const float dod = io<T>::ld((const T*)a.dout + (size_t)b * a.bso + (size_t)qi * a.ldo + h * ATTN_D + lane);
    float s = wave_sum(qd * kd) * a.scale + km;
    if (a.bias) s += a.bias[((size_t)b * a.Lq + qi) * a.Lk + key];
    const size_t ridx = ((size_t)b * a.nh + h) * a.Lq + qi;
    const float p = __expf(s - a.lse[ridx]);
    float dp = wave_sum(dod * vd);
    float pd = p;
    if (a.drop_p > 0.f) {
      const bool keep = bb_keep(a.drop_key, attn_elem(a, b, h, qi, key), a.drop_thr);
      dp = keep ? dp * keep_scale : 0.f;
      pd = keep ? p * keep_scale : 0.f;
    }
    const float ds = p * (dp - a.delta[ridx]);
    dv += pd * dod;
    dk += ds * qd;
  }
  io<T>::st((T*)a.dk + (size_t)b * a.bsk + (size_t)key * a.ldk + h * ATTN_D + lane, dk * a.scale);
  io<T>::st((T*)a.dv + (size_t)b * a.bsv + (size_t)key * a.ldv + h * ATTN_D + lane, dv);
}

int attn_simple_fwd(const AttnArgs& a, int dtype, hipStream_t st) {
  BB_REQUIRE(a.Lk <= LK_MAX, "attention (exact path): Lk=%d exceeds %d", a.Lk, LK_MAX);
  const dim3 grid((a.Lq + 3) / 4, a.nh, a.B);
  if (dtype == BB_F32) hipLaunchKernelGGL(attn_simple_fwd_kernel<float>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(attn_simple_fwd_kernel<bf16_raw>, grid, dim3(256), 0, st, a);
  BB_CHECK_LAUNCH("attn_fwd(exact)");
  return BB_OK;
}

int attn_delta(const AttnArgs& a, float* delta, int dtype, hipStream_t st) {
  const dim3 grid((a.Lq + 3) / 4, a.nh, a.B);
  if (dtype == BB_F32) hipLaunchKernelGGL(attn_delta_kernel<float>, grid, dim3(256), 0, st, a, delta);
  else hipLaunchKernelGGL(attn_delta_kernel<bf16_raw>, grid, dim3(256), 0, st, a, delta);
  BB_CHECK_LAUNCH("attn_delta");
  return BB_OK;
}

int attn_simple_bwd(const AttnArgs& a, int dtype, hipStream_t st) {
  BB_REQUIRE(a.Lk <= LK_MAX, "attention (exact path): Lk=%d exceeds %d", a.Lk, LK_MAX);
  const dim3 gq((a.Lq + 3) / 4, a.nh, a.B), gk((a.Lk + 3) / 4, a.nh, a.B);
  if (dtype == BB_F32) {
    hipLaunchKernelGGL(attn_simple_dq_kernel<float>, gq, dim3(256), 0, st, a);
    hipLaunchKernelGGL(attn_simple_dkv_kernel<float>, gk, dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(attn_simple_dq_kernel<bf16_raw>, gq, dim3(256), 0, st, a);
    hipLaunchKernelGGL(attn_simple_dkv_kernel<bf16_raw>, gk, dim3(256), 0, st, a);
  }
  BB_CHECK_LAUNCH("attn_bwd(exact)");
  return BB_OK;
}
