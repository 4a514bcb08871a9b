// HBM-bound row kernels: K3 bias+dropout+residual+LayerNorm (fwd/bwd), K4 bias+erf-GELU (fwd/bwd),
// K5 embedding-sum+LayerNorm, K6 weighted segment gather (gmap aggregation fwd and bwd), column sums,
// and the flat-arena optimiser kernels (K7 fused AdamW, grad-norm, clip coefficient).
//
// Reference call sites replaced:
//   BertSelfOutput / BertOutput            pretrain_src/model/vilmodel.py:143-154,182-193  (dense bias + dropout + add + LN)
//   BertIntermediate + gelu                vilmodel.py:31-37,168-180
//   BertEmbeddings                         vilmodel.py:48-77
//   _aggregate_gmap_features               vilmodel.py:632-666 (Python dict loops -> one CSR gather)
//   clip_grad_norm_ + AdamW.step           pretrain_src/train_r2r.py:295-306, pretrain_src/optim/adamw.py:53-112
//
// Layout: activations are (rows, H) row-major with H % 256 == 0; one 64-lane wave owns one row and keeps it in
// registers (H/64 values per lane as float4s), so each tensor is read once and written once; statistics are fp32.
#include <math.h>

#include "common.h"

// =============================================================================================
// LayerNorm forward:  z = dropout(x + bias) + residual ;  y = (z - mean) * rstd * gamma + beta
// GATHER: x row = word[ids[row]] + pos[row % L] + type_row   (BertEmbeddings)
// =============================================================================================
template <typename T, int NV, bool GATHER>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                     const T* __restrict__ residual, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     T* __restrict__ z_out, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int rows, float eps, float drop_p,
                                                     uint32_t drop_thr, uint32_t drop_key,
                                                     const int64_t* __restrict__ ids, const T* __restrict__ word,
                                                     const T* __restrict__ pos, const T* __restrict__ type_row, int L,
                                                     const uint32_t* __restrict__ salt) {
  constexpr int H = NV * 256;
  if (drop_p > 0.f) drop_key = bb_salted(drop_key, salt);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4 v[NV];
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = (i * 64 + lane) * 4;
    float4 a;
    if (GATHER) {
      const int64_t id = ids[row];
      a = ld4<T>(word + (size_t)id * H + col);
      const float4 p = ld4<T>(pos + (size_t)(row % L) * H + col);
      const float4 t = ld4<T>(type_row + col);
      a.x = (a.x + p.x) + t.x; a.y = (a.y + p.y) + t.y; a.z = (a.z + p.z) + t.z; a.w = (a.w + p.w) + t.w;
    } else {
      a = ld4<T>(x + (size_t)row * H + col);
    }
    if (bias != nullptr) {
      const float4 bb = *reinterpret_cast<const float4*>(bias + col);
      a.x += bb.x; a.y += bb.y; a.z += bb.z; a.w += bb.w;
    }
    if (drop_p > 0.f) {
      const uint32_t pr = ((uint32_t)row * H + col) >> 1;        // col % 4 == 0: two index pairs
      const uint32_t b0 = bb_pair_bits(drop_key, pr), b1 = bb_pair_bits(drop_key, pr + 1);
      a.x = bb_keep_lo(b0, drop_thr) ? a.x * keep_scale : 0.f;
      a.y = bb_keep_hi(b0, drop_thr) ? a.y * keep_scale : 0.f;
      a.z = bb_keep_lo(b1, drop_thr) ? a.z * keep_scale : 0.f;
      a.w = bb_keep_hi(b1, drop_thr) ? a.w * keep_scale : 0.f;
    }
    if (residual != nullptr) {
      const float4 r = ld4<T>(residual + (size_t)row * H + col);
      a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
    }
    v[i] = a;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) * (1.0f / H);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / H) + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = (i * 64 + lane) * 4;
    if (z_out != nullptr) st4<T>(z_out + (size_t)row * H + col, v[i]);
    const float4 g = *reinterpret_cast<const float4*>(gamma + col);
    const float4 b = *reinterpret_cast<const float4*>(beta + col);
    float4 o;
    o.x = (v[i].x - mean) * rstd * g.x + b.x;
    o.y = (v[i].y - mean) * rstd * g.y + b.y;
    o.z = (v[i].z - mean) * rstd * g.z + b.z;
    o.w = (v[i].w - mean) * rstd * g.w + b.w;
    if (!GATHER) {   // post-normalisation terms (bevbert_layernorm_post_fwd): y = LN(..) + post1 + post2, in that order --
      // the GATHER-only pointers carry them, so the plain variant costs two null tests
      if (word != nullptr) {
        const float4 r = ld4<T>(word + (size_t)row * H + col);
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      if (pos != nullptr) {
        const float4 r = ld4<T>(pos + (size_t)row * H + col);
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
    }
    st4<T>(y + (size_t)row * H + col, o);
  }
}

// =============================================================================================
// LayerNorm backward.  Given dy, the saved z, mean, rstd:
//   dz = rstd * (g*dy - mean_H(g*dy) - xhat * mean_H(g*dy*xhat))        -> d(residual)
//   dx = dz * keep/(1-p)                                                  -> d(dense output)   (== dz when p == 0)
//   per-column partial sums of dy*xhat (dgamma), dy (dbeta), dx (dbias) -> partials[block][3][H]
// Persistent-style grid: each wave strides over rows and keeps its column partials in registers.
// =============================================================================================
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, T* __restrict__ dz_out,
                                                     T* __restrict__ dx_out, float* __restrict__ partials, int rows,
                                                     float drop_p, uint32_t drop_thr, uint32_t drop_key,
                                                     const uint32_t* __restrict__ salt, const T* __restrict__ dz_add) {
  constexpr int H = NV * 256;
  if (drop_p > 0.f) drop_key = bb_salted(drop_key, salt);
  __shared__ float4 s_red[3][4][NV * 64];  // [which][wave][lane-major float4]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  float4 ag[NV], ab[NV], ax[NV];
  float4 g[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ag[i] = ab[i] = ax[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    g[i] = *reinterpret_cast<const float4*>(gamma + (i * 64 + lane) * 4);
  }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float4 d[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int col = (i * 64 + lane) * 4;
      d[i] = ld4<T>(dy + (size_t)row * H + col);
      const float4 zz = ld4<T>(z + (size_t)row * H + col);
      xh[i] = make_float4((zz.x - mu) * rs, (zz.y - mu) * rs, (zz.z - mu) * rs, (zz.w - mu) * rs);
      ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y; ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
      ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
      d[i].x *= g[i].x; d[i].y *= g[i].y; d[i].z *= g[i].z; d[i].w *= g[i].w;
      s1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
      s2 += (d[i].x * xh[i].x + d[i].y * xh[i].y) + (d[i].z * xh[i].z + d[i].w * xh[i].w);
    }
    s1 = wave_sum(s1) * (1.0f / H);
    s2 = wave_sum(s2) * (1.0f / H);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int col = (i * 64 + lane) * 4;
      float4 o;
      o.x = rs * (d[i].x - s1 - xh[i].x * s2);
      o.y = rs * (d[i].y - s1 - xh[i].y * s2);
      o.z = rs * (d[i].z - s1 - xh[i].z * s2);
      o.w = rs * (d[i].w - s1 - xh[i].w * s2);
      if (dz_add != nullptr) {       // z has a second consumer (pre-norm residual stream): its gradient joins here
        const float4 e = ld4<T>(dz_add + (size_t)row * H + col);
        o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
      }
      if (dz_out != nullptr) st4<T>(dz_out + (size_t)row * H + col, o);
      if (drop_p > 0.f) {
        const uint32_t pr = ((uint32_t)row * H + col) >> 1;
        const uint32_t b0 = bb_pair_bits(drop_key, pr), b1 = bb_pair_bits(drop_key, pr + 1);
        o.x = bb_keep_lo(b0, drop_thr) ? o.x * keep_scale : 0.f;
        o.y = bb_keep_hi(b0, drop_thr) ? o.y * keep_scale : 0.f;
        o.z = bb_keep_lo(b1, drop_thr) ? o.z * keep_scale : 0.f;
        o.w = bb_keep_hi(b1, drop_thr) ? o.w * keep_scale : 0.f;
      }
      if (dx_out != nullptr) st4<T>(dx_out + (size_t)row * H + col, o);
      ax[i].x += o.x; ax[i].y += o.y; ax[i].z += o.z; ax[i].w += o.w;
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    s_red[0][wave][i * 64 + lane] = ag[i];
    s_red[1][wave][i * 64 + lane] = ab[i];
    s_red[2][wave][i * 64 + lane] = ax[i];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 3 * NV * 64; k += 256) {
    const int which = k / (NV * 64), j = k % (NV * 64);
    float4 a = s_red[which][0][j];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 t = s_red[which][w][j];
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    *reinterpret_cast<float4*>(partials + ((size_t)blockIdx.x * 3 + which) * H + j * 4) = a;
  }
}

// =============================================================================================
// The same pair with an fp32 RESIDUAL STREAM around bf16 matrix operands (round 5; torch.autocast keeps LayerNorm outputs
// and residual sums in fp32 and only rounds what enters a GEMM: pretrain_src/train_r2r.py:256-258).
//   forward:  z = dropout(x + bias) + residual, x bf16 (a GEMM output), residual fp32 (the previous LayerNorm's fp32 output)
//             or bf16 (a stream that starts here); writes y16 = bf16(y) for the next GEMMs, y32 = y for the next residual
//             add, z in fp32 for the backward
//   backward: dy = dy16 (bf16: from the GEMMs that read y16) + dy32 (fp32: from the residual add that read y32), either
//             may be null; dz32 (fp32) continues the residual stream's gradient, dx16 = bf16(dz through the dropout mask)
//             feeds the dense layer's backward GEMMs; column partials as in ln_bwd_kernel
// =============================================================================================
template <typename TR, int NV>
__global__ __launch_bounds__(256) void ln_res32_fwd_kernel(const bf16_raw* __restrict__ x, const float* __restrict__ bias,
                                                           const TR* __restrict__ residual, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, bf16_raw* __restrict__ y16,
                                                           float* __restrict__ y32, float* __restrict__ z_out,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           int rows, float eps, float drop_p, uint32_t drop_thr,
                                                           uint32_t drop_key, const uint32_t* __restrict__ salt) {
  constexpr int H = NV * 256;
  if (drop_p > 0.f) drop_key = bb_salted(drop_key, salt);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4 v[NV];
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = (i * 64 + lane) * 4;
    float4 a = ld4<bf16_raw>(x + (size_t)row * H + col);
    if (bias != nullptr) {
      const float4 bb = *reinterpret_cast<const float4*>(bias + col);
      a.x += bb.x; a.y += bb.y; a.z += bb.z; a.w += bb.w;
    }
    if (drop_p > 0.f) {
      const uint32_t pr = ((uint32_t)row * H + col) >> 1;
      const uint32_t b0 = bb_pair_bits(drop_key, pr), b1 = bb_pair_bits(drop_key, pr + 1);
      a.x = bb_keep_lo(b0, drop_thr) ? a.x * keep_scale : 0.f;
      a.y = bb_keep_hi(b0, drop_thr) ? a.y * keep_scale : 0.f;
      a.z = bb_keep_lo(b1, drop_thr) ? a.z * keep_scale : 0.f;
      a.w = bb_keep_hi(b1, drop_thr) ? a.w * keep_scale : 0.f;
    }
    if (residual != nullptr) {
      const float4 r = ld4<TR>(residual + (size_t)row * H + col);
      a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
    }
    v[i] = a;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) * (1.0f / H);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / H) + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = (i * 64 + lane) * 4;
    if (z_out != nullptr) st4<float>(z_out + (size_t)row * H + col, v[i]);
    const float4 g = *reinterpret_cast<const float4*>(gamma + col);
    const float4 b = *reinterpret_cast<const float4*>(beta + col);
    float4 o;
    o.x = (v[i].x - mean) * rstd * g.x + b.x;
    o.y = (v[i].y - mean) * rstd * g.y + b.y;
    o.z = (v[i].z - mean) * rstd * g.z + b.z;
    o.w = (v[i].w - mean) * rstd * g.w + b.w;
    st4<bf16_raw>(y16 + (size_t)row * H + col, o);
    if (y32 != nullptr) st4<float>(y32 + (size_t)row * H + col, o);
  }
}

template <int NV>
__global__ __launch_bounds__(256) void ln_res32_bwd_kernel(const bf16_raw* __restrict__ dy16, const float* __restrict__ dy32,
                                                           const float* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           float* __restrict__ dz_out, bf16_raw* __restrict__ dx_out,
                                                           float* __restrict__ partials, int rows, float drop_p,
                                                           uint32_t drop_thr, uint32_t drop_key,
                                                           const uint32_t* __restrict__ salt, int dz_bf16) {
  // dz_bf16: dz_out is a bf16 tensor (the residual was a bf16 tensor -- where an fp32 residual stream starts)
  constexpr int H = NV * 256;
  if (drop_p > 0.f) drop_key = bb_salted(drop_key, salt);
  __shared__ float4 s_red[3][4][NV * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  float4 ag[NV], ab[NV], ax[NV];
  float4 g[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ag[i] = ab[i] = ax[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    g[i] = *reinterpret_cast<const float4*>(gamma + (i * 64 + lane) * 4);
  }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float4 d[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int col = (i * 64 + lane) * 4;
      d[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (dy16 != nullptr) d[i] = ld4<bf16_raw>(dy16 + (size_t)row * H + col);
      if (dy32 != nullptr) {
        const float4 e = ld4<float>(dy32 + (size_t)row * H + col);
        d[i].x += e.x; d[i].y += e.y; d[i].z += e.z; d[i].w += e.w;
      }
      const float4 zz = ld4<float>(z + (size_t)row * H + col);
      xh[i] = make_float4((zz.x - mu) * rs, (zz.y - mu) * rs, (zz.z - mu) * rs, (zz.w - mu) * rs);
      ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y; ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
      ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
      d[i].x *= g[i].x; d[i].y *= g[i].y; d[i].z *= g[i].z; d[i].w *= g[i].w;
      s1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
      s2 += (d[i].x * xh[i].x + d[i].y * xh[i].y) + (d[i].z * xh[i].z + d[i].w * xh[i].w);
    }
    s1 = wave_sum(s1) * (1.0f / H);
    s2 = wave_sum(s2) * (1.0f / H);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int col = (i * 64 + lane) * 4;
      float4 o;
      o.x = rs * (d[i].x - s1 - xh[i].x * s2);
      o.y = rs * (d[i].y - s1 - xh[i].y * s2);
      o.z = rs * (d[i].z - s1 - xh[i].z * s2);
      o.w = rs * (d[i].w - s1 - xh[i].w * s2);
      if (dz_out != nullptr) {
        if (dz_bf16) st4<bf16_raw>(reinterpret_cast<bf16_raw*>(dz_out) + (size_t)row * H + col, o);
        else st4<float>(dz_out + (size_t)row * H + col, o);
      }
      if (drop_p > 0.f) {
        const uint32_t pr = ((uint32_t)row * H + col) >> 1;
        const uint32_t b0 = bb_pair_bits(drop_key, pr), b1 = bb_pair_bits(drop_key, pr + 1);
        o.x = bb_keep_lo(b0, drop_thr) ? o.x * keep_scale : 0.f;
        o.y = bb_keep_hi(b0, drop_thr) ? o.y * keep_scale : 0.f;
        o.z = bb_keep_lo(b1, drop_thr) ? o.z * keep_scale : 0.f;
        o.w = bb_keep_hi(b1, drop_thr) ? o.w * keep_scale : 0.f;
      }
      if (dx_out != nullptr) st4<bf16_raw>(dx_out + (size_t)row * H + col, o);
      ax[i].x += o.x; ax[i].y += o.y; ax[i].z += o.z; ax[i].w += o.w;
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    s_red[0][wave][i * 64 + lane] = ag[i];
    s_red[1][wave][i * 64 + lane] = ab[i];
    s_red[2][wave][i * 64 + lane] = ax[i];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 3 * NV * 64; k += 256) {
    const int which = k / (NV * 64), j = k % (NV * 64);
    float4 a = s_red[which][0][j];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 t = s_red[which][w][j];
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    *reinterpret_cast<float4*>(partials + ((size_t)blockIdx.x * 3 + which) * H + j * 4) = a;
  }
}

// out_w[c] (+)= sum_b partials[b][w][c] for every w with a non-null output.  Block = 64 columns x 16 partial groups:
// coalesced 256-B reads, 16 independent running sums per column, fixed combination order => deterministic.
struct FinalizeOut { float* out[3]; };
__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ partials, int nblocks,
                                                              int nwhich, int C, FinalizeOut o, int accumulate) {
  __shared__ float sh[16][64];
  const int which = blockIdx.y;
  float* out = o.out[which];
  if (out == nullptr) return;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int c = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (c < C)
    for (int b = ty; b < nblocks; b += 16) s += partials[((size_t)b * nwhich + which) * C + c];
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < C) {
    float t = sh[0][tx];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += sh[k][tx];
    out[c] = accumulate ? out[c] + t : t;
  }
}
static void launch_finalize(const float* partials, int nblocks, int nwhich, int C, float* o0, float* o1, float* o2,
                            int accumulate, hipStream_t stream) {
  FinalizeOut o;
  o.out[0] = o0; o.out[1] = o1; o.out[2] = o2;
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 63) / 64, nwhich), dim3(64, 16), 0, stream, partials, nblocks,
                     nwhich, C, o, accumulate);
}

// Batched second stage: ONE launch finishes every pending column reduction of a backward pass (LayerNorm gamma / beta /
// bias, GELU bias, projection biases: ~110 per training step, each a 6-8 us launch of a few dozen workgroups when issued
// one by one).  A task = 64 columns of one output vector; the host builds the task table once per distinct step shape
// (ops.ReduceQueue) and keeps it in device memory.  Same arithmetic and summation order as colsum_finalize_kernel.
struct FinalizeTask {
  const float* partials;   // [nblocks][row_stride] floats
  float* out;              // 64-column slice of the output vector
  int nblocks, row_stride, col0, ncols, accumulate, pad;
};
__global__ __launch_bounds__(1024) void multi_finalize_kernel(const FinalizeTask* __restrict__ tasks) {
  __shared__ float sh[16][64];
  const FinalizeTask t = tasks[blockIdx.x];
  const int tx = threadIdx.x, ty = threadIdx.y;
  float s = 0.f;
  if (tx < t.ncols)
    for (int b = ty; b < t.nblocks; b += 16) s += t.partials[(size_t)b * t.row_stride + t.col0 + tx];
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && tx < t.ncols) {
    float v = sh[0][tx];
#pragma unroll
    for (int k = 1; k < 16; ++k) v += sh[k][tx];
    t.out[tx] = t.accumulate ? t.out[tx] + v : v;
  }
}

// =============================================================================================
// bias + erf-GELU forward / backward, and a plain column sum (QKV bias grads)
// =============================================================================================
// Same decomposition as the backward below: grid (row groups, 1024-column slices), a thread owns 4 columns (its bias is
// loaded once) and walks CONTIGUOUS rows with 8 independent loads in flight.  The first version -- a flat grid-stride
// loop, one chunk per thread and iteration, the column recomputed as (i * 4) % C on a 64-bit index (~60 instructions
// per chunk) -- ran at 3.2 TB/s.
// ACT 0: erf-GELU (BertIntermediate, vilmodel.py:31-37); ACT 1: ReLU (the prediction heads' Linear - ReLU - LayerNorm - Linear,
// pretrain_src/model/pretrain_cmt.py:34-71)
template <typename T, int ACT> __device__ __forceinline__ float4 act4_of(float4 a, float4 b) {
  if (ACT == 0) return gelu4_of<T>(a, b);
  return make_float4(fmaxf(a.x + b.x, 0.f), fmaxf(a.y + b.y, 0.f), fmaxf(a.z + b.z, 0.f), fmaxf(a.w + b.w, 0.f));
}
template <typename T, int ACT> __device__ __forceinline__ float4 act_grad4_of(float4 d, float4 a, float4 b) {
  if (ACT == 0) return gelu_grad4_of<T>(d, a, b);
  return make_float4(a.x + b.x > 0.f ? d.x : 0.f, a.y + b.y > 0.f ? d.y : 0.f, a.z + b.z > 0.f ? d.z : 0.f,
                     a.w + b.w > 0.f ? d.w : 0.f);
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void bias_gelu_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                            T* __restrict__ y, int rows, int C) {
  const int c0 = blockIdx.y * 1024 + threadIdx.x * 4;
  if (c0 >= C) return;
  const float4 b = *reinterpret_cast<const float4*>(bias + c0);
  const int rpb = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
  int r = blockIdx.x * rpb;
  rows = (r + rpb < rows) ? r + rpb : rows;
  constexpr int R = 8;
  for (; r + R - 1 < rows; r += R) {
    float4 a[R];
#pragma unroll
    for (int k = 0; k < R; ++k) a[k] = ld4<T>(x + (size_t)(r + k) * C + c0);
#pragma unroll
    for (int k = 0; k < R; ++k) st4<T>(y + (size_t)(r + k) * C + c0, act4_of<T, ACT>(a[k], b));
  }
  for (; r < rows; ++r) st4<T>(y + (size_t)r * C + c0, act4_of<T, ACT>(ld4<T>(x + (size_t)r * C + c0), b));
}

// bf16, 16-byte accesses: a thread owns 8 columns (128 threads per 1024-column slice)
__device__ __forceinline__ void bf16x8_to_f32(const uint4& u, float4& lo, float4& hi) {
  lo = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                   __uint_as_float(u.y & 0xffff0000u));
  hi = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                   __uint_as_float(u.w & 0xffff0000u));
}
__device__ __forceinline__ uint4 f32_to_bf16x8(const float4& lo, const float4& hi) {
  return make_uint4(pack_bf16x2(lo.x, lo.y), pack_bf16x2(lo.z, lo.w), pack_bf16x2(hi.x, hi.y), pack_bf16x2(hi.z, hi.w));
}
__global__ __launch_bounds__(128) void bias_gelu_fwd8_kernel(const bf16_raw* __restrict__ x, const float* __restrict__ bias,
                                                             bf16_raw* __restrict__ y, int rows, int C) {
  const int c0 = blockIdx.y * 1024 + threadIdx.x * 8;
  if (c0 >= C) return;
  const float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
  const int rpb = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
  int r = blockIdx.x * rpb;
  rows = (r + rpb < rows) ? r + rpb : rows;
  constexpr int R = 8;
  for (; r + R - 1 < rows; r += R) {
    uint4 a[R];
#pragma unroll
    for (int k = 0; k < R; ++k) a[k] = *reinterpret_cast<const uint4*>(x + (size_t)(r + k) * C + c0);
#pragma unroll
    for (int k = 0; k < R; ++k) {
      float4 lo, hi;
      bf16x8_to_f32(a[k], lo, hi);
      *reinterpret_cast<uint4*>(y + (size_t)(r + k) * C + c0) =
          f32_to_bf16x8(gelu4_of<bf16_raw>(lo, b0), gelu4_of<bf16_raw>(hi, b1));
    }
  }
  for (; r < rows; ++r) {
    float4 lo, hi;
    bf16x8_to_f32(*reinterpret_cast<const uint4*>(x + (size_t)r * C + c0), lo, hi);
    *reinterpret_cast<uint4*>(y + (size_t)r * C + c0) = f32_to_bf16x8(gelu4_of<bf16_raw>(lo, b0), gelu4_of<bf16_raw>(hi, b1));
  }
}

// MODE 0: dx = dy * gelu'(x + bias) ; partial column sums of dx.   MODE 1: plain column sums of dy (no dx).
// MODE 2: as 0 with ReLU (dx = dy where x + bias > 0).
// Grid (row groups, column slices of 1024): thread t owns 4 columns and walks its row group 4 rows at a time
// (8 independent loads per tensor in flight); partials[blockIdx.x][C].
template <typename T, int MODE>
__global__ __launch_bounds__(256) void colwise_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ bias, T* __restrict__ dx,
                                                          float* __restrict__ partials, int rows, int C) {
  const int c0 = blockIdx.y * 1024 + threadIdx.x * 4;
  if (c0 >= C) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE != 1) b = *reinterpret_cast<const float4*>(bias + c0);
  // block b owns the CONTIGUOUS rows [b * rpb, (b + 1) * rpb): the eight rows a thread has in flight are neighbours
  const int rpb = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
  const int stride = 1;
  int r = blockIdx.x * rpb;
  rows = (r + rpb < rows) ? r + rpb : rows;
  constexpr int R = 8;                   // rows in flight per thread
  for (; r + (R - 1) * stride < rows; r += R * stride) {
    float4 d[R], a[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      d[k] = ld4<T>(dy + (size_t)(r + k * stride) * C + c0);
      if (MODE != 1) a[k] = ld4<T>(x + (size_t)(r + k * stride) * C + c0);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (MODE != 1) {
        d[k] = act_grad4_of<T, MODE == 2 ? 1 : 0>(d[k], a[k], b);
        st4<T>(dx + (size_t)(r + k * stride) * C + c0, d[k]);
      }
      acc.x += d[k].x; acc.y += d[k].y; acc.z += d[k].z; acc.w += d[k].w;
    }
  }
  for (; r < rows; r += stride) {
    float4 d = ld4<T>(dy + (size_t)r * C + c0);
    if (MODE != 1) {
      const float4 a = ld4<T>(x + (size_t)r * C + c0);
      d = act_grad4_of<T, MODE == 2 ? 1 : 0>(d, a, b);
      st4<T>(dx + (size_t)r * C + c0, d);
    }
    acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
  }
  *reinterpret_cast<float4*>(partials + (size_t)blockIdx.x * C + c0) = acc;
}

// =============================================================================================
// K6: weighted segment gather: out[r] = sum_{e in [rowptr[r], rowptr[r+1])} w[e] * src[idx[e]]
// (gmap node aggregation with the CSR built on the host from the vpid lists; its backward is the same
//  kernel over the transposed CSR).
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(192) void gather_wsum_kernel(const T* __restrict__ src, const int* __restrict__ rowptr,
                                                          const int* __restrict__ idx, const float* __restrict__ w,
                                                          T* __restrict__ out, int H) {
  const int r = blockIdx.x;
  const int e0 = rowptr[r], e1 = rowptr[r + 1];
  for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = e0; e < e1; ++e) {
      const float ww = w[e];
      const float4 v = ld4<T>(src + (size_t)idx[e] * H + c);
      acc.x += ww * v.x; acc.y += ww * v.y; acc.z += ww * v.z; acc.w += ww * v.w;
    }
    st4<T>(out + (size_t)r * H + c, acc);
  }
}

// sink[i] += sum_s partials[s][i]: the reduction step of the host-side split-K weight-gradient GEMMs, fused with the
// accumulation into the fp32 gradient arena (replaces a torch sum + add_ pair).  S is small (<= 32).
template <typename T>
__global__ __launch_bounds__(256) void accum_partials_kernel(const T* __restrict__ partials, float* __restrict__ sink,
                                                             int S, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 acc = *reinterpret_cast<const float4*>(sink + i * 4);
    for (int s = 0; s < S; ++s) {
      const float4 v = ld4<T>(partials + ((size_t)s * n4 + i) * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(sink + i * 4) = acc;
  }
}

// Batched form of accum_partials_kernel: one launch folds the split-K partial products of EVERY weight-gradient GEMM of
// a backward pass into the fp32 gradient arena (~100 launches of 7-9 us per training step when issued one by one).
// A task = up to 4096 float4 of one weight; the host keeps the table on the device (ops.ReduceQueue).
struct AccumTask {
  const void* partials;    // [S][n4_total] float4-groups of T
  float* sink;             // this task's first element in the arena
  unsigned long long n4_total;   // stride between the S partial slices, in float4 groups
  unsigned int off4, n4;   // range of this task inside a slice, in float4 groups
  int S, dtype;            // dtype: BB_F32 / BB_BF16
};
template <typename T>
__device__ __forceinline__ void accum_task(const AccumTask& t) {
  const T* part = (const T*)t.partials;
  for (unsigned i = threadIdx.x; i < t.n4; i += 256) {
    float4 acc = *reinterpret_cast<const float4*>(t.sink + (size_t)i * 4);
    for (int s = 0; s < t.S; ++s) {
      const float4 v = ld4<T>(part + ((size_t)s * t.n4_total + t.off4 + i) * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(t.sink + (size_t)i * 4) = acc;
  }
}
__global__ __launch_bounds__(256) void multi_accum_kernel(const AccumTask* __restrict__ tasks) {
  const AccumTask t = tasks[blockIdx.x];
  if (t.dtype == BB_BF16) accum_task<bf16_raw>(t);
  else accum_task<float>(t);
}

// table_grad[t] += sum of d[r] over the rows r with ids[r] == t, WITHOUT atomics and in a fixed order (training runs are
// bit-reproducible; the reference's index_add is not).  One workgroup per gradient row r: it returns at once unless r
// is the FIRST row carrying its id (and the id is not padding_idx: nn.Embedding(padding_idx=0) gives the padding row no
// lookup gradient, vilmodel.py:50).  The leader's four waves each scan one contiguous quarter of the rows [r, rows) for
// the same id -- 64 ids per ballot, four ballots' loads in flight -- and add the matching rows of d in ascending row
// order, every lane owning its columns (no LDS list, no barrier inside the scan); the four partial sums are folded as
// (w0 + w1) + (w2 + w3) and the leader alone read-modify-writes the table row.  The result is a pure function of (ids, d).
// rows^2 / 256 id comparisons per launch (5 120 rows: the id vector stays in L2).
#define EG_MAXJ 4          // float4 column groups per lane: H <= 64 lanes * 4 floats * EG_MAXJ = 1024
// TO / ACCUM: the destination rows are fp32 and accumulated into (embedding tables in the gradient arena), or of the
// activations' type -- rows of a zero-initialised activation gradient, the backward of a row gather (bevbert_rows_scatter):
// ACCUM false stores the id's sum, true adds it to what the row holds (a second scatter onto the same tensor).
template <typename T, typename TO = float, bool ACCUM = true>
__global__ __launch_bounds__(256) void embedding_grad_kernel(const int64_t* __restrict__ ids, const T* __restrict__ d,
                                                             TO* __restrict__ table_grad, int rows, int H,
                                                             int padding_idx) {
  __shared__ int s_flag;
  __shared__ float s_part[4][256 * EG_MAXJ];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t id = ids[r];
  if (id == (int64_t)padding_idx) return;
  if (tid == 0) s_flag = 0;
  __syncthreads();
  int hit = 0;
  for (int i = tid; i < r; i += 256) hit |= (ids[i] == id);
  if (hit) s_flag = 1;
  __syncthreads();
  if (s_flag) return;                       // an earlier row leads this id
  float4 acc[EG_MAXJ];
#pragma unroll
  for (int j = 0; j < EG_MAXJ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int span = rows - r, q = (span + 3) >> 2;
  const int lo = r + wave * q, hi = min(rows, lo + q);
  for (int base = lo; base < hi; base += 256) {
    unsigned long long bal[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {           // four independent loads, then four ballots
      const int i = base + 64 * u + lane;
      bal[u] = __ballot(i < hi && ids[i] == id);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      unsigned long long b = bal[u];
      while (b) {                           // wave-uniform: the matching rows of these 64, in ascending order
        const int row = base + 64 * u + (__ffsll((long long)b) - 1);
        b &= b - 1;
#pragma unroll
        for (int j = 0; j < EG_MAXJ; ++j) {
          const int c = (lane + 64 * j) * 4;
          if (c < H) {
            const float4 v = ld4<T>(d + (size_t)row * H + c);
            acc[j].x += v.x; acc[j].y += v.y; acc[j].z += v.z; acc[j].w += v.w;
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < EG_MAXJ; ++j) *reinterpret_cast<float4*>(&s_part[wave][(lane + 64 * j) * 4]) = acc[j];
  __syncthreads();
  TO* dst = table_grad + (size_t)id * H;
  for (int c = tid * 4; c < H; c += 1024) {
    const float4 p0 = *reinterpret_cast<const float4*>(&s_part[0][c]), p1 = *reinterpret_cast<const float4*>(&s_part[1][c]);
    const float4 p2 = *reinterpret_cast<const float4*>(&s_part[2][c]), p3 = *reinterpret_cast<const float4*>(&s_part[3][c]);
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ACCUM) t = ld4<TO>(dst + c);
    t.x += (p0.x + p1.x) + (p2.x + p3.x); t.y += (p0.y + p1.y) + (p2.y + p3.y);
    t.z += (p0.z + p1.z) + (p2.z + p3.z); t.w += (p0.w + p1.w) + (p2.w + p3.w);
    st4<TO>(dst + c, t);
  }
}

// out[i, :] = src[ids[i], :]: one workgroup per output row (index_select of activation rows: the masked tokens of the
// MLM head, the candidate cells of the SAP head, the supervised cells of the semantic head)
template <typename T>
__global__ __launch_bounds__(256) void rows_gather_kernel(const T* __restrict__ src, const int64_t* __restrict__ ids,
                                                          T* __restrict__ out, int H) {
  const int64_t id = ids[blockIdx.x];
  for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4)
    st4<T>(out + (size_t)blockIdx.x * H + c, ld4<T>(src + (size_t)id * H + c));
}

// Small tables (2 .. ~128 rows: navigation types, step ids, token types) with MANY gradient rows: the atomic version
// piles 28 224 rows onto two destination rows.  Here workgroup (t, s) sums the rows of slice s whose id is t and writes
// partials[s][t][:]; the slices are folded into the fp32 table gradient by the step's batched second reduction stage
// (bevbert_multi_finalize): no atomics, a fixed summation order.  The id test is wave-uniform; the loads of a group of
// four rows are issued together.
template <typename T>
__global__ __launch_bounds__(256) void embedding_grad_sliced_kernel(const int64_t* __restrict__ ids, const T* __restrict__ d,
                                                                    float* __restrict__ partials, int rows, int H,
                                                                    int rows_per_slice, int ntab) {
  const int t = blockIdx.x, s = blockIdx.y;
  const int r0 = s * rows_per_slice, r1 = min(rows, r0 + rows_per_slice);
  for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = r0; r < r1; r += 4) {
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r + j < r1 && ids[r + j] == t) v[j] = ld4<T>(d + (size_t)(r + j) * H + c);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    *reinterpret_cast<float4*>(partials + ((size_t)s * ntab + t) * H + c) = acc;
  }
}

// =============================================================================================
// Flat-arena optimiser kernels.  The arena is padded so every tensor starts on a 1024-element
// boundary; flags[i] describes chunk i: bit0 = apply weight decay, bit1 = tensor has ever had a
// gradient (the reference skips params whose .grad is None: adamw.py:66-67).
// =============================================================================================
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n4, float* __restrict__ partials) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = *reinterpret_cast<const float4*>(g + i * 4);
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  __shared__ float sh[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// scalars[0] = total L2 norm of (pre_scale * g) ; scalars[1] = multiplier the optimiser applies to g:
//   pre_scale * min(1, max_norm / (norm + 1e-6))      (torch.nn.utils.clip_grad_norm_)
__global__ void clip_coef_kernel(const float* __restrict__ partials, int n, float pre_scale, float max_norm,
                                 float* __restrict__ scalars) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(sh[0]) * pre_scale;
    float coef = 1.0f;
    if (max_norm > 0.f) {
      coef = max_norm / (norm + 1e-6f);
      coef = coef < 1.0f ? coef : 1.0f;
    }
    scalars[0] = norm;
    scalars[1] = pre_scale * coef;
  }
}

// One block per 1024-element chunk.  The bias-correction step count is PER CHUNK (the reference keeps state["step"]
// per parameter and only advances it when the parameter has a gradient: adamw.py:66-67,84,96-100).
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16_raw* __restrict__ p_bf16, const uint8_t* __restrict__ flags,
                                                    int* __restrict__ chunk_steps, size_t nchunks,
                                                    const float* __restrict__ gscale_ptr,
                                                    const float* __restrict__ lr_ptr, float lr, float beta1,
                                                    float beta2, float eps, float wd, float log2_beta1,
                                                    float log2_beta2) {
  if (lr_ptr) lr = *lr_ptr;            // device-resident learning rate: a captured hipGraph replays with the current one
  const size_t chunk = blockIdx.x;
  if (chunk >= nchunks) return;
  const uint8_t f = flags[chunk];
  if (!(f & 2)) return;
  const int t = chunk_steps[chunk] + 1;          // every thread reads the old value before thread 0 bumps it
  __syncthreads();
  if (threadIdx.x == 0) chunk_steps[chunk] = t;
  // beta^t = 2^(t * log2(beta)) on the transcendental unit (two instructions per thread, exact to ~1e-7)
  const float bc1 = 1.0f - __builtin_amdgcn_exp2f((float)t * log2_beta1);
  const float bc2 = 1.0f - __builtin_amdgcn_exp2f((float)t * log2_beta2);
  const float step_size = lr * sqrtf(bc2) / bc1;
  const float gs = gscale_ptr ? *gscale_ptr : 1.0f;
  const float decay = (f & 1) ? lr * wd : 0.f;
  const size_t i = chunk * 1024 + threadIdx.x * 4;
  const float4 gg = *reinterpret_cast<const float4*>(g + i);
  float4 pp = *reinterpret_cast<float4*>(p + i);
  float4 mm = *reinterpret_cast<float4*>(m + i);
  float4 vv = *reinterpret_cast<float4*>(v + i);
#define UPD(c)                                                    \
  {                                                               \
    const float gr = gg.c * gs;                                   \
    mm.c = mm.c * beta1 + (1.0f - beta1) * gr;                    \
    vv.c = vv.c * beta2 + (1.0f - beta2) * gr * gr;               \
    pp.c = pp.c - step_size * (mm.c / (sqrtf(vv.c) + eps));       \
    pp.c = pp.c - decay * pp.c;                                   \
  }
  UPD(x) UPD(y) UPD(z) UPD(w)
#undef UPD
  *reinterpret_cast<float4*>(p + i) = pp;
  *reinterpret_cast<float4*>(m + i) = mm;
  *reinterpret_cast<float4*>(v + i) = vv;
  if (p_bf16 != nullptr) st4<bf16_raw>(p_bf16 + i, pp);
}

template <typename TO>
__global__ __launch_bounds__(256) void cast_f32_kernel(const float* __restrict__ src, TO* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
    st4<TO>(dst + i * 4, *reinterpret_cast<const float4*>(src + i * 4));
}

// y = residual + dropout(x)   (residual optional; input and output dtypes independent: fuses the fp32 -> bf16 cast of
// the loader's features into their feature dropout).  Mask = the library's counter-based stream (common.h).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void dropout_add_kernel(const TI* __restrict__ x, const TO* __restrict__ residual,
                                                          TO* __restrict__ y, size_t n4, float keep_scale,
                                                          uint32_t thr, uint32_t key, const uint32_t* __restrict__ salt) {
  key = bb_salted(key, salt);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 a = ld4<TI>(x + i * 4);
    const uint32_t pr = (uint32_t)(i * 2);
    const uint32_t b0 = bb_pair_bits(key, pr), b1 = bb_pair_bits(key, pr + 1);
    a.x = bb_keep_lo(b0, thr) ? a.x * keep_scale : 0.f;
    a.y = bb_keep_hi(b0, thr) ? a.y * keep_scale : 0.f;
    a.z = bb_keep_lo(b1, thr) ? a.z * keep_scale : 0.f;
    a.w = bb_keep_hi(b1, thr) ? a.w * keep_scale : 0.f;
    if (residual != nullptr) {
      const float4 r = ld4<TO>(residual + i * 4);
      a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
    }
    st4<TO>(y + i * 4, a);
  }
}

// Column sums for ANY column count (the 30 522-wide vocabulary bias, the 1-wide heads: C % 4 != 0 rules out the float4
// kernels above): out[c] (+)= sum_r dy[r][c].  Wide: workgroup = 64 columns x 4 row groups, folded pairwise.  Narrow
// (C <= 4): the whole workgroup walks the rows.  Fixed summation order in both.
template <typename T>
__global__ __launch_bounds__(256) void colsum_any_kernel(const T* __restrict__ dy, float* __restrict__ out, int rows, int C,
                                                         int accumulate) {
  __shared__ float s_part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (C <= 4) {
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
      for (int r = threadIdx.x; r < rows; r += 256) s += io<T>::ld(dy + (size_t)r * C + c);
      s = wave_sum(s);
      if (lane == 0) s_part[wave][0] = s;
      __syncthreads();
      if (threadIdx.x == 0) {
        const float t = (s_part[0][0] + s_part[1][0]) + (s_part[2][0] + s_part[3][0]);
        out[c] = accumulate ? out[c] + t : t;
      }
      __syncthreads();
    }
    return;
  }
  const int c = blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    int r = wave;
    for (; r + 12 < rows; r += 16) {
      s0 += io<T>::ld(dy + (size_t)r * C + c);
      s1 += io<T>::ld(dy + (size_t)(r + 4) * C + c);
      s2 += io<T>::ld(dy + (size_t)(r + 8) * C + c);
      s3 += io<T>::ld(dy + (size_t)(r + 12) * C + c);
    }
    for (; r < rows; r += 4) s0 += io<T>::ld(dy + (size_t)r * C + c);
  }
  s_part[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && c < C) {
    const float t = (s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]);
    out[c] = accumulate ? out[c] + t : t;
  }
}

BEVBERT_API int bevbert_colsum_any(const void* dy, float* out, int rows, int C, int dtype, int accumulate,
                                   hipStream_t stream) {
  BB_REQUIRE(rows >= 0 && C > 0, "colsum_any: rows=%d C=%d", rows, C);
  if (rows == 0) return BB_OK;
  const dim3 grid(C <= 4 ? 1 : (C + 63) / 64);
  if (dtype == BB_F32)
    hipLaunchKernelGGL(colsum_any_kernel<float>, grid, dim3(256), 0, stream, (const float*)dy, out, rows, C, accumulate);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL(colsum_any_kernel<bf16_raw>, grid, dim3(256), 0, stream, (const bf16_raw*)dy, out, rows, C, accumulate);
  else {
    bb_set_error("colsum_any: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("colsum_any");
  return BB_OK;
}

// loss.mean() of the step (pretrain_src/train_r2r.py:263) with the zero-weight padding rows of a static batch:
// out = sum_i w[i] * x[i] / denom   (w NULL: all ones; denom from device memory when denom_dev is given -- the number of
// real rows of the batch that currently sits in the buffers).  One workgroup (n is at most a few thousand), fixed order.
__global__ __launch_bounds__(1024) void weighted_mean_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ denom_dev, float denom, int n,
                                                                 float* __restrict__ out) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += (w ? w[i] : 1.0f) * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += sh[i];
    *out = t / (denom_dev ? *denom_dev : denom);
  }
}
__global__ __launch_bounds__(256) void weighted_mean_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                                const float* __restrict__ denom_dev, float denom, int n,
                                                                float* __restrict__ dx) {
  const float g = *dout / (denom_dev ? *denom_dev : denom);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dx[i] = (w ? w[i] : 1.0f) * g;
}

BEVBERT_API int bevbert_weighted_mean_fwd(const float* x, const float* w, const float* denom_dev, float denom, int n,
                                          float* out, hipStream_t stream) {
  BB_REQUIRE(n > 0 && x != nullptr && out != nullptr, "weighted_mean_fwd: empty input");
  hipLaunchKernelGGL(weighted_mean_fwd_kernel, dim3(1), dim3(1024), 0, stream, x, w, denom_dev, denom, n, out);
  BB_CHECK_LAUNCH("weighted_mean_fwd");
  return BB_OK;
}

BEVBERT_API int bevbert_weighted_mean_bwd(const float* dout, const float* w, const float* denom_dev, float denom, int n,
                                          float* dx, hipStream_t stream) {
  BB_REQUIRE(n > 0 && dout != nullptr && dx != nullptr, "weighted_mean_bwd: empty input");
  int nb = (n + 255) / 256;
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(weighted_mean_bwd_kernel, dim3(nb), dim3(256), 0, stream, dout, w, denom_dev, denom, n, dx);
  BB_CHECK_LAUNCH("weighted_mean_bwd");
  return BB_OK;
}

// Semantic head (pretrain_cmt.py:391-441, MaskSEM / SEM): the supervised cells are those of a boolean mask (two masks ANDed
// for MaskSEM) -- a data-dependent count only the device knows.  sem_select: ONE workgroup compacts their row numbers into a
// fixed-capacity index (padding: row 0, weight 0), and leaves the weights and the divisor of the mean (count x classes).
// bce_rows: sum over the classes of binary_cross_entropy_with_logits for the selected rows, labels read through the index.
__global__ __launch_bounds__(1024) void sem_select_kernel(const uint8_t* __restrict__ m1, const uint8_t* __restrict__ m2, int n,
                                                         int cap, int classes, int64_t* __restrict__ idx,
                                                         float* __restrict__ valid, float* __restrict__ denom) {
  __shared__ int s_cnt[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024, lo = t * per, hi = min(n, lo + per);
  int c = 0;
  for (int i = lo; i < hi; ++i) c += (m1[i] != 0) && (m2 == nullptr || m2[i] != 0);
  s_cnt[t] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {       // inclusive scan (Hillis-Steele; 10 rounds on 1024 counters)
    const int v = (t >= off) ? s_cnt[t - off] : 0;
    __syncthreads();
    s_cnt[t] += v;
    __syncthreads();
  }
  const int total = s_cnt[1023];
  int pos = s_cnt[t] - c;
  for (int i = lo; i < hi; ++i)
    if ((m1[i] != 0) && (m2 == nullptr || m2[i] != 0)) {
      if (pos < cap) idx[pos] = i;
      ++pos;
    }
  const int kept = total < cap ? total : cap;
  for (int i = t; i < cap; i += 1024) {
    if (i >= kept) idx[i] = 0;
    valid[i] = i < kept ? 1.0f : 0.0f;
  }
  if (t == 0) *denom = (float)total * (float)classes;
}

template <typename T>
__global__ __launch_bounds__(256) void bce_rows_fwd_kernel(const T* __restrict__ x, const uint8_t* __restrict__ labels,
                                                           const int64_t* __restrict__ idx, float* __restrict__ out, int rows,
                                                           int C) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const uint8_t* lab = labels + (size_t)(idx ? idx[row] : row) * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = io<T>::ld(x + (size_t)row * C + c), y = lab[c] ? 1.0f : 0.0f;
    s += fmaxf(v, 0.f) - v * y + log1pf(__expf(-fabsf(v)));       // the numerically stable form torch uses
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}
template <typename T>
__global__ __launch_bounds__(256) void bce_rows_bwd_kernel(const T* __restrict__ x, const uint8_t* __restrict__ labels,
                                                           const int64_t* __restrict__ idx, const float* __restrict__ g,
                                                           T* __restrict__ dx, int rows, int C) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const uint8_t* lab = labels + (size_t)(idx ? idx[row] : row) * C;
  const float gr = g[row];
  for (int c = lane; c < C; c += 64) {
    const float v = io<T>::ld(x + (size_t)row * C + c), y = lab[c] ? 1.0f : 0.0f;
    io<T>::st(dx + (size_t)row * C + c, (1.0f / (1.0f + __expf(-v)) - y) * gr);
  }
}

BEVBERT_API int bevbert_sem_select(const uint8_t* mask1, const uint8_t* mask2, int n, int cap, int classes, int64_t* idx,
                                   float* valid, float* denom, hipStream_t stream) {
  BB_REQUIRE(mask1 != nullptr && n > 0 && cap > 0 && classes > 0, "sem_select: bad arguments");
  hipLaunchKernelGGL(sem_select_kernel, dim3(1), dim3(1024), 0, stream, mask1, mask2, n, cap, classes, idx, valid, denom);
  BB_CHECK_LAUNCH("sem_select");
  return BB_OK;
}

BEVBERT_API int bevbert_bce_rows_fwd(const void* logits, const uint8_t* labels, const int64_t* idx, float* out, int rows,
                                     int C, int dtype, hipStream_t stream) {
  BB_REQUIRE(rows >= 0 && C > 0, "bce_rows_fwd: rows=%d C=%d", rows, C);
  if (rows == 0) return BB_OK;
  const dim3 grid((rows + 3) / 4);
  if (dtype == BB_F32) hipLaunchKernelGGL(bce_rows_fwd_kernel<float>, grid, dim3(256), 0, stream, (const float*)logits, labels, idx, out, rows, C);
  else if (dtype == BB_BF16) hipLaunchKernelGGL(bce_rows_fwd_kernel<bf16_raw>, grid, dim3(256), 0, stream, (const bf16_raw*)logits, labels, idx, out, rows, C);
  else { bb_set_error("bce_rows_fwd: dtype %d unsupported", dtype); return BB_EUNSUPPORTED; }
  BB_CHECK_LAUNCH("bce_rows_fwd");
  return BB_OK;
}

BEVBERT_API int bevbert_bce_rows_bwd(const void* logits, const uint8_t* labels, const int64_t* idx, const float* g, void* dlogits,
                                     int rows, int C, int dtype, hipStream_t stream) {
  BB_REQUIRE(rows >= 0 && C > 0, "bce_rows_bwd: rows=%d C=%d", rows, C);
  if (rows == 0) return BB_OK;
  const dim3 grid((rows + 3) / 4);
  if (dtype == BB_F32) hipLaunchKernelGGL(bce_rows_bwd_kernel<float>, grid, dim3(256), 0, stream, (const float*)logits, labels, idx, g, (float*)dlogits, rows, C);
  else if (dtype == BB_BF16) hipLaunchKernelGGL(bce_rows_bwd_kernel<bf16_raw>, grid, dim3(256), 0, stream, (const bf16_raw*)logits, labels, idx, g, (bf16_raw*)dlogits, rows, C);
  else { bb_set_error("bce_rows_bwd: dtype %d unsupported", dtype); return BB_EUNSUPPORTED; }
  BB_CHECK_LAUNCH("bce_rows_bwd");
  return BB_OK;
}

// Graph-aware attention bias of the global-map encoder (vilmodel.py:543-546, 575-577: sprel_linear = nn.Linear(1, 1) on
// the pairwise node distances): bias = dists * w + b with w, b read from the parameters in device memory; and its backward
// over the per-head, per-layer bias gradients the attention kernels leave: dw = sum dbias * dists, db = sum dbias.  One
// workgroup, a fixed summation order (the tensors are a few MB and the global-map branch runs beside the BEV encoder).
__global__ __launch_bounds__(256) void graph_bias_fwd_kernel(const float* __restrict__ dists, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ out, int n) {
  const float ww = *w, bb = *b;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = dists[i] * ww + bb;
}
// stage 1: workgroup g sums a contiguous run of elements -> partials[g] = (sum dbias * dists, sum dbias); stage 2: one
// wave folds the partials in ascending order and adds them to the two gradients
__global__ __launch_bounds__(256) void graph_bias_bwd_kernel(const float* __restrict__ dbias, const float* __restrict__ dists,
                                                             uint32_t n, uint32_t per_sample, uint32_t per_layer,
                                                             uint32_t nh_gg, uint32_t gg, float2* __restrict__ partials) {
  // dbias: (layers, B, nh, G, G); element e -> sample (e % per_layer) / nh_gg, pair e % gg
  const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = blockIdx.x * per, hi = min(n, lo + per);
  float sw = 0.f, sb = 0.f;
  for (uint32_t e = lo + threadIdx.x; e < hi; e += 256) {
    const float d = dbias[e];
    const uint32_t r = e % per_layer;
    sw = fmaf(d, dists[(r / nh_gg) * per_sample + r % gg], sw);
    sb += d;
  }
  __shared__ float s_w[4], s_b[4];
  sw = wave_sum(sw);
  sb = wave_sum(sb);
  if ((threadIdx.x & 63) == 0) { s_w[threadIdx.x >> 6] = sw; s_b[threadIdx.x >> 6] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = make_float2((s_w[0] + s_w[1]) + (s_w[2] + s_w[3]), (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]));
}
__global__ __launch_bounds__(64) void graph_bias_finalize_kernel(const float2* __restrict__ partials, int nb,
                                                                 float* __restrict__ dw, float* __restrict__ db) {
  float sw = 0.f, sb = 0.f;
  for (int i = threadIdx.x; i < nb; i += 64) { sw += partials[i].x; sb += partials[i].y; }
  sw = wave_sum(sw);
  sb = wave_sum(sb);
  if (threadIdx.x == 0) {
    if (dw) *dw += sw;
    if (db) *db += sb;
  }
}

BEVBERT_API int bevbert_graph_bias_fwd(const float* dists, const float* w, const float* b, float* out, int64_t n,
                                       hipStream_t stream) {
  BB_REQUIRE(n >= 0 && n < 2147483647 && w != nullptr && b != nullptr, "graph_bias_fwd: bad arguments");
  if (n == 0) return BB_OK;
  int nb = (int)((n + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(graph_bias_fwd_kernel, dim3(nb), dim3(256), 0, stream, dists, w, b, out, (int)n);
  BB_CHECK_LAUNCH("graph_bias_fwd");
  return BB_OK;
}

BEVBERT_API int bevbert_graph_bias_bwd(const float* dbias, const float* dists, int layers, int B, int nh, int G, float* dw,
                                       float* db, float* workspace, hipStream_t stream) {
  BB_REQUIRE(layers > 0 && B > 0 && nh > 0 && G > 0 && workspace != nullptr, "graph_bias_bwd: empty problem or no workspace");
  const int64_t per_layer = (int64_t)B * nh * G * G, n = per_layer * layers;
  BB_REQUIRE(n < 2147483647, "graph_bias_bwd: %lld elements do not fit 31 bits", (long long)n);
  int nb = (int)((n + 4095) / 4096);
  if (nb > 512) nb = 512;                                 // workspace: 2 * 512 floats
  hipLaunchKernelGGL(graph_bias_bwd_kernel, dim3(nb), dim3(256), 0, stream, dbias, dists, (uint32_t)n, (uint32_t)(G * G),
                     (uint32_t)per_layer, (uint32_t)(nh * G * G), (uint32_t)(G * G), reinterpret_cast<float2*>(workspace));
  BB_CHECK_LAUNCH("graph_bias_bwd");
  hipLaunchKernelGGL(graph_bias_finalize_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<const float2*>(workspace), nb, dw, db);
  BB_CHECK_LAUNCH("graph_bias_bwd finalize");
  return BB_OK;
}

// =============================================================================================
// C ABI
// =============================================================================================
template <typename T, bool GATHER>
static int ln_fwd_dispatch(int NV, dim3 grid, hipStream_t st, const void* x, const float* bias, const void* residual,
                           const float* gamma, const float* beta, void* y, void* z_out, float* mean, float* rstd,
                           int rows, float eps, float p, uint64_t seed, uint64_t offset, const int64_t* ids,
                           const void* word, const void* pos, const void* type_row, int L) {
  const uint32_t thr = bb_drop_threshold(p);
#define GO(N)                                                                                                        \
  case N:                                                                                                            \
    hipLaunchKernelGGL((ln_fwd_kernel<T, N, GATHER>), grid, dim3(256), 0, st, (const T*)x, bias, (const T*)residual, \
                       gamma, beta, (T*)y, (T*)z_out, mean, rstd, rows, eps, p, thr, bb_site_key(seed, offset), ids, \
                       (const T*)word, (const T*)pos, (const T*)type_row, L, bb_step_salt());                        \
    break;
  switch (NV) {
    GO(1) GO(2) GO(3) GO(4) GO(6) GO(8)
    default:
      bb_set_error("layernorm: H=%d unsupported (need H in {256,512,768,1024,1536,2048})", NV * 256);
      return BB_EUNSUPPORTED;
  }
#undef GO
  return BB_OK;
}

BEVBERT_API int bevbert_bias_dropout_residual_layernorm_fwd(const void* x, const float* bias, const void* residual,
                                                           const float* gamma, const float* beta, void* y, void* z_out,
                                                           float* mean, float* rstd, int rows, int H, float eps,
                                                           int dtype, float drop_p, uint64_t seed, uint64_t offset,
                                                           hipStream_t stream) {
  BB_REQUIRE(rows >= 0 && H % 256 == 0, "layernorm_fwd: H=%d must be a multiple of 256", H);
  BB_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "layernorm_fwd: dropout p=%f", drop_p);
  if (rows == 0) return BB_OK;
  const dim3 grid((rows + 3) / 4);
  int rc;
  if (dtype == BB_F32)
    rc = ln_fwd_dispatch<float, false>(H / 256, grid, stream, x, bias, residual, gamma, beta, y, z_out, mean, rstd,
                                       rows, eps, drop_p, seed, offset, nullptr, nullptr, nullptr, nullptr, 1);
  else if (dtype == BB_BF16)
    rc = ln_fwd_dispatch<bf16_raw, false>(H / 256, grid, stream, x, bias, residual, gamma, beta, y, z_out, mean, rstd,
                                          rows, eps, drop_p, seed, offset, nullptr, nullptr, nullptr, nullptr, 1);
  else {
    bb_set_error("layernorm_fwd: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  if (rc != BB_OK) return rc;
  BB_CHECK_LAUNCH("layernorm_fwd");
  return BB_OK;
}

// y = LayerNorm(x + bias) + post1 + post2 (either may be NULL): the element-wise sums that follow a LayerNorm in the
// embedding compositions (vilmodel.py:494-532 image embeddings, :589-593 BEV input embedding) ride on its store.
BEVBERT_API int bevbert_layernorm_post_fwd(const void* x, const float* bias, const float* gamma, const float* beta,
                                           const void* post1, const void* post2, void* y, void* z_out, float* mean,
                                           float* rstd, int rows, int H, float eps, int dtype, hipStream_t stream) {
  BB_REQUIRE(rows >= 0 && H % 256 == 0, "layernorm_post_fwd: H=%d must be a multiple of 256", H);
  if (rows == 0) return BB_OK;
  const dim3 grid((rows + 3) / 4);
  int rc;
  if (dtype == BB_F32)
    rc = ln_fwd_dispatch<float, false>(H / 256, grid, stream, x, bias, nullptr, gamma, beta, y, z_out, mean, rstd, rows,
                                       eps, 0.f, 0, 0, nullptr, post1, post2, nullptr, 1);
  else if (dtype == BB_BF16)
    rc = ln_fwd_dispatch<bf16_raw, false>(H / 256, grid, stream, x, bias, nullptr, gamma, beta, y, z_out, mean, rstd,
                                          rows, eps, 0.f, 0, 0, nullptr, post1, post2, nullptr, 1);
  else {
    bb_set_error("layernorm_post_fwd: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  if (rc != BB_OK) return rc;
  BB_CHECK_LAUNCH("layernorm_post_fwd");
  return BB_OK;
}

BEVBERT_API int bevbert_embed_sum_layernorm_fwd(const int64_t* ids, const void* word, const void* pos,
                                                const void* type_row, const float* gamma, const float* beta, void* y,
                                                void* z_out, float* mean, float* rstd, int rows, int L, int H,
                                                float eps, int dtype, float drop_p, uint64_t seed, uint64_t offset,
                                                hipStream_t stream) {
  BB_REQUIRE(rows >= 0 && H % 256 == 0 && L > 0, "embed_sum_layernorm_fwd: bad shape rows=%d L=%d H=%d", rows, L, H);
  if (rows == 0) return BB_OK;
  const dim3 grid((rows + 3) / 4);
  int rc;
  if (dtype == BB_F32)
    rc = ln_fwd_dispatch<float, true>(H / 256, grid, stream, nullptr, nullptr, nullptr, gamma, beta, y, z_out, mean,
                                      rstd, rows, eps, drop_p, seed, offset, ids, word, pos, type_row, L);
  else if (dtype == BB_BF16)
    rc = ln_fwd_dispatch<bf16_raw, true>(H / 256, grid, stream, nullptr, nullptr, nullptr, gamma, beta, y, z_out, mean,
                                         rstd, rows, eps, drop_p, seed, offset, ids, word, pos, type_row, L);
  else {
    bb_set_error("embed_sum_layernorm_fwd: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  if (rc != BB_OK) return rc;
  BB_CHECK_LAUNCH("embed_sum_layernorm_fwd");
  return BB_OK;
}

// row groups of the column kernels (and of the LayerNorm backward, whose partial rows go through the same second stage):
// enough blocks to fill the chip, <= 1024 partial rows (BEVBERT_COLWISE_MAX_BLOCKS).  Rows per block: 16 from 8 192 rows
// up, i.e. the cap for the 28 224-row BEV problems: 1 024 blocks = four waves per SIMD.  Until round 6 the cap was 512 (two
// waves per SIMD): the fp32-stream LayerNorm backward, three input streams of 2 + 4 + 4 bytes per element and a reduction
// between its loads and its stores, ran at 4.1 TB/s; with 1 024 blocks 5.15 TB/s (84.3 -> 67.4 us; 768: 70.3, 1 280: 78.6 --
// an uneven last round --, 2 048: 69.2), the bf16 LayerNorm backward and the GELU backward do not care (31.4 -> 30.0,
// 105.4 -> 104.1 us), the bare column sums lose 0.8 us (r06al).  Below 8 192 rows the kernels are latency-bound -- a wave walks its rows one after the other, every row
// a dependent load -> reduce -> store chain -- and 10 rows per block (two or three per wave; 512 blocks for the 5 120 text
// rows instead of 320) shortens that chain.  BEVBERT_ROWS_PER_BLOCK overrides (A/B measurements).
static int colwise_max_blocks() {
  static const int cap = [] { const char* v = getenv("BEVBERT_COLWISE_MAX_BLOCKS"); const int c = v ? atoi(v) : 0; return c > 0 ? c : 1024; }();
  return cap;
}
static int colwise_blocks(int rows) {
  static const int env = [] { const char* v = getenv("BEVBERT_ROWS_PER_BLOCK"); return v ? atoi(v) : 0; }();
  const int per = env > 0 ? env : (rows >= 8192 ? 16 : 10);
  int nb = (rows + per - 1) / per;
  if (nb > colwise_max_blocks()) nb = colwise_max_blocks();
  return nb < 1 ? 1 : nb;
}
static int partial_blocks(int rows, int) { return colwise_blocks(rows); }

// row groups of a purely elementwise row kernel (no partial rows to bound): 8 rows each, at most 4096 groups
static int elementwise_row_groups(int rows) {
  int nb = (rows + 7) / 8;
  if (nb > 4096) nb = 4096;
  return nb < 1 ? 1 : nb;
}

// workspace: >= bevbert_colsum_workspace_floats(3*H) floats
BEVBERT_API int64_t bevbert_colsum_workspace_floats(int total_cols) { return (int64_t)colwise_max_blocks() * total_cols; }

// Two-stage column reductions, split: layernorm_bwd / bias_gelu_bwd called with NULL parameter-gradient outputs leave
// their per-block partial sums in `workspace` ([bevbert_colsum_partial_rows(rows)][nwhich][C] floats); this entry is the
// second stage.  The host runs it on the weight-gradient stream, off the activation-gradient critical path.
BEVBERT_API int bevbert_colsum_partial_rows(int rows) { return colwise_blocks(rows); }

BEVBERT_API int bevbert_colsum_finalize(const float* partials, int nblocks, int nwhich, int C, float* out0, float* out1,
                                        float* out2, int accumulate, hipStream_t stream) {
  BB_REQUIRE(nblocks >= 1 && nwhich >= 1 && nwhich <= 3 && C >= 1, "colsum_finalize: bad shape (%d, %d, %d)", nblocks, nwhich, C);
  launch_finalize(partials, nblocks, nwhich, C, out0, out1, out2, accumulate, stream);
  BB_CHECK_LAUNCH("colsum_finalize");
  return BB_OK;
}

static int layernorm_bwd_impl(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                              void* dz, void* dx, const void* dz_add, float* dgamma, float* dbeta, float* dbias,
                              float* workspace, int rows, int H, int dtype, float drop_p, uint64_t seed, uint64_t offset,
                              int accumulate, hipStream_t stream);

BEVBERT_API int bevbert_layernorm_bwd(const void* dy, const void* z, const float* mean, const float* rstd,
                                      const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta,
                                      float* dbias, float* workspace, int rows, int H, int dtype, float drop_p,
                                      uint64_t seed, uint64_t offset, int accumulate, hipStream_t stream) {
  return layernorm_bwd_impl(dy, z, mean, rstd, gamma, dz, dx, nullptr, dgamma, dbeta, dbias, workspace, rows, H, dtype,
                            drop_p, seed, offset, accumulate, stream);
}

// The same with an addend: dz (and dx behind the dropout mask) receive LayerNorm's input gradient PLUS dz_add -- the
// gradient that reaches z through its other consumer when z is also an output (the pre-norm residual stream of the
// panorama encoder: src = src + dropout(sublayer(norm(src))), transformer.py:170-182).
BEVBERT_API int bevbert_layernorm_bwd_add(const void* dy, const void* z, const float* mean, const float* rstd,
                                          const float* gamma, void* dz, void* dx, const void* dz_add, float* dgamma,
                                          float* dbeta, float* dbias, float* workspace, int rows, int H, int dtype,
                                          float drop_p, uint64_t seed, uint64_t offset, int accumulate,
                                          hipStream_t stream) {
  return layernorm_bwd_impl(dy, z, mean, rstd, gamma, dz, dx, dz_add, dgamma, dbeta, dbias, workspace, rows, H, dtype,
                            drop_p, seed, offset, accumulate, stream);
}

static int layernorm_bwd_impl(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                              void* dz, void* dx, const void* dz_add, float* dgamma, float* dbeta, float* dbias,
                              float* workspace, int rows, int H, int dtype, float drop_p, uint64_t seed, uint64_t offset,
                              int accumulate, hipStream_t stream) {
  BB_REQUIRE(H % 256 == 0, "layernorm_bwd: H=%d must be a multiple of 256", H);
  if (rows <= 0) return BB_OK;
  const int nb = partial_blocks(rows, 16);
  const uint32_t thr = bb_drop_threshold(drop_p);
#define GO(T, N)                                                                                              \
  hipLaunchKernelGGL((ln_bwd_kernel<T, N>), dim3(nb), dim3(256), 0, stream, (const T*)dy, (const T*)z, mean, \
                     rstd, gamma, (T*)dz, (T*)dx, workspace, rows, drop_p, thr, bb_site_key(seed, offset), bb_step_salt(), \
                     (const T*)dz_add)
#define SW(T)                                                                     \
  switch (H / 256) {                                                              \
    case 1: GO(T, 1); break;                                                      \
    case 2: GO(T, 2); break;                                                      \
    case 3: GO(T, 3); break;                                                      \
    case 4: GO(T, 4); break;                                                      \
    default: bb_set_error("layernorm_bwd: H=%d unsupported", H); return BB_EUNSUPPORTED; \
  }
  if (dtype == BB_F32) { SW(float) } else if (dtype == BB_BF16) { SW(bf16_raw) } else {
    bb_set_error("layernorm_bwd: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
#undef SW
#undef GO
  BB_CHECK_LAUNCH("layernorm_bwd");
  if (dgamma || dbeta || dbias) launch_finalize(workspace, nb, 3, H, dgamma, dbeta, dbias, accumulate, stream);
  BB_CHECK_LAUNCH("layernorm_bwd finalize");
  return BB_OK;
}

// bf16 activations around an fp32 residual stream (see ln_res32_*_kernel): residual_dtype BB_F32 / BB_BF16 (ignored when
// residual is NULL); y32 / z32 may be NULL (inference: no backward, or a consumer that only wants the bf16 copy)
BEVBERT_API int bevbert_layernorm_res32_fwd(const void* x, const float* bias, const void* residual, int residual_dtype,
                                            const float* gamma, const float* beta, void* y16, float* y32, float* z32,
                                            float* mean, float* rstd, int rows, int H, float eps, float drop_p,
                                            uint64_t seed, uint64_t offset, hipStream_t stream) {
  BB_REQUIRE(rows >= 0 && H % 256 == 0 && H / 256 <= 4, "layernorm_res32_fwd: H=%d must be 256, 512, 768 or 1024", H);
  BB_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "layernorm_res32_fwd: dropout p=%f", drop_p);
  BB_REQUIRE(residual == nullptr || residual_dtype == BB_F32 || residual_dtype == BB_BF16, "layernorm_res32_fwd: residual dtype %d", residual_dtype);
  if (rows == 0) return BB_OK;
  const dim3 grid((rows + 3) / 4);
  const uint32_t thr = bb_drop_threshold(drop_p);
#define GO(TR, N)                                                                                                       \
  hipLaunchKernelGGL((ln_res32_fwd_kernel<TR, N>), grid, dim3(256), 0, stream, (const bf16_raw*)x, bias, (const TR*)residual, \
                     gamma, beta, (bf16_raw*)y16, y32, z32, mean, rstd, rows, eps, drop_p, thr, bb_site_key(seed, offset),   \
                     bb_step_salt())
#define SW(TR)                \
  switch (H / 256) {          \
    case 1: GO(TR, 1); break; \
    case 2: GO(TR, 2); break; \
    case 3: GO(TR, 3); break; \
    default: GO(TR, 4); break; \
  }
  if (residual != nullptr && residual_dtype == BB_BF16) { SW(bf16_raw) } else { SW(float) }
#undef SW
#undef GO
  BB_CHECK_LAUNCH("layernorm_res32_fwd");
  return BB_OK;
}

BEVBERT_API int bevbert_layernorm_res32_bwd(const void* dy16, const float* dy32, const float* z32, const float* mean,
                                            const float* rstd, const float* gamma, void* dz, void* dx16, float* dgamma,
                                            float* dbeta, float* dbias, float* workspace, int rows, int H, float drop_p,
                                            uint64_t seed, uint64_t offset, int accumulate, int dz_dtype,
                                            hipStream_t stream) {
  BB_REQUIRE(dz_dtype == BB_F32 || dz_dtype == BB_BF16, "layernorm_res32_bwd: dz dtype %d (fp32 or bf16)", dz_dtype);
  BB_REQUIRE(H % 256 == 0 && H / 256 <= 4, "layernorm_res32_bwd: H=%d must be 256, 512, 768 or 1024", H);
  BB_REQUIRE(dy16 != nullptr || dy32 != nullptr, "layernorm_res32_bwd: no output gradient");
  if (rows <= 0) return BB_OK;
  const int nb = partial_blocks(rows, 16);
  const uint32_t thr = bb_drop_threshold(drop_p);
#define GO(N)                                                                                                          \
  hipLaunchKernelGGL((ln_res32_bwd_kernel<N>), dim3(nb), dim3(256), 0, stream, (const bf16_raw*)dy16, dy32, z32, mean, rstd, \
                     gamma, (float*)dz, (bf16_raw*)dx16, workspace, rows, drop_p, thr, bb_site_key(seed, offset), bb_step_salt(), \
                     dz_dtype == BB_BF16 ? 1 : 0)
  switch (H / 256) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 3: GO(3); break;
    default: GO(4); break;
  }
#undef GO
  BB_CHECK_LAUNCH("layernorm_res32_bwd");
  if (dgamma || dbeta || dbias) launch_finalize(workspace, nb, 3, H, dgamma, dbeta, dbias, accumulate, stream);
  BB_CHECK_LAUNCH("layernorm_res32_bwd finalize");
  return BB_OK;
}

static int bias_act_fwd(const void* x, const float* bias, void* y, int rows, int C, int dtype, int act, hipStream_t stream) {
  BB_REQUIRE(C % 4 == 0, "bias_act_fwd: C=%d must be a multiple of 4", C);
  BB_REQUIRE(act == 0 || act == 1, "bias_act_fwd: activation %d (0 erf-GELU, 1 ReLU)", act);
  if (rows <= 0) return BB_OK;
  const dim3 grid(elementwise_row_groups(rows), (C + 1023) / 1024);
  if (dtype == BB_F32 && act == 0)
    hipLaunchKernelGGL((bias_gelu_fwd_kernel<float, 0>), grid, dim3(256), 0, stream, (const float*)x, bias, (float*)y, rows, C);
  else if (dtype == BB_F32)
    hipLaunchKernelGGL((bias_gelu_fwd_kernel<float, 1>), grid, dim3(256), 0, stream, (const float*)x, bias, (float*)y, rows, C);
  else if (dtype == BB_BF16 && act == 0 && C % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0)   // 71 vs 76 us at 28224 x 3072
    hipLaunchKernelGGL(bias_gelu_fwd8_kernel, grid, dim3(128), 0, stream, (const bf16_raw*)x, bias, (bf16_raw*)y, rows, C);
  else if (dtype == BB_BF16 && act == 0)
    hipLaunchKernelGGL((bias_gelu_fwd_kernel<bf16_raw, 0>), grid, dim3(256), 0, stream, (const bf16_raw*)x, bias, (bf16_raw*)y, rows, C);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL((bias_gelu_fwd_kernel<bf16_raw, 1>), grid, dim3(256), 0, stream, (const bf16_raw*)x, bias, (bf16_raw*)y, rows, C);
  else {
    bb_set_error("bias_act_fwd: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("bias_act_fwd");
  return BB_OK;
}

static int bias_act_bwd(const void* dy, const void* x, const float* bias, void* dx, float* dbias, float* workspace, int rows,
                        int C, int dtype, int accumulate, int act, hipStream_t stream) {
  BB_REQUIRE(C % 4 == 0, "bias_act_bwd: C=%d must be a multiple of 4", C);
  BB_REQUIRE(act == 0 || act == 1, "bias_act_bwd: activation %d (0 erf-GELU, 1 ReLU)", act);
  if (rows <= 0) return BB_OK;
  const int nb = colwise_blocks(rows);
  const dim3 grid(nb, (C + 1023) / 1024);
  if (dtype == BB_F32 && act == 0)
    hipLaunchKernelGGL((colwise_bwd_kernel<float, 0>), grid, dim3(256), 0, stream, (const float*)dy, (const float*)x, bias, (float*)dx, workspace, rows, C);
  else if (dtype == BB_F32)
    hipLaunchKernelGGL((colwise_bwd_kernel<float, 2>), grid, dim3(256), 0, stream, (const float*)dy, (const float*)x, bias, (float*)dx, workspace, rows, C);
  else if (dtype == BB_BF16 && act == 0)
    hipLaunchKernelGGL((colwise_bwd_kernel<bf16_raw, 0>), grid, dim3(256), 0, stream, (const bf16_raw*)dy, (const bf16_raw*)x, bias, (bf16_raw*)dx, workspace, rows, C);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL((colwise_bwd_kernel<bf16_raw, 2>), grid, dim3(256), 0, stream, (const bf16_raw*)dy, (const bf16_raw*)x, bias, (bf16_raw*)dx, workspace, rows, C);
  else {
    bb_set_error("bias_act_bwd: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("bias_act_bwd");
  if (dbias) {
    launch_finalize(workspace, nb, 1, C, dbias, nullptr, nullptr, accumulate, stream);
    BB_CHECK_LAUNCH("bias_act_bwd finalize");
  }
  return BB_OK;
}

BEVBERT_API int bevbert_bias_gelu_fwd(const void* x, const float* bias, void* y, int rows, int C, int dtype,
                                      hipStream_t stream) {
  return bias_act_fwd(x, bias, y, rows, C, dtype, 0, stream);
}

BEVBERT_API int bevbert_bias_gelu_bwd(const void* dy, const void* x, const float* bias, void* dx, float* dbias,
                                      float* workspace, int rows, int C, int dtype, int accumulate,
                                      hipStream_t stream) {
  return bias_act_bwd(dy, x, bias, dx, dbias, workspace, rows, C, dtype, accumulate, 0, stream);
}

// The prediction heads' Linear -> ReLU (pretrain_src/model/pretrain_cmt.py:34-71) on the same kernels.
BEVBERT_API int bevbert_bias_relu_fwd(const void* x, const float* bias, void* y, int rows, int C, int dtype,
                                      hipStream_t stream) {
  return bias_act_fwd(x, bias, y, rows, C, dtype, 1, stream);
}

BEVBERT_API int bevbert_bias_relu_bwd(const void* dy, const void* x, const float* bias, void* dx, float* dbias,
                                      float* workspace, int rows, int C, int dtype, int accumulate,
                                      hipStream_t stream) {
  return bias_act_bwd(dy, x, bias, dx, dbias, workspace, rows, C, dtype, accumulate, 1, stream);
}

BEVBERT_API int bevbert_colsum(const void* dy, float* out, float* workspace, int rows, int C, int dtype, int accumulate,
                               hipStream_t stream) {
  BB_REQUIRE(C % 4 == 0, "colsum: C=%d must be a multiple of 4", C);
  if (rows <= 0) return BB_OK;
  const int nb = colwise_blocks(rows);
  const dim3 grid(nb, (C + 1023) / 1024);
  if (dtype == BB_F32)
    hipLaunchKernelGGL((colwise_bwd_kernel<float, 1>), grid, dim3(256), 0, stream, (const float*)dy, nullptr, nullptr, nullptr, workspace, rows, C);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL((colwise_bwd_kernel<bf16_raw, 1>), grid, dim3(256), 0, stream, (const bf16_raw*)dy, nullptr, nullptr, nullptr, workspace, rows, C);
  else {
    bb_set_error("colsum: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("colsum");
  launch_finalize(workspace, nb, 1, C, out, nullptr, nullptr, accumulate, stream);
  BB_CHECK_LAUNCH("colsum finalize");
  return BB_OK;
}

// First stage of bevbert_colsum alone: partials [bevbert_colsum_partial_rows(rows)][C]; the second stage goes through
// bevbert_multi_finalize together with the other pending reductions of the step.
BEVBERT_API int bevbert_colsum_partials(const void* dy, float* partials, int rows, int C, int dtype, hipStream_t stream) {
  BB_REQUIRE(C % 4 == 0 && rows > 0, "colsum_partials: C=%d must be a multiple of 4, rows=%d positive", C, rows);
  const int nb = colwise_blocks(rows);
  const dim3 grid(nb, (C + 1023) / 1024);
  if (dtype == BB_F32)
    hipLaunchKernelGGL((colwise_bwd_kernel<float, 1>), grid, dim3(256), 0, stream, (const float*)dy, nullptr, nullptr, nullptr, partials, rows, C);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL((colwise_bwd_kernel<bf16_raw, 1>), grid, dim3(256), 0, stream, (const bf16_raw*)dy, nullptr, nullptr, nullptr, partials, rows, C);
  else {
    bb_set_error("colsum_partials: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("colsum_partials");
  return BB_OK;
}

// tasks: device array of `ntasks` 40-byte records {u64 partials, u64 sink, u64 n4_total, u32 off4, u32 n4, i32 S, i32 dtype}
BEVBERT_API int bevbert_multi_accum(const void* tasks, int ntasks, hipStream_t stream) {
  static_assert(sizeof(AccumTask) == 40, "AccumTask is part of the C ABI");
  if (ntasks <= 0) return BB_OK;
  BB_REQUIRE(tasks != nullptr, "multi_accum: null task table");
  hipLaunchKernelGGL(multi_accum_kernel, dim3(ntasks), dim3(256), 0, stream, (const AccumTask*)tasks);
  BB_CHECK_LAUNCH("multi_accum");
  return BB_OK;
}

// tasks: device array of `ntasks` 40-byte records {u64 partials, u64 out, i32 nblocks, row_stride, col0, ncols, accumulate, pad}
BEVBERT_API int bevbert_multi_finalize(const void* tasks, int ntasks, hipStream_t stream) {
  static_assert(sizeof(FinalizeTask) == 40, "FinalizeTask is part of the C ABI");
  if (ntasks <= 0) return BB_OK;
  BB_REQUIRE(tasks != nullptr, "multi_finalize: null task table");
  hipLaunchKernelGGL(multi_finalize_kernel, dim3(ntasks), dim3(64, 16), 0, stream, (const FinalizeTask*)tasks);
  BB_CHECK_LAUNCH("multi_finalize");
  return BB_OK;
}

BEVBERT_API int bevbert_segment_wsum(const void* src, const int* rowptr, const int* idx, const float* w, void* out,
                                     int out_rows, int H, int dtype, hipStream_t stream) {
  BB_REQUIRE(H % 4 == 0, "segment_wsum: H=%d must be a multiple of 4", H);
  if (out_rows <= 0) return BB_OK;
  if (dtype == BB_F32)
    hipLaunchKernelGGL(gather_wsum_kernel<float>, dim3(out_rows), dim3(192), 0, stream, (const float*)src, rowptr, idx, w, (float*)out, H);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL(gather_wsum_kernel<bf16_raw>, dim3(out_rows), dim3(192), 0, stream, (const bf16_raw*)src, rowptr, idx, w, (bf16_raw*)out, H);
  else {
    bb_set_error("segment_wsum: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("segment_wsum");
  return BB_OK;
}

BEVBERT_API int bevbert_accum_partials(const void* partials, float* sink, int S, int64_t n, int dtype,
                                       hipStream_t stream) {
  BB_REQUIRE(n % 4 == 0 && S >= 1, "accum_partials: n must be a multiple of 4 and S >= 1");
  if (n == 0) return BB_OK;
  size_t nb = ((size_t)n / 4 + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (dtype == BB_F32)
    hipLaunchKernelGGL(accum_partials_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)partials, sink, S, (size_t)n / 4);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL(accum_partials_kernel<bf16_raw>, dim3(nb), dim3(256), 0, stream, (const bf16_raw*)partials, sink, S, (size_t)n / 4);
  else {
    bb_set_error("accum_partials: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("accum_partials");
  return BB_OK;
}

BEVBERT_API int bevbert_rows_gather(const void* src, const int64_t* ids, void* out, int rows, int H, int dtype,
                                    hipStream_t stream) {
  BB_REQUIRE(H % 4 == 0 && H > 0, "rows_gather: H=%d must be a positive multiple of 4", H);
  if (rows <= 0) return BB_OK;
  const int nt = H / 4 >= 256 ? 256 : (H / 4 + 63) / 64 * 64;
  if (dtype == BB_F32)
    hipLaunchKernelGGL(rows_gather_kernel<float>, dim3(rows), dim3(nt), 0, stream, (const float*)src, ids, (float*)out, H);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL(rows_gather_kernel<bf16_raw>, dim3(rows), dim3(nt), 0, stream, (const bf16_raw*)src, ids, (bf16_raw*)out, H);
  else {
    bb_set_error("rows_gather: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("rows_gather");
  return BB_OK;
}

BEVBERT_API int bevbert_rows_scatter(const int64_t* ids, const void* d, void* out, int rows, int H, int dtype, int accumulate,
                                     hipStream_t stream) {
  BB_REQUIRE(H % 4 == 0 && H <= 256 * EG_MAXJ, "rows_scatter: H=%d must be a multiple of 4 and <= %d", H, 256 * EG_MAXJ);
  if (rows <= 0) return BB_OK;
  if (dtype == BB_F32 && accumulate)
    hipLaunchKernelGGL((embedding_grad_kernel<float, float, true>), dim3(rows), dim3(256), 0, stream, ids, (const float*)d, (float*)out, rows, H, -1);
  else if (dtype == BB_F32)
    hipLaunchKernelGGL((embedding_grad_kernel<float, float, false>), dim3(rows), dim3(256), 0, stream, ids, (const float*)d, (float*)out, rows, H, -1);
  else if (dtype == BB_BF16 && accumulate)
    hipLaunchKernelGGL((embedding_grad_kernel<bf16_raw, bf16_raw, true>), dim3(rows), dim3(256), 0, stream, ids, (const bf16_raw*)d, (bf16_raw*)out, rows, H, -1);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL((embedding_grad_kernel<bf16_raw, bf16_raw, false>), dim3(rows), dim3(256), 0, stream, ids, (const bf16_raw*)d, (bf16_raw*)out, rows, H, -1);
  else {
    bb_set_error("rows_scatter: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("rows_scatter");
  return BB_OK;
}

BEVBERT_API int bevbert_embedding_grad(const int64_t* ids, const void* d, float* table_grad, int rows, int H,
                                       int padding_idx, int dtype, hipStream_t stream) {
  BB_REQUIRE(H % 4 == 0 && H <= 256 * EG_MAXJ, "embedding_grad: H=%d must be a multiple of 4 and <= %d", H, 256 * EG_MAXJ);
  if (rows <= 0) return BB_OK;
  if (dtype == BB_F32)
    hipLaunchKernelGGL(embedding_grad_kernel<float>, dim3(rows), dim3(256), 0, stream, ids, (const float*)d, table_grad, rows, H,
                       padding_idx);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL(embedding_grad_kernel<bf16_raw>, dim3(rows), dim3(256), 0, stream, ids, (const bf16_raw*)d, table_grad,
                       rows, H, padding_idx);
  else {
    bb_set_error("embedding_grad: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("embedding_grad");
  return BB_OK;
}

BEVBERT_API int bevbert_embedding_grad_sliced(const int64_t* ids, const void* d, float* partials, int rows, int H,
                                              int table_rows, int rows_per_slice, int dtype, hipStream_t stream) {
  BB_REQUIRE(H % 4 == 0, "embedding_grad_sliced: H=%d must be a multiple of 4", H);
  BB_REQUIRE(table_rows > 0 && rows_per_slice > 0 && rows > 0, "embedding_grad_sliced: empty problem (%d table rows, %d rows)",
             table_rows, rows);
  const int slices = (rows + rows_per_slice - 1) / rows_per_slice;
  BB_REQUIRE(slices <= 65535, "embedding_grad_sliced: %d slices exceed the grid limit", slices);
  const int nt = H / 4 < 256 ? ((H / 4 + 63) / 64) * 64 : 256;
  const dim3 grid(table_rows, slices);
  if (dtype == BB_F32)
    hipLaunchKernelGGL(embedding_grad_sliced_kernel<float>, grid, dim3(nt), 0, stream, ids, (const float*)d, partials, rows, H,
                       rows_per_slice, table_rows);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL(embedding_grad_sliced_kernel<bf16_raw>, grid, dim3(nt), 0, stream, ids, (const bf16_raw*)d, partials,
                       rows, H, rows_per_slice, table_rows);
  else {
    bb_set_error("embedding_grad_sliced: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("embedding_grad_sliced");
  return BB_OK;
}

// scalars: device float[2] -> {norm, grad multiplier}; partials: device float[1024]
BEVBERT_API int bevbert_grad_norm_clip(const float* grads, int64_t n, float pre_scale, float max_norm,
                                       float* partials, float* scalars, hipStream_t stream) {
  BB_REQUIRE(n % 4 == 0, "grad_norm_clip: n must be a multiple of 4");
  hipLaunchKernelGGL(sumsq_kernel, dim3(1024), dim3(256), 0, stream, grads, (size_t)n / 4, partials);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, stream, partials, 1024, pre_scale, max_norm, scalars);
  BB_CHECK_LAUNCH("grad_norm_clip");
  return BB_OK;
}

BEVBERT_API int bevbert_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                   void* params_bf16, const uint8_t* chunk_flags, int* chunk_steps, int64_t n,
                                   const float* grad_scale_dev, const float* lr_dev, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, hipStream_t stream) {
  BB_REQUIRE(n % 1024 == 0, "adamw_step: arena length must be a multiple of 1024 elements");
  BB_REQUIRE(chunk_steps != nullptr, "adamw_step: per-chunk step counters are required");
  const size_t nchunks = (size_t)n / 1024;
  hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq,
                     (bf16_raw*)params_bf16, chunk_flags, chunk_steps, nchunks, grad_scale_dev, lr_dev, lr, beta1, beta2, eps,
                     weight_decay, (float)log2((double)beta1), (float)log2((double)beta2));
  BB_CHECK_LAUNCH("adamw_step");
  return BB_OK;
}

BEVBERT_API int bevbert_cast_f32(const float* src, void* dst, int64_t n, int dst_dtype, hipStream_t stream) {
  BB_REQUIRE(n % 4 == 0, "cast_f32: n must be a multiple of 4");
  if (n == 0) return BB_OK;
  size_t nb = ((size_t)n / 4 + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (dst_dtype == BB_BF16)
    hipLaunchKernelGGL(cast_f32_kernel<bf16_raw>, dim3(nb), dim3(256), 0, stream, src, (bf16_raw*)dst, (size_t)n / 4);
  else if (dst_dtype == BB_F16)
    hipLaunchKernelGGL(cast_f32_kernel<_Float16>, dim3(nb), dim3(256), 0, stream, src, (_Float16*)dst, (size_t)n / 4);
  else {
    bb_set_error("cast_f32: dst dtype %d unsupported", dst_dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("cast_f32");
  return BB_OK;
}

BEVBERT_API int bevbert_dropout_add(const void* x, const void* residual, void* y, int64_t n, int in_dtype,
                                    int out_dtype, float drop_p, uint64_t seed, uint64_t offset, hipStream_t stream) {
  BB_REQUIRE(n % 4 == 0 && n < ((int64_t)1 << 32), "dropout_add: n=%ld must be a multiple of 4 below 2^32", (long)n);
  BB_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "dropout_add: p=%f out of range", drop_p);
  if (n == 0) return BB_OK;
  size_t nb = ((size_t)n / 4 + 255) / 256;
  if (nb > 8192) nb = 8192;
  const float ks = 1.0f / (1.0f - drop_p);
  const uint32_t thr = bb_drop_threshold(drop_p), key = bb_site_key(seed, offset);
#define GO(TI, TO)                                                                                                   \
  hipLaunchKernelGGL((dropout_add_kernel<TI, TO>), dim3(nb), dim3(256), 0, stream, (const TI*)x, (const TO*)residual, \
                     (TO*)y, (size_t)n / 4, ks, thr, key, bb_step_salt())
  if (in_dtype == BB_F32 && out_dtype == BB_F32) GO(float, float);
  else if (in_dtype == BB_F32 && out_dtype == BB_BF16) GO(float, bf16_raw);
  else if (in_dtype == BB_BF16 && out_dtype == BB_BF16) GO(bf16_raw, bf16_raw);
  else if (in_dtype == BB_F16 && out_dtype == BB_F16) GO(_Float16, _Float16);
  else if (in_dtype == BB_F32 && out_dtype == BB_F16) GO(float, _Float16);
  else {
    bb_set_error("dropout_add: dtype pair %d -> %d unsupported", in_dtype, out_dtype);
    return BB_EUNSUPPORTED;
  }
#undef GO
  BB_CHECK_LAUNCH("dropout_add");
  return BB_OK;
}
