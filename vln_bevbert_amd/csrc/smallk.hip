// Feature projections with a tiny inner dimension fused into the LayerNorm that follows them:
//
//     y = LayerNorm(feat W^T + b) + post1 + table[idx]            feat (rows, K) fp32, K <= 16, W (H, K)
//
// Reference sites (pretrain_src/model/vilmodel.py): ImageEmbeddings  :507-515  loc_layer_norm(loc_linear(loc_fts))   K = 7
//                                                   LocalBEVEncoder  :577-583  bev_pos_embeddings(bev_pos_fts)       K = 10
//                                                   GlobalMapEncoder :589-593  gmap_pos_embeddings(gmap_pos_fts)     K = 7
// each followed by the sum with the image / BEV / map features (post1) and a nav-type / step embedding row (table[idx]).
//
// Rounds 1-5 ran these as library GEMMs on the fp32 masters (K is not MFMA-tileable) + a cast to bf16 + the LayerNorm
// kernel, and in backward a cast back + a (H x rows) x (rows x K) GEMM that the library serves with a 16 x 16 x 256 tile
// (76 us at 28 224 rows).  The product feat W^T is 2 K flops per output element: cheaper to RECOMPUTE inside the row
// kernels than to move -- the forward writes only y, the backward reads only dy (no z, no dz tensor):
//
//   forward   W^T staged once per workgroup in LDS ([K][H] fp32); a wave owns a row, its features are broadcast from
//             lanes 0..K-1; statistics, affine, post terms and the gathered table row as in ln_fwd_kernel (rowops.hip);
//   backward  phase A (wave per row): z recomputed, LayerNorm backward -> dz (fp32) into an LDS tile of R rows, running
//             column sums of dy * xhat (dgamma), dy (dbeta), dz (dbias) in registers;
//             phase B (thread per column triple): dW[c][k] += dz[r][c] * feat[r][k] over the R rows of the tile;
//             per-workgroup partials [K + 3][H], folded into the gradient arena by smallk_finalize_kernel in a fixed
//             order (no atomics).
#include "common.h"

#define SK_MAXK 16
#define SK_R 8            // rows per LDS tile of the backward (two per wave)

template <typename T, int NV>
__global__ __launch_bounds__(256) void smallk_ln_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ W,
                                                            const float* __restrict__ bias, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const T* __restrict__ post1,
                                                            const T* __restrict__ table, const int64_t* __restrict__ idx,
                                                            T* __restrict__ y, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int rows, int K, float eps) {
  constexpr int H = NV * 256;
  extern __shared__ __attribute__((aligned(16))) float s_wt[];      // [K][H]
  for (int i = threadIdx.x; i < K * H; i += 256) {
    const int k = i / H, c = i - k * H;
    s_wt[i] = W[c * K + k];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float f = lane < K ? feat[(size_t)row * K + lane] : 0.f;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias != nullptr) v[i] = *reinterpret_cast<const float4*>(bias + (i * 64 + lane) * 4);
    }
    for (int k = 0; k < K; ++k) {
      const float fk = __shfl(f, k, 64);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 w = *reinterpret_cast<const float4*>(s_wt + k * H + (i * 64 + lane) * 4);
        v[i].x = fmaf(fk, w.x, v[i].x); v[i].y = fmaf(fk, w.y, v[i].y);
        v[i].z = fmaf(fk, w.z, v[i].z); v[i].w = fmaf(fk, w.w, v[i].w);
      }
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
    const T* trow = table != nullptr ? table + (size_t)idx[row] * H : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int col = (i * 64 + lane) * 4;
      const float4 g = *reinterpret_cast<const float4*>(gamma + col);
      const float4 b = *reinterpret_cast<const float4*>(beta + col);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (post1 != nullptr) {       // y = (LN(..) + post1) + table row, in that order (vilmodel.py:516-518, 583, 592-593)
        const float4 r = ld4<T>(post1 + (size_t)row * H + col);
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      if (trow != nullptr) {
        const float4 r = ld4<T>(trow + col);
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      st4<T>(y + (size_t)row * H + col, o);
    }
  }
}

// partials: [gridDim.x][K + 3][H]: planes 0..K-1 = dW^T, K = dgamma, K + 1 = dbeta, K + 2 = dbias
template <typename T, int NV>
__global__ __launch_bounds__(256) void smallk_ln_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ feat,
                                                            const float* __restrict__ W, const float* __restrict__ bias,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, float* __restrict__ partials,
                                                            int rows, int K) {
  constexpr int H = NV * 256;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* const s_wt = sm;                          // [K][H]
  float* const s_dz = sm + K * H;                  // [SK_R][H]  (reused for the cross-wave fold of the column sums)
  float* const s_f = s_dz + SK_R * H;              // [SK_R][SK_MAXK]
  for (int i = threadIdx.x; i < K * H; i += 256) {
    const int k = i / H, c = i - k * H;
    s_wt[i] = W[c * K + k];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
  float4 ag[NV], ab[NV], ax[NV], g[NV], bi[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ag[i] = ab[i] = ax[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    g[i] = *reinterpret_cast<const float4*>(gamma + (i * 64 + lane) * 4);
    bi[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias != nullptr) bi[i] = *reinterpret_cast<const float4*>(bias + (i * 64 + lane) * 4);
  }
  float acc[NV][SK_MAXK];                          // phase B: columns tid + 256 j, all k
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int k = 0; k < SK_MAXK; ++k) acc[j][k] = 0.f;
  __syncthreads();
  const int ntiles = (rows + SK_R - 1) / SK_R;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // ---- phase A: two rows per wave
#pragma unroll
    for (int rr = 0; rr < SK_R / 4; ++rr) {
      const int lr = wave * (SK_R / 4) + rr, row = tile * SK_R + lr;
      if (row < rows) {
        const float f = lane < K ? feat[(size_t)row * K + lane] : 0.f;
        if (lane < SK_MAXK) s_f[lr * SK_MAXK + lane] = f;
        const float mu = mean[row], rs = rstd[row];
        float4 xh[NV], d[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) xh[i] = bi[i];
        for (int k = 0; k < K; ++k) {
          const float fk = __shfl(f, k, 64);
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            const float4 w = *reinterpret_cast<const float4*>(s_wt + k * H + (i * 64 + lane) * 4);
            xh[i].x = fmaf(fk, w.x, xh[i].x); xh[i].y = fmaf(fk, w.y, xh[i].y);
            xh[i].z = fmaf(fk, w.z, xh[i].z); xh[i].w = fmaf(fk, w.w, xh[i].w);
          }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int col = (i * 64 + lane) * 4;
          xh[i] = make_float4((xh[i].x - mu) * rs, (xh[i].y - mu) * rs, (xh[i].z - mu) * rs, (xh[i].w - mu) * rs);
          d[i] = ld4<T>(dy + (size_t)row * H + col);
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
          float4 o;
          o.x = rs * (d[i].x - s1 - xh[i].x * s2);
          o.y = rs * (d[i].y - s1 - xh[i].y * s2);
          o.z = rs * (d[i].z - s1 - xh[i].z * s2);
          o.w = rs * (d[i].w - s1 - xh[i].w * s2);
          ax[i].x += o.x; ax[i].y += o.y; ax[i].z += o.z; ax[i].w += o.w;
          *reinterpret_cast<float4*>(s_dz + lr * H + (i * 64 + lane) * 4) = o;
        }
      } else {
        if (lane < SK_MAXK) s_f[lr * SK_MAXK + lane] = 0.f;      // rows past the end contribute nothing
#pragma unroll
        for (int i = 0; i < NV; ++i)
          *reinterpret_cast<float4*>(s_dz + lr * H + (i * 64 + lane) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    __syncthreads();
    // ---- phase B: dW[c][k] += dz[r][c] * feat[r][k]
#pragma unroll
    for (int r = 0; r < SK_R; ++r) {
      float dzv[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) dzv[j] = s_dz[r * H + tid + 256 * j];
#pragma unroll
      for (int k4 = 0; k4 < SK_MAXK / 4; ++k4) {
        if (k4 * 4 < K) {
          const float4 f4 = *reinterpret_cast<const float4*>(s_f + r * SK_MAXK + k4 * 4);
#pragma unroll
          for (int j = 0; j < NV; ++j) {
            acc[j][k4 * 4 + 0] = fmaf(dzv[j], f4.x, acc[j][k4 * 4 + 0]);
            acc[j][k4 * 4 + 1] = fmaf(dzv[j], f4.y, acc[j][k4 * 4 + 1]);
            acc[j][k4 * 4 + 2] = fmaf(dzv[j], f4.z, acc[j][k4 * 4 + 2]);
            acc[j][k4 * 4 + 3] = fmaf(dzv[j], f4.w, acc[j][k4 * 4 + 3]);
          }
        }
      }
    }
    __syncthreads();
  }
  float* out = partials + (size_t)blockIdx.x * (K + 3) * H;
#pragma unroll
  for (int k = 0; k < SK_MAXK; ++k)
    if (k < K) {
#pragma unroll
      for (int j = 0; j < NV; ++j) out[(size_t)k * H + tid + 256 * j] = acc[j][k];
    }
  // column sums of the LayerNorm part: fold the four waves through LDS, one plane at a time
#pragma unroll
  for (int which = 0; which < 3; ++which) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float4 v = which == 0 ? ag[i] : (which == 1 ? ab[i] : ax[i]);
      *reinterpret_cast<float4*>(s_dz + wave * H + (i * 64 + lane) * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = tid + 256 * j;
      out[(size_t)(K + which) * H + c] = (s_dz[c] + s_dz[H + c]) + (s_dz[2 * H + c] + s_dz[3 * H + c]);
    }
    __syncthreads();
  }
}

// grid (H / 64, K + 3), 256 threads: plane p of 64 columns; wave w sums the workgroups b = w, w + 4, ... in ascending
// order, the four partial sums are folded pairwise -- a fixed order -- and added to the destination
__global__ __launch_bounds__(256) void smallk_finalize_kernel(const float* __restrict__ partials, int nblocks, int K, int H,
                                                              float* __restrict__ dW, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ dbias) {
  __shared__ float s_part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, p = blockIdx.y;
  float s0 = 0.f, s1 = 0.f;
  int b = wave;
  for (; b + 4 < nblocks; b += 8) {
    s0 += partials[((size_t)b * (K + 3) + p) * H + c];
    s1 += partials[((size_t)(b + 4) * (K + 3) + p) * H + c];
  }
  for (; b < nblocks; b += 4) s0 += partials[((size_t)b * (K + 3) + p) * H + c];
  s_part[wave][lane] = s0 + s1;
  __syncthreads();
  if (wave != 0) return;
  const float s = (s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]);
  if (p < K) {
    if (dW != nullptr) dW[(size_t)c * K + p] += s;
  } else {
    float* dst = p == K ? dgamma : (p == K + 1 ? dbeta : dbias);
    if (dst != nullptr) dst[c] += s;
  }
}

static int sk_bwd_blocks(int rows) {
  const int tiles = (rows + SK_R - 1) / SK_R;
  return tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256;
}

BEVBERT_API int64_t bevbert_smallk_workspace_floats(int rows, int K, int H) {
  return (int64_t)sk_bwd_blocks(rows) * (K + 3) * H;
}

BEVBERT_API int bevbert_smallk_linear_layernorm_fwd(const float* feat, const float* weight, const float* bias,
                                                    const float* gamma, const float* beta, const void* post1,
                                                    const void* table, const int64_t* idx, void* y, float* mean,
                                                    float* rstd, int rows, int K, int H, float eps, int dtype,
                                                    hipStream_t stream) {
  BB_REQUIRE(K >= 1 && K <= SK_MAXK, "smallk_linear_layernorm_fwd: K=%d outside 1..%d", K, SK_MAXK);
  BB_REQUIRE(H == 256 || H == 512 || H == 768 || H == 1024, "smallk_linear_layernorm_fwd: H=%d unsupported", H);
  BB_REQUIRE((table == nullptr) == (idx == nullptr), "smallk_linear_layernorm_fwd: table and idx come together");
  if (rows <= 0) return BB_OK;
  int nb = (rows + 3) / 4;
  if (nb > 1024) nb = 1024;
  const size_t lds = (size_t)K * H * sizeof(float);
  BB_REQUIRE(lds <= 64 * 1024, "smallk_linear_layernorm_fwd: K=%d x H=%d needs more than 64 KB of LDS", K, H);
#define GO(TT, N)                                                                                                     \
  hipLaunchKernelGGL((smallk_ln_fwd_kernel<TT, N>), dim3(nb), dim3(256), lds, stream, feat, weight, bias, gamma, beta,  \
                     (const TT*)post1, (const TT*)table, idx, (TT*)y, mean, rstd, rows, K, eps)
  if (dtype == BB_F32) {
    switch (H / 256) { case 1: GO(float, 1); break; case 2: GO(float, 2); break; case 3: GO(float, 3); break; default: GO(float, 4); }
  } else if (dtype == BB_BF16) {
    switch (H / 256) { case 1: GO(bf16_raw, 1); break; case 2: GO(bf16_raw, 2); break; case 3: GO(bf16_raw, 3); break; default: GO(bf16_raw, 4); }
  } else {
    bb_set_error("smallk_linear_layernorm_fwd: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
#undef GO
  BB_CHECK_LAUNCH("smallk_linear_layernorm_fwd");
  return BB_OK;
}

BEVBERT_API int bevbert_smallk_linear_layernorm_bwd(const void* dy, const float* feat, const float* weight,
                                                    const float* bias, const float* mean, const float* rstd,
                                                    const float* gamma, float* dweight, float* dbias, float* dgamma,
                                                    float* dbeta, float* workspace, int rows, int K, int H, int dtype,
                                                    hipStream_t stream) {
  BB_REQUIRE(K >= 1 && K <= SK_MAXK, "smallk_linear_layernorm_bwd: K=%d outside 1..%d", K, SK_MAXK);
  BB_REQUIRE(H == 256 || H == 512 || H == 768 || H == 1024, "smallk_linear_layernorm_bwd: H=%d unsupported", H);
  BB_REQUIRE(workspace != nullptr && mean != nullptr && rstd != nullptr, "smallk_linear_layernorm_bwd: workspace / statistics missing");
  if (rows <= 0) return BB_OK;
  const int nb = sk_bwd_blocks(rows);
  const size_t lds = ((size_t)K * H + (size_t)SK_R * H + (size_t)SK_R * SK_MAXK) * sizeof(float);
  BB_REQUIRE(lds <= 64 * 1024, "smallk_linear_layernorm_bwd: K=%d x H=%d needs more than 64 KB of LDS", K, H);
#define GO(TT, N)                                                                                                     \
  hipLaunchKernelGGL((smallk_ln_bwd_kernel<TT, N>), dim3(nb), dim3(256), lds, stream, (const TT*)dy, feat, weight, bias, \
                     mean, rstd, gamma, workspace, rows, K)
  if (dtype == BB_F32) {
    switch (H / 256) { case 1: GO(float, 1); break; case 2: GO(float, 2); break; case 3: GO(float, 3); break; default: GO(float, 4); }
  } else if (dtype == BB_BF16) {
    switch (H / 256) { case 1: GO(bf16_raw, 1); break; case 2: GO(bf16_raw, 2); break; case 3: GO(bf16_raw, 3); break; default: GO(bf16_raw, 4); }
  } else {
    bb_set_error("smallk_linear_layernorm_bwd: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
#undef GO
  BB_CHECK_LAUNCH("smallk_linear_layernorm_bwd");
  hipLaunchKernelGGL(smallk_finalize_kernel, dim3(H / 64, K + 3), dim3(256), 0, stream, workspace, nb, K, H,
                     dweight, dgamma, dbeta, dbias);
  BB_CHECK_LAUNCH("smallk_linear_layernorm_bwd finalize");
  return BB_OK;
}
