// K2 in fp32 on the matrix cores: v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate -- an fmaf chain bit for bit, at the
// fp32 vector rate of 64 FLOP / clk / SIMD) for the parity mode of the library, the reference's DEFAULT precision
// (configs/r2r_pretrain.json: "fp16": false).  Until round 5 that mode ran attention on the wave-per-row vector kernels
// of attn_simple.hip: 28.7 ms per 441 x 441 backward launch at batch 64, 65 % of an fp32 training step
// (profiles/r05_bench_fp32_before.json).  Same math and interface (two-kernel backward behind attn_delta, the dropout
// mask from the element-indexed hash every exact consumer uses, additive key mask and per-element bias, per-head bias
// gradients), so attn_simple.hip stays as the on-GPU cross-check (BEVBERT_ATTN_F32=simple).
//
// One 16 x 16 x 4 instruction: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15] -- ONE float per lane each;
// D[row = 4 (lane >> 4) + r][col = lane & 15].  With c = lane & 15, g = lane >> 4:
//   * contractions over the head dimension (S, dP): the k-slot g of instruction kk stands for d = 16 g + kk, so a lane's
//     16 operands are 16 CONTIGUOUS floats of its row (four 16-byte LDS reads, or registers for the resident side);
//   * S^T = K Q^T (forward, dQ kernel: lane = query c, rows = keys 4 g + r) resp. S = Q K^T (dK / dV kernel: lane = key c,
//     rows = queries 4 g + r): the softmax results P / dS of instruction r sit exactly where the B operand of the second
//     contraction (over keys resp. queries, k-slot g <-> row 4 g + r) wants them -- they never leave their lanes.
#include "attn_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define F32_LD 68      // LDS row stride in floats (64 + 4): consecutive rows start 4 banks apart
#define F32_TR 64      // rows of a staged tile

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// [64 rows][64 floats] tile: global -> LDS, rows past `nrows` zero filled; thread owns 16-byte chunks t + 256 i
__device__ __forceinline__ void f32_stage(float* dst, const float* src, int64_t ld, int row0, int nrows, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i, row = id >> 4, c4 = (id & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + row < nrows) v = *reinterpret_cast<const float4*>(src + (size_t)(row0 + row) * ld + c4);
    *reinterpret_cast<float4*>(dst + row * F32_LD + c4) = v;
  }
}
// the 16 operands of lane (c, g) for the head-dimension contractions: row `row` of a tile, floats 16 g .. 16 g + 15
__device__ __forceinline__ void f32_row16(float (&f)[16], const float* tile, int row, int g) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(tile + row * F32_LD + 16 * g + 4 * j);
    f[4 * j] = v.x; f[4 * j + 1] = v.y; f[4 * j + 2] = v.z; f[4 * j + 3] = v.w;
  }
}
__device__ __forceinline__ void f32_row16_global(float (&f)[16], const float* base, int64_t ld, int row, int g) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(base + (size_t)row * ld + 16 * g + 4 * j);
    f[4 * j] = v.x; f[4 * j + 1] = v.y; f[4 * j + 2] = v.z; f[4 * j + 3] = v.w;
  }
}
__device__ __forceinline__ float g_max(float v) {       // over the four lanes c, c + 16, c + 32, c + 48
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float g_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// =============================================================================================
// forward: workgroup = 4 waves x 16 queries, K / V tiles of 64 keys through LDS, online softmax
// =============================================================================================
__global__ __launch_bounds__(256) void attn_f32_fwd_kernel(AttnArgs a) {
  if (a.drop_p > 0.f) a.drop_key = bb_salted(a.drop_key, a.salt);
  __shared__ __attribute__((aligned(16))) float s_k[F32_TR * F32_LD];
  __shared__ __attribute__((aligned(16))) float s_v[F32_TR * F32_LD];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15, w = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qi = blockIdx.x * 64 + w * 16 + c;                   // this lane's query
  const int qr = qi < a.Lq ? qi : a.Lq - 1;
  const float* qp = (const float*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const float* kp = (const float*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const float* vp = (const float*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  float qf[16];
  f32_row16_global(qf, qp, a.ldq, qr, g);
  f32x4 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;                          // l_run: this lane's share (its keys 4 g + r) of the row sum
  const float keep_scale = a.drop_p > 0.f ? a.keep_scale : 1.0f;
  for (int kv0 = 0; kv0 < a.Lk; kv0 += F32_TR) {
    __syncthreads();                                             // the previous tile's readers are done
    f32_stage(s_k, kp, a.ldk, kv0, a.Lk, tid);
    f32_stage(s_v, vp, a.ldv, kv0, a.Lk, tid);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
      if (kv0 + 16 * t >= a.Lk) break;
      float kf[16];
      f32_row16(kf, s_k, 16 * t + c, g);
      f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) sacc = mfma4(kf[kk], qf[kk], sacc);          // S^T[key 4 g + r][query c]
      float s[4], tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kv0 + 16 * t + 4 * g + r;
        float v = -INFINITY;
        if (key < a.Lk) {
          v = sacc[r] * a.scale;
          if (a.key_mask) v += a.key_mask[(size_t)b * a.Lk + key];
          if (a.bias) v += a.bias[((size_t)b * a.Lq + qr) * a.Lk + key];
        }
        s[r] = v;
        tmax = fmaxf(tmax, v);
      }
      tmax = g_max(tmax);
      const float m_new = fmaxf(m_run, tmax);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __expf(m_run - m_use);                 // first tile: exp(-inf) = 0 on zero accumulators
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[dt] *= alpha;
      float pd[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(s[r] - m_use);
        l_run += p;
        pd[r] = p;
        if (a.drop_p > 0.f) {
          const int key = kv0 + 16 * t + 4 * g + r;
          pd[r] = (key < a.Lk && bb_keep(a.drop_key, attn_elem(a, b, h, qr, key), a.drop_thr)) ? p * keep_scale : 0.f;
        }
      }
      // O^T[d = 16 dt + c'][query c] += V^T P^T: A = V[key 4 g + r][16 dt + c], B = P[key 4 g + r][query c]
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[dt] = mfma4(s_v[(16 * t + 4 * g + r) * F32_LD + 16 * dt + c], pd[r], oacc[dt]);
    }
  }
  const float l = g_sum(l_run);
  if (qi < a.Lq) {
    const float inv = 1.0f / l;
    float* op = (float*)a.o + (size_t)b * a.bso + (size_t)qi * a.ldo + h * ATTN_D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)       // lane holds O^T[16 dt + 4 g + r][query c]
      *reinterpret_cast<float4*>(op + 16 * dt + 4 * g) = make_float4(oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv);
    if (a.lse && g == 0) a.lse[((size_t)b * a.nh + h) * a.Lq + qi] = ((m_run == -INFINITY) ? 0.f : m_run) + __logf(l);
  }
}

// =============================================================================================
// dQ (and the per-head bias gradients): workgroup = 4 waves x 16 queries, K / V tiles through LDS
// =============================================================================================
__global__ __launch_bounds__(256) void attn_f32_dq_kernel(AttnArgs a) {
  if (a.drop_p > 0.f) a.drop_key = bb_salted(a.drop_key, a.salt);
  __shared__ __attribute__((aligned(16))) float s_k[F32_TR * F32_LD];
  __shared__ __attribute__((aligned(16))) float s_v[F32_TR * F32_LD];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15, w = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qi = blockIdx.x * 64 + w * 16 + c;
  const int qr = qi < a.Lq ? qi : a.Lq - 1;
  const float* qp = (const float*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const float* kp = (const float*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const float* vp = (const float*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const float* dop = (const float*)a.dout + (size_t)b * a.bso + h * ATTN_D;
  float qf[16], dof[16];
  f32_row16_global(qf, qp, a.ldq, qr, g);
  f32_row16_global(dof, dop, a.ldo, qr, g);
  const size_t ridx = ((size_t)b * a.nh + h) * a.Lq + qr;
  const float lse = a.lse[ridx], delta = a.delta[ridx];
  const float keep_scale = a.drop_p > 0.f ? a.keep_scale : 1.0f;
  f32x4 dqacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dqacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int kv0 = 0; kv0 < a.Lk; kv0 += F32_TR) {
    __syncthreads();
    f32_stage(s_k, kp, a.ldk, kv0, a.Lk, tid);
    f32_stage(s_v, vp, a.ldv, kv0, a.Lk, tid);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
      if (kv0 + 16 * t >= a.Lk) break;
      float kf[16], vf[16];
      f32_row16(kf, s_k, 16 * t + c, g);
      f32_row16(vf, s_v, 16 * t + c, g);
      f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f}, dpacc = sacc;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        sacc = mfma4(kf[kk], qf[kk], sacc);                      // S^T[key 4 g + r][query c]
        dpacc = mfma4(vf[kk], dof[kk], dpacc);                   // dP^T
      }
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kv0 + 16 * t + 4 * g + r;
        float v = 0.f;
        if (key < a.Lk) {
          float sc = sacc[r] * a.scale;
          if (a.key_mask) sc += a.key_mask[(size_t)b * a.Lk + key];
          if (a.bias) sc += a.bias[((size_t)b * a.Lq + qr) * a.Lk + key];
          const float p = __expf(sc - lse);
          float dp = dpacc[r];
          if (a.drop_p > 0.f) dp = bb_keep(a.drop_key, attn_elem(a, b, h, qr, key), a.drop_thr) ? dp * keep_scale : 0.f;
          v = p * (dp - delta);
          if (a.dbias && qi < a.Lq) a.dbias[(((size_t)b * a.nh + h) * a.Lq + qi) * a.Lk + key] = v;
        }
        ds[r] = v;
      }
      // dQ^T[d][query c] += K^T dS^T: A = K[key 4 g + r][16 dt + c], B = dS[key 4 g + r][query c]
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dqacc[dt] = mfma4(s_k[(16 * t + 4 * g + r) * F32_LD + 16 * dt + c], ds[r], dqacc[dt]);
    }
  }
  if (qi < a.Lq) {
    float* dqp = (float*)a.dq + (size_t)b * a.bsq + (size_t)qi * a.ldq + h * ATTN_D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<float4*>(dqp + 16 * dt + 4 * g) = make_float4(dqacc[dt][0] * a.scale, dqacc[dt][1] * a.scale,
                                                                      dqacc[dt][2] * a.scale, dqacc[dt][3] * a.scale);
  }
}

// =============================================================================================
// dK, dV: workgroup = 4 waves x 16 keys, Q / dO tiles (+ lse, delta) of 64 queries through LDS
// =============================================================================================
__global__ __launch_bounds__(256) void attn_f32_dkv_kernel(AttnArgs a) {
  if (a.drop_p > 0.f) a.drop_key = bb_salted(a.drop_key, a.salt);
  __shared__ __attribute__((aligned(16))) float s_q[F32_TR * F32_LD];
  __shared__ __attribute__((aligned(16))) float s_do[F32_TR * F32_LD];
  __shared__ float s_lse[F32_TR], s_dl[F32_TR];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15, w = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int key = blockIdx.x * 64 + w * 16 + c;                  // this lane's key
  const int kr = key < a.Lk ? key : a.Lk - 1;
  const float* qp = (const float*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const float* kp = (const float*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const float* vp = (const float*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const float* dop = (const float*)a.dout + (size_t)b * a.bso + h * ATTN_D;
  float kf[16], vf[16];
  f32_row16_global(kf, kp, a.ldk, kr, g);
  f32_row16_global(vf, vp, a.ldv, kr, g);
  const float km = a.key_mask ? a.key_mask[(size_t)b * a.Lk + kr] : 0.f;
  const float keep_scale = a.drop_p > 0.f ? a.keep_scale : 1.0f;
  f32x4 dkacc[4], dvacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dkacc[dt] = dvacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int q0 = 0; q0 < a.Lq; q0 += F32_TR) {
    __syncthreads();
    f32_stage(s_q, qp, a.ldq, q0, a.Lq, tid);
    f32_stage(s_do, dop, a.ldo, q0, a.Lq, tid);
    if (tid < F32_TR) {
      const bool ok = q0 + tid < a.Lq;
      const size_t ridx = ((size_t)b * a.nh + h) * a.Lq + (ok ? q0 + tid : 0);
      s_lse[tid] = ok ? a.lse[ridx] : INFINITY;                  // rows past the end: p = exp(-inf) = 0
      s_dl[tid] = ok ? a.delta[ridx] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
      if (q0 + 16 * t >= a.Lq) break;
      float qa[16], da[16];
      f32_row16(qa, s_q, 16 * t + c, g);
      f32_row16(da, s_do, 16 * t + c, g);
      f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f}, dpacc = sacc;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        sacc = mfma4(qa[kk], kf[kk], sacc);                      // S[query 4 g + r][key c]
        dpacc = mfma4(da[kk], vf[kk], dpacc);                    // dP
      }
      float ds[4], pd[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 16 * t + 4 * g + r, qi = q0 + ql;
        const int qr = qi < a.Lq ? qi : a.Lq - 1;
        float sc = sacc[r] * a.scale + km;
        if (a.bias) sc += a.bias[((size_t)b * a.Lq + qr) * a.Lk + kr];
        const float p = __expf(sc - s_lse[ql]);
        float dp = dpacc[r], pk = p;
        if (a.drop_p > 0.f) {
          const bool keep = bb_keep(a.drop_key, attn_elem(a, b, h, qr, kr), a.drop_thr);
          dp = keep ? dp * keep_scale : 0.f;
          pk = keep ? p * keep_scale : 0.f;
        }
        ds[r] = p * (dp - s_dl[ql]);
        pd[r] = pk;
      }
      // dV^T[d][key c] += dO^T P, dK^T[d][key c] += Q^T dS: A = dO / Q [query 4 g + r][16 dt + c], B = P / dS [query 4 g + r][key c]
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dvacc[dt] = mfma4(s_do[(16 * t + 4 * g + r) * F32_LD + 16 * dt + c], pd[r], dvacc[dt]);
          dkacc[dt] = mfma4(s_q[(16 * t + 4 * g + r) * F32_LD + 16 * dt + c], ds[r], dkacc[dt]);
        }
    }
  }
  if (key < a.Lk) {
    float* dkp = (float*)a.dk + (size_t)b * a.bsk + (size_t)key * a.ldk + h * ATTN_D;
    float* dvp = (float*)a.dv + (size_t)b * a.bsv + (size_t)key * a.ldv + h * ATTN_D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      *reinterpret_cast<float4*>(dkp + 16 * dt + 4 * g) = make_float4(dkacc[dt][0] * a.scale, dkacc[dt][1] * a.scale,
                                                                      dkacc[dt][2] * a.scale, dkacc[dt][3] * a.scale);
      *reinterpret_cast<float4*>(dvp + 16 * dt + 4 * g) = make_float4(dvacc[dt][0], dvacc[dt][1], dvacc[dt][2], dvacc[dt][3]);
    }
  }
}

// =============================================================================================
// launchers
// =============================================================================================
static bool f32_layout_ok(const AttnArgs& a, bool bwd) {
  bool ok = a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldo % 4 == 0 && a.bsq % 4 == 0 && a.bsk % 4 == 0 &&
            a.bsv % 4 == 0 && a.bso % 4 == 0 && ((uintptr_t)a.q % 16) == 0 && ((uintptr_t)a.k % 16) == 0 &&
            ((uintptr_t)a.v % 16) == 0 && ((uintptr_t)a.o % 16) == 0;
  if (bwd) ok = ok && ((uintptr_t)a.dout % 16) == 0 && ((uintptr_t)a.dq % 16) == 0 && ((uintptr_t)a.dk % 16) == 0 && ((uintptr_t)a.dv % 16) == 0;
  return ok;
}
// fp32 tensors with 16-byte aligned rows; anything else stays on the wave-per-row kernels
bool attn_f32_supported(const AttnArgs& a, int dtype, bool bwd) {
  static const bool off = [] { const char* v = getenv("BEVBERT_ATTN_F32"); return v && v[0] == 's'; }();
  return !off && dtype == BB_F32 && f32_layout_ok(a, bwd);
}

int attn_f32_fwd(const AttnArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(attn_f32_fwd_kernel, dim3((a.Lq + 63) / 64, a.nh, a.B), dim3(256), 0, st, a);
  BB_CHECK_LAUNCH("attn_fwd(fp32 mfma)");
  return BB_OK;
}

int attn_f32_bwd(const AttnArgs& a, hipStream_t st) {       // a.delta filled by attn_delta beforehand
  hipLaunchKernelGGL(attn_f32_dq_kernel, dim3((a.Lq + 63) / 64, a.nh, a.B), dim3(256), 0, st, a);
  hipLaunchKernelGGL(attn_f32_dkv_kernel, dim3((a.Lk + 63) / 64, a.nh, a.B), dim3(256), 0, st, a);
  BB_CHECK_LAUNCH("attn_bwd(fp32 mfma)");
  return BB_OK;
}
