// K2 forward, second generation (bf16 in / fp32 accumulate): the same transposed MFMA formulation as attn_mfma.hip
//   S^T = K Q^T  (lane q = l & 15 holds S[q][key = 16 t + 4 g + r]),   O^T = V^T P^T  (P goes D -> B operand in place)
// on a VALU diet.  The round-2 kernel issued 23 VALU instructions per matrix instruction with dropout (13 without):
// it was bound by the vector pipe, not by MFMA, LDS or HBM (profiles/r02f_pmc_attn_sq_counters.txt).  Here:
//   * additive key mask (and the -inf of keys beyond Lk) enters as the C operand of the first Q K^T instruction of each
//     key tile, in raw-score units (mask / scale): zero vector instructions per score element;
//   * lazy running maximum: a tile only triggers the rescale of O and l when some query's maximum grows by more than
//     FWD2_THR (log2 domain); otherwise P = exp2(s * scale*log2e - m_stale) <= 2^FWD2_THR, which bf16 (relative rounding)
//     and the fp32 accumulators take without loss.  The quad reductions and the 16 multiplies per query tile leave the
//     common path; the decision is wave-uniform (__any) and is taken before the tile's P exists (textbook order);
//   * dropout: the keep bits of the whole call are produced beforehand by attn_drop_bits_kernel (one hash per element
//     pair, the same counter-based stream as every other dropout site) as 64-bit lane masks in exactly the layout of the
//     S accumulators.  The forward fetches them with scalar loads (constant address space -> s_load_dwordx16) and drops
//     with ONE v_cndmask_b32 per element (SGPR-pair condition); 1 / (1 - p) is folded into the final normalisation;
//   * workgroups of NW waves x 32 queries: NW = 7 covers the 441 BEV cells in two workgroups of 224 queries with 1.6 %
//     padding (the 128-query blocks of round 2 wasted 14 % of the grid); 32-key halves keep the kernel at <= 128 VGPRs so
//     that two 7-wave workgroups share a CU (3.5 waves per SIMD).
// Per score element: max3 (1/2), fma, exp2, add, cndmask (dropout only), cvt_pk (1/2)  =  4 - 5 vector instructions per
// matrix instruction.
#include "attn_mfma_common.h"

#define FWD2_THR 5.0f

typedef const __attribute__((address_space(4))) uint64_t* bb_cu64p;   // uniform loads through the scalar cache

// keep ? p : 0 with the keep decisions of the 64 lanes in a scalar register pair: one v_cndmask_b32 (the compiler sees
// the instruction, so the transcendental-result hazard behind v_exp_f32 is padded by it, unlike inside inline asm)
__device__ __forceinline__ float drop_select(float p, uint64_t lane_mask) {
  return __builtin_amdgcn_inverse_ballot_w64(lane_mask) ? p : 0.f;
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// K / V tiles of this kernel: [64 rows][64 bf16] at a row stride of 72 elements (144 B).  Measured on the MI355X at
// B = 64, 441 x 441 (gpurun_out r03d / r03k / r03l): 144-byte rows with four 4-wave workgroups per CU 87.7 / 76.4 us
// (dropout 0.1 / none); 160-byte rows (conflict-free in scripts/lds_bank_sim.py, but 41.5 KB per workgroup -> three per
// CU) 116 / 100 us; 128-byte rows with an XOR swizzle of the 16-byte chunks (conflict-free, 33 KB, but the per-lane
// swizzled offsets cost the registers that the 128-VGPR budget of four waves per SIMD does not have) 95 / 82 us.
// Occupancy beats bank conflicts here: LDS is busy a quarter of the time, the waves wait on latency.
#define F2_LD 72
#define F2_TILE (TK * F2_LD)
__device__ __forceinline__ int f2_off(int row, int chunk) { return row * F2_LD + (chunk << 3); }
__device__ __forceinline__ bf16x8 f2_frag_rows(const bf16_raw* tile, int t, int ks, int lane) {
  return as_bf16x8(*reinterpret_cast<const uint4*>(tile + f2_off(t * 16 + (lane & 15), ks * 4 + (lane >> 4))));
}
// fragment whose eight k-slots of lane group g are image rows row_lo + (0..3) and row_hi + (0..3), at columns col0 .. +15
__device__ __forceinline__ bf16x8 f2_frag_tr(const bf16_raw* img, int row_lo, int row_hi, int col0, int lane) {
  const int i = lane & 15, col = col0 + 4 * (i & 3);
  const int rl = row_lo + (i >> 2), rh = row_hi + (i >> 2);
  const uint2 lo = lds_tr16(img + f2_off(rl, col >> 3) + (col & 7));
  const uint2 hi = lds_tr16(img + f2_off(rh, col >> 3) + (col & 7));
  return as_bf16x8(make_uint4(lo.x, lo.y, hi.x, hi.y));
}

template <int NW, bool DROP>
__global__ __launch_bounds__(64 * NW, 4) void attn_fwd2_kernel(AttnArgs a) {
  constexpr int QT = 2;                      // 16-query tiles per wave
  constexpr int NT = 64 * NW;
  constexpr int CPT = (512 + NT - 1) / NT;   // 16-byte chunks of a [64][64] bf16 tile per thread
  __shared__ __attribute__((aligned(16))) bf16_raw s_k[2][F2_TILE];
  __shared__ __attribute__((aligned(16))) bf16_raw s_v[2][F2_TILE];
  __shared__ __attribute__((aligned(16))) float s_mask[2][TK];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int blk, h, b;
  attn_decode_block(a, blk, h, b);
  const int qbase = blk * (16 * QT * NW) + w * (16 * QT);
  const bf16_raw* qp = (const bf16_raw*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const bf16_raw* kp = (const bf16_raw*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const bf16_raw* vp = (const bf16_raw*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const float sc2 = a.scale * LOG2E;
  const float inv_scale = 1.0f / a.scale;

  bf16x8 qf[QT][2];
  int qrow[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qrow[qt] = qbase + qt * 16 + c;
    const int r = qrow[qt] < a.Lq ? qrow[qt] : a.Lq - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = as_bf16x8(ld_frag_global(qp, a.ldq, r, ks * 32 + g * 8));
  }
  // keep-bit words of this wave's query tiles: 16 words (t, r) per (query tile, 64-key tile), 64-key tiles contiguous
  bb_cu64p wq[QT];
  if (DROP) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      int q16 = (qbase >> 4) + qt;
      q16 = q16 < a.nq16 ? q16 : a.nq16 - 1;       // query tiles past the end: any valid words will do
      wq[qt] = (bb_cu64p)(uintptr_t)(a.drop_bits + ((size_t)(b * a.nh + h) * a.nq16 + q16) * a.nk64 * 16);
    }
  }
  f32x4 oacc[QT][4];
  float m_run[QT], nm[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = -INFINITY;     // running maximum (log2 domain) shared by the four lanes of a query
    nm[qt] = 0.f;              // -(m_run), 0 while m_run is still -inf
    l_run[qt] = 0.f;           // this lane's share of the row sum
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // additive key mask of a tile in raw-score units (the C operand of Q K^T), one key per thread; -inf beyond Lk
  auto mask_of = [&](int kv0) -> float {
    const int key = kv0 + tid;
    if (tid >= TK || key >= a.Lk) return -INFINITY;
    return a.key_mask ? a.key_mask[(size_t)b * a.Lk + key] * inv_scale : 0.f;
  };
  uint4 kreg[CPT], vreg[CPT];
  float mreg;
  auto stage_load = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int ch = tid + i * NT, row = ch >> 3, d0 = (ch & 7) * 8;
      kreg[i] = vreg[i] = make_uint4(0, 0, 0, 0);                  // rows past the end are zero filled
      if (ch < 512 && kv0 + row < a.Lk) {
        kreg[i] = ld_frag_global(kp, a.ldk, kv0 + row, d0);
        vreg[i] = ld_frag_global(vp, a.ldv, kv0 + row, d0);
      }
    }
    mreg = mask_of(kv0);
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int ch = tid + i * NT, row = ch >> 3, d0 = (ch & 7) * 8;
      if (ch < 512) {
        *reinterpret_cast<uint4*>(s_k[buf] + f2_off(row, ch & 7)) = kreg[i];
        *reinterpret_cast<uint4*>(s_v[buf] + f2_off(row, ch & 7)) = vreg[i];
      }
    }
    if (tid < TK) s_mask[buf][tid] = mreg;
  };
  stage_load(0);
  stage_store(0);
  if (TK < a.Lk) stage_load(TK);
  __syncthreads();

  // keep-bit words travel one (half, query tile) unit ahead of their use: every word is read once, so each scalar load is
  // a cache miss with an L2 / HBM round trip to hide; 8 words per unit, 16 + 16 scalar registers
  uint64_t kw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) kw[i] = DROP ? wq[0][i] : 0;
  for (int kv0 = 0, cur = 0; kv0 < a.Lk; kv0 += TK, cur ^= 1) {
    const bf16_raw* ck = s_k[cur];
    const bf16_raw* cv = s_v[cur];
    const float* cmask = s_mask[cur];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {       // halves of 32 keys: key tiles t = 2 hh, 2 hh + 1
      // ---- S^T = K Q^T + mask
      f32x4 sacc[QT][2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * hh + tt;
        const float4 mk = *reinterpret_cast<const float4*>(&cmask[t * 16 + g * 4]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) sacc[qt][tt] = (f32x4){mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 kf = f2_frag_rows(ck, t, ks, lane);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) sacc[qt][tt] = mfma16(kf, qf[qt][ks], sacc[qt][tt]);
        }
      }
      // ---- first half: stage tile j+1 into the other buffer (its last readers passed the barrier that closed
      //      iteration j-1) and start the global loads of tile j+2; both overlap the arithmetic below
      if (hh == 0 && kv0 + TK < a.Lk) {
        stage_store(cur ^ 1);
        if (kv0 + 2 * TK < a.Lk) stage_load(kv0 + 2 * TK);
      }
      // ---- lazy running maximum (decision before this half's P exists)
      float lm[QT];
      bool grow = false;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const f32x4 s0 = sacc[qt][0], s1 = sacc[qt][1];
        lm[qt] = max3f(max3f(s0[0], s0[1], s0[2]), max3f(s0[3], s1[0], s1[1]), fmaxf(s1[2], s1[3])) * sc2;
        grow |= lm[qt] > m_run[qt] + FWD2_THR;      // m_run = -inf: any finite score triggers the first update
      }
      if (__any(grow)) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const float m_new = fmaxf(m_run[qt], quad_max(lm[qt]));
          const float nm_new = (m_new == -INFINITY) ? 0.f : -m_new;
          const float alpha = fast_exp2(m_run[qt] + nm_new);  // exp2(m_old - m_new); first update: exp2(-inf) = 0, O = l = 0
          m_run[qt] = m_new;
          nm[qt] = nm_new;
          l_run[qt] *= alpha;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] *= alpha;
        }
      }
      // ---- P = exp2(s * sc2 - m); row sums before dropout; keep bits straight from scalar registers
      bf16x8 pb[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        uint64_t kwn[8];
        if (DROP) {      // next unit: the other query tile of this half, or query tile 0 of the next half / key tile
          const int kvn = (qt + 1 < QT || hh == 0) ? kv0 : (kv0 + TK < a.Lk ? kv0 + TK : kv0);
          const int hn = qt + 1 < QT ? hh : hh ^ 1, qn = qt + 1 < QT ? qt + 1 : 0;
#pragma unroll
          for (int i = 0; i < 8; ++i) kwn[i] = wq[qn][(kvn >> 6) * 16 + hn * 8 + i];
          __builtin_amdgcn_sched_barrier(0);      // the loads stay HERE, ahead of this unit's arithmetic
        }
        float psum = 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = fast_exp2(fmaf(sacc[qt][tt][r], sc2, nm[qt]));
            psum += p;
            if (DROP) p = drop_select(p, kw[tt * 4 + r]);
            sacc[qt][tt][r] = p;
          }
        l_run[qt] += psum;
        pb[qt] = pack_pair(sacc[qt][0], sacc[qt][1]);
        if (DROP) {
#pragma unroll
          for (int i = 0; i < 8; ++i) kw[i] = kwn[i];
        }
      }
      // ---- O^T += V^T P^T over this half's 32 keys: k-slot (g, j) <-> key 32 hh + 16 (j >> 2) + 4 g + (j & 3)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 vf = f2_frag_tr(cv, 32 * hh + 4 * g, 32 * hh + 16 + 4 * g, dt * 16, lane);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) oacc[qt][dt] = mfma16(vf, pb[qt], oacc[qt][dt]);
      }

    }
    __syncthreads();  // one barrier per tile: buffer `cur` is free again, buffer `cur^1` is complete
  }

  // ---- epilogue: normalise (dropout scaling folded in), store O[q][h*64 + dt*16 + g*4 .. +3] and the log-sum-exp
  const float ks = DROP ? a.keep_scale : 1.0f;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l = quad_sum(l_run[qt]);
    const float inv = ks / l;
    if (qrow[qt] < a.Lq) {
      bf16_raw* op = (bf16_raw*)a.o + (size_t)b * a.bso + (size_t)qrow[qt] * a.ldo + h * ATTN_D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        st4<bf16_raw>(op + dt * 16 + g * 4, make_float4(oacc[qt][dt][0] * inv, oacc[qt][dt][1] * inv,
                                                         oacc[qt][dt][2] * inv, oacc[qt][dt][3] * inv));
      if (a.lse && g == 0) a.lse[((size_t)b * a.nh + h) * a.Lq + qrow[qt]] = (log2f(l) - nm[qt]) * LN2;
    }
  }
}

// Round 5: the same kernel on a further diet (see the comments inside): fused multiply-add before the maximum, select
// before the packed conversion, keep-bit words of a whole tile requested in front of the tile barrier.
// ABL (diagnostics, BEVBERT_FWD_ABL; results WRONG): 1 = no softmax arithmetic, 2 = no P V products, 4 = no Q K^T products,
// 8 = no K / V staging after tile 1, 16 = no running-maximum check, 64 = one key tile only
template <int NW, bool DROP, int ABL = 0>
__global__ __launch_bounds__(64 * NW, 4) void attn_fwd3_kernel(AttnArgs a) {
  constexpr int QT = 2;                      // 16-query tiles per wave
  constexpr int NT = 64 * NW;
  constexpr int CPT = (512 + NT - 1) / NT;   // 16-byte chunks of a [64][64] bf16 tile per thread
  __shared__ __attribute__((aligned(16))) bf16_raw s_k[2][F2_TILE];
  __shared__ __attribute__((aligned(16))) bf16_raw s_v[2][F2_TILE];
  __shared__ __attribute__((aligned(16))) float s_mask[2][TK];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int blk, h, b;
  attn_decode_block(a, blk, h, b);
  const int qbase = blk * (16 * QT * NW) + w * (16 * QT);
  const bf16_raw* qp = (const bf16_raw*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const bf16_raw* kp = (const bf16_raw*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const bf16_raw* vp = (const bf16_raw*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const float sc2 = a.scale * LOG2E;
  const float inv_scale = 1.0f / a.scale;

  bf16x8 qf[QT][2];
  int qrow[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qrow[qt] = qbase + qt * 16 + c;
    const int r = qrow[qt] < a.Lq ? qrow[qt] : a.Lq - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = as_bf16x8(ld_frag_global(qp, a.ldq, r, ks * 32 + g * 8));
  }
  // keep-bit words of this wave's query tiles: 16 words (t, r) per (query tile, 64-key tile), 64-key tiles contiguous
  bb_cu64p wq[QT];
  if (DROP) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      int q16 = (qbase >> 4) + qt;
      q16 = q16 < a.nq16 ? q16 : a.nq16 - 1;       // query tiles past the end: any valid words will do
      wq[qt] = (bb_cu64p)(uintptr_t)(a.drop_bits + ((size_t)(b * a.nh + h) * a.nq16 + q16) * a.nk64 * 16);
    }
  }
  f32x4 oacc[QT][4];
  float m_run[QT], nm[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = -INFINITY;     // running maximum (log2 domain) shared by the four lanes of a query
    nm[qt] = 0.f;              // -(m_run), 0 while m_run is still -inf
    l_run[qt] = 0.f;           // this lane's share of the row sum
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // additive key mask of a tile in raw-score units (the C operand of Q K^T), one key per thread; -inf beyond Lk
  auto mask_of = [&](int kv0) -> float {
    const int key = kv0 + tid;
    if (tid >= TK || key >= a.Lk) return -INFINITY;
    return a.key_mask ? a.key_mask[(size_t)b * a.Lk + key] * inv_scale : 0.f;
  };
  uint4 kreg[CPT], vreg[CPT];
  float mreg;
  auto stage_load = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int ch = tid + i * NT, row = ch >> 3, d0 = (ch & 7) * 8;
      kreg[i] = vreg[i] = make_uint4(0, 0, 0, 0);                  // rows past the end are zero filled
      if (ch < 512 && kv0 + row < a.Lk) {
        kreg[i] = ld_frag_global(kp, a.ldk, kv0 + row, d0);
        vreg[i] = ld_frag_global(vp, a.ldv, kv0 + row, d0);
      }
    }
    mreg = mask_of(kv0);
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int ch = tid + i * NT, row = ch >> 3, d0 = (ch & 7) * 8;
      if (ch < 512) {
        *reinterpret_cast<uint4*>(s_k[buf] + f2_off(row, ch & 7)) = kreg[i];
        *reinterpret_cast<uint4*>(s_v[buf] + f2_off(row, ch & 7)) = vreg[i];
      }
    }
    if (tid < TK) s_mask[buf][tid] = mreg;
  };
  stage_load(0);
  stage_store(0);
  if (TK < a.Lk) stage_load(TK);
  __syncthreads();

  // keep-bit words of a whole 64-key tile (2 query tiles x 16 words (t, r)) in 64 scalar registers, requested right in
  // front of the barrier that closes the previous tile: scalar loads share lgkmcnt with LDS and return out of order, so
  // any LDS wait behind an outstanding scalar load waits for it too (every word is read once: an L2 / HBM round trip)
  uint64_t kw[QT][16];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int i = 0; i < 16; ++i) kw[qt][i] = DROP ? wq[qt][i] : 0;
  const int Lk_run = (ABL & 64) ? (a.Lk < TK ? a.Lk : TK) : a.Lk;
  for (int kv0 = 0, cur = 0; kv0 < Lk_run; kv0 += TK, cur ^= 1) {
    const bf16_raw* ck = s_k[cur];
    const bf16_raw* cv = s_v[cur];
    const float* cmask = s_mask[cur];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {       // halves of 32 keys: key tiles t = 2 hh, 2 hh + 1
      // ---- S^T = K Q^T + mask
      f32x4 sacc[QT][2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * hh + tt;
        const float4 mk = *reinterpret_cast<const float4*>(&cmask[t * 16 + g * 4]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) sacc[qt][tt] = (f32x4){mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 kf = f2_frag_rows(ck, t, ks, lane);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) if (!(ABL & 4)) sacc[qt][tt] = mfma16(kf, qf[qt][ks], sacc[qt][tt]);
        }
      }
      // ---- first half: stage tile j+1 into the other buffer (its last readers passed the barrier that closed
      //      iteration j-1) and start the global loads of tile j+2; both overlap the arithmetic below
      if (hh == 0 && kv0 + TK < a.Lk && !((ABL & 8) && kv0 > 0)) {
        stage_store(cur ^ 1);
        if (kv0 + 2 * TK < a.Lk) stage_load(kv0 + 2 * TK);
      }
      // ---- t = s * scale * log2e - m_stale for every element first (the fused multiply-add the exponential needs anyway;
      //      its results are canonical numbers, so the maximum below needs no canonicalising v_max per matrix result)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[qt][tt][r] = fmaf(sacc[qt][tt][r], sc2, nm[qt]);
      // ---- lazy running maximum (decision before this half's P exists): a rescale only when some t exceeds FWD2_THR
      float lm[QT];
      bool grow = false;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const f32x4 s0 = sacc[qt][0], s1 = sacc[qt][1];
        lm[qt] = max3f(max3f(s0[0], s0[1], s0[2]), max3f(s0[3], s1[0], s1[1]), fmaxf(s1[2], s1[3]));
        grow |= (lm[qt] > FWD2_THR) | ((m_run[qt] == -INFINITY) & (lm[qt] > -INFINITY));   // no maximum yet: any finite score sets it
      }
      if ((ABL & 16) ? (kv0 == 0 && hh == 0) : __any(grow)) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const float m_new = fmaxf(m_run[qt], quad_max(lm[qt]) - nm[qt]);      // lm is relative to the stale maximum
          const float nm_new = (m_new == -INFINITY) ? 0.f : -m_new;
          const float alpha = fast_exp2(m_run[qt] + nm_new);  // exp2(m_old - m_new); first update: exp2(-inf) = 0, O = l = 0
          const float shift = nm_new - nm[qt];
          m_run[qt] = m_new;
          nm[qt] = nm_new;
          l_run[qt] *= alpha;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] *= alpha;
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[qt][tt][r] += shift;
        }
      }
      // ---- P = exp2(t); row sums before dropout; keep bits straight from scalar registers; the empty asm pins the
      //      select in front of the bf16 conversion (the compiler otherwise converts each element alone, selects on the
      //      16-bit halves and merges them with v_perm: 2.5 instead of 1.5 vector instructions per element)
      bf16x8 pb[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float psum = 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = (ABL & 1) ? sacc[qt][tt][r] : fast_exp2(sacc[qt][tt][r]);
            psum += p;
            if (DROP && !(ABL & 1)) {
              p = drop_select(p, kw[qt][(2 * hh + tt) * 4 + r]);
              asm("" : "+v"(p));
            }
            sacc[qt][tt][r] = p;
          }
        l_run[qt] += psum;
        pb[qt] = pack_pair(sacc[qt][0], sacc[qt][1]);
      }
      // ---- O^T += V^T P^T over this half's 32 keys: k-slot (g, j) <-> key 32 hh + 16 (j >> 2) + 4 g + (j & 3)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 vf = f2_frag_tr(cv, 32 * hh + 4 * g, 32 * hh + 16 + 4 * g, dt * 16, lane);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) if (!(ABL & 2)) oacc[qt][dt] = mfma16(vf, pb[qt], oacc[qt][dt]);
      }

    }
    // one barrier per tile: buffer `cur` is free again, buffer `cur^1` is complete.  This wave's LDS traffic has landed
    // (the pointer operand ties the scalar loads below to this point), the next tile's words are requested, then the
    // bare barrier: no fence, which would wait for the scalar loads -- and for the global loads of tile j + 2 -- first.
    __builtin_amdgcn_sched_barrier(0);
    if (DROP) {
      const int kvn = kv0 + TK < a.Lk ? kv0 + TK : kv0;
      bb_cu64p w0 = wq[0] + (kvn >> 6) * 16, w1 = wq[1] + (kvn >> 6) * 16;
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w0), "+s"(w1) : : "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) { kw[0][i] = w0[i]; kw[1][i] = w1[i]; }
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" : : : "memory");
  }

  // ---- epilogue: normalise (dropout scaling folded in), store O[q][h*64 + dt*16 + g*4 .. +3] and the log-sum-exp
  const float ks = DROP ? a.keep_scale : 1.0f;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l = quad_sum(l_run[qt]);
    const float inv = ks / l;
    if (qrow[qt] < a.Lq) {
      bf16_raw* op = (bf16_raw*)a.o + (size_t)b * a.bso + (size_t)qrow[qt] * a.ldo + h * ATTN_D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        st4<bf16_raw>(op + dt * 16 + g * 4, make_float4(oacc[qt][dt][0] * inv, oacc[qt][dt][1] * inv,
                                                         oacc[qt][dt][2] * inv, oacc[qt][dt][3] * inv));
      if (a.lse && g == 0) a.lse[((size_t)b * a.nh + h) * a.Lq + qrow[qt]] = (log2f(l) - nm[qt]) * LN2;
    }
  }
}

// =============================================================================================
// Keep-bit matrices of one attention call.  keep(q, k) = bits(hash(pair index ^ site key)) >= threshold with the
// element index ((b*nh + h)*Lq + q)*Lk2 + k -- the stream every other consumer (exact kernels, the test hook) uses.
//   F layout (forward, and the nibble reader of attn_bwd1): word (bh, q16, k64, t, r), bit l = keep(q = 16 q16 + (l & 15),
//            key = 64 k64 + 16 t + 4 (l >> 4) + r)                                   -- attn_common.h
//   B layout (single-pass backward, lanes <-> keys): word (bh, q32, k64, tt, t, r), bit l = keep(q = 32 q32 + 16 tt +
//            4 (l >> 4) + r, key = 64 k64 + 16 t + (l & 15)); nq32 = nq16 / 2 query blocks.
// One wave per (bh, q16, k64): 8 hashes per lane give the 16 F words as wave-wide compare masks; the B words are the
// same bits transposed inside the wave (4 ds_bpermute of the lane's 16-bit mask + 16 compares), not hashed again.
// =============================================================================================
// lanes LANE0 .. LANE0 + 3 of the (lo, hi) register pair <- four 64-bit wave masks.  v_writelane_b32 has no builtin in
// this compiler; the ONE leading s_nop covers the "VALU wrote the SGPR (v_cmp) -> v_writelane reads it" wait states of
// the most recent compare, which the compiler does not insert for instructions inside an asm statement.
template <int LANE0> __device__ __forceinline__ void put_words4(uint32_t& lo, uint32_t& hi, uint64_t m0, uint64_t m1,
                                                              uint64_t m2, uint64_t m3) {
  asm volatile("s_nop 3\n\t"
               "v_writelane_b32 %0, %2, %10\n\tv_writelane_b32 %1, %3, %10\n\t"
               "v_writelane_b32 %0, %4, %11\n\tv_writelane_b32 %1, %5, %11\n\t"
               "v_writelane_b32 %0, %6, %12\n\tv_writelane_b32 %1, %7, %12\n\t"
               "v_writelane_b32 %0, %8, %13\n\tv_writelane_b32 %1, %9, %13"
               : "+v"(lo), "+v"(hi)
               : "s"((uint32_t)m0), "s"((uint32_t)(m0 >> 32)), "s"((uint32_t)m1), "s"((uint32_t)(m1 >> 32)),
                 "s"((uint32_t)m2), "s"((uint32_t)(m2 >> 32)), "s"((uint32_t)m3), "s"((uint32_t)(m3 >> 32)),
                 "n"(LANE0), "n"(LANE0 + 1), "n"(LANE0 + 2), "n"(LANE0 + 3));
}

template <bool WITH_B>
__global__ __launch_bounds__(256) void attn_drop_bits_kernel(AttnArgs a, uint64_t* bits_f, uint64_t* bits_b, uint32_t* bits_l) {
  constexpr int NQT = WITH_B ? 4 : 1;            // 16-query tiles per task: the L words span the 64 queries of a fwd4 wave
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const uint32_t key = bb_salted(a.drop_key, a.salt);
  const uint32_t thr_hi = a.drop_thr << 16;      // (bits >> 16) >= thr  <=>  bits >= thr << 16;  low field: shift it up first
  const int nqg = a.nq16 / NQT, per_bh = nqg * a.nk64, total = a.B * a.nh * per_bh;
  for (int gtask = wave; gtask < total; gtask += nwaves) {
    const int bh = gtask / per_bh, rem = gtask - bh * per_bh, qg = rem / a.nk64, k64 = rem - qg * a.nk64;
    uint32_t wl0 = 0, wl1 = 0;                   // L layout: this lane's elements of the two 32-key halves
#pragma unroll 1
   for (int qt = 0; qt < NQT; ++qt) {
    const int q16 = qg * NQT + qt;
    const int task = (bh * a.nq16 + q16) * a.nk64 + k64;
    const int q = q16 * 16 + c;
    const uint32_t rbase = (uint32_t)(((uint32_t)bh * a.Lq + (q < a.Lq ? q : a.Lq - 1)) * (uint32_t)a.Lk2);
    uint32_t mine = 0;                   // bit (4 t + r) = keep of (q, key 64 k64 + 16 t + 4 g + r)       (WITH_B only)
    uint32_t flo = 0, fhi = 0;           // lane j < 16 ends up holding F word j = 4 t + r (v_writelane of the compare masks)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t pr = (rbase + (uint32_t)(k64 * 64 + t * 16 + g * 4)) >> 1;
      const uint32_t b0 = bb_pair_bits(key, pr), b1 = bb_pair_bits(key, pr + 1);
      const bool k0 = (b0 << 16) >= thr_hi, k1 = b0 >= thr_hi, k2 = (b1 << 16) >= thr_hi, k3 = b1 >= thr_hi;
      const uint64_t m0 = __ballot(k0), m1 = __ballot(k1), m2 = __ballot(k2), m3 = __ballot(k3);
      if (t == 0) put_words4<0>(flo, fhi, m0, m1, m2, m3);
      if (t == 1) put_words4<4>(flo, fhi, m0, m1, m2, m3);
      if (t == 2) put_words4<8>(flo, fhi, m0, m1, m2, m3);
      if (t == 3) put_words4<12>(flo, fhi, m0, m1, m2, m3);
      if (WITH_B) {
        const uint32_t nib = (k0 ? 1u : 0u) | (k1 ? 2u : 0u) | (k2 ? 4u : 0u) | (k3 ? 8u : 0u);
        mine |= nib << (4 * t);
      }
    }
    if (lane < 16) bits_f[(size_t)task * 16 + lane] = ((uint64_t)fhi << 32) | flo;
    if (WITH_B) {
      // destination lane (c' = c, g' = g) of word (t, r') wants keep(q = 16 q16 + 4 g' + r', key = 64 k64 + 16 t + c'):
      // source lane (c' >> 2) * 16 + 4 g' + r', source bit 4 t + (c' & 3)
      uint32_t blo = 0, bhi = 0;         // lane j < 16 ends up holding B word (t = j >> 2, r' = j & 3)
      // word index 4 t + r'; the four words of one r' sit 4 apart: assemble per t after all four r' have been fetched
      uint32_t got[4];
#pragma unroll
      for (int rp = 0; rp < 4; ++rp)
        got[rp] = (uint32_t)__builtin_amdgcn_ds_bpermute(((c >> 2) * 16 + 4 * g + rp) << 2, (int)mine) >> (c & 3);
#define BB_B_WORDS(T) put_words4<4 * T>(blo, bhi, __ballot((got[0] >> (4 * T)) & 1u), __ballot((got[1] >> (4 * T)) & 1u), \
                                        __ballot((got[2] >> (4 * T)) & 1u), __ballot((got[3] >> (4 * T)) & 1u));
      BB_B_WORDS(0) BB_B_WORDS(1) BB_B_WORDS(2) BB_B_WORDS(3)
#undef BB_B_WORDS
      // word (bh, q32 = q16 >> 1, k64, tt = q16 & 1, t, r'): the 16 words of one 16-query tile against the 64 keys of a
      // key wave are contiguous (two s_load_dwordx16 fetch a wave's 32 words of a step)
      if (lane < 16) {
        const int t = lane >> 2, rp = lane & 3;
        const size_t wi = ((((size_t)bh * (a.nq16 >> 1) + (q16 >> 1)) * a.nk64 + k64) * 2 + (q16 & 1)) * 16 + t * 4 + rp;
        bits_b[wi] = ((uint64_t)bhi << 32) | blo;
      }
      // even / odd elements of the 16 (t, r) bits compressed to 8 bits each: bits 0..3 = pairs of the first 32-key half, 4..7 = second
      uint32_t ev = mine & 0x5555u, od = (mine >> 1) & 0x5555u;
      ev = (ev | (ev >> 1)) & 0x3333u; od = (od | (od >> 1)) & 0x3333u;
      ev = (ev | (ev >> 2)) & 0x0f0fu; od = (od | (od >> 2)) & 0x0f0fu;
      ev = (ev | (ev >> 4)) & 0x00ffu; od = (od | (od >> 4)) & 0x00ffu;
      wl0 |= ((ev & 15u) | ((od & 15u) << 16)) << (4 * qt);
      wl1 |= ((ev >> 4) | ((od >> 4) << 16)) << (4 * qt);
    }
   }
    // L layout (attn_fwd4.hip): word (bh, q64, k64, half, lane); element e = 8 qt + 4 tt + r, i.e. keep(q = 64 q64 + 16 qt +
    // (lane & 15), key = 64 k64 + 32 half + 16 tt + 4 (lane >> 4) + r), at bit (e >> 1) + 16 (e & 1): the reader turns the
    // two halves into the AND mask of a packed bf16 pair with two packed 16-bit shifts
    if (WITH_B) {
      uint32_t* dst = bits_l + (((size_t)bh * nqg + qg) * a.nk64 + k64) * 128 + lane;
      dst[0] = wl0;
      dst[64] = wl1;
    }
  }
}

int attn_drop_bits(const AttnArgs& a, uint64_t* bits_f, uint64_t* bits_b, uint32_t* bits_l, hipStream_t st) {
  // the backward layout is only read by the 7+1-wave backward (attn_bwd3.hip: 256 < Lk <= 448, no bias), the per-lane
  // layout by the one-workgroup-per-head forward (attn_fwd4.hip, same key range): both are written for that key range
  const bool with_b = bits_b != nullptr && bits_l != nullptr && a.bias == nullptr && a.Lk > 256 && a.Lk <= 448;
  const int total = a.B * a.nh * (with_b ? a.nq16 / 4 : a.nq16) * a.nk64;
  int nb = (total + 3) / 4;
  if (nb > 8192) nb = 8192;
  if (with_b) hipLaunchKernelGGL(attn_drop_bits_kernel<true>, dim3(nb), dim3(256), 0, st, a, bits_f, bits_b, bits_l);
  else hipLaunchKernelGGL(attn_drop_bits_kernel<false>, dim3(nb), dim3(256), 0, st, a, bits_f, bits_b, bits_l);
  BB_CHECK_LAUNCH("attn_drop_bits");
  return BB_OK;
}

// =============================================================================================
// launcher
// =============================================================================================
template <int NW>
static int launch_fwd2(const AttnArgs& a_in, hipStream_t st) {
  AttnArgs a = a_in;
  a.nblk = (a.Lq + 32 * NW - 1) / (32 * NW);
  const dim3 grid((unsigned)a.nblk * a.nh * a.B);
  // BEVBERT_FWD_VAR=0: the round-3 kernel (A/B measurements, on-GPU cross-check)
  static const int var = [] { const char* v = getenv("BEVBERT_FWD_VAR"); return v ? atoi(v) : 1; }();
  if (var == 0) {
    if (a.drop_p > 0.f) hipLaunchKernelGGL((attn_fwd2_kernel<NW, true>), grid, dim3(64 * NW), 0, st, a);
    else hipLaunchKernelGGL((attn_fwd2_kernel<NW, false>), grid, dim3(64 * NW), 0, st, a);
  } else {
    static const int abl = [] { const char* v = getenv("BEVBERT_FWD_ABL"); return v ? atoi(v) : 0; }();
    if (abl && a.drop_p > 0.f && NW == 4) {
      switch (abl) {
        case 1: hipLaunchKernelGGL((attn_fwd3_kernel<4, true, 1>), grid, dim3(256), 0, st, a); return BB_OK;
        case 2: hipLaunchKernelGGL((attn_fwd3_kernel<4, true, 2>), grid, dim3(256), 0, st, a); return BB_OK;
        case 4: hipLaunchKernelGGL((attn_fwd3_kernel<4, true, 4>), grid, dim3(256), 0, st, a); return BB_OK;
        case 6: hipLaunchKernelGGL((attn_fwd3_kernel<4, true, 6>), grid, dim3(256), 0, st, a); return BB_OK;
        case 7: hipLaunchKernelGGL((attn_fwd3_kernel<4, true, 7>), grid, dim3(256), 0, st, a); return BB_OK;
        case 8: hipLaunchKernelGGL((attn_fwd3_kernel<4, true, 8>), grid, dim3(256), 0, st, a); return BB_OK;
        case 16: hipLaunchKernelGGL((attn_fwd3_kernel<4, true, 16>), grid, dim3(256), 0, st, a); return BB_OK;
        case 17: hipLaunchKernelGGL((attn_fwd3_kernel<4, true, 17>), grid, dim3(256), 0, st, a); return BB_OK;
        case 31: hipLaunchKernelGGL((attn_fwd3_kernel<4, true, 31>), grid, dim3(256), 0, st, a); return BB_OK;
        case 64: hipLaunchKernelGGL((attn_fwd3_kernel<4, true, 64>), grid, dim3(256), 0, st, a); return BB_OK;
        case 95: hipLaunchKernelGGL((attn_fwd3_kernel<4, true, 95>), grid, dim3(256), 0, st, a); return BB_OK;
        default: break;
      }
    }
    if (a.drop_p > 0.f) hipLaunchKernelGGL((attn_fwd3_kernel<NW, true>), grid, dim3(64 * NW), 0, st, a);
    else hipLaunchKernelGGL((attn_fwd3_kernel<NW, false>), grid, dim3(64 * NW), 0, st, a);
  }
  BB_CHECK_LAUNCH("attn_fwd(mfma, gen 2)");
  return BB_OK;
}

// The second-generation forward covers everything without a per-element additive bias (the graph bias of the global
// map encoder: a few dozen nodes, stays on attn_mfma_fwd_kernel) and, with dropout, needs the keep-bit matrix.
bool attn_fwd2_supported(const AttnArgs& a) { return a.bias == nullptr && (a.drop_p <= 0.f || a.drop_bits != nullptr); }

int attn_fwd2(const AttnArgs& a, hipStream_t st) {
  BB_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.bsq % 8 == 0 && a.bsk % 8 == 0 &&
                 a.bsv % 8 == 0 && a.bso % 8 == 0 && ((uintptr_t)a.q % 16) == 0 && ((uintptr_t)a.k % 16) == 0 &&
                 ((uintptr_t)a.v % 16) == 0 && ((uintptr_t)a.o % 16) == 0,
             "attention (MFMA path): pointers must be 16-byte aligned and strides multiples of 8 elements");
  // BEVBERT_FWD2_NW in {4, 7}: force a workgroup shape (A/B measurements)
  static const int force = [] { const char* v = getenv("BEVBERT_FWD2_NW"); return v ? atoi(v) : 0; }();
  // 4 waves x 32 queries: four workgroups share a CU (16 waves).  The 7-wave shape (two workgroups of 224 queries cover
  // the 441 BEV cells without the 14 % padding of 128-query blocks) measured slower: 97 vs 88 us (BEVBERT_FWD2_NW=7)
  if (force == 7) return launch_fwd2<7>(a, st);
  if (force == 8) return launch_fwd2<8>(a, st);
  return launch_fwd2<4>(a, st);
}
