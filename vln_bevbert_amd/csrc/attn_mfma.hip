// K2 (fast path): fused multi-head attention forward/backward on the CDNA4 matrix cores, bf16 in / fp32 accumulate.
//
// MFMA is used for exactly the contractions north_star names -- Q K^T and P V (plus their backward twins);
// softmax statistics, masking, dropout and the rescale are fp32 VALU work on the accumulator registers.
//
// Mapping (v_mfma_f32_16x16x32_bf16, 64-lane wave):
//   A operand: lane l holds row (l & 15), eight k-slots of group g = l >> 4          (8 bf16 = 4 VGPR)
//   B operand: lane l holds col (l & 15), the same eight k-slots of group g
//   C/D      : lane l holds col (l & 15), rows g*4 + r, r = 0..3                     (4 fp32)
// Everything is computed TRANSPOSED so that the per-query softmax state lives in the lane that owns the query:
//   S^T = K Q^T   (A = K tile rows from LDS, B = Q fragments kept in registers for the whole kernel)
//         -> lane (q = l&15) holds S[q][key = 16 t + 4 g + r]
//   O^T = V^T P^T (A = V^T tile rows from LDS, B = P packed to bf16 straight from the S accumulators)
// The contraction index of the second product is *defined* as  k-slot (g, j) <-> key 16 (2m + (j>>2)) + 4 g + (j&3)
// for both operands, which is exactly the order the S accumulators already have: P never moves between lanes and
// never touches LDS.  K and V tiles are staged row-major [key][d]; the V^T A fragments of the second product are two
// transposing 8-byte LDS reads (ds_read_b64_tr_b16) of the row-major V tile -- the forward stores no transposed image.
// Row stride 72 bf16 (144 B) keeps 16-byte reads of 16 consecutive rows on distinct banks.
#include "attn_mfma_common.h"

// =============================================================================================
// Forward.  Workgroup = 4 waves; wave w owns QT query tiles of 16 rows: rows q0 + (w*QT + qt)*16 + (l&15).
// =============================================================================================
template <int QT, bool BIAS, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_mfma_fwd_kernel(AttnArgs a) {
  // double-buffered tiles: while tile j is consumed from buffer j&1, tile j+1 is written to the other one
  __shared__ __attribute__((aligned(16))) bf16_raw s_k[2][TK * LDT];
  __shared__ __attribute__((aligned(16))) bf16_raw s_v[2][TK * LDT];      // row-major; the V^T operand comes out of it by transposing reads
  __shared__ __attribute__((aligned(16))) float s_mask[2][TK];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
  int blk, h, b;
  attn_decode_block(a, blk, h, b);
  if (DROP) a.drop_key = bb_salted(a.drop_key, a.salt);
  const int qbase = blk * (64 * QT) + w * (16 * QT);
  const bf16_raw* qp = (const bf16_raw*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const bf16_raw* kp = (const bf16_raw*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const bf16_raw* vp = (const bf16_raw*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const float sc2 = a.scale * LOG2E;
  const float keep_scale = DROP ? 1.0f / (1.0f - a.drop_p) : 1.0f;

  bf16x8 qf[QT][2];
  int qrow[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qrow[qt] = qbase + qt * 16 + c;
    const int r = qrow[qt] < a.Lq ? qrow[qt] : a.Lq - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = as_bf16x8(ld_frag_global(qp, a.ldq, r, ks * 32 + g * 8));
  }
  f32x4 oacc[QT][4];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = -INFINITY;
    l_run[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // additive key mask of a tile (log2 domain), one key per thread of the first wave; -inf beyond Lk
  auto mask_of = [&](int kv0) -> float {
    const int key = kv0 + tid;
    if (tid >= TK || key >= a.Lk) return -INFINITY;
    return a.key_mask ? a.key_mask[(size_t)b * a.Lk + key] * LOG2E : 0.f;
  };
  TileRegs kreg, vreg;
  float mreg;
  tile_load(kreg, kp, a.ldk, 0, a.Lk, tid);
  tile_load(vreg, vp, a.ldv, 0, a.Lk, tid);
  mreg = mask_of(0);
  tile_store_rows(s_k[0], kreg, tid);
  tile_store_rows(s_v[0], vreg, tid);
  if (tid < TK) s_mask[0][tid] = mreg;
  if (TK < a.Lk) {
    tile_load(kreg, kp, a.ldk, TK, a.Lk, tid);
    tile_load(vreg, vp, a.ldv, TK, a.Lk, tid);
    mreg = mask_of(TK);
  }
  __syncthreads();
  for (int kv0 = 0, cur = 0; kv0 < a.Lk; kv0 += TK, cur ^= 1) {
    const bf16_raw* ck = s_k[cur];
    const bf16_raw* cv = s_v[cur];
    const float* cmask = s_mask[cur];

    // ---- S^T = K Q^T
    f32x4 sacc[QT][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) sacc[qt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kf = lds_frag_rows(ck, t, ks, lane);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) sacc[qt][t] = mfma16(kf, qf[qt][ks], sacc[qt][t]);
      }
    }
    // ---- stage tile j+1 into the other buffer (its last readers passed the barrier that closed iteration j-1);
    //      the LDS writes overlap the softmax arithmetic below; then start the global loads of tile j+2
    if (kv0 + TK < a.Lk) {
      tile_store_rows(s_k[cur ^ 1], kreg, tid);
      tile_store_rows(s_v[cur ^ 1], vreg, tid);
      if (tid < TK) s_mask[cur ^ 1][tid] = mreg;
      if (kv0 + 2 * TK < a.Lk) {
        tile_load(kreg, kp, a.ldk, kv0 + 2 * TK, a.Lk, tid);
        tile_load(vreg, vp, a.ldv, kv0 + 2 * TK, a.Lk, tid);
        mreg = mask_of(kv0 + 2 * TK);
      }
    }
    // ---- online softmax (log2 domain), per owned query
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 mk = *reinterpret_cast<const float4*>(&cmask[t * 16 + g * 4]);
        const float mkv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = sacc[qt][t][r] * sc2 + mkv[r];
          if (BIAS) {
            const int key = kv0 + t * 16 + g * 4 + r;
            if (key < a.Lk && qrow[qt] < a.Lq) s += a.bias[((size_t)b * a.Lq + qrow[qt]) * a.Lk + key] * LOG2E;
          }
          sacc[qt][t][r] = s;
          mx = fmaxf(mx, s);
        }
      }
      mx = quad_max(mx);
      const float m_new = fmaxf(m_run[qt], mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fast_exp2(m_run[qt] - m_use);  // first tile: exp2(-inf) = 0
      m_run[qt] = m_new;
      float psum = 0.f;
      const uint32_t rbase = attn_row_base(a, b, h, qrow[qt]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = fast_exp2(sacc[qt][t][r] - m_use);
          psum += p[r];
        }
        if (DROP) {   // keys kv0+16t+4g .. +3: two index pairs, one hash each
          const uint32_t pr = (rbase + (uint32_t)(kv0 + t * 16 + g * 4)) >> 1;
          const uint32_t b0 = bb_pair_bits(a.drop_key, pr), b1 = bb_pair_bits(a.drop_key, pr + 1);
          const bool k0 = bb_keep_lo(b0, a.drop_thr), k1 = bb_keep_hi(b0, a.drop_thr);
          const bool k2 = bb_keep_lo(b1, a.drop_thr), k3 = bb_keep_hi(b1, a.drop_thr);
          p[0] = k0 ? p[0] * keep_scale : 0.f;
          p[1] = k1 ? p[1] * keep_scale : 0.f;
          p[2] = k2 ? p[2] * keep_scale : 0.f;
          p[3] = k3 ? p[3] * keep_scale : 0.f;
          if (a.drop_bits != nullptr) {   // the four wave-wide compare masks ARE the keep bits of 16 queries x 16 keys
            const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1), m2 = __ballot(k2), m3 = __ballot(k3);
            if (lane == 0) {
              uint64_t* wp = a.drop_bits + attn_bits_word(a, b * a.nh + h, (qbase >> 4) + qt, kv0 >> 6, t, 0);
              wp[0] = m0; wp[1] = m1; wp[2] = m2; wp[3] = m3;
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[qt][t][r] = p[r];
      }
      l_run[qt] = l_run[qt] * alpha + psum;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] *= alpha;
    }
    // ---- O^T += V^T P^T
    bf16x8 pb[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      pb[qt][0] = pack_pair(sacc[qt][0], sacc[qt][1]);
      pb[qt][1] = pack_pair(sacc[qt][2], sacc[qt][3]);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        // rows d = 16 dt + (l & 15), k-slots (g, j) <-> key 32 m + 16 (j >> 2) + 4 g + (j & 3): ds_read_b64_tr_b16 x 2
        const bf16x8 vf = lds_frag_tr(cv, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) oacc[qt][dt] = mfma16(vf, pb[qt][m], oacc[qt][dt]);
      }
    __syncthreads();  // one barrier per tile: buffer `cur` is free again, buffer `cur^1` is complete
  }

  // ---- epilogue: normalise, store O[q][h*64 + dt*16 + g*4 .. +3] and the log-sum-exp
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l = quad_sum(l_run[qt]);
    const float inv = 1.0f / l;
    if (qrow[qt] < a.Lq) {
      bf16_raw* op = (bf16_raw*)a.o + (size_t)b * a.bso + (size_t)qrow[qt] * a.ldo + h * ATTN_D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        st4<bf16_raw>(op + dt * 16 + g * 4, make_float4(oacc[qt][dt][0] * inv, oacc[qt][dt][1] * inv,
                                                         oacc[qt][dt][2] * inv, oacc[qt][dt][3] * inv));
      if (a.lse && g == 0) {
        const float mu = (m_run[qt] == -INFINITY) ? 0.f : m_run[qt];
        a.lse[((size_t)b * a.nh + h) * a.Lq + qrow[qt]] = (mu + log2f(l)) * LN2;
      }
    }
  }
}

// =============================================================================================
// Backward, part 1: dQ (and dbias).  Same ownership as the forward; per key tile
//   S^T, P = exp2(S2 - lse2);  dP^T = V dO^T;  dS = P * (drop(dP) - delta);  dQ^T += K^T dS^T
// =============================================================================================
template <int QT, bool BIAS, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_mfma_dq_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_raw s_k[2][TK * LDT];
  __shared__ __attribute__((aligned(16))) bf16_raw s_v[2][TK * LDT];
  __shared__ __attribute__((aligned(16))) bf16_raw s_kt[2][ATTN_D * LDT];
  __shared__ __attribute__((aligned(16))) float s_mask[2][TK];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
  int blk, h, b;
  attn_decode_block(a, blk, h, b);
  if (DROP) a.drop_key = bb_salted(a.drop_key, a.salt);
  const int qbase = blk * (64 * QT) + w * (16 * QT);
  const bf16_raw* qp = (const bf16_raw*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const bf16_raw* kp = (const bf16_raw*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const bf16_raw* vp = (const bf16_raw*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const bf16_raw* dop = (const bf16_raw*)a.dout + (size_t)b * a.bso + h * ATTN_D;
  const float sc2 = a.scale * LOG2E;
  const float keep_scale = DROP ? 1.0f / (1.0f - a.drop_p) : 1.0f;

  const bf16_raw* op = (const bf16_raw*)a.o + (size_t)b * a.bso + h * ATTN_D;
  bf16x8 qf[QT][2], dof[QT][2];
  int qrow[QT];
  float lse2[QT], dlt[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qrow[qt] = qbase + qt * 16 + c;
    const int r = qrow[qt] < a.Lq ? qrow[qt] : a.Lq - 1;
    float dsum = 0.f;      // delta[q] = sum_d dO[q][d] * O[q][d]: this lane's 16 dims, then the 4 lanes of the query
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[qt][ks] = as_bf16x8(ld_frag_global(qp, a.ldq, r, ks * 32 + g * 8));
      const uint4 du = ld_frag_global(dop, a.ldo, r, ks * 32 + g * 8);
      const uint4 ou = ld_frag_global(op, a.ldo, r, ks * 32 + g * 8);
      dof[qt][ks] = as_bf16x8(du);
      const uint32_t dw[4] = {du.x, du.y, du.z, du.w}, ow[4] = {ou.x, ou.y, ou.z, ou.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        dsum += __uint_as_float(dw[j] << 16) * __uint_as_float(ow[j] << 16) +
                __uint_as_float(dw[j] & 0xffff0000u) * __uint_as_float(ow[j] & 0xffff0000u);
    }
    dlt[qt] = quad_sum(dsum);
    const size_t ridx = ((size_t)b * a.nh + h) * a.Lq + r;
    lse2[qt] = a.lse[ridx] * LOG2E;
    if (g == 0 && qrow[qt] < a.Lq) const_cast<float*>(a.delta)[ridx] = dlt[qt];   // consumed by the dK/dV kernel
  }
  f32x4 dqacc[QT][4];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dqacc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto mask_of = [&](int kv0) -> float {
    const int key = kv0 + tid;
    if (tid >= TK || key >= a.Lk) return -INFINITY;
    return a.key_mask ? a.key_mask[(size_t)b * a.Lk + key] * LOG2E : 0.f;
  };
  TileRegs kreg, vreg;
  float mreg;
  tile_load(kreg, kp, a.ldk, 0, a.Lk, tid);
  tile_load(vreg, vp, a.ldv, 0, a.Lk, tid);
  mreg = mask_of(0);
  tile_store_rows(s_k[0], kreg, tid);
  tile_store_cols(s_kt[0], kreg, tid);
  tile_store_rows(s_v[0], vreg, tid);
  if (tid < TK) s_mask[0][tid] = mreg;
  if (TK < a.Lk) {
    tile_load(kreg, kp, a.ldk, TK, a.Lk, tid);
    tile_load(vreg, vp, a.ldv, TK, a.Lk, tid);
    mreg = mask_of(TK);
  }
  __syncthreads();
  for (int kv0 = 0, cur = 0; kv0 < a.Lk; kv0 += TK, cur ^= 1) {
    const bf16_raw* ck = s_k[cur];
    const bf16_raw* cv = s_v[cur];
    const bf16_raw* ckt = s_kt[cur];
    const float* cmask = s_mask[cur];
    bool staged = false;

    // two halves of 32 keys: scores + dP for key tiles (2m, 2m+1), then their contribution to dQ^T
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x4 sacc[QT][2], dpacc[QT][2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * m + tt;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          sacc[qt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          dpacc[qt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 kf = lds_frag_rows(ck, t, ks, lane);
          const bf16x8 vf = lds_frag_rows(cv, t, ks, lane);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            sacc[qt][tt] = mfma16(kf, qf[qt][ks], sacc[qt][tt]);
            dpacc[qt][tt] = mfma16(vf, dof[qt][ks], dpacc[qt][tt]);
          }
        }
      }
      if (!staged && kv0 + TK < a.Lk) {   // next tile -> other buffer, overlapping the VALU work below
        tile_store_rows(s_k[cur ^ 1], kreg, tid);
        tile_store_cols(s_kt[cur ^ 1], kreg, tid);
        tile_store_rows(s_v[cur ^ 1], vreg, tid);
        if (tid < TK) s_mask[cur ^ 1][tid] = mreg;
        if (kv0 + 2 * TK < a.Lk) {
          tile_load(kreg, kp, a.ldk, kv0 + 2 * TK, a.Lk, tid);
          tile_load(vreg, vp, a.ldv, kv0 + 2 * TK, a.Lk, tid);
          mreg = mask_of(kv0 + 2 * TK);
        }
      }
      staged = true;
      bf16x8 dsb[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const uint32_t rbase = attn_row_base(a, b, h, qrow[qt]);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * m + tt;
          const float4 mk = *reinterpret_cast<const float4*>(&cmask[t * 16 + g * 4]);
          const float mkv[4] = {mk.x, mk.y, mk.z, mk.w};
          bool keep[4] = {true, true, true, true};
          if (DROP) {
            const uint32_t pr = (rbase + (uint32_t)(kv0 + t * 16 + g * 4)) >> 1;
            const uint32_t b0 = bb_pair_bits(a.drop_key, pr), b1 = bb_pair_bits(a.drop_key, pr + 1);
            keep[0] = bb_keep_lo(b0, a.drop_thr); keep[1] = bb_keep_hi(b0, a.drop_thr);
            keep[2] = bb_keep_lo(b1, a.drop_thr); keep[3] = bb_keep_hi(b1, a.drop_thr);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kv0 + t * 16 + g * 4 + r;
            const bool valid = key < a.Lk && qrow[qt] < a.Lq;
            float sv = sacc[qt][tt][r] * sc2 + mkv[r];
            if (BIAS && valid) sv += a.bias[((size_t)b * a.Lq + qrow[qt]) * a.Lk + key] * LOG2E;
            const float p = fast_exp2(sv - lse2[qt]);
            const float dp = keep[r] ? dpacc[qt][tt][r] * keep_scale : 0.f;
            const float ds = valid ? p * (dp - dlt[qt]) : 0.f;
            sacc[qt][tt][r] = ds;
            if (a.dbias && valid) a.dbias[(((size_t)b * a.nh + h) * a.Lq + qrow[qt]) * a.Lk + key] = ds;
          }
        }
        dsb[qt] = pack_pair(sacc[qt][0], sacc[qt][1]);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 kt = lds_frag_cols(ckt, dt, m, lane);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) dqacc[qt][dt] = mfma16(kt, dsb[qt], dqacc[qt][dt]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
    if (qrow[qt] < a.Lq) {
      bf16_raw* dqp = (bf16_raw*)a.dq + (size_t)b * a.bsq + (size_t)qrow[qt] * a.ldq + h * ATTN_D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        st4<bf16_raw>(dqp + dt * 16 + g * 4,
                      make_float4(dqacc[qt][dt][0] * a.scale, dqacc[qt][dt][1] * a.scale, dqacc[qt][dt][2] * a.scale,
                                  dqacc[qt][dt][3] * a.scale));
    }
}

// =============================================================================================
// Backward, part 2: dK, dV.  Wave w owns KT key tiles: keys k0 + (w*KT + kt)*16 + (l&15); loop over query tiles.
//   S = Q K^T (A = Q tile rows, B = K fragments in registers) -> lane (key = l&15) holds S[q = 16 t + 4 g + r][key]
//   dV^T += dO^T Pd ;  dK^T += Q^T dS      (A = transposed dO / Q tiles, B = packed Pd / dS, k-slots <-> queries)
// =============================================================================================
template <int KT, bool BIAS, bool DROP>
__global__ __launch_bounds__(256, KT == 1 ? 2 : 1) void attn_mfma_dkv_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dkv_smem[];
  bf16_raw* const tiles = reinterpret_cast<bf16_raw*>(dkv_smem);            // [2 buffers][4 tiles][TK * LDT]
  float* const stats = reinterpret_cast<float*>(dkv_smem + DKV_TILE_BYTES);   // [2 buffers][2][TK]
  auto s_q = [&](int bf) { return tiles + (bf * 4 + 0) * (TK * LDT); };
  auto s_do = [&](int bf) { return tiles + (bf * 4 + 1) * (TK * LDT); };
  auto s_qt = [&](int bf) { return tiles + (bf * 4 + 2) * (TK * LDT); };
  auto s_dot = [&](int bf) { return tiles + (bf * 4 + 3) * (TK * LDT); };
  auto s_lse2 = [&](int bf) { return stats + (bf * 2 + 0) * TK; };
  auto s_dlt = [&](int bf) { return stats + (bf * 2 + 1) * TK; };
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
  int blk, h, b;
  attn_decode_block(a, blk, h, b);
  if (DROP) a.drop_key = bb_salted(a.drop_key, a.salt);
  const int kbase = blk * (64 * KT) + w * (16 * KT);
  const bf16_raw* qp = (const bf16_raw*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const bf16_raw* kp = (const bf16_raw*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const bf16_raw* vp = (const bf16_raw*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const bf16_raw* dop = (const bf16_raw*)a.dout + (size_t)b * a.bso + h * ATTN_D;
  const float sc2 = a.scale * LOG2E;
  const float keep_scale = DROP ? 1.0f / (1.0f - a.drop_p) : 1.0f;

  bf16x8 kf[KT][2], vf[KT][2];
  int krow[KT];
  float mask2[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    krow[kt] = kbase + kt * 16 + c;
    const int r = krow[kt] < a.Lk ? krow[kt] : a.Lk - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[kt][ks] = as_bf16x8(ld_frag_global(kp, a.ldk, r, ks * 32 + g * 8));
      vf[kt][ks] = as_bf16x8(ld_frag_global(vp, a.ldv, r, ks * 32 + g * 8));
    }
    mask2[kt] = a.key_mask ? a.key_mask[(size_t)b * a.Lk + r] * LOG2E : 0.f;
  }
  f32x4 dkacc[KT][4], dvacc[KT][4];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dkacc[kt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dvacc[kt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  // per-query statistics of a tile, one query per thread of the first wave: lse (log2 domain; +inf for padding
  // rows so that their probabilities vanish) and delta
  auto stats_of = [&](int q0, float& l2, float& dl) {
    const int qi = q0 + tid;
    l2 = INFINITY;
    dl = 0.f;
    if (tid < TK && qi < a.Lq) {
      const size_t ridx = ((size_t)b * a.nh + h) * a.Lq + qi;
      l2 = a.lse[ridx] * LOG2E;
      dl = a.delta[ridx];
    }
  };
  TileRegs qreg, doreg;
  float lreg, dreg;
  tile_load(qreg, qp, a.ldq, 0, a.Lq, tid);
  tile_load(doreg, dop, a.ldo, 0, a.Lq, tid);
  stats_of(0, lreg, dreg);
  tile_store_rows(s_q(0), qreg, tid);
  tile_store_cols(s_qt(0), qreg, tid);
  tile_store_rows(s_do(0), doreg, tid);
  tile_store_cols(s_dot(0), doreg, tid);
  if (tid < TK) { s_lse2(0)[tid] = lreg; s_dlt(0)[tid] = dreg; }
  if (TK < a.Lq) {
    tile_load(qreg, qp, a.ldq, TK, a.Lq, tid);
    tile_load(doreg, dop, a.ldo, TK, a.Lq, tid);
    stats_of(TK, lreg, dreg);
  }
  __syncthreads();
  for (int q0 = 0, cur = 0; q0 < a.Lq; q0 += TK, cur ^= 1) {
    const bf16_raw *cq = s_q(cur), *cdo = s_do(cur), *cqt = s_qt(cur), *cdot = s_dot(cur);
    const float *clse = s_lse2(cur), *cdlt = s_dlt(cur);
    bool staged = false;

    // two halves of 32 queries: S and dP for query tiles (2m, 2m+1), then their contribution to dK^T / dV^T
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x4 sacc[KT][2], dpacc[KT][2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * m + tt;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          sacc[kt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          dpacc[kt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 qa = lds_frag_rows(cq, t, ks, lane);
          const bf16x8 da = lds_frag_rows(cdo, t, ks, lane);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            sacc[kt][tt] = mfma16(qa, kf[kt][ks], sacc[kt][tt]);
            dpacc[kt][tt] = mfma16(da, vf[kt][ks], dpacc[kt][tt]);
          }
        }
      }
      if (!staged && q0 + TK < a.Lq) {   // next query tile -> other buffer, overlapping the VALU work below
        tile_store_rows(s_q(cur ^ 1), qreg, tid);
        tile_store_cols(s_qt(cur ^ 1), qreg, tid);
        tile_store_rows(s_do(cur ^ 1), doreg, tid);
        tile_store_cols(s_dot(cur ^ 1), doreg, tid);
        if (tid < TK) { s_lse2(cur ^ 1)[tid] = lreg; s_dlt(cur ^ 1)[tid] = dreg; }
        if (q0 + 2 * TK < a.Lq) {
          tile_load(qreg, qp, a.ldq, q0 + 2 * TK, a.Lq, tid);
          tile_load(doreg, dop, a.ldo, q0 + 2 * TK, a.Lq, tid);
          stats_of(q0 + 2 * TK, lreg, dreg);
        }
      }
      staged = true;
      // sacc -> dS, dpacc -> dropped P
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * m + tt;
        const float4 l4 = *reinterpret_cast<const float4*>(&clse[t * 16 + g * 4]);
        const float4 d4 = *reinterpret_cast<const float4*>(&cdlt[t * 16 + g * 4]);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = q0 + t * 16 + g * 4 + r;
            float sv = sacc[kt][tt][r] * sc2 + mask2[kt];
            if (BIAS && qi < a.Lq && krow[kt] < a.Lk)
              sv += a.bias[((size_t)b * a.Lq + qi) * a.Lk + krow[kt]] * LOG2E;
            const float p = fast_exp2(sv - lv[r]);
            float dp = dpacc[kt][tt][r], pd = p;
            if (DROP) {
              const bool keep = bb_keep(a.drop_key, attn_elem(a, b, h, qi, krow[kt]), a.drop_thr);
              dp = keep ? dp * keep_scale : 0.f;
              pd = keep ? p * keep_scale : 0.f;
            }
            sacc[kt][tt][r] = p * (dp - dv[r]);
            dpacc[kt][tt][r] = pd;
          }
      }
      bf16x8 dsb[KT], pdb[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        dsb[kt] = pack_pair(sacc[kt][0], sacc[kt][1]);
        pdb[kt] = pack_pair(dpacc[kt][0], dpacc[kt][1]);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 qt_ = lds_frag_cols(cqt, dt, m, lane);
        const bf16x8 dot_ = lds_frag_cols(cdot, dt, m, lane);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          dkacc[kt][dt] = mfma16(qt_, dsb[kt], dkacc[kt][dt]);
          dvacc[kt][dt] = mfma16(dot_, pdb[kt], dvacc[kt][dt]);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
    if (krow[kt] < a.Lk) {
      bf16_raw* dkp = (bf16_raw*)a.dk + (size_t)b * a.bsk + (size_t)krow[kt] * a.ldk + h * ATTN_D;
      bf16_raw* dvp = (bf16_raw*)a.dv + (size_t)b * a.bsv + (size_t)krow[kt] * a.ldv + h * ATTN_D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        st4<bf16_raw>(dkp + dt * 16 + g * 4,
                      make_float4(dkacc[kt][dt][0] * a.scale, dkacc[kt][dt][1] * a.scale, dkacc[kt][dt][2] * a.scale,
                                  dkacc[kt][dt][3] * a.scale));
        st4<bf16_raw>(dvp + dt * 16 + g * 4,
                      make_float4(dvacc[kt][dt][0], dvacc[kt][dt][1], dvacc[kt][dt][2], dvacc[kt][dt][3]));
      }
    }
}

// =============================================================================================
// launchers
// =============================================================================================
static bool aligned8(const AttnArgs& a) {
  return a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.bsq % 8 == 0 && a.bsk % 8 == 0 &&
         a.bsv % 8 == 0 && a.bso % 8 == 0 && ((uintptr_t)a.q % 16) == 0 && ((uintptr_t)a.k % 16) == 0 &&
         ((uintptr_t)a.v % 16) == 0 && ((uintptr_t)a.o % 16) == 0;
}

// Tile-shape knobs for A/B measurements (read once): BEVBERT_FWD_QT / BEVBERT_DQ_QT / BEVBERT_DKV_KT in {1, 2}.
static int env_knob(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && (v[0] == '1' || v[0] == '2')) ? v[0] - '0' : dflt;
}

// compile-time specialisation on (graph bias present, dropout active): the common launches carry neither
#define BB_DISPATCH_FLAGS(KERNEL, N, NBLK, SMEM)                                                         \
  do {                                                                                                  \
    AttnArgs a = a_in;                                                                                  \
    a.nblk = (NBLK);                                                                                    \
    const dim3 GRID((unsigned)a.nblk * a.nh * a.B);                                                     \
    const bool hb = a.bias != nullptr, hd = a.drop_p > 0.f;                                             \
    if (hb && hd) hipLaunchKernelGGL((KERNEL<N, true, true>), GRID, dim3(256), SMEM, st, a);            \
    else if (hb) hipLaunchKernelGGL((KERNEL<N, true, false>), GRID, dim3(256), SMEM, st, a);            \
    else if (hd) hipLaunchKernelGGL((KERNEL<N, false, true>), GRID, dim3(256), SMEM, st, a);            \
    else hipLaunchKernelGGL((KERNEL<N, false, false>), GRID, dim3(256), SMEM, st, a);                   \
  } while (0)

int attn_mfma_fwd(const AttnArgs& a_in, hipStream_t st) {
  const AttnArgs& a = a_in;
  static const int fwd_qt = env_knob("BEVBERT_FWD_QT", 2);
  BB_REQUIRE(aligned8(a), "attention (MFMA path): pointers must be 16-byte aligned and strides multiples of 8 elements");
  if (a.Lq > 64 && fwd_qt == 2)
    BB_DISPATCH_FLAGS(attn_mfma_fwd_kernel, 2, (a.Lq + 127) / 128, 0);
  else
    BB_DISPATCH_FLAGS(attn_mfma_fwd_kernel, 1, (a.Lq + 63) / 64, 0);
  BB_CHECK_LAUNCH("attn_fwd(mfma)");
  return BB_OK;
}

template <int KT, bool B_, bool D_>
static bool raise_lds_limit() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_dkv_kernel<KT, B_, D_>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, DKV_SMEM_BYTES) == hipSuccess;
}

int attn_mfma_bwd(const AttnArgs& a_in, hipStream_t st) {
  const AttnArgs& a = a_in;
  BB_REQUIRE(aligned8(a), "attention (MFMA path): pointers must be 16-byte aligned and strides multiples of 8 elements");
  BB_REQUIRE(((uintptr_t)a.dout % 16) == 0 && ((uintptr_t)a.dq % 16) == 0 && ((uintptr_t)a.dk % 16) == 0 &&
                 ((uintptr_t)a.dv % 16) == 0, "attention bwd (MFMA path): gradient pointers must be 16-byte aligned");
  static const int dq_qt = env_knob("BEVBERT_DQ_QT", 2), dkv_kt = env_knob("BEVBERT_DKV_KT", 1);
  if (a.Lq > 64 && dq_qt == 2)
    BB_DISPATCH_FLAGS(attn_mfma_dq_kernel, 2, (a.Lq + 127) / 128, 0);
  else
    BB_DISPATCH_FLAGS(attn_mfma_dq_kernel, 1, (a.Lq + 63) / 64, 0);
  static const bool attr_ok = [] {   // 73.7 KB of dynamic LDS exceeds the default 64 KB cap: opt in once per variant
    bool ok = true;
    ok &= raise_lds_limit<1, false, false>() && raise_lds_limit<1, false, true>();
    ok &= raise_lds_limit<1, true, false>() && raise_lds_limit<1, true, true>();
    ok &= raise_lds_limit<2, false, false>() && raise_lds_limit<2, false, true>();
    ok &= raise_lds_limit<2, true, false>() && raise_lds_limit<2, true, true>();
    return ok;
  }();
  BB_REQUIRE(attr_ok, "attention bwd: cannot raise the dynamic LDS limit to %d bytes", DKV_SMEM_BYTES);
  if (a.Lk > 64 && dkv_kt == 2)
    BB_DISPATCH_FLAGS(attn_mfma_dkv_kernel, 2, (a.Lk + 127) / 128, DKV_SMEM_BYTES);
  else
    BB_DISPATCH_FLAGS(attn_mfma_dkv_kernel, 1, (a.Lk + 63) / 64, DKV_SMEM_BYTES);
  BB_CHECK_LAUNCH("attn_bwd(mfma)");
  return BB_OK;
}
