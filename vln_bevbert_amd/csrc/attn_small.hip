// K2 for SHORT key sequences (Lk <= 96, no graph bias): the 80-token text, the 36-view panoramas, the 17-node global map
// as keys -- 80 x 80 text self-attention alone is 35 of the 71 attention sites of three pre-training steps.
//
// The tiled kernels (attn_mfma.hip / attn_bwd1.hip) treat such a problem as one 64-key tile and a half: a 256-thread
// workgroup per (batch, head) that spends its time in barriers and in the latency of three dependent
// global -> LDS -> MFMA round trips, on 128 x 128 padded scores for 80 x 80 real ones (r03x: 25.9 us forward, 33.0 us
// backward at B = 64; the operands are 31 MB, 4 us of HBM time).  Here the whole key range is ONE tile set:
//
//   forward   a wave owns 16-query tiles; the K fragments of all keys (A operand of S^T = K Q^T) stay in its registers,
//             V is staged once per workgroup in LDS (row-major; V^T fragments by ds_read_b64_tr_b16).  No online
//             softmax: all scores of a query are in one lane group at once -- max, exp2, sum, dropout (inline hash; the
//             wave-wide compare masks are stored as the keep-bit matrix for the backward), P V, store.  One barrier in
//             the whole kernel.
//   backward  ONE WAVE per (batch, head), no barrier at all (a wave's LDS operations execute in order).  Scores are
//             recomputed NON-transposed, S = Q K^T: lane <-> key, registers <-> queries, so that P and dS ARE the B
//             operands of dV^T += dO^T P and dK^T += Q^T dS (contraction over queries; Q^T / dO^T fragments by
//             transposing reads of the row-major tiles) without leaving their lanes.  dK^T / dV^T of all keys live in
//             registers for the whole kernel.  Only dS crosses lanes: it is written once as a [key][query] bf16 image
//             and read back by transposing reads as the B operand of dQ^T = K^T dS^T.  Queries advance in chunks of 32;
//             the global loads of chunk i + 1 are in flight while chunk i is computed.
//
// STATUS: opt-in (BEVBERT_ATTN_SMALL=1), parity-tested, NOT the default.  Measured on the MI355X (r03y, B = 64, p = 0.1):
// 80 x 80 forward 17.6 us (tiled forward 12.3 us + 5.9 us bit generation), backward 30.6 us (single-pass tiled 27.7 us);
// 441 x 80: 51.7 / 79.0 us against 38.6 / 68.9 us.  With at most three waves per CU nothing hides a wave's own LDS and
// global round trips -- the one-wave backward spends ~24 k cycles per 32-query chunk on ~2 k cycles of MFMA work.  What
// these shapes need is more waves per (batch, head), not fewer barriers ("short key sequences").
//
// Dropout in the backward reads the keep bits the forward left (forward layout, attn_common.h): for a (query tile, key
// tile) pair a lane needs the 4 bits of its key and its 4 queries, which sit next to each other in ONE 64-bit word.
#include "attn_mfma_common.h"

#define SM_MAXKT 6      // 16-key tiles: Lk <= 96
#define SM_DS_LD 40     // row stride (bf16) of the dS image: 32 queries + 8 (rows stay 8-byte aligned for the tr reads)

__device__ __forceinline__ bf16x8 join_pair(uint2 a, uint2 b) { return as_bf16x8(make_uint4(a.x, a.y, b.x, b.y)); }
__device__ __forceinline__ uint2 pack4(const f32x4& v) {
  return make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}

// =============================================================================================
// Forward.  Workgroup = NW waves of one (batch, head, query block); wave w owns the 16-query tiles
// (blk * NW + w) * qpw .. + qpw - 1.
// =============================================================================================
template <int NKT, bool DROP, bool KMASK>
__global__ __launch_bounds__(512) void attn_small_fwd_kernel(AttnArgs a, int qpw) {
  constexpr int NC = (NKT + 1) / 2;                  // 32-key chunks of the second product
  __shared__ __attribute__((aligned(16))) bf16_raw s_v[32 * NC * LDT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
  const int nw = blockDim.x >> 6;
  int blk, h, b;
  attn_decode_block(a, blk, h, b);
  if (DROP) a.drop_key = bb_salted(a.drop_key, a.salt);
  const bf16_raw* qp = (const bf16_raw*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const bf16_raw* kp = (const bf16_raw*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const bf16_raw* vp = (const bf16_raw*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const float sc2 = a.scale * LOG2E, inv_sc2 = 1.0f / sc2;

  // V -> LDS (rows beyond Lk are zero: they meet P = 0, and 0 * garbage could be NaN)
  for (int ci = tid; ci < 32 * NC * 8; ci += blockDim.x) {
    const int row = ci >> 3, ch = ci & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < a.Lk) v = ld_frag_global(vp, a.ldv, row, ch * 8);
    *reinterpret_cast<uint4*>(s_v + row * LDT + ch * 8) = v;
  }
  // K fragments of every key tile, and the additive key mask in RAW score units (it enters as the C operand of the
  // first product: s_raw = q.k + mask / (scale log2 e), so the softmax needs one FMA per element)
  bf16x8 kf[NKT][2];
  f32x4 minit[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int key = t * 16 + c, kr = key < a.Lk ? key : a.Lk - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) kf[t][ks] = as_bf16x8(ld_frag_global(kp, a.ldk, kr, ks * 32 + g * 8));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = t * 16 + g * 4 + r;             // the accumulator rows of this lane: keys 16 t + 4 g + r
      float m = -INFINITY;
      if (kk < a.Lk) m = KMASK ? a.key_mask[(size_t)b * a.Lk + kk] * (LOG2E * inv_sc2) : 0.f;
      minit[t][r] = m;
    }
  }
  __syncthreads();

  const int qt0 = (blk * nw + w) * qpw;
  for (int i = 0; i < qpw; ++i) {
    const int qt = qt0 + i;
    if (qt * 16 >= a.Lq) break;
    const int qrow = qt * 16 + c;
    const int qr = qrow < a.Lq ? qrow : a.Lq - 1;
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = as_bf16x8(ld_frag_global(qp, a.ldq, qr, ks * 32 + g * 8));
    f32x4 s[NKT];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      s[t] = mfma16(kf[t][0], qf[0], minit[t]);
      s[t] = mfma16(kf[t][1], qf[1], s[t]);
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][r]);
    }
    mx = quad_max(mx);
    const float m2 = (mx == -INFINITY) ? 0.f : mx * sc2;
    float psum = 0.f;
    const uint32_t rbase = attn_row_base(a, b, h, qrow);
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = fast_exp2(fmaf(s[t][r], sc2, -m2));
        psum += p[r];
      }
      if (DROP) {   // keys 16 t + 4 g .. + 3: two index pairs, one hash each (same element numbering as every other kernel)
        const uint32_t pr = (rbase + (uint32_t)(t * 16 + g * 4)) >> 1;
        const uint32_t b0 = bb_pair_bits(a.drop_key, pr), b1 = bb_pair_bits(a.drop_key, pr + 1);
        const bool k0 = bb_keep_lo(b0, a.drop_thr), k1 = bb_keep_hi(b0, a.drop_thr);
        const bool k2 = bb_keep_lo(b1, a.drop_thr), k3 = bb_keep_hi(b1, a.drop_thr);
        p[0] = k0 ? p[0] : 0.f;
        p[1] = k1 ? p[1] : 0.f;
        p[2] = k2 ? p[2] : 0.f;
        p[3] = k3 ? p[3] : 0.f;
        if (a.drop_bits != nullptr && t * 16 < a.Lk) {   // (a tile wholly beyond Lk has no words: NKT rounds the tile count up)
          const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1), m2b = __ballot(k2), m3 = __ballot(k3);
          if (lane == 0) {
            uint64_t* wp = a.drop_bits + attn_bits_word(a, b * a.nh + h, qt, t >> 2, t & 3, 0);
            wp[0] = m0; wp[1] = m1; wp[2] = m2b; wp[3] = m3;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) s[t][r] = p[r];
    }
    // O^T = V^T P^T, contraction over 32-key chunks (the keep scale 1 / (1 - p) is applied with the normalisation)
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < NC; ++m) {
      const bf16x8 pb = pack_pair(s[2 * m], (2 * m + 1 < NKT) ? s[(2 * m + 1 < NKT) ? 2 * m + 1 : 0] : zero);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = mfma16(lds_frag_tr(s_v, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane), pb, o[dt]);
    }
    const float l = quad_sum(psum);
    const float inv = (DROP ? a.keep_scale : 1.0f) / l;
    if (qrow < a.Lq) {
      bf16_raw* op = (bf16_raw*)a.o + (size_t)b * a.bso + (size_t)qrow * a.ldo + h * ATTN_D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        st4<bf16_raw>(op + dt * 16 + g * 4, make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv));
      if (a.lse && g == 0) a.lse[((size_t)b * a.nh + h) * a.Lq + qrow] = (m2 + log2f(l)) * LN2;
    }
  }
}

// =============================================================================================
// Backward: one wave per (batch, head).
// =============================================================================================
template <int NKT> struct SmLds {
  static constexpr int NC = (NKT + 1) / 2;
  static constexpr int k_off = 0;                                   // [32 NC][LDT]   K row-major (rows >= Lk zero)
  static constexpr int v_off = k_off + 32 * NC * LDT * 2;           // [16 NKT][LDT]  V row-major
  static constexpr int q_off = v_off + 16 * NKT * LDT * 2;          // [32][LDT]      Q rows of the chunk
  static constexpr int do_off = q_off + 32 * LDT * 2;               // [32][LDT]      dO rows of the chunk
  static constexpr int ds_off = do_off + 32 * LDT * 2;              // [32 NC][SM_DS_LD] dS of the chunk, [key][query]
  static constexpr int stat_off = ds_off + 32 * NC * SM_DS_LD * 2;  // [2][32] float  lse (log2 domain), delta
  static constexpr int bytes = stat_off + 2 * 32 * 4;
};

template <int CTRL> __device__ __forceinline__ float sm_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sm_sum8(float v) {   // sum over the 8 lanes that hold one row of a staged tile
  v += sm_dpp<0xB1>(v);           // quad_perm [1,0,3,2]
  v += sm_dpp<0x4E>(v);           // quad_perm [2,3,0,1]
  return v + sm_dpp<0x141>(v);    // row_half_mirror: lane i <-> 7 - i of its half row
}

template <int NKT, bool DROP, bool KMASK>
__global__ __launch_bounds__(64, 1) void attn_small_bwd_kernel(AttnArgs a) {
  typedef SmLds<NKT> L;
  constexpr int NC = L::NC;
  __shared__ __attribute__((aligned(16))) unsigned char sm[L::bytes];
  bf16_raw* const s_k = reinterpret_cast<bf16_raw*>(sm + L::k_off);
  bf16_raw* const s_v = reinterpret_cast<bf16_raw*>(sm + L::v_off);
  bf16_raw* const s_q = reinterpret_cast<bf16_raw*>(sm + L::q_off);
  bf16_raw* const s_do = reinterpret_cast<bf16_raw*>(sm + L::do_off);
  bf16_raw* const s_ds = reinterpret_cast<bf16_raw*>(sm + L::ds_off);
  float* const s_lse2 = reinterpret_cast<float*>(sm + L::stat_off);
  float* const s_dlt = s_lse2 + 32;
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  int blk, h, b;
  attn_decode_block(a, blk, h, b);
  const int bh = b * a.nh + h;
  const bf16_raw* qp = (const bf16_raw*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const bf16_raw* kp = (const bf16_raw*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const bf16_raw* vp = (const bf16_raw*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const bf16_raw* op = (const bf16_raw*)a.o + (size_t)b * a.bso + h * ATTN_D;
  const bf16_raw* dop = (const bf16_raw*)a.dout + (size_t)b * a.bso + h * ATTN_D;
  const float* lsep = a.lse + (size_t)bh * a.Lq;
  const float sc2 = a.scale * LOG2E;
  const float ks = DROP ? a.keep_scale : 1.0f;

  // ---- chunk loads: global -> registers (issued one chunk ahead), committed to LDS when the chunk starts
  struct ChunkRegs {
    uint4 q[4], d[4], o[4];
    float lse;
    uint64_t bits[DROP ? 2 * NKT : 1];
  };
  auto chunk_issue = [&](ChunkRegs& r, int q0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ci = lane + 64 * i, row = q0 + (ci >> 3), ch = ci & 7;
      const int rr = row < a.Lq ? row : a.Lq - 1;
      r.q[i] = ld_frag_global(qp, a.ldq, rr, ch * 8);
      r.d[i] = ld_frag_global(dop, a.ldo, rr, ch * 8);
      r.o[i] = ld_frag_global(op, a.ldo, rr, ch * 8);
    }
    const int lq = q0 + (lane & 31);
    r.lse = lsep[lq < a.Lq ? lq : a.Lq - 1];
    if (DROP) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < NKT; ++t)
          r.bits[j * NKT + t] = (t * 16 < a.Lk) ? a.drop_bits[attn_bits_word(a, bh, (q0 >> 4) + j, t >> 2, t & 3, c & 3)] : 0ull;
    }
  };
  auto chunk_commit = [&](const ChunkRegs& r, int q0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ci = lane + 64 * i, row = ci >> 3, ch = ci & 7;
      const bool ok = q0 + row < a.Lq;
      const uint32_t km = ok ? 0xffffffffu : 0u;        // rows beyond Lq: Q = dO = 0 (and lse = +inf below) -> P = dS = 0
      const uint4 qv = make_uint4(r.q[i].x & km, r.q[i].y & km, r.q[i].z & km, r.q[i].w & km);
      const uint4 dv = make_uint4(r.d[i].x & km, r.d[i].y & km, r.d[i].z & km, r.d[i].w & km);
      *reinterpret_cast<uint4*>(s_q + row * LDT + ch * 8) = qv;
      *reinterpret_cast<uint4*>(s_do + row * LDT + ch * 8) = dv;
      const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w}, ow[4] = {r.o[i].x, r.o[i].y, r.o[i].z, r.o[i].w};
      float dsum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        dsum += __uint_as_float(dw[j] << 16) * __uint_as_float(ow[j] << 16) +
                __uint_as_float(dw[j] & 0xffff0000u) * __uint_as_float(ow[j] & 0xffff0000u);
      dsum = sm_sum8(dsum);
      if ((lane & 7) == 0) s_dlt[row] = dsum;
    }
    if (lane < 32) s_lse2[lane] = (q0 + lane < a.Lq) ? r.lse * LOG2E : INFINITY;
  };

  ChunkRegs cr;
  chunk_issue(cr, 0);
  // ---- K, V -> LDS; zero the dS rows the chunks never write (they meet zero K rows, but 0 * garbage could be NaN)
  for (int ci = lane; ci < 32 * NC * 8; ci += 64) {
    const int row = ci >> 3, ch = ci & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < a.Lk) v = ld_frag_global(kp, a.ldk, row, ch * 8);
    *reinterpret_cast<uint4*>(s_k + row * LDT + ch * 8) = v;
  }
  for (int ci = lane; ci < 16 * NKT * 8; ci += 64) {
    const int row = ci >> 3, ch = ci & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < a.Lk) v = ld_frag_global(vp, a.ldv, row, ch * 8);
    *reinterpret_cast<uint4*>(s_v + row * LDT + ch * 8) = v;
  }
  if (NKT & 1)
    for (int i = lane; i < 16 * SM_DS_LD / 4; i += 64)
      reinterpret_cast<uint2*>(s_ds + 16 * NKT * SM_DS_LD)[i] = make_uint2(0, 0);
  // additive key mask (log2 domain) of this lane's key in every key tile; -inf beyond Lk
  float mk2[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int key = 16 * t + c;
    mk2[t] = -INFINITY;
    if (key < a.Lk) mk2[t] = KMASK ? a.key_mask[(size_t)b * a.Lk + key] * LOG2E : 0.f;
  }

  f32x4 dk[4][NKT], dv[4][NKT];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int t = 0; t < NKT; ++t) dk[dt][t] = dv[dt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int q0 = 0; q0 < a.Lq; q0 += 32) {
    chunk_commit(cr, q0);
    uint64_t bw[DROP ? 2 * NKT : 1];
    if (DROP) {
#pragma unroll
      for (int i = 0; i < 2 * NKT; ++i) bw[i] = cr.bits[i];
    }
    if (q0 + 32 < a.Lq) chunk_issue(cr, q0 + 32);

    uint2 pp[2][NKT], pds[2][NKT];         // bf16 P (dropped, scaled) and dS of the two query tiles of the chunk
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16x8 qa0 = lds_frag_rows(s_q, j, 0, lane), qa1 = lds_frag_rows(s_q, j, 1, lane);
      const bf16x8 da0 = lds_frag_rows(s_do, j, 0, lane), da1 = lds_frag_rows(s_do, j, 1, lane);
      const float4 l4 = *reinterpret_cast<const float4*>(s_lse2 + 16 * j + 4 * g);
      const float4 d4 = *reinterpret_cast<const float4*>(s_dlt + 16 * j + 4 * g);
      const float lrow[4] = {l4.x, l4.y, l4.z, l4.w}, drow[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 s = mfma16(qa0, lds_frag_rows(s_k, t, 0, lane), zero);
        s = mfma16(qa1, lds_frag_rows(s_k, t, 1, lane), s);
        f32x4 dp = mfma16(da0, lds_frag_rows(s_v, t, 0, lane), zero);
        dp = mfma16(da1, lds_frag_rows(s_v, t, 1, lane), dp);
        uint32_t kb = 0xfu;
        if (DROP) kb = (uint32_t)(bw[j * NKT + t] >> (16 * (c >> 2) + 4 * g));
        f32x4 pd, dsv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = fast_exp2(fmaf(s[r], sc2, mk2[t] - lrow[r]));
          const bool keep = !DROP || ((kb >> r) & 1u);
          pd[r] = keep ? p * ks : 0.f;
          const float dpd = keep ? dp[r] * ks : 0.f;
          dsv[r] = p * (dpd - drow[r]);
        }
        pp[j][t] = pack4(pd);
        pds[j][t] = pack4(dsv);
        *reinterpret_cast<uint2*>(s_ds + (16 * t + c) * SM_DS_LD + 16 * j + 4 * g) = pds[j][t];
      }
    }
    // ---- dV^T += dO^T P, dK^T += Q^T dS: contraction over the 32 queries of the chunk
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const bf16x8 dot = lds_frag_tr(s_do, LDT, 4 * g, 16 + 4 * g, dt * 16, lane);
      const bf16x8 qtf = lds_frag_tr(s_q, LDT, 4 * g, 16 + 4 * g, dt * 16, lane);
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        dv[dt][t] = mfma16(dot, join_pair(pp[0][t], pp[1][t]), dv[dt][t]);
        dk[dt][t] = mfma16(qtf, join_pair(pds[0][t], pds[1][t]), dk[dt][t]);
      }
    }
    // ---- dQ^T = K^T dS^T over all keys, stored straight away
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 dq[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < NC; ++m) {
        const bf16x8 dsf = lds_frag_tr(s_ds, SM_DS_LD, 32 * m + 4 * g, 32 * m + 16 + 4 * g, 16 * j, lane);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dq[dt] = mfma16(lds_frag_tr(s_k, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane), dsf, dq[dt]);
      }
      const int qrow = q0 + 16 * j + c;
      if (qrow < a.Lq) {
        bf16_raw* dqp = (bf16_raw*)a.dq + (size_t)b * a.bsq + (size_t)qrow * a.ldq + h * ATTN_D;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          st4<bf16_raw>(dqp + dt * 16 + g * 4, make_float4(dq[dt][0] * a.scale, dq[dt][1] * a.scale,
                                                            dq[dt][2] * a.scale, dq[dt][3] * a.scale));
      }
    }
  }
  // ---- dK, dV: lane <-> key 16 t + c, rows d = 16 dt + 4 g + r
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int key = 16 * t + c;
    if (key < a.Lk) {
      bf16_raw* dkp = (bf16_raw*)a.dk + (size_t)b * a.bsk + (size_t)key * a.ldk + h * ATTN_D;
      bf16_raw* dvp = (bf16_raw*)a.dv + (size_t)b * a.bsv + (size_t)key * a.ldv + h * ATTN_D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        st4<bf16_raw>(dkp + dt * 16 + g * 4, make_float4(dk[dt][t][0] * a.scale, dk[dt][t][1] * a.scale,
                                                          dk[dt][t][2] * a.scale, dk[dt][t][3] * a.scale));
        st4<bf16_raw>(dvp + dt * 16 + g * 4, make_float4(dv[dt][t][0], dv[dt][t][1], dv[dt][t][2], dv[dt][t][3]));
      }
    }
  }
}

// =============================================================================================
// Backward for SHORT query AND key sequences (Lq, Lk <= 96): waves that never talk to each other.
//
// One workgroup per (batch, head); K, Q, dO (row-major), lse and delta = rowsum(dO * O) are staged in LDS once, ONE
// barrier, and from there on every wave works alone on data nobody writes:
//   * query-owner waves (one per 16-query tile): S^T = K Q^T and dP^T = V dO^T (lane <-> query, registers <-> keys, the
//     forward's orientation) -> dS^T, which IS the B operand of dQ^T = K^T dS^T (K^T by transposing reads) -> store dQ;
//   * key-owner waves (one per 16-key tile): S = Q K^T and dP = dO V^T (lane <-> key, registers <-> queries) -> P and dS,
//     which ARE the B operands of dV^T += dO^T P and dK^T += Q^T dS (Q^T / dO^T by transposing reads) -> store dK, dV.
// The scores are computed twice (7 tile products instead of 5) -- the price of 10 independent waves per (batch, head)
// for an 80 x 80 problem where the single-pass kernel has 4 waves and 4 barriers per 64-query tile.
// =============================================================================================
typedef const __attribute__((address_space(4))) uint64_t* sm_cu64p;   // uniform loads through the scalar cache

#define SM2_ROWS 96

template <int NKT, bool DROP, bool KMASK>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(5))) void attn_small_bwd2_kernel(AttnArgs a) {
  constexpr int NC = (NKT + 1) / 2;
  __shared__ __attribute__((aligned(16))) bf16_raw s_k[SM2_ROWS * LDT];
  __shared__ __attribute__((aligned(16))) bf16_raw s_q[SM2_ROWS * LDT];
  __shared__ __attribute__((aligned(16))) bf16_raw s_do[SM2_ROWS * LDT];
  __shared__ __attribute__((aligned(16))) bf16_raw s_v[16 * NKT * LDT];
  __shared__ __attribute__((aligned(16))) float s_lse2[SM2_ROWS];
  __shared__ __attribute__((aligned(16))) float s_dlt[SM2_ROWS];
  __shared__ __attribute__((aligned(16))) float s_mk[SM2_ROWS];      // additive key mask, log2 domain; -inf beyond Lk
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqt = (a.Lq + 15) >> 4;                  // query-owner waves 0 .. nqt-1, key-owner waves nqt ..
  int blk, h, b;
  attn_decode_block(a, blk, h, b);
  const int bh = b * a.nh + h;
  const bf16_raw* qp = (const bf16_raw*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const bf16_raw* kp = (const bf16_raw*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const bf16_raw* vp = (const bf16_raw*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const bf16_raw* op = (const bf16_raw*)a.o + (size_t)b * a.bso + h * ATTN_D;
  const bf16_raw* dop = (const bf16_raw*)a.dout + (size_t)b * a.bso + h * ATTN_D;
  const float sc2 = a.scale * LOG2E;
  const float ks = DROP ? a.keep_scale : 1.0f;

  // ---- staging (all waves): rows beyond Lq / Lk are zero, their lse is +inf (P = 0), their delta 0
  for (int ci = tid; ci < SM2_ROWS * 8; ci += blockDim.x) {
    const int row = ci >> 3, ch = ci & 7;
    uint4 kv = make_uint4(0, 0, 0, 0), qv = kv, dv = kv, ov = kv, vv = kv;
    if (row < a.Lk) {
      kv = ld_frag_global(kp, a.ldk, row, ch * 8);
      vv = ld_frag_global(vp, a.ldv, row, ch * 8);
    }
    if (row < 16 * NKT) *reinterpret_cast<uint4*>(s_v + row * LDT + ch * 8) = vv;
    if (row < a.Lq) {
      qv = ld_frag_global(qp, a.ldq, row, ch * 8);
      dv = ld_frag_global(dop, a.ldo, row, ch * 8);
      ov = ld_frag_global(op, a.ldo, row, ch * 8);
    }
    *reinterpret_cast<uint4*>(s_k + row * LDT + ch * 8) = kv;
    *reinterpret_cast<uint4*>(s_q + row * LDT + ch * 8) = qv;
    *reinterpret_cast<uint4*>(s_do + row * LDT + ch * 8) = dv;
    const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
    float dsum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      dsum += __uint_as_float(dw[j] << 16) * __uint_as_float(ow[j] << 16) +
              __uint_as_float(dw[j] & 0xffff0000u) * __uint_as_float(ow[j] & 0xffff0000u);
    dsum = sm_sum8(dsum);
    if ((lane & 7) == 0) s_dlt[row] = dsum;
  }
  if (tid < SM2_ROWS) {
    s_lse2[tid] = tid < a.Lq ? a.lse[(size_t)bh * a.Lq + tid] * LOG2E : INFINITY;
    float m = -INFINITY;
    if (tid < a.Lk) m = KMASK ? a.key_mask[(size_t)b * a.Lk + tid] * LOG2E : 0.f;
    s_mk[tid] = m;
  }

  if (w < nqt) {
    // =================================================================== query-owner wave: tile qt = w -> dQ
    const int qt = w;
    sm_cu64p wq = nullptr;
    if (DROP) wq = (sm_cu64p)(uintptr_t)(a.drop_bits + attn_bits_word(a, bh, qt, 0, 0, 0));
    __syncthreads();
    const bf16x8 qb0 = lds_frag_rows(s_q, qt, 0, lane), qb1 = lds_frag_rows(s_q, qt, 1, lane);
    const bf16x8 db0 = lds_frag_rows(s_do, qt, 0, lane), db1 = lds_frag_rows(s_do, qt, 1, lane);
    const float lse2 = s_lse2[qt * 16 + c], dlt = s_dlt[qt * 16 + c];
    f32x4 ds[NKT];
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      f32x4 s = mfma16(lds_frag_rows(s_k, t, 0, lane), qb0, zero);
      s = mfma16(lds_frag_rows(s_k, t, 1, lane), qb1, s);
      f32x4 dp = mfma16(lds_frag_rows(s_v, t, 0, lane), db0, zero);
      dp = mfma16(lds_frag_rows(s_v, t, 1, lane), db1, dp);
      const float4 m4 = *reinterpret_cast<const float4*>(s_mk + 16 * t + 4 * g);
      const float mk[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = fast_exp2(fmaf(s[r], sc2, mk[r] - lse2));
        float dpd = dp[r] * ks;
        if (DROP) {
          const uint64_t word = (t * 16 < a.Lk) ? wq[(t >> 2) * 16 + (t & 3) * 4 + r] : 0ull;
          dpd = __builtin_amdgcn_inverse_ballot_w64(word) ? dpd : 0.f;
        }
        ds[t][r] = p * (dpd - dlt);
      }
    }
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = zero;
#pragma unroll
    for (int m = 0; m < NC; ++m) {
      const bf16x8 dsb = pack_pair(ds[2 * m], (2 * m + 1 < NKT) ? ds[(2 * m + 1 < NKT) ? 2 * m + 1 : 0] : zero);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        dq[dt] = mfma16(lds_frag_tr(s_k, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane), dsb, dq[dt]);
    }
    const int qrow = qt * 16 + c;
    if (qrow < a.Lq) {
      bf16_raw* dqp = (bf16_raw*)a.dq + (size_t)b * a.bsq + (size_t)qrow * a.ldq + h * ATTN_D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        st4<bf16_raw>(dqp + dt * 16 + g * 4, make_float4(dq[dt][0] * a.scale, dq[dt][1] * a.scale,
                                                          dq[dt][2] * a.scale, dq[dt][3] * a.scale));
    }
  } else {
    // =================================================================== key-owner wave: tile t = w - nqt -> dK, dV
    const int t = w - nqt;
    const int key = t * 16 + c;
    uint64_t bw[DROP ? 6 : 1];                         // keep bits of (query tile, this key tile): one word per lane and tile
    if (DROP) {
#pragma unroll
      for (int j = 0; j < 6; ++j)
        bw[j] = (j < nqt) ? a.drop_bits[attn_bits_word(a, bh, j, t >> 2, t & 3, c & 3)] : 0ull;
    }
    __syncthreads();
    const bf16x8 kb0 = lds_frag_rows(s_k, t, 0, lane), kb1 = lds_frag_rows(s_k, t, 1, lane);
    const bf16x8 vb0 = lds_frag_rows(s_v, t, 0, lane), vb1 = lds_frag_rows(s_v, t, 1, lane);
    const float mk2 = s_mk[key];
    f32x4 dk[4], dv[4];
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = zero;
#pragma unroll
    for (int m = 0; m < 3; ++m) {                      // 32-query chunks
      if (m * 32 < a.Lq) {
        uint2 pp[2], pds[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int qt = 2 * m + j;
          f32x4 s = mfma16(lds_frag_rows(s_q, qt, 0, lane), kb0, zero);
          s = mfma16(lds_frag_rows(s_q, qt, 1, lane), kb1, s);
          f32x4 dp = mfma16(lds_frag_rows(s_do, qt, 0, lane), vb0, zero);
          dp = mfma16(lds_frag_rows(s_do, qt, 1, lane), vb1, dp);
          const float4 l4 = *reinterpret_cast<const float4*>(s_lse2 + 16 * qt + 4 * g);
          const float4 d4 = *reinterpret_cast<const float4*>(s_dlt + 16 * qt + 4 * g);
          const float lrow[4] = {l4.x, l4.y, l4.z, l4.w}, drow[4] = {d4.x, d4.y, d4.z, d4.w};
          uint32_t kbits = 0xfu;
          if (DROP) kbits = (uint32_t)(bw[qt] >> (16 * (c >> 2) + 4 * g));
          f32x4 pd, dsv;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = fast_exp2(fmaf(s[r], sc2, mk2 - lrow[r]));
            const bool keep = !DROP || ((kbits >> r) & 1u);
            pd[r] = keep ? p * ks : 0.f;
            const float dpd = keep ? dp[r] * ks : 0.f;
            dsv[r] = p * (dpd - drow[r]);
          }
          pp[j] = pack4(pd);
          pds[j] = pack4(dsv);
        }
        const bf16x8 pb = join_pair(pp[0], pp[1]), dsb = join_pair(pds[0], pds[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = mfma16(lds_frag_tr(s_do, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane), pb, dv[dt]);
          dk[dt] = mfma16(lds_frag_tr(s_q, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane), dsb, dk[dt]);
        }
      }
    }
    if (key < a.Lk) {
      bf16_raw* dkp = (bf16_raw*)a.dk + (size_t)b * a.bsk + (size_t)key * a.ldk + h * ATTN_D;
      bf16_raw* dvp = (bf16_raw*)a.dv + (size_t)b * a.bsv + (size_t)key * a.ldv + h * ATTN_D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        st4<bf16_raw>(dkp + dt * 16 + g * 4, make_float4(dk[dt][0] * a.scale, dk[dt][1] * a.scale, dk[dt][2] * a.scale,
                                                          dk[dt][3] * a.scale));
        st4<bf16_raw>(dvp + dt * 16 + g * 4, make_float4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]));
      }
    }
  }
}

// =============================================================================================
// Host side
// =============================================================================================
static bool sm_aligned(const AttnArgs& a) {
  return a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.bsq % 8 == 0 && a.bsk % 8 == 0 &&
         a.bsv % 8 == 0 && a.bso % 8 == 0 && ((uintptr_t)a.q % 16) == 0 && ((uintptr_t)a.k % 16) == 0 &&
         ((uintptr_t)a.v % 16) == 0 && ((uintptr_t)a.o % 16) == 0;
}

bool attn_small_fwd_supported(const AttnArgs& a) { return a.bias == nullptr && a.Lk <= 16 * SM_MAXKT && sm_aligned(a); }

bool attn_small_bwd_supported(const AttnArgs& a) {
  return a.bias == nullptr && a.dbias == nullptr && a.Lk <= 16 * SM_MAXKT && sm_aligned(a) &&
         (a.drop_p <= 0.f || a.drop_bits != nullptr) && ((uintptr_t)a.dout % 16) == 0 && ((uintptr_t)a.dq % 16) == 0 &&
         ((uintptr_t)a.dk % 16) == 0 && ((uintptr_t)a.dv % 16) == 0;
}

template <int NKT>
static void launch_fwd(const AttnArgs& a, dim3 grid, dim3 block, int qpw, hipStream_t st) {
  const bool hd = a.drop_p > 0.f, hm = a.key_mask != nullptr;
  if (hd && hm) hipLaunchKernelGGL((attn_small_fwd_kernel<NKT, true, true>), grid, block, 0, st, a, qpw);
  else if (hd) hipLaunchKernelGGL((attn_small_fwd_kernel<NKT, true, false>), grid, block, 0, st, a, qpw);
  else if (hm) hipLaunchKernelGGL((attn_small_fwd_kernel<NKT, false, true>), grid, block, 0, st, a, qpw);
  else hipLaunchKernelGGL((attn_small_fwd_kernel<NKT, false, false>), grid, block, 0, st, a, qpw);
}

template <int NKT>
static void launch_bwd(const AttnArgs& a, dim3 grid, hipStream_t st) {
  const bool hd = a.drop_p > 0.f, hm = a.key_mask != nullptr;
  if (hd && hm) hipLaunchKernelGGL((attn_small_bwd_kernel<NKT, true, true>), grid, dim3(64), 0, st, a);
  else if (hd) hipLaunchKernelGGL((attn_small_bwd_kernel<NKT, true, false>), grid, dim3(64), 0, st, a);
  else if (hm) hipLaunchKernelGGL((attn_small_bwd_kernel<NKT, false, true>), grid, dim3(64), 0, st, a);
  else hipLaunchKernelGGL((attn_small_bwd_kernel<NKT, false, false>), grid, dim3(64), 0, st, a);
}

static int sm_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && v[0] >= '1' && v[0] <= '8') ? v[0] - '0' : dflt;
}

int attn_small_fwd(const AttnArgs& a_in, hipStream_t st) {
  AttnArgs a = a_in;
  const int nqt = (a.Lq + 15) / 16, nkt = (a.Lk + 15) / 16;
  // waves per workgroup / query tiles per wave (BEVBERT_SMALL_NW / BEVBERT_SMALL_QPW for A/B measurements):
  // short query sequences get one tile per wave (the whole (batch, head) in one workgroup), long ones (the 441 BEV
  // cells against the text) four waves of two tiles
  static const int env_nw = sm_env("BEVBERT_SMALL_NW", 0), env_qpw = sm_env("BEVBERT_SMALL_QPW", 0);
  int nw = nqt <= 6 ? nqt : 4, qpw = nqt <= 6 ? 1 : 2;
  if (env_nw) nw = env_nw < nqt ? env_nw : nqt;
  if (env_qpw) qpw = env_qpw;
  a.nblk = (nqt + nw * qpw - 1) / (nw * qpw);
  const dim3 grid((unsigned)a.nblk * a.nh * a.B), block(64 * nw);
  if (nkt <= 2) launch_fwd<2>(a, grid, block, qpw, st);
  else if (nkt <= 3) launch_fwd<3>(a, grid, block, qpw, st);
  else if (nkt <= 5) launch_fwd<5>(a, grid, block, qpw, st);
  else launch_fwd<6>(a, grid, block, qpw, st);
  BB_CHECK_LAUNCH("attn_fwd(small)");
  return BB_OK;
}

template <int NKT>
static void launch_bwd2(const AttnArgs& a, dim3 grid, dim3 block, hipStream_t st) {
  const bool hd = a.drop_p > 0.f, hm = a.key_mask != nullptr;
  if (hd && hm) hipLaunchKernelGGL((attn_small_bwd2_kernel<NKT, true, true>), grid, block, 0, st, a);
  else if (hd) hipLaunchKernelGGL((attn_small_bwd2_kernel<NKT, true, false>), grid, block, 0, st, a);
  else if (hm) hipLaunchKernelGGL((attn_small_bwd2_kernel<NKT, false, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((attn_small_bwd2_kernel<NKT, false, false>), grid, block, 0, st, a);
}

// query AND key sequences up to 96: independent query-owner and key-owner waves (attn_small_bwd2_kernel)
bool attn_small_bwd2_supported(const AttnArgs& a) { return attn_small_bwd_supported(a) && a.Lq <= SM2_ROWS; }

int attn_small_bwd2(const AttnArgs& a_in, hipStream_t st) {
  AttnArgs a = a_in;
  const int nkt = (a.Lk + 15) / 16, nqt = (a.Lq + 15) / 16;
  a.nblk = 1;
  const dim3 grid((unsigned)a.nh * a.B), block(64 * (nqt + nkt));
  if (nkt <= 2) launch_bwd2<2>(a, grid, block, st);
  else if (nkt <= 3) launch_bwd2<3>(a, grid, block, st);
  else if (nkt <= 5) launch_bwd2<5>(a, grid, block, st);
  else launch_bwd2<6>(a, grid, block, st);
  BB_CHECK_LAUNCH("attn_bwd(small, independent waves)");
  return BB_OK;
}

int attn_small_bwd(const AttnArgs& a_in, hipStream_t st) {
  AttnArgs a = a_in;
  const int nkt = (a.Lk + 15) / 16;
  a.nblk = 1;
  const dim3 grid((unsigned)a.nh * a.B);
  if (nkt <= 2) launch_bwd<2>(a, grid, st);
  else if (nkt <= 3) launch_bwd<3>(a, grid, st);
  else if (nkt <= 5) launch_bwd<5>(a, grid, st);
  else launch_bwd<6>(a, grid, st);
  BB_CHECK_LAUNCH("attn_bwd(small)");
  return BB_OK;
}
