// K2 for SHORT sequences, round 6: the 80-token text (35 of the 71 attention sites of three pre-training steps), the
// 36-view panoramas, the <= 20-node global map and its cross-attentions with the text.  BertSelfAttention /
// BertOutAttention (pretrain_src/model/vilmodel.py:79-141, 301-352), nn.MultiheadAttention of the panorama encoder
// (pretrain_src/model/transformer.py:138,174-178).
//
// These problems are not MFMA- or VALU-bound: one (batch, head) item is 1.6 MFLOP of products on 30 KB (forward) /
// 50 KB (backward) of operands, and a launch has B x 12 = 768 of them -- three per CU.  What the time goes to is the chain
// global load -> LDS -> MFMA -> softmax -> MFMA -> global store, which every workgroup of a launch walks at the same time,
// and the number of ROUNDS of workgroups a launch needs (rounds 3-5: the tiled kernels pad 80 x 80 to 128 x 128 and hash
// 2.5 x the elements; the 10-wave backward of attn_small.hip needs 59 KB of LDS, so two workgroups share a CU and 768
// items take 1.5 rounds).  Design here:
//
//   * one workgroup per (batch, head) [x query block in the forward]; LDS footprint sized to the REAL tile counts so that
//     three (80 x 80) or more workgroups are resident per CU and a launch is ONE round of workgroups;
//   * forward: one wave per 16-query tile, Q fragments straight from global (in flight before anything else), K and V
//     staged once per workgroup, ONE barrier; all scores of a query live in one lane group: max, exp2, sum, dropout
//     (hash inline -- only the real elements, the keep words are left for the backward -- or stored words when the caller
//     generated them), P V by transposing LDS reads of V, store;
//   * backward: ONE pass, scores computed once.  Wave t first OWNS KEY TILE t: S = Q K^T and dP = dO V^T (lane <-> key,
//     registers <-> queries; its K / V rows are B operands), P and dS are the B operands of dV^T += dO^T P and
//     dK^T += Q^T dS without leaving their lanes; dS is also written as a [key][query] bf16 image.  Barrier.  Then wave q
//     OWNS QUERY TILE q: dQ^T = K^T dS^T with both operands by transposing reads.  Two barriers in the whole kernel.
//
// Dropout: same element numbering, hash and keep-word layout as every other attention kernel (attn_common.h).
#include "attn_mfma_common.h"

typedef const __attribute__((address_space(4))) uint64_t* sh_cu64p;   // uniform loads through the scalar cache

template <int CTRL> __device__ __forceinline__ float sh_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sh_sum8(float v) {   // sum over the 8 lanes that hold one 64-element row
  v += sh_dpp<0xB1>(v);           // quad_perm [1,0,3,2]
  v += sh_dpp<0x4E>(v);           // quad_perm [2,3,0,1]
  return v + sh_dpp<0x141>(v);    // row_half_mirror
}
__device__ __forceinline__ bf16x8 sh_join(uint2 a, uint2 b) { return as_bf16x8(make_uint4(a.x, a.y, b.x, b.y)); }
__device__ __forceinline__ uint2 sh_pack4(const f32x4& v) {
  return make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}
// Transposing fragment read whose second group of four k-slot rows may lie beyond the image (odd tile counts): zeros.
template <bool HI>
__device__ __forceinline__ bf16x8 sh_frag_tr(const bf16_raw* img, int stride, int row_lo, int row_hi, int col0, int lane) {
  const int i = lane & 15;
  const int off = (i >> 2) * stride + col0 + 4 * (i & 3);
  const uint2 lo = lds_tr16(img + row_lo * stride + off);
  uint2 hi = make_uint2(0u, 0u);
  if (HI) hi = lds_tr16(img + row_hi * stride + off);
  return as_bf16x8(make_uint4(lo.x, lo.y, hi.x, hi.y));
}

// A wave's 16 x 64 result tile leaves through LDS as WHOLE 128-byte rows: the accumulators hold, per lane, four consecutive
// head-dim elements of row (lane & 15) for each of the four 16-wide column tiles -- stored directly that is 32-byte runs, four
// store instructions per tile; through a [16][72] bf16 image in LDS (8-byte writes, 16-byte reads) it is two store
// instructions of eight full rows each.  Measured (round 6, B = 64, 80 x 80): the output stores were 5.1 of the backward's
// 18.0 us.  ``stage``: 16 * SH_ST_LD bf16 of LDS private to the wave (a wave's LDS operations execute in order: no barrier).
#define SH_ST_LD 72
__device__ __forceinline__ void sh_store_tile(bf16_raw* stage, const f32x4 (&acc)[4], float mul, bf16_raw* gtile, int64_t ld,
                                              int nrows, int lane) {
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
    *reinterpret_cast<uint2*>(stage + c * SH_ST_LD + dt * 16 + g * 4) =
        make_uint2(pack_bf16x2(acc[dt][0] * mul, acc[dt][1] * mul), pack_bf16x2(acc[dt][2] * mul, acc[dt][3] * mul));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (lane >> 3) + 8 * i, ch = lane & 7;
    const uint4 v = *reinterpret_cast<const uint4*>(stage + row * SH_ST_LD + ch * 8);
    if (row < nrows) *reinterpret_cast<uint4*>(gtile + (size_t)row * ld + ch * 8) = v;
  }
}

// =============================================================================================
// Forward.  Workgroup = (batch, head, query block of nw tiles); wave w owns query tile blk * nw + w.
// DROP: 0 none, 1 hash inline (and leave the keep words), 2 read the keep words the caller generated.
// =============================================================================================
template <int NKT, int DROP>
__global__ __launch_bounds__(512) void attn_short_fwd_kernel(AttnArgs a) {
  constexpr int NC = (NKT + 1) / 2;
  __shared__ __attribute__((aligned(16))) bf16_raw s_k[16 * NKT * LDT];
  __shared__ __attribute__((aligned(16))) bf16_raw s_v[16 * NKT * LDT];
  __shared__ __attribute__((aligned(16))) float s_mk[16 * NKT];     // additive key mask in RAW score units; -inf beyond Lk
  extern __shared__ __attribute__((aligned(16))) bf16_raw s_stage[];      // output tiles, one [16][SH_ST_LD] image per wave
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  int blk, h, b;
  attn_decode_block(a, blk, h, b);
  const int bh = b * a.nh + h;
  if (DROP == 1) a.drop_key = bb_salted(a.drop_key, a.salt);
  const bf16_raw* qp = (const bf16_raw*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const bf16_raw* kp = (const bf16_raw*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const bf16_raw* vp = (const bf16_raw*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const float sc2 = a.scale * LOG2E, inv_sc2 = 1.0f / sc2;

  // Q fragments of this wave's tile: issued first, they need no staging
  const int qt = blk * nw + w;
  const int qrow = qt * 16 + c;
  const int qr = qrow < a.Lq ? qrow : a.Lq - 1;
  bf16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qf[ks] = as_bf16x8(ld_frag_global(qp, a.ldq, qr, ks * 32 + g * 8));

  // K, V -> LDS, row-major, rows beyond Lk zero (they meet P = 0; 0 * garbage could be NaN).  Two chunk iterations are
  // loaded before either is stored: with one wave per query tile the 128 NKT chunks are two iterations per thread, and a
  // loop with a runtime trip count would serialise their global round trips
  for (int ci0 = tid; ci0 < 16 * NKT * 8; ci0 += 2 * blockDim.x) {
    uint4 kv[2], vv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ci = ci0 + i * blockDim.x, row = ci >> 3, ch = ci & 7;
      kv[i] = vv[i] = make_uint4(0, 0, 0, 0);
      if (row < a.Lk) {
        kv[i] = ld_frag_global(kp, a.ldk, row, ch * 8);
        vv[i] = ld_frag_global(vp, a.ldv, row, ch * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ci = ci0 + i * blockDim.x, row = ci >> 3, ch = ci & 7;
      if (row < 16 * NKT) {
        *reinterpret_cast<uint4*>(s_k + row * LDT + ch * 8) = kv[i];
        *reinterpret_cast<uint4*>(s_v + row * LDT + ch * 8) = vv[i];
      }
    }
  }
  for (int i = tid; i < 16 * NKT; i += blockDim.x) {      // (a one-wave workgroup has fewer threads than keys)
    float m = -INFINITY;
    if (i < a.Lk) m = a.key_mask ? a.key_mask[(size_t)b * a.Lk + i] * (LOG2E * inv_sc2) : 0.f;
    s_mk[i] = m;
  }
  __syncthreads();
  if (qt * 16 >= a.Lq) return;

  // S^T = K Q^T: lane <-> query c, registers <-> keys 16 t + 4 g + r; the mask enters as the C operand
  f32x4 s[NKT];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const float4 m4 = *reinterpret_cast<const float4*>(s_mk + 16 * t + 4 * g);
    s[t] = mfma16(lds_frag_rows(s_k, t, 0, lane), qf[0], (f32x4){m4.x, m4.y, m4.z, m4.w});
    s[t] = mfma16(lds_frag_rows(s_k, t, 1, lane), qf[1], s[t]);
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][r]);
  }
  mx = quad_max(mx);
  const float m2 = (mx == -INFINITY) ? 0.f : mx * sc2;
  float psum = 0.f;
  const uint32_t rbase = attn_row_base(a, b, h, qrow);
  sh_cu64p wq = nullptr;
  if (DROP == 2) wq = (sh_cu64p)(uintptr_t)(a.drop_bits + attn_bits_word(a, bh, qt, 0, 0, 0));
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = fast_exp2(fmaf(s[t][r], sc2, -m2));
      psum += p[r];
    }
    if (DROP == 1) {   // keys 16 t + 4 g .. + 3: two index pairs, one hash each (the element numbering of every kernel)
      const uint32_t pr = (rbase + (uint32_t)(t * 16 + g * 4)) >> 1;
      const uint32_t b0 = bb_pair_bits(a.drop_key, pr), b1 = bb_pair_bits(a.drop_key, pr + 1);
      const bool k0 = bb_keep_lo(b0, a.drop_thr), k1 = bb_keep_hi(b0, a.drop_thr);
      const bool k2 = bb_keep_lo(b1, a.drop_thr), k3 = bb_keep_hi(b1, a.drop_thr);
      p[0] = k0 ? p[0] : 0.f;
      p[1] = k1 ? p[1] : 0.f;
      p[2] = k2 ? p[2] : 0.f;
      p[3] = k3 ? p[3] : 0.f;
      if (a.drop_bits != nullptr && t * 16 < a.Lk) {   // (a tile wholly beyond Lk has no words: NKT rounds the tile count up)
        const unsigned long long w0 = __ballot(k0), w1 = __ballot(k1), w2 = __ballot(k2), w3 = __ballot(k3);
        if (lane == 0) {
          uint64_t* wp = a.drop_bits + attn_bits_word(a, bh, qt, t >> 2, t & 3, 0);
          wp[0] = w0; wp[1] = w1; wp[2] = w2; wp[3] = w3;
        }
      }
    } else if (DROP == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint64_t word = (t * 16 < a.Lk) ? wq[(t >> 2) * 16 + (t & 3) * 4 + r] : 0ull;
        p[r] = __builtin_amdgcn_inverse_ballot_w64(word) ? p[r] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s[t][r] = p[r];
  }
  // O^T = V^T P^T over 32-key chunks (the keep scale 1 / (1 - p) is applied with the normalisation)
  const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = zero;
#pragma unroll
  for (int m = 0; m < NC; ++m) {
    constexpr bool dummy = true;
    (void)dummy;
    if (2 * m + 1 < NKT) {
      const bf16x8 pb = pack_pair(s[2 * m], s[(2 * m + 1 < NKT) ? 2 * m + 1 : 0]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = mfma16(sh_frag_tr<true>(s_v, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane), pb, o[dt]);
    } else {
      const bf16x8 pb = pack_pair(s[2 * m], zero);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = mfma16(sh_frag_tr<false>(s_v, LDT, 32 * m + 4 * g, 0, dt * 16, lane), pb, o[dt]);
    }
  }
  const float l = quad_sum(psum);
  const float inv = (DROP ? a.keep_scale : 1.0f) / l;
  {   // each lane's normalisation factor belongs to ITS query (row c of the tile): scale, then leave as whole rows
    f32x4 on[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) on[dt] = o[dt] * inv;
    bf16_raw* op = (bf16_raw*)a.o + (size_t)b * a.bso + (size_t)(qt * 16) * a.ldo + h * ATTN_D;
    sh_store_tile(s_stage + w * 16 * SH_ST_LD, on, 1.0f, op, a.ldo, a.Lq - qt * 16, lane);
  }
  if (qrow < a.Lq && a.lse && g == 0) a.lse[(size_t)bh * a.Lq + qrow] = (m2 + log2f(l)) * LN2;
}

// =============================================================================================
// Backward, Lq and Lk <= 96: one workgroup of max(NKT, NQT) waves per (batch, head).
// =============================================================================================
template <int NKT, int NQT> struct ShLds {
  static constexpr int NR = NKT > NQT ? NKT : NQT;
  static constexpr int DS_LD = 16 * NQT + 8;                         // dS image row stride (bf16): rows stay 8-byte aligned
  static constexpr int k_off = 0;                                    // [16 NKT][LDT]  K row-major (rows >= Lk zero)
  static constexpr int q_off = k_off + 16 * NKT * LDT * 2;           // [16 NQT][LDT]  Q row-major (rows >= Lq zero)
  static constexpr int do_off = q_off + 16 * NQT * LDT * 2;          // [16 NQT][LDT]  dO
  static constexpr int ds_off = do_off + 16 * NQT * LDT * 2;         // [16 NKT][DS_LD] dS, [key][query]
  static constexpr int lse_off = ds_off + 16 * NKT * DS_LD * 2;      // [16 NQT] float lse (log2 domain; +inf beyond Lq)
  static constexpr int dlt_off = lse_off + 16 * NQT * 4;             // [16 NQT] float delta = rowsum(dO * O)
  static constexpr int mk_off = dlt_off + 16 * NQT * 4;              // [16 NKT] float key mask (log2 domain; -inf beyond Lk)
  static constexpr int bytes = mk_off + 16 * NKT * 4;
};

template <int NKT, int NQT, bool DROP>
__global__ __launch_bounds__(384) void attn_short_bwd_kernel(AttnArgs a) {
  typedef ShLds<NKT, NQT> L;
  constexpr int NR = L::NR, DS_LD = L::DS_LD;
  constexpr int NCQ = (NQT + 1) / 2, NCK = (NKT + 1) / 2;           // 32-query / 32-key chunks
  __shared__ __attribute__((aligned(16))) unsigned char sm[L::bytes];
  bf16_raw* const s_k = reinterpret_cast<bf16_raw*>(sm + L::k_off);
  bf16_raw* const s_q = reinterpret_cast<bf16_raw*>(sm + L::q_off);
  bf16_raw* const s_do = reinterpret_cast<bf16_raw*>(sm + L::do_off);
  bf16_raw* const s_ds = reinterpret_cast<bf16_raw*>(sm + L::ds_off);
  float* const s_lse2 = reinterpret_cast<float*>(sm + L::lse_off);
  float* const s_dlt = reinterpret_cast<float*>(sm + L::dlt_off);
  float* const s_mk = reinterpret_cast<float*>(sm + L::mk_off);
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  // ---- what a key-owner wave needs from global alone: its V rows (B operand of dP = dO V^T) and its keep words
  const int t = w;                                      // key tile (phase 1) and query tile (phase 2) of this wave
  bf16x8 vb[2];
  uint64_t bw[DROP ? NQT : 1];
  if (w < NKT) {
    const int key = t * 16 + c, kr = key < a.Lk ? key : a.Lk - 1;
#pragma unroll
    for (int ks_ = 0; ks_ < 2; ++ks_) vb[ks_] = as_bf16x8(ld_frag_global(vp, a.ldv, kr, ks_ * 32 + g * 8));
    if (DROP) {
#pragma unroll
      for (int j = 0; j < NQT; ++j)
        bw[j] = (j * 16 < a.Lq && t * 16 < a.Lk) ? a.drop_bits[attn_bits_word(a, bh, j, t >> 2, t & 3, c & 3)] : 0ull;
    }
  }
  // ---- staging (all waves): K, Q, dO row-major, delta, lse, key mask.  64 NR threads, 128 NR chunks: exactly two per
  // thread, all eight global loads of a thread in flight before the first LDS store
  {
    uint4 kv[2], qv[2], dv[2], ov[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ci = tid + i * 64 * NR, row = ci >> 3, ch = ci & 7;
      kv[i] = qv[i] = dv[i] = ov[i] = make_uint4(0, 0, 0, 0);
      if (row < a.Lk) kv[i] = ld_frag_global(kp, a.ldk, row, ch * 8);
      if (row < a.Lq) {
        qv[i] = ld_frag_global(qp, a.ldq, row, ch * 8);
        dv[i] = ld_frag_global(dop, a.ldo, row, ch * 8);
        if (!(a.dbg & 2)) ov[i] = ld_frag_global(op, a.ldo, row, ch * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ci = tid + i * 64 * NR, row = ci >> 3, ch = ci & 7;
      if (row < 16 * NKT) *reinterpret_cast<uint4*>(s_k + row * LDT + ch * 8) = kv[i];
      if (row < 16 * NQT) {
        *reinterpret_cast<uint4*>(s_q + row * LDT + ch * 8) = qv[i];
        *reinterpret_cast<uint4*>(s_do + row * LDT + ch * 8) = dv[i];
        const uint32_t dw[4] = {dv[i].x, dv[i].y, dv[i].z, dv[i].w}, ow[4] = {ov[i].x, ov[i].y, ov[i].z, ov[i].w};
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          dsum += __uint_as_float(dw[j] << 16) * __uint_as_float(ow[j] << 16) +
                  __uint_as_float(dw[j] & 0xffff0000u) * __uint_as_float(ow[j] & 0xffff0000u);
        dsum = sh_sum8(dsum);
        if ((lane & 7) == 0) s_dlt[row] = dsum;
      }
    }
  }
  if (tid < 16 * NQT) s_lse2[tid] = tid < a.Lq ? a.lse[(size_t)bh * a.Lq + tid] * LOG2E : INFINITY;
  if (tid < 16 * NKT) {
    float m = -INFINITY;
    if (tid < a.Lk) m = a.key_mask ? a.key_mask[(size_t)b * a.Lk + tid] * LOG2E : 0.f;
    s_mk[tid] = m;
  }
  __syncthreads();

  const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 dk[4], dv[4];          // live across the second barrier: the stores go out behind it (a barrier waits for the
                               // wave's outstanding stores, which would put their latency in front of phase 2)
  if (w < NKT) {
    // =================================================================== phase 1: key tile t -> dK, dV, dS image
    const bf16x8 kb0 = lds_frag_rows(s_k, t, 0, lane), kb1 = lds_frag_rows(s_k, t, 1, lane);
    const float mk2 = s_mk[t * 16 + c];
  #pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = zero;
#pragma unroll
    for (int m = 0; m < NCQ; ++m) {                    // 32-query chunks
      uint2 pp[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)}, pds[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int qt = 2 * m + j;
        if (qt < NQT) {
          f32x4 s = mfma16(lds_frag_rows(s_q, qt, 0, lane), kb0, zero);
          s = mfma16(lds_frag_rows(s_q, qt, 1, lane), kb1, s);
          f32x4 dp = mfma16(lds_frag_rows(s_do, qt, 0, lane), vb[0], zero);
          dp = mfma16(lds_frag_rows(s_do, qt, 1, lane), vb[1], dp);
          const float4 l4 = *reinterpret_cast<const float4*>(s_lse2 + 16 * qt + 4 * g);
          const float4 d4 = *reinterpret_cast<const float4*>(s_dlt + 16 * qt + 4 * g);
          const float lrow[4] = {l4.x, l4.y, l4.z, l4.w}, drow[4] = {d4.x, d4.y, d4.z, d4.w};
          uint32_t kbits = 0xfu;
          if (DROP) kbits = (uint32_t)(bw[qt < NQT ? qt : 0] >> (16 * (c >> 2) + 4 * g));
          f32x4 pd, dsv;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = fast_exp2(fmaf(s[r], sc2, mk2 - lrow[r]));
            const bool keep = !DROP || ((kbits >> r) & 1u);
            pd[r] = keep ? p * ks : 0.f;
            const float dpd = keep ? dp[r] * ks : 0.f;
            dsv[r] = p * (dpd - drow[r]);
          }
          pp[j] = sh_pack4(pd);
          pds[j] = sh_pack4(dsv);
          *reinterpret_cast<uint2*>(s_ds + (16 * t + c) * DS_LD + 16 * qt + 4 * g) = pds[j];
        }
      }
      const bf16x8 pb = sh_join(pp[0], pp[1]), dsb = sh_join(pds[0], pds[1]);
      if (2 * m + 1 < NQT) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = mfma16(sh_frag_tr<true>(s_do, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane), pb, dv[dt]);
          dk[dt] = mfma16(sh_frag_tr<true>(s_q, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane), dsb, dk[dt]);
        }
      } else {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = mfma16(sh_frag_tr<false>(s_do, LDT, 32 * m + 4 * g, 0, dt * 16, lane), pb, dv[dt]);
          dk[dt] = mfma16(sh_frag_tr<false>(s_q, LDT, 32 * m + 4 * g, 0, dt * 16, lane), dsb, dk[dt]);
        }
      }
    }
  }
  __syncthreads();
  // Q and dO are dead from here on (phase 2 reads K and dS): their LDS is the staging space of the output tiles, one
  // [16][SH_ST_LD] image per wave -- where it fits (it does unless the key tiles outnumber the query tiles 2.2 : 1)
  constexpr bool STAGED = NR * 16 * SH_ST_LD * 2 <= 2 * 16 * NQT * LDT * 2;
  bf16_raw* const stage = s_q + w * 16 * SH_ST_LD;
  if (w < NKT) {
    const int key = t * 16 + c;
    if (STAGED) {
      if (t * 16 < a.Lk && !(a.dbg & 1)) {
        bf16_raw* dkp = (bf16_raw*)a.dk + (size_t)b * a.bsk + (size_t)(t * 16) * a.ldk + h * ATTN_D;
        bf16_raw* dvp = (bf16_raw*)a.dv + (size_t)b * a.bsv + (size_t)(t * 16) * a.ldv + h * ATTN_D;
        sh_store_tile(stage, dk, a.scale, dkp, a.ldk, a.Lk - t * 16, lane);
        sh_store_tile(stage, dv, 1.0f, dvp, a.ldv, a.Lk - t * 16, lane);
      }
    } else if (key < a.Lk && !((a.dbg & 1) && lane != 0)) {
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
  if (w < NQT && w * 16 < a.Lq) {
    // =================================================================== phase 2: query tile w -> dQ^T = K^T dS^T
    const int qt = w;
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = zero;
#pragma unroll
    for (int m = 0; m < NCK; ++m) {
      if (2 * m + 1 < NKT) {
        const bf16x8 dsf = sh_frag_tr<true>(s_ds, DS_LD, 32 * m + 4 * g, 32 * m + 16 + 4 * g, 16 * qt, lane);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dq[dt] = mfma16(sh_frag_tr<true>(s_k, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane), dsf, dq[dt]);
      } else {
        const bf16x8 dsf = sh_frag_tr<false>(s_ds, DS_LD, 32 * m + 4 * g, 0, 16 * qt, lane);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dq[dt] = mfma16(sh_frag_tr<false>(s_k, LDT, 32 * m + 4 * g, 0, dt * 16, lane), dsf, dq[dt]);
      }
    }
    const int qrow = qt * 16 + c;
    if (STAGED) {
      if (!(a.dbg & 1)) {
        bf16_raw* dqp = (bf16_raw*)a.dq + (size_t)b * a.bsq + (size_t)(qt * 16) * a.ldq + h * ATTN_D;
        sh_store_tile(stage, dq, a.scale, dqp, a.ldq, a.Lq - qt * 16, lane);
      }
    } else if (qrow < a.Lq && !((a.dbg & 1) && lane != 0)) {
      bf16_raw* dqp = (bf16_raw*)a.dq + (size_t)b * a.bsq + (size_t)qrow * a.ldq + h * ATTN_D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        st4<bf16_raw>(dqp + dt * 16 + g * 4, make_float4(dq[dt][0] * a.scale, dq[dt][1] * a.scale,
                                                          dq[dt][2] * a.scale, dq[dt][3] * a.scale));
    }
  }
}

// =============================================================================================
// Host side
// =============================================================================================
#define SH_MAX_LEN 96

static bool sh_aligned(const AttnArgs& a) {
  return a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.bsq % 8 == 0 && a.bsk % 8 == 0 &&
         a.bsv % 8 == 0 && a.bso % 8 == 0 && ((uintptr_t)a.q % 16) == 0 && ((uintptr_t)a.k % 16) == 0 &&
         ((uintptr_t)a.v % 16) == 0 && ((uintptr_t)a.o % 16) == 0;
}

static int sh_round_tiles(int n) {     // tile counts the kernels are instantiated for
  return n <= 2 ? 2 : n <= 3 ? 3 : n <= 5 ? 5 : 6;
}

// BEVBERT_ATTN_SHORT=0 (read per call): the kernels of rounds 2-5 (A/B measurements, the on-GPU cross-check)
static bool sh_on() {
  const char* v = getenv("BEVBERT_ATTN_SHORT");
  return !(v && v[0] == '0');
}

bool attn_short_fwd_supported(const AttnArgs& a, bool bits_ready) {
  if (!sh_on()) return false;
  return a.bias == nullptr && a.Lk <= SH_MAX_LEN && sh_aligned(a) &&
         !(a.drop_p > 0.f && bits_ready && a.drop_bits == nullptr);
}

template <int NKT>
static void sh_launch_fwd(const AttnArgs& a, dim3 grid, dim3 block, int drop, hipStream_t st) {
  const size_t stage = (size_t)(block.x / 64) * 16 * SH_ST_LD * sizeof(bf16_raw);
  if (drop == 1) hipLaunchKernelGGL((attn_short_fwd_kernel<NKT, 1>), grid, block, stage, st, a);
  else if (drop == 2) hipLaunchKernelGGL((attn_short_fwd_kernel<NKT, 2>), grid, block, stage, st, a);
  else hipLaunchKernelGGL((attn_short_fwd_kernel<NKT, 0>), grid, block, stage, st, a);
}

int attn_short_fwd(const AttnArgs& a_in, bool bits_ready, hipStream_t st) {
  AttnArgs a = a_in;
  const int nqt = (a.Lq + 15) / 16, nkt = sh_round_tiles((a.Lk + 15) / 16);
  // waves (= query tiles) per workgroup: the whole query range when it is short; 7 tiles = 112 queries for long ones
  // (441 BEV cells -> 4 workgroups per (batch, head)).  BEVBERT_SHORT_NW overrides (A/B measurements).
  static const int env_nw = [] { const char* v = getenv("BEVBERT_SHORT_NW"); return v ? atoi(v) : 0; }();
  int nw = nqt <= 6 ? nqt : 7;
  if (env_nw >= 1 && env_nw <= 8) nw = env_nw < nqt ? env_nw : nqt;
  a.nblk = (nqt + nw - 1) / nw;
  const int drop = a.drop_p > 0.f ? (bits_ready ? 2 : 1) : 0;
  const dim3 grid((unsigned)a.nblk * a.nh * a.B), block(64 * nw);
  switch (nkt) {
    case 2: sh_launch_fwd<2>(a, grid, block, drop, st); break;
    case 3: sh_launch_fwd<3>(a, grid, block, drop, st); break;
    case 5: sh_launch_fwd<5>(a, grid, block, drop, st); break;
    default: sh_launch_fwd<6>(a, grid, block, drop, st); break;
  }
  BB_CHECK_LAUNCH("attn_fwd(short)");
  return BB_OK;
}

bool attn_short_bwd_supported(const AttnArgs& a) {
  if (!sh_on()) return false;
  return a.bias == nullptr && a.dbias == nullptr && a.Lk <= SH_MAX_LEN && a.Lq <= SH_MAX_LEN && sh_aligned(a) &&
         (a.drop_p <= 0.f || a.drop_bits != nullptr) && ((uintptr_t)a.dout % 16) == 0 && ((uintptr_t)a.dq % 16) == 0 &&
         ((uintptr_t)a.dk % 16) == 0 && ((uintptr_t)a.dv % 16) == 0;
}

template <int NKT, int NQT>
static void sh_launch_bwd(const AttnArgs& a, dim3 grid, hipStream_t st) {
  const dim3 block(64 * (NKT > NQT ? NKT : NQT));
  if (a.drop_p > 0.f) hipLaunchKernelGGL((attn_short_bwd_kernel<NKT, NQT, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((attn_short_bwd_kernel<NKT, NQT, false>), grid, block, 0, st, a);
}

template <int NKT>
static void sh_launch_bwd_q(const AttnArgs& a, int nqt, dim3 grid, hipStream_t st) {
  switch (nqt) {
    case 2: sh_launch_bwd<NKT, 2>(a, grid, st); break;
    case 3: sh_launch_bwd<NKT, 3>(a, grid, st); break;
    case 5: sh_launch_bwd<NKT, 5>(a, grid, st); break;
    default: sh_launch_bwd<NKT, 6>(a, grid, st); break;
  }
}

int attn_short_bwd(const AttnArgs& a_in, hipStream_t st) {
  AttnArgs a = a_in;
  static const int dbg = [] { const char* v = getenv("BEVBERT_SHORT_DBG"); return v ? atoi(v) : 0; }();
  a.dbg = dbg;
  const int nqt = sh_round_tiles((a.Lq + 15) / 16), nkt = sh_round_tiles((a.Lk + 15) / 16);
  a.nblk = 1;
  const dim3 grid((unsigned)a.nh * a.B);
  switch (nkt) {
    case 2: sh_launch_bwd_q<2>(a, nqt, grid, st); break;
    case 3: sh_launch_bwd_q<3>(a, nqt, grid, st); break;
    case 5: sh_launch_bwd_q<5>(a, nqt, grid, st); break;
    default: sh_launch_bwd_q<6>(a, nqt, grid, st); break;
  }
  BB_CHECK_LAUNCH("attn_bwd(short)");
  return BB_OK;
}
