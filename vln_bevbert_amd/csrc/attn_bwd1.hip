// K2 backward in ONE pass (bf16 in / fp32 accumulate): dQ, dK, dV of a whole (batch, head) from one workgroup.
//
// The two-kernel backward (attn_mfma.hip) recomputes S, P, the dropout mask and dP twice -- once per kernel -- and reads
// Q, K, V, dO twice: 7 tile GEMMs and ~2x the softmax arithmetic for 5 GEMMs' worth of algorithmic work, on kernels that
// are VALU-bound.  Here a workgroup owns ALL keys of one (batch, head) (Lk <= 64 * KT <= 448: the 21x21 BEV has 441 cells):
//
//   * NW waves; wave w owns the 16*KT keys [w*16*KT, (w+1)*16*KT): its V fragments (B operand of dP) and its dK^T / dV^T
//     accumulators [64 d][16*KT keys] stay in registers for the whole kernel.  Default geometry: NW = 4, one wave per
//     SIMD with the whole 512-entry register file (KT = 7: 224 accumulator + 56 operand registers).  An NW = 8 x 64-key
//     form (two waves per SIMD) exists behind BEVBERT_BWD1_WAVES=8; it has to re-read every fragment per key tile to fit
//     256 registers and measured slower;
//   * K is staged once, row-major, in LDS: B operand of S = Q K^T by plain 16-byte reads, and -- through the
//     transposing LDS read of gfx950 (ds_read_b64_tr_b16) -- the K^T A operand of dQ^T = K^T dS^T from the SAME image;
//   * loop over 64-query tiles: S and dP on the matrix cores, then ONE pass of softmax-backward arithmetic per score
//     element (p, dropout from the forward's keep-bit matrix, dS), dK^T += Q^T dS and dV^T += dO^T P with the D->B
//     register hand-over of the two-kernel path (P / dS never leave their lanes); Q^T / dO^T A operands come out of
//     the row-major Q / dO tiles by transposing reads -- no transposed images are ever stored;
//   * dS (bf16) additionally goes to a shared [key][query] LDS image; after a barrier wave w computes
//     dQ^T[16 w .. 16 w + 15][64 queries] over ALL keys (A = K^T, B = dS^T, both by transposing reads) and stores it:
//     dQ needs no atomics, no partial buffers and no second kernel, every operand is read from HBM exactly once;
//   * delta = rowsum(dO * O) is computed while the Q / dO tile is staged.
//
// Work per score element: 5 contractions x 64 MACs on MFMA and ~10 VALU instructions (two-kernel path: 7 and ~60).
#include "attn_mfma_common.h"

#define B1_LDS_DS 68     // row stride (bf16) of the dS image: 64 queries + 4 (rows stay 8-byte aligned for the tr reads)

// dynamic LDS carve (bytes); NK = 64 * NKT keys
template <int NKT> struct B1Lds {
  static constexpr int NK = 64 * NKT;
  static constexpr int q_off = 0;                                  // [64][LDT] bf16   Q tile (row-major)
  static constexpr int do_off = q_off + TK * LDT * 2;              // [64][LDT] bf16   dO tile
  static constexpr int k_off = do_off + TK * LDT * 2;              // [NK][LDT] bf16   K (row-major), whole kernel
  static constexpr int ds_off = k_off + NK * LDT * 2;              // [NK][B1_LDS_DS]  dS of the current query tile
  static constexpr int bits_off = ds_off + NK * B1_LDS_DS * 2;     // [4][KT][16] u64  keep bits of the current query tile
  static constexpr int stat_off = bits_off + 4 * NKT * 16 * 8;      // [2][64] float    lse (log2 domain), delta
  static constexpr int bytes = stat_off + 2 * TK * 4;
};

// KT: 16-key tiles per wave; NW: waves per workgroup (4 or 8); NKT: 64-key tiles of the workgroup (LDS images)
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KMASK: an additive key mask exists (a.key_mask).  Without one the scores need no mask term at all: keys beyond Lk have
// zero K rows in LDS (their dS rows meet zeros in the dQ product) and their dK / dV rows are never stored.
template <int KT, int NW, int NKT, bool BIAS, bool DROP, bool EARLY, bool KMASK>
__global__ __launch_bounds__(64 * NW, (NW == 8 || (KT <= 2 && !BIAS)) ? 2 : 1) void attn_mfma_bwd1_kernel(AttnArgs a) {
  typedef B1Lds<NKT> L;
  constexpr int NT = 64 * NW;             // threads
  constexpr int CPT = 512 / NT;           // 16-byte chunks of a [64][64] bf16 tile per thread (2 or 1)
  constexpr bool HOLD = (NW == 4 && KT >= 4);   // keep the fragments of a half in registers across the key tiles
  //                                             (short key sequences: two workgroups per CU matter more than LDS reads)
  extern __shared__ __attribute__((aligned(16))) unsigned char b1_smem[];
  bf16_raw* const s_q = reinterpret_cast<bf16_raw*>(b1_smem + L::q_off);
  bf16_raw* const s_do = reinterpret_cast<bf16_raw*>(b1_smem + L::do_off);
  bf16_raw* const s_k = reinterpret_cast<bf16_raw*>(b1_smem + L::k_off);
  bf16_raw* const s_ds = reinterpret_cast<bf16_raw*>(b1_smem + L::ds_off);
  uint2* const s_bits = reinterpret_cast<uint2*>(b1_smem + L::bits_off);
  float* const s_lse2 = reinterpret_cast<float*>(b1_smem + L::stat_off);
  float* const s_dlt = s_lse2 + TK;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
  const int bh = blockIdx.x, b = bh / a.nh, h = bh - b * a.nh;
  const bf16_raw* qp = (const bf16_raw*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const bf16_raw* kp = (const bf16_raw*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const bf16_raw* vp = (const bf16_raw*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const bf16_raw* op = (const bf16_raw*)a.o + (size_t)b * a.bso + h * ATTN_D;
  const bf16_raw* dop = (const bf16_raw*)a.dout + (size_t)b * a.bso + h * ATTN_D;
  const float sc2 = a.scale * LOG2E;
  constexpr bool use_bits = DROP;          // with dropout the launcher requires the forward's keep-bit matrix
  const bool bits_hi = (c >> 2) >= 2;      // this lane's bits sit at 16 (c >> 2) + 4 g + r of its 64-bit words
  const int bits_sh = (16 * (c >> 2) + 4 * g) & 31;
  // dropout: with ks = 1 / (1 - p_drop) the kernel works on dS / ks = P (dP keep - delta / ks) and on P keep; the factor ks
  // goes into the three output scalings (dQ, dK, dV), so the keep decision enters as a plain 0.0 / 1.0 factor
  const float inv_ks = DROP ? 1.0f / a.keep_scale : 1.0f;
  const float out_ks = DROP ? a.keep_scale : 1.0f;
  const float dq_scale = a.scale * out_ks;

  // ---- per-wave key state: V fragments, additive key mask (log2 domain; -inf beyond Lk), accumulators
  const int key0 = w * (16 * KT);                  // first key of this wave
  const bool has_keys = key0 < L::NK;              // NW = 8 on 448 keys: the eighth wave only stages tiles and computes dQ
  bf16x8 vf[KT][2];
  float mask2[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = key0 + kt * 16 + c;
    const int r = key < a.Lk ? key : a.Lk - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) vf[kt][ks] = as_bf16x8(ld_frag_global(vp, a.ldv, r, ks * 32 + g * 8));      // (unused rows: clamped)
    mask2[kt] = 0.f;
    if (KMASK || BIAS) mask2[kt] = key < a.Lk ? (a.key_mask ? a.key_mask[(size_t)b * a.Lk + key] * LOG2E : 0.f) : -INFINITY;
  }
  f32x4 dkacc[KT][4], dvacc[KT][4];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dkacc[kt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dvacc[kt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  // ---- staging of one 64-query tile: Q, dO rows -> LDS; delta = rowsum(dO * O); lse; keep bits
  // thread t owns 16-byte chunks ch = t and t + 256: row ch >> 3, dims 8 (ch & 7) .. +7; a row is covered by 8 lanes
  uint4 qreg[CPT], doreg[CPT], oreg[CPT];
  float lreg = 0.f;
  uint2 breg[CPT];
  auto tile_issue = [&](int q0) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int ch = tid + i * NT, row = ch >> 3, d0 = (ch & 7) * 8;
      qreg[i] = doreg[i] = oreg[i] = make_uint4(0, 0, 0, 0);           // rows past the end are zero filled
      if (q0 + row < a.Lq) {
        qreg[i] = ld_frag_global(qp, a.ldq, q0 + row, d0);
        doreg[i] = ld_frag_global(dop, a.ldo, q0 + row, d0);
        oreg[i] = ld_frag_global(op, a.ldo, q0 + row, d0);
      }
    }
    lreg = INFINITY;                            // padding rows: p = exp2(-inf) = 0
    if (tid < TK && q0 + tid < a.Lq) lreg = a.lse[((size_t)b * a.nh + h) * a.Lq + q0 + tid] * LOG2E;
    if (use_bits) {                             // words of (4 query-16-tiles) x (nk64 key tiles) x 16, 8 bytes each
      const int per_q16 = a.nk64 * 16;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int wi = tid + i * NT, q16l = wi / per_q16, rest = wi - q16l * per_q16;
        breg[i] = make_uint2(0u, 0u);
        if (q16l < 4 && (q0 >> 4) + q16l < a.nq16)
          breg[i] = *reinterpret_cast<const uint2*>(a.drop_bits + ((size_t)bh * a.nq16 + (q0 >> 4) + q16l) * per_q16 + rest);
      }
    }
  };
  auto tile_commit = [&]() {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int ch = tid + i * NT, row = ch >> 3, d0 = (ch & 7) * 8;
      *reinterpret_cast<uint4*>(s_q + row * LDT + d0) = qreg[i];
      *reinterpret_cast<uint4*>(s_do + row * LDT + d0) = doreg[i];
      const uint32_t dw[4] = {doreg[i].x, doreg[i].y, doreg[i].z, doreg[i].w};
      const uint32_t ow[4] = {oreg[i].x, oreg[i].y, oreg[i].z, oreg[i].w};
      float dsum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        dsum += __uint_as_float(dw[j] << 16) * __uint_as_float(ow[j] << 16) +
                __uint_as_float(dw[j] & 0xffff0000u) * __uint_as_float(ow[j] & 0xffff0000u);
      dsum += __shfl_xor(dsum, 1, 64);
      dsum += __shfl_xor(dsum, 2, 64);
      dsum += __shfl_xor(dsum, 4, 64);
      if ((tid & 7) == 0) s_dlt[row] = dsum * inv_ks;
    }
    if (tid < TK) s_lse2[tid] = lreg;
    if (use_bits) {
      const int per_q16 = a.nk64 * 16;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int wi = tid + i * NT, q16l = wi / per_q16, rest = wi - q16l * per_q16;
        if (q16l < 4) s_bits[q16l * (NKT * 16) + rest] = breg[i];
      }
    }
  };
  // ---- prologue: the first query tile's loads and ALL of K go out together (one memory round trip, not two);
  //      K -> LDS row-major, zero rows beyond Lk
  tile_issue(0);
  {
    constexpr int KCH = L::NK * 8 / NT;         // 16-byte chunks of K per thread
    uint4 kbuf[KCH];
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int c16 = tid + i * NT, row = c16 >> 3, ch = c16 & 7;
      kbuf[i] = make_uint4(0, 0, 0, 0);
      if (row < a.Lk) kbuf[i] = ld_frag_global(kp, a.ldk, row, ch * 8);
    }
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int c16 = tid + i * NT, row = c16 >> 3, ch = c16 & 7;
      *reinterpret_cast<uint4*>(s_k + row * LDT + ch * 8) = kbuf[i];
    }
  }
  tile_commit();
  __syncthreads();

  for (int q0 = 0; q0 < a.Lq; q0 += TK) {
    // ================= phase 1: S, dP, softmax backward, dK^T / dV^T, dS -> LDS =================
    const bool more = q0 + TK < a.Lq;
#pragma unroll 1
    for (int m = 0; m < (has_keys ? 2 : 0); ++m) {       // halves of 32 queries: query tiles t = 2m, 2m + 1
      // EARLY: the next tile's global loads go out under this tile's arithmetic (KT = 7: under its second half, the
      // registers are full before) instead of under the short dQ phase
      if (EARLY && m == (KT == 7 ? 1 : 0) && more) tile_issue(q0 + TK);
      bf16x8 qa[2][2], da[2][2], qtf[HOLD ? 4 : 1], dotf[HOLD ? 4 : 1];
      float lv[2][4], ndl[2][4];
      auto load_rows = [&]() {          // A operands of S / dP (rows of this half's two query tiles) and their statistics
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * m + tt;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            qa[tt][ks] = lds_frag_rows(s_q, t, ks, lane);
            da[tt][ks] = lds_frag_rows(s_do, t, ks, lane);
          }
          const float4 l4 = *reinterpret_cast<const float4*>(&s_lse2[t * 16 + g * 4]);
          const float4 d4 = *reinterpret_cast<const float4*>(&s_dlt[t * 16 + g * 4]);
          lv[tt][0] = -l4.x; lv[tt][1] = -l4.y; lv[tt][2] = -l4.z; lv[tt][3] = -l4.w;      // -lse (log2 domain)
          ndl[tt][0] = -d4.x; ndl[tt][1] = -d4.y; ndl[tt][2] = -d4.z; ndl[tt][3] = -d4.w;
        }
      };
      if (HOLD) load_rows();
      // A operands of the two "contract over queries" products: rows d, k-slots (g, j) <-> query 32 m + 16 (j >> 2) + 4 g + (j & 3)
      if (HOLD) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          qtf[dt] = lds_frag_tr(s_q, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane);
          dotf[dt] = lds_frag_tr(s_do, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane);
        }
      }
      // K fragments (B operand of S) and keep-bit words of a key tile are fetched from LDS one tile AHEAD, behind the
      // matrix instructions of the current tile and in front of its ~100 VALU instructions: with one wave per SIMD an
      // LDS round trip placed right before its first use is fully exposed (57 % of the wave cycles of the first version
      // of this kernel were s_waitcnt time, profiles/r02a_pmc_attn_sq_counters.txt)
      auto bits_of = [&](int t, int ktg) -> uint2 {
        // word (q16 = t, k64 = ktg >> 2, t' = ktg & 3, r' = c & 3), bits 16 (c >> 2) + 4 g + r
        return s_bits[((t * NKT + (ktg >> 2)) * 4 + (ktg & 3)) * 4 + (c & 3)];
      };
      bf16x8 kf0 = lds_frag_rows(s_k, w * KT, 0, lane), kf1 = lds_frag_rows(s_k, w * KT, 1, lane);
      uint2 wdc[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
      if (DROP) { wdc[0] = bits_of(2 * m, w * KT); wdc[1] = bits_of(2 * m + 1, w * KT); }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int ktg = w * KT + kt;                 // 16-key tile index within the (batch, head)
        if (!HOLD) load_rows();                      // two waves per SIMD: re-read per key tile, keep the registers free
        f32x4 sacc[2], dpacc[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          sacc[tt] = mfma16(qa[tt][0], kf0, (f32x4){0.f, 0.f, 0.f, 0.f});
          sacc[tt] = mfma16(qa[tt][1], kf1, sacc[tt]);
          dpacc[tt] = mfma16(da[tt][0], vf[kt][0], (f32x4){0.f, 0.f, 0.f, 0.f});
          dpacc[tt] = mfma16(da[tt][1], vf[kt][1], dpacc[tt]);
        }
        const uint2 wd0 = wdc[0], wd1 = wdc[1];
        if (kt + 1 < KT) {
          kf0 = lds_frag_rows(s_k, ktg + 1, 0, lane);
          kf1 = lds_frag_rows(s_k, ktg + 1, 1, lane);
          if (DROP) { wdc[0] = bits_of(2 * m, ktg + 1); wdc[1] = bits_of(2 * m + 1, ktg + 1); }
          __builtin_amdgcn_sched_barrier(0);         // keep the prefetch HERE (the scheduler would sink it to its use)
        }
        // lane (key = key0 + 16 kt + c) holds S[q = q0 + 16 t + 4 g + r][key], r = 0..3.  The MFMA results are wanted in
        // VGPRs (the VALU cannot read the accumulation registers; left alone the allocator parks them there and copies
        // every element out again), and the arithmetic runs on pairs (v_pk_fma_f32 / v_pk_mul_f32).
        asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]), "+v"(dpacc[0]), "+v"(dpacc[1]));
        const float mk = mask2[kt];
        const f32x2 sc2v = {sc2, sc2}, mkv = {mk, mk};
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * m + tt;
          uint32_t spread = 0x01010101u;
          if (DROP) {     // keep bits of the forward (prefetched above): bit r of the nibble -> byte r (0 / 1)
            const uint2 wd = tt == 0 ? wd0 : wd1;
            const uint32_t nib = ((bits_hi ? wd.y : wd.x) >> bits_sh) & 0xfu;
            spread = (nib * 0x00204081u) & 0x01010101u;
          }
          float keepf[4] = {1.f, 1.f, 1.f, 1.f};
          if (DROP) {
            // (written as instructions: from the C expression the compiler re-derives each byte from the nibble with a
            //  bit-field extract + convert, two operations per element instead of one)
            asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(keepf[0]) : "v"(spread));
            asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(keepf[1]) : "v"(spread));
            asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(keepf[2]) : "v"(spread));
            asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(keepf[3]) : "v"(spread));
          }
#pragma unroll
          for (int rp = 0; rp < 2; ++rp) {
            const int r = 2 * rp;
            f32x2 s2 = {sacc[tt][r], sacc[tt][r + 1]};
            const f32x2 nl2 = {lv[tt][r], lv[tt][r + 1]};
            f32x2 e2;
            if (KMASK || BIAS) e2 = s2 * sc2v + mkv;
            if (BIAS) {
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const int qi = q0 + t * 16 + g * 4 + r + j, key = key0 + kt * 16 + c;
                if (qi < a.Lq && key < a.Lk) e2[j] += a.bias[((size_t)b * a.Lq + qi) * a.Lk + key] * LOG2E;
              }
            }
            if (KMASK || BIAS) e2 = e2 + nl2;
            else e2 = s2 * sc2v + nl2;
            const f32x2 p2 = {fast_exp2(e2[0]), fast_exp2(e2[1])};
            const f32x2 dp2 = {dpacc[tt][r], dpacc[tt][r + 1]};
            const f32x2 nd2 = {ndl[tt][r], ndl[tt][r + 1]};
            f32x2 u2, pd2;
            if (DROP) {     // keep: u = dp - delta / ks, pd = p;  dropped: u = -delta / ks, pd = 0   (ks: output scalings)
              const f32x2 k2 = {keepf[r], keepf[r + 1]};
              u2 = dp2 * k2 + nd2;
              pd2 = p2 * k2;
            } else {
              u2 = dp2 + nd2;
              pd2 = p2;
            }
            const f32x2 ds2 = p2 * u2;
            sacc[tt][r] = ds2[0]; sacc[tt][r + 1] = ds2[1];
            dpacc[tt][r] = pd2[0]; dpacc[tt][r + 1] = pd2[1];
            if (BIAS) {
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const int qi = q0 + t * 16 + g * 4 + r + j, key = key0 + kt * 16 + c;
                if (a.dbias && qi < a.Lq && key < a.Lk)        // this (batch, head)'s own slice: a plain store
                  a.dbias[(((size_t)b * a.nh + h) * a.Lq + qi) * a.Lk + key] = ds2[j] * out_ks;
              }
            }
          }
        }
        const bf16x8 dsb = pack_pair(sacc[0], sacc[1]), pdb = pack_pair(dpacc[0], dpacc[1]);
        {   // dS image [key][query]: this lane's 4 consecutive queries of tile t at row key  (2 x 8 bytes)
          const uint4 dsu = __builtin_bit_cast(uint4, dsb);
          bf16_raw* row = s_ds + (key0 + kt * 16 + c) * B1_LDS_DS + 32 * m + 4 * g;
          *reinterpret_cast<uint2*>(row) = make_uint2(dsu.x, dsu.y);
          *reinterpret_cast<uint2*>(row + 16) = make_uint2(dsu.z, dsu.w);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          if (HOLD) {
            dkacc[kt][dt] = mfma16(qtf[dt], dsb, dkacc[kt][dt]);
            dvacc[kt][dt] = mfma16(dotf[dt], pdb, dvacc[kt][dt]);
          } else {      // two resident waves per SIMD hide the extra transposing reads; the registers are what is scarce
            const bf16x8 qt1 = lds_frag_tr(s_q, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane);
            const bf16x8 dot1 = lds_frag_tr(s_do, LDT, 32 * m + 4 * g, 32 * m + 16 + 4 * g, dt * 16, lane);
            dkacc[kt][dt] = mfma16(qt1, dsb, dkacc[kt][dt]);
            dvacc[kt][dt] = mfma16(dot1, pdb, dvacc[kt][dt]);
          }
        }
      }
    }
    __syncthreads();   // dS image complete; nobody reads s_q / s_do / stats / bits of this tile any more

    // ================= phase 2: next tile's loads in flight, dQ^T = K^T dS^T for d rows 16 w .. 16 w + 15 =================
    if (more && !(EARLY && has_keys)) tile_issue(q0 + TK);
    // NW = 4: wave w -> d rows 16 w .. +15, all four query tiles;  NW = 8: d rows 16 (w & 3), query tiles 2 (w >> 2), +1
    constexpr int NQT = (NW == 4) ? 4 : 2;
    const int dq_d0 = 16 * (w & 3), dq_qt0 = (NW == 4) ? 0 : 2 * (w >> 2);
    f32x4 dqacc[NQT];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) dqacc[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int ks = 0; ks < 2 * NKT; ++ks) {         // k-steps of 32 keys
      const bf16x8 ka = lds_frag_tr(s_k, LDT, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, dq_d0, lane);
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) {
        const bf16x8 dsf = lds_frag_tr(s_ds, B1_LDS_DS, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * (dq_qt0 + qt), lane);
        dqacc[qt] = mfma16(ka, dsf, dqacc[qt]);
      }
    }
    // The next tile goes to LDS BEFORE this tile's dQ is stored: vmcnt counts stores as well as loads on gfx950, so a
    // commit placed behind the (scattered, 8-byte) dQ stores would wait for their write acknowledgements too -- this
    // way the stores drain underneath the next tile's phase 1.
    if (more) tile_commit();
    // lane (query = q0 + 16 qt + c) holds dQ^T[d = dq_d0 + 4 g + r][query]
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      const int qi = q0 + (dq_qt0 + qt) * 16 + c;
      if (qi < a.Lq) {
        bf16_raw* dqp = (bf16_raw*)a.dq + (size_t)b * a.bsq + (size_t)qi * a.ldq + h * ATTN_D + dq_d0 + 4 * g;
        st4<bf16_raw>(dqp, make_float4(dqacc[qt][0] * dq_scale, dqacc[qt][1] * dq_scale, dqacc[qt][2] * dq_scale,
                                       dqacc[qt][3] * dq_scale));
      }
    }
    __syncthreads();   // next tile staged; dS image free again
  }

  // ---- epilogue: dK = scale * dK^T, dV = dV^T / (1 - p_drop)
  const float vs = out_ks;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = key0 + kt * 16 + c;
    if (has_keys && key < a.Lk) {
      bf16_raw* dkp = (bf16_raw*)a.dk + (size_t)b * a.bsk + (size_t)key * a.ldk + h * ATTN_D;
      bf16_raw* dvp = (bf16_raw*)a.dv + (size_t)b * a.bsv + (size_t)key * a.ldv + h * ATTN_D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        st4<bf16_raw>(dkp + dt * 16 + g * 4,
                      make_float4(dkacc[kt][dt][0] * dq_scale, dkacc[kt][dt][1] * dq_scale, dkacc[kt][dt][2] * dq_scale,
                                  dkacc[kt][dt][3] * dq_scale));
        st4<bf16_raw>(dvp + dt * 16 + g * 4, make_float4(dvacc[kt][dt][0] * vs, dvacc[kt][dt][1] * vs,
                                                         dvacc[kt][dt][2] * vs, dvacc[kt][dt][3] * vs));
      }
    }
  }
}

// =============================================================================================
// launcher
// =============================================================================================
template <int KT, int NW, int NKT, bool B_, bool D_, bool E_, bool M_>
static int launch_bwd1(const AttnArgs& a, hipStream_t st) {
  typedef B1Lds<NKT> L;
  static const bool ok =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_bwd1_kernel<KT, NW, NKT, B_, D_, E_, M_>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, L::bytes) == hipSuccess;
  BB_REQUIRE(ok, "attention bwd (single pass): cannot raise the dynamic LDS limit to %d bytes", L::bytes);
  hipLaunchKernelGGL((attn_mfma_bwd1_kernel<KT, NW, NKT, B_, D_, E_, M_>), dim3((unsigned)a.B * a.nh), dim3(64 * NW),
                     L::bytes, st, a);
  BB_CHECK_LAUNCH("attn_bwd(single pass)");
  return BB_OK;
}

template <int KT, int NW, int NKT>
static int dispatch_bwd1(const AttnArgs& a, hipStream_t st) {
  const bool hb = a.bias != nullptr, hd = a.drop_p > 0.f, km = a.key_mask != nullptr;
  // BEVBERT_BWD1_EARLY=0: issue the next query tile's loads only in the dQ phase (A/B knob for the 448-key form)
  static const bool late = [] { const char* v = getenv("BEVBERT_BWD1_EARLY"); return v && v[0] == '0'; }();
  if constexpr (NKT <= 2) {      // the additive graph bias only occurs on the global map (a few dozen nodes)
    if (hb && hd) return launch_bwd1<KT, NW, NKT, true, true, true, true>(a, st);
    if (hb) return launch_bwd1<KT, NW, NKT, true, false, true, true>(a, st);
  }
  if constexpr (KT == 7) {
    if (late) {
      if (hd) return launch_bwd1<KT, NW, NKT, false, true, false, true>(a, st);
      return launch_bwd1<KT, NW, NKT, false, false, false, true>(a, st);
    }
  }
  if (hd) return km ? launch_bwd1<KT, NW, NKT, false, true, true, true>(a, st)
                    : launch_bwd1<KT, NW, NKT, false, true, true, false>(a, st);
  return km ? launch_bwd1<KT, NW, NKT, false, false, true, true>(a, st)
            : launch_bwd1<KT, NW, NKT, false, false, true, false>(a, st);
}

// The single-pass kernel covers Lk <= 448 (a bias: Lk <= 128) and, with dropout, needs the forward's keep-bit matrix;
// everything else stays on the two-kernel path (attn_mfma.hip).
bool attn_mfma_bwd1_supported(const AttnArgs& a) {
  return a.Lk <= (a.bias ? 128 : 448) && (a.drop_p <= 0.f || a.drop_bits != nullptr);
}

int attn_mfma_bwd1(const AttnArgs& a, hipStream_t st) {
  BB_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.bsq % 8 == 0 && a.bsk % 8 == 0 &&
                 a.bsv % 8 == 0 && a.bso % 8 == 0 && ((uintptr_t)a.q % 16) == 0 && ((uintptr_t)a.k % 16) == 0 &&
                 ((uintptr_t)a.v % 16) == 0 && ((uintptr_t)a.o % 16) == 0 && ((uintptr_t)a.dout % 16) == 0 &&
                 ((uintptr_t)a.dq % 16) == 0 && ((uintptr_t)a.dk % 16) == 0 && ((uintptr_t)a.dv % 16) == 0,
             "attention bwd (MFMA path): pointers must be 16-byte aligned and strides multiples of 8 elements");
  // BEVBERT_BWD1_WAVES=8: 8 waves x 64 keys (two waves per SIMD, no fragments held in registers) instead of 4 x 112.
  // Measured on the BEV shape (B = 64, 441 x 441, p = 0.1, same box): 303 us vs 262 us -- the re-read fragments cost more
  // LDS time than the second wave hides, so one wave per SIMD is the default.
  static const bool eight = [] { const char* v = getenv("BEVBERT_BWD1_WAVES"); return v && v[0] == '8'; }();
  if (a.Lk <= 64) return dispatch_bwd1<1, 4, 1>(a, st);
  // 65..128 keys (the 80 text tokens): BEVBERT_BWD1_KEYS128=8 -> eight waves of 16 keys instead of four of 32
  static const bool k128_8 = [] { const char* v = getenv("BEVBERT_BWD1_KEYS128"); return v && v[0] == '8'; }();
  if (a.Lk <= 128) return k128_8 ? dispatch_bwd1<1, 8, 2>(a, st) : dispatch_bwd1<2, 4, 2>(a, st);
  if (a.Lk <= 256) return dispatch_bwd1<4, 4, 4>(a, st);
  if (eight) return dispatch_bwd1<4, 8, 7>(a, st);
  return dispatch_bwd1<7, 4, 7>(a, st);
}
