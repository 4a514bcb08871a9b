// K2 backward in one pass, third generation (bf16 in / fp32 accumulate): the 7 + 1-wave workgroup of attn_bwd2.hip
// (7 key waves own 64 keys each, dK^T / dV^T in registers; wave 7 stages the 32-query tiles and contracts dQ^T = K^T dS^T
// one step behind from the dS image) with its step re-cut after the measurements of round 5 (profiles/r05_attention_ablations_and_arms.txt, r05a_bwd2_phase_timeline_round3_kernel.txt):
//
//   * the round-3 loop spent 54 % of its wave cycles in s_waitcnt: per key tile it re-read the Q^T / dO^T fragments of the
//     dK / dV products from LDS (64 transposing reads per step) right in front of their use.  Here a step has two
//     phases: PHASE 1 walks the 8 chains (16-query tile, key tile) S / dP -> softmax backward -> packed P, dS as a
//     two-deep software pipeline (the matrix instructions of chain i + 1 in front of the vector arithmetic of chain i),
//     PHASE 2 issues the step's 32 dK / dV matrix instructions back to back, d-tile outermost: each Q^T / dO^T fragment is
//     read ONCE per step (16 transposing reads instead of 64; 48 LDS instructions per wave and step instead of 88);
//   * keep-bit words: the 32 words of a step in 64 scalar registers, requested after the step's last LDS store has landed
//     and right in front of the barrier (scalar loads share lgkmcnt with LDS and return out of order: any LDS wait
//     behind an outstanding scalar load waits for it -- the round-3 loop paid that round trip four times per step);
//   * prologue: every wave issues ALL its first loads (K chunks, V fragments / tile 0) before anything waits;
//   * epilogue: dK / dV leave through LDS as whole 128-byte rows, 16 bytes per lane (the accumulator layout gives sixteen
//     32-byte pieces per store instruction; with one workgroup per CU nothing overlaps that tail);
//   * dQ wave: delta = rowsum(dO o O) on v_dot2c_f32_bf16, zero fill only on the partial tile.
//
// What was measured and NOT kept (same file history, profiles/r05_attention_ablations_and_arms.txt): running phase 2 of waves
// 4 .. 6 one step late under their SIMD partner's phase 1 (+3 % time; every ablation of this kernel removes about its part's
// own issue time -- the parts add up -- although the chip can hide matrix instructions behind the other wave's plain vector
// instructions: scripts/probes/mfma_valu_overlap.hip.  Its packed fp32 lines (soft_bwd) are one suspect: packed fp32 does not
// overlap with v_mfma; the scalar form of those lines spilled 20 - 60 registers at the 256-register limit); staging the
// query tiles from waves 0 .. 3 with wave 7 touching the lines into L2 (the 13 extra live registers made the compiler
// spill the V fragments and the staged chunks: 2 x slower).
//
// Covers what attn_bwd2.hip covers (no per-element bias, 256 < Lk <= 448); BEVBERT_ATTN_BWD3=0 selects the round-3 kernel.
#include "attn_bwd7p1.h"

// ABL (diagnostics, BEVBERT_B3_ABL; results are WRONG with any bit set): 1 = no dK / dV products, 2 = no dQ products,
// 4 = no softmax arithmetic, 8 = no tile staging after tile 0
template <bool DROP, bool KMASK, int ABL>
__global__ __launch_bounds__(512, 2) void attn_bwd3_kernel(AttnArgs a) {
  typedef B2Lds L;
  extern __shared__ __attribute__((aligned(16))) unsigned char b3_smem[];
  bf16_raw* const s_k = reinterpret_cast<bf16_raw*>(b3_smem + L::k_off);
  bf16_raw* const s_ds = reinterpret_cast<bf16_raw*>(b3_smem + L::ds_off);
  bf16_raw* const s_q = reinterpret_cast<bf16_raw*>(b3_smem + L::q_off);
  bf16_raw* const s_do = reinterpret_cast<bf16_raw*>(b3_smem + L::do_off);
  float* const s_stat = reinterpret_cast<float*>(b3_smem + L::stat_off);

  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x, b = bh / a.nh, h = bh - b * a.nh;
  const bf16_raw* qp = (const bf16_raw*)a.q + (size_t)b * a.bsq + h * ATTN_D;
  const bf16_raw* kp = (const bf16_raw*)a.k + (size_t)b * a.bsk + h * ATTN_D;
  const bf16_raw* vp = (const bf16_raw*)a.v + (size_t)b * a.bsv + h * ATTN_D;
  const bf16_raw* op = (const bf16_raw*)a.o + (size_t)b * a.bso + h * ATTN_D;
  const bf16_raw* dop = (const bf16_raw*)a.dout + (size_t)b * a.bso + h * ATTN_D;
  const float sc2 = a.scale * LOG2E;
  const float inv_ks = DROP ? 1.0f / a.keep_scale : 1.0f;
  const float out_ks = DROP ? a.keep_scale : 1.0f;
  const float dq_scale = a.scale * out_ks;
  const int nsteps = (a.Lq + 31) >> 5;
  const int ldq = (int)a.ldq, ldo = (int)a.ldo;   // 32-bit element offsets: a (batch, head) slice is far below 2^31 elements

  // ---- prologue, all waves: the K chunks stay in registers until each wave has issued its other first loads as well
  constexpr int KCH = B2_NK * 8 / 512;            // 16-byte chunks of K per thread (7)
  uint4 kbuf[KCH];
#pragma unroll
  for (int i = 0; i < KCH; ++i) {
    const int c16 = tid + i * 512, row = c16 >> 3, ch = c16 & 7;
    kbuf[i] = make_uint4(0, 0, 0, 0);
    if (row < a.Lk) kbuf[i] = ld_frag_global(kp, a.ldk, row, ch * 8);
  }
  auto k_to_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int c16 = tid + i * 512, row = c16 >> 3, ch = c16 & 7;
      *reinterpret_cast<uint4*>(s_k + row * LDT + ch * 8) = kbuf[i];
    }
  };

  if (w == B2_NKEYW) {
    // =====================================================================================================
    // dQ wave: staging of the 32-query tiles and dQ^T = K^T dS^T one step behind the key waves
    // =====================================================================================================
    __builtin_amdgcn_s_setprio(3);                 // one wave, 1/5 of the matrix work, on a SIMD it shares with a key wave
    // dQ^T[64 d][32 q] += K^T dS^T over all 448 key rows of the images (rows beyond Lk are zero in both), 14 k-steps of
    // 32 keys.  A lone wave issues an 8-byte LDS read every ~10 cycles at best: the K^T fragments of the upper NRES
    // k-steps live in registers for the whole kernel, the reads of k-step ks + 1 are issued before the matrix
    // instructions of k-step ks (two-deep software pipeline).
    constexpr int NKS = 2 * B2_NKEYW, NRES = 3;
    struct DqFrags { bf16x8 ka[4], d0, d1; };
    auto dq_load = [&](DqFrags& f, const bf16_raw* img, int ks) __attribute__((always_inline)) {
      f.d0 = lds_frag_tr(img, B2_LDS_DS, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 0, lane);
      f.d1 = lds_frag_tr(img, B2_LDS_DS, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16, lane);
      if (ks < NKS - NRES) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) f.ka[dt] = lds_frag_tr(s_k, LDT, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * dt, lane);
      }
    };
    bf16x8 kres[NRES][4];
    auto dq_mma = [&](f32x4 (&acc)[2][4], const DqFrags& f, int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 ka = ks < NKS - NRES ? f.ka[dt] : kres[ks - (NKS - NRES) < 0 ? 0 : ks - (NKS - NRES)][dt];
        acc[0][dt] = mfma16(ka, f.d0, acc[0][dt]);
        acc[1][dt] = mfma16(ka, f.d1, acc[1][dt]);
      }
    };
    // staging: lane owns 16-byte chunks ch = lane + 64 i (i = 0..3): row ch >> 3, dims 8 (ch & 7) .. + 7; a row is covered
    // by 8 neighbouring lanes.  (Threading these pieces through dq_step's k-steps -- loads on the first four, commits on
    // the last four -- measured 13 % SLOWER, r05l: the branches cut its software pipeline into ten basic blocks.)
    uint4 qreg[4], doreg[4], oreg[4];
    float lreg = INFINITY;
    const int row0 = lane >> 3, dcol = (lane & 7) * 8;
    auto issue_piece = [&](int i, int q0) __attribute__((always_inline)) {      // loads only
      int row = q0 + row0 + 8 * i;
      row = row < a.Lq ? row : a.Lq - 1;                    // rows past the end: clamped load, zeroed when committed
      qreg[i] = *reinterpret_cast<const uint4*>(qp + dcol + row * ldq);
      doreg[i] = *reinterpret_cast<const uint4*>(dop + dcol + row * ldo);
      oreg[i] = *reinterpret_cast<const uint4*>(op + dcol + row * ldo);
      if (i == 3) {
        lreg = INFINITY;                          // padding rows: p = exp2(-inf) = 0
        if (lane < 32 && q0 + lane < a.Lq) lreg = a.lse[((size_t)b * a.nh + h) * a.Lq + q0 + lane];
      }
    };
    auto commit_piece = [&](int i, int buf, int q0) __attribute__((always_inline)) {
      bf16_raw* tq = s_q + buf * (32 * LDT);
      bf16_raw* tdo = s_do + buf * (32 * LDT);
      float* st = s_stat + buf * 64;
      const int ch = lane + i * 64, row = ch >> 3, d0 = (ch & 7) * 8;
      uint4 qv = qreg[i], dv = doreg[i];
      if (q0 + 32 > a.Lq) {                                 // only the last tile has rows past the end (wave-uniform): zero fill
        const uint32_t m = q0 + row < a.Lq ? 0xffffffffu : 0u;
        qv = make_uint4(qv.x & m, qv.y & m, qv.z & m, qv.w & m);
        dv = make_uint4(dv.x & m, dv.y & m, dv.z & m, dv.w & m);
      }
      *reinterpret_cast<uint4*>(tq + row * LDT + d0) = qv;
      *reinterpret_cast<uint4*>(tdo + row * LDT + d0) = dv;
      // delta = rowsum(dO o O): v_dot2c_f32_bf16 multiplies two packed pairs and accumulates in fp32 (products of bf16
      // values are exact in fp32); the 8 lanes of a row are neighbours: one DPP sum
      float dsum = dot2_bf16(dv.x, oreg[i].x, 0.f);
      dsum = dot2_bf16(dv.y, oreg[i].y, dsum);
      dsum = dot2_bf16(dv.z, oreg[i].z, dsum);
      dsum = dot2_bf16(dv.w, oreg[i].w, dsum);
      dsum = sum8(dsum);
      if ((lane & 7) == 0) st[32 + row] = dsum * inv_ks;
      if (i == 3 && lane < 32) st[lane] = lreg * LOG2E;
    };
    auto dq_step = [&](int q0, int buf) __attribute__((always_inline)) {
      if (ABL & 2) return;
      const bf16_raw* img = s_ds + buf * (B2_NK * B2_LDS_DS);
      f32x4 dqacc[2][4];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dqacc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      DqFrags fa, fb;
      dq_load(fa, img, 0);
#pragma unroll
      for (int ks = 0; ks < NKS; ks += 2) {
        dq_load(fb, img, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        dq_mma(dqacc, fa, ks);
        if (ks + 2 < NKS) dq_load(fa, img, ks + 2);
        __builtin_amdgcn_sched_barrier(0);
        dq_mma(dqacc, fb, ks + 1);
      }
      // lane (query = q0 + 16 qt + c) holds dQ^T[d = 16 dt + 4 g + r][query]
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int qi = q0 + qt * 16 + c;
        if (qi < a.Lq) {
          bf16_raw* dqp = (bf16_raw*)a.dq + (size_t)b * a.bsq + (size_t)qi * a.ldq + h * ATTN_D + 4 * g;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            st4<bf16_raw>(dqp + 16 * dt, make_float4(dqacc[qt][dt][0] * dq_scale, dqacc[qt][dt][1] * dq_scale,
                                                     dqacc[qt][dt][2] * dq_scale, dqacc[qt][dt][3] * dq_scale));
        }
      }
    };

#pragma unroll
    for (int i = 0; i < 4; ++i) issue_piece(i, 0);
    k_to_lds();
#pragma unroll
    for (int i = 0; i < 4; ++i) commit_piece(i, 0, 0);
    __syncthreads();                                   // K image and tile 0 visible
#pragma unroll
    for (int i = 0; i < NRES; ++i)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int ks = NKS - NRES + i;
        kres[i][dt] = lds_frag_tr(s_k, LDT, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * dt, lane);
      }
    for (int k = 0; k < nsteps; ++k) {
      const bool more = k + 1 < nsteps && !(ABL & 8);
      if (more) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_piece(i, 32 * (k + 1));
      }
      if (k > 0) dq_step(32 * (k - 1), (k - 1) & 1);
      if (more) {
#pragma unroll
        for (int i = 0; i < 4; ++i) commit_piece(i, (k + 1) & 1, 32 * (k + 1));
      }
      __syncthreads();                                 // closes step k
    }
    dq_step(32 * (nsteps - 1), (nsteps - 1) & 1);
    return;
  }

  // =====================================================================================================
  // key waves
  // =====================================================================================================
  const int key0 = w * 64;
  const bool has_keys = key0 < a.Lk;
  float mask2[4];
  bf16x8 vf[4][2];             // V fragments (B operand of dP), resident
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int key = key0 + kt * 16 + c;
    const int r = key < a.Lk ? key : a.Lk - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) vf[kt][ks] = as_bf16x8(ld_frag_global(vp, a.ldv, r, ks * 32 + g * 8));   // (unused rows: clamped)
    mask2[kt] = 0.f;
    if (KMASK) mask2[kt] = key < a.Lk ? a.key_mask[(size_t)b * a.Lk + key] * LOG2E : -INFINITY;
  }
  k_to_lds();

  f32x4 dkacc[4][4], dvacc[4][4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dkacc[kt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dvacc[kt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  // keep bits, lanes <-> keys: 32 words (tt, kt, r) per (32-query block, 64-key tile of one wave), see attn_common.h
  bb_cu64p wbits = nullptr;
  const size_t bits_step = (size_t)a.nk64 * 32;         // words per 32-query block
  if (DROP) wbits = (bb_cu64p)(uintptr_t)(a.drop_bits_b + (size_t)bh * (a.nq16 >> 1) * bits_step + (size_t)w * 32);
  if (!has_keys) {      // no keys (Lk <= 384): this wave's rows of both dS images stay zero for the dQ contraction
    for (int i = lane; i < 64 * B2_LDS_DS / 4; i += 64) {
      reinterpret_cast<uint2*>(s_ds + key0 * B2_LDS_DS)[i] = make_uint2(0u, 0u);
      reinterpret_cast<uint2*>(s_ds + (B2_NK + key0) * B2_LDS_DS)[i] = make_uint2(0u, 0u);
    }
  }
  uint64_t bw[32];
  bb_cu64p wnext = wbits;
#pragma unroll
  for (int i = 0; i < 32; ++i) bw[i] = (DROP && has_keys) ? wnext[i] : 0;

  // softmax backward of one chain (key tile x 16-query tile): 4 score elements per lane, P / dS stay in their lanes
  auto soft_bwd = [&](const f32x4& sacc, const f32x4& dpacc, const float (&nl)[4], const f32x4& ndl, float mk,
                      const uint64_t* bwords, uint2& dsu, uint2& pdu) __attribute__((always_inline)) {
    if (ABL & 4) {
      dsu = make_uint2(__float_as_uint(sacc[0]), __float_as_uint(dpacc[1]));
      pdu = make_uint2(__float_as_uint(sacc[2]), __float_as_uint(dpacc[3]));
      return;
    }
    // Two elements per instruction where the operation allows (v_pk_fma_f32 / v_pk_mul_f32: the accumulator quad is two
    // register pairs): scale-and-shift, P delta, and the final multiply-add -- 1.5 instead of 3 instruction slots per
    // element; the exponential and the keep select stay per element.
    uint32_t dsw[2], pdw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bb_f32x2 s2 = {sacc[2 * j], sacc[2 * j + 1]}, d2 = {dpacc[2 * j], dpacc[2 * j + 1]};
      const bb_f32x2 o2 = {KMASK ? mk + nl[2 * j] : nl[2 * j], KMASK ? mk + nl[2 * j + 1] : nl[2 * j + 1]};
      const bb_f32x2 t2 = s2 * sc2 + o2;
      const bb_f32x2 p2 = {fast_exp2(t2[0]), fast_exp2(t2[1])};
      bb_f32x2 ds2, pd2;
      if (DROP) {
        pd2 = (bb_f32x2){keep_select(p2[0], bwords[2 * j]), keep_select(p2[1], bwords[2 * j + 1])};
        const bb_f32x2 n2 = {ndl[2 * j], ndl[2 * j + 1]};
        ds2 = pd2 * d2 + p2 * n2;                           // P keep dP - P delta / ks
      } else {
        pd2 = p2;
        ds2 = p2 * d2;                                      // dP' = dP - delta came out of the matrix unit
      }
      dsw[j] = pack_bf16x2(ds2[0], ds2[1]);
      pdw[j] = pack_bf16x2(pd2[0], pd2[1]);
    }
    dsu = make_uint2(dsw[0], dsw[1]);
    pdu = make_uint2(pdw[0], pdw[1]);
  };
  __syncthreads();                                       // K image and tile 0 visible

  for (int k = 0; k < nsteps; ++k) {
    const int buf = k & 1;
    const bf16_raw* tq = s_q + buf * (32 * LDT);
    const bf16_raw* tdo = s_do + buf * (32 * LDT);
    const float* st = s_stat + buf * 64;
    bf16_raw* img = s_ds + buf * (B2_NK * B2_LDS_DS);
    if (has_keys) {
      uint2 dsu[4][2], pdu[4][2];
      // ---- phase 1: two-deep software pipeline over the chains i = (tt, kt); a scheduling barrier after each chain keeps
      //      the compiler from hoisting every chain's LDS reads to the top (that version needed 290 registers)
      bf16x8 qa0, qa1, da0, da1, qb0, qb1, db0, db1;
      float nl[4], nlb[4];
      f32x4 ndl, ndlb;
      auto load_rows = [&](int tt, bf16x8& q0, bf16x8& q1, bf16x8& d0, bf16x8& d1, float (&l)[4], f32x4& dl)
                           __attribute__((always_inline)) {
        q0 = lds_frag_rows(tq, tt, 0, lane); q1 = lds_frag_rows(tq, tt, 1, lane);
        d0 = lds_frag_rows(tdo, tt, 0, lane); d1 = lds_frag_rows(tdo, tt, 1, lane);
        const float4 l4 = *reinterpret_cast<const float4*>(&st[tt * 16 + g * 4]);
        const float4 d4 = *reinterpret_cast<const float4*>(&st[32 + tt * 16 + g * 4]);
        l[0] = -l4.x; l[1] = -l4.y; l[2] = -l4.z; l[3] = -l4.w;               // -lse (log2 domain)
        dl = (f32x4){-d4.x, -d4.y, -d4.z, -d4.w};                             // -delta / ks
      };
      auto chain_mma = [&](int kt, const bf16x8& q0, const bf16x8& q1, const bf16x8& d0, const bf16x8& d1, const f32x4& dl,
                           f32x4& sacc, f32x4& dpacc) __attribute__((always_inline)) {
        const bf16x8 kf0 = lds_frag_rows(s_k, w * 4 + kt, 0, lane), kf1 = lds_frag_rows(s_k, w * 4 + kt, 1, lane);
        sacc = mfma16(q0, kf0, (f32x4){0.f, 0.f, 0.f, 0.f});
        sacc = mfma16(q1, kf1, sacc);
        dpacc = mfma16(d0, vf[kt][0], DROP ? (f32x4){0.f, 0.f, 0.f, 0.f} : dl);
        dpacc = mfma16(d1, vf[kt][1], dpacc);
      };
      load_rows(0, qa0, qa1, da0, da1, nl, ndl);
      f32x4 sa, dpa, sb, dpb;
      chain_mma(0, qa0, qa1, da0, da1, ndl, sa, dpa);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int tt = i >> 2, kt = i & 3;
        if (i == 3) load_rows(1, qb0, qb1, db0, db1, nlb, ndlb);
        if (i < 7) {               // chain i + 1's matrix instructions
          const int ktn = (i + 1) & 3;
          if (i + 1 < 4) chain_mma(ktn, qa0, qa1, da0, da1, ndl, (i & 1) ? sa : sb, (i & 1) ? dpa : dpb);
          else chain_mma(ktn, qb0, qb1, db0, db1, ndlb, (i & 1) ? sa : sb, (i & 1) ? dpa : dpb);
        }
        const uint64_t* bcur = bw + tt * 16 + kt * 4;      // chain i's vector arithmetic
        if (tt == 0) soft_bwd((i & 1) ? sb : sa, (i & 1) ? dpb : dpa, nl, ndl, mask2[kt], bcur, dsu[kt][tt], pdu[kt][tt]);
        else soft_bwd((i & 1) ? sb : sa, (i & 1) ? dpb : dpa, nlb, ndlb, mask2[kt], bcur, dsu[kt][tt], pdu[kt][tt]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- phase 2: dK^T += Q^T dS, dV^T += dO^T P, d-tile outermost; A operands rows d, k-slots = the 32 queries (g, j) <->
      //      query 16 (j >> 2) + 4 g + (j & 3), fragments read one d-tile ahead.  Program order: first d-tile's fragments,
      //      THEN the dS image (transposing reads may not be hoisted over LDS stores the compiler cannot prove disjoint)
      bf16x8 fq = lds_frag_tr(tq, LDT, 4 * g, 16 + 4 * g, 0, lane), fdo = lds_frag_tr(tdo, LDT, 4 * g, 16 + 4 * g, 0, lane);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          *reinterpret_cast<uint2*>(img + (key0 + kt * 16 + c) * B2_LDS_DS + 16 * tt + 4 * g) = dsu[kt][tt];
#pragma unroll
      for (int dt = 0; dt < ((ABL & 1) ? 0 : 4); ++dt) {
        bf16x8 fqn = fq, fdon = fdo;
        if (dt < 3) {
          fqn = lds_frag_tr(tq, LDT, 4 * g, 16 + 4 * g, (dt + 1) * 16, lane);
          fdon = lds_frag_tr(tdo, LDT, 4 * g, 16 + 4 * g, (dt + 1) * 16, lane);
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const bf16x8 dsb = as_bf16x8(make_uint4(dsu[kt][0].x, dsu[kt][0].y, dsu[kt][1].x, dsu[kt][1].y));
          const bf16x8 pdb = as_bf16x8(make_uint4(pdu[kt][0].x, pdu[kt][0].y, pdu[kt][1].x, pdu[kt][1].y));
          dkacc[kt][dt] = mfma16(fq, dsb, dkacc[kt][dt]);
          dvacc[kt][dt] = mfma16(fdo, pdb, dvacc[kt][dt]);
        }
        fq = fqn;
        fdo = fdon;
      }
    }
    // ---- closes step k: this wave's LDS stores have landed (the pointer operand ties the scalar loads below to this
    //      point: loads from the constant address space would otherwise be free to move up), the next step's keep-bit
    //      words are requested, then the bare barrier -- an acquire fence would wait for the scalar loads first
    __builtin_amdgcn_sched_barrier(0);
    if (DROP && has_keys) {
      wnext = wbits + (size_t)(k + 1 < nsteps ? k + 1 : k) * bits_step;
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wnext) : : "memory");
#pragma unroll
      for (int i = 0; i < 32; ++i) bw[i] = wnext[i];
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" : : : "memory");
  }

  // ---- epilogue: dK = scale * ks * dK^T, dV = ks * dV^T as whole 128-byte rows.  Each wave transposes through LDS it
  //      owns: its 64 rows of the dS buffer the dQ wave is NOT reading during its last contraction (buffer nsteps & 1,
  //      last read before the final barrier) hold exactly a [32 keys][72] bf16 tile; four passes (dK / dV x key-tile
  //      pair), each 8 ds_write_b64 + 4 ds_read_b128 + 4 global_store_dwordx4.  A wave's LDS operations execute in order.
  if (has_keys) {
    bf16_raw* stg = s_ds + (nsteps & 1) * (B2_NK * B2_LDS_DS) + key0 * B2_LDS_DS;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int hp = pass & 1;
      const bool is_dk = pass < 2;
      const float sc = is_dk ? dq_scale : out_ks;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const f32x4 v = is_dk ? dkacc[2 * hp + kk][dt] : dvacc[2 * hp + kk][dt];
          *reinterpret_cast<uint2*>(stg + (kk * 16 + c) * 72 + dt * 16 + g * 4) =
              make_uint2(pack_bf16x2(v[0] * sc, v[1] * sc), pack_bf16x2(v[2] * sc, v[3] * sc));
        }
      bf16_raw* dst = (bf16_raw*)(is_dk ? a.dk : a.dv) + (size_t)b * (is_dk ? a.bsk : a.bsv) + h * ATTN_D;
      const int64_t ld = is_dk ? a.ldk : a.ldv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ch = lane + 64 * j, row = ch >> 3, d0 = (ch & 7) * 8;
        const uint4 val = *reinterpret_cast<const uint4*>(stg + row * 72 + d0);
        const int key = key0 + hp * 32 + row;
        if (key < a.Lk) *reinterpret_cast<uint4*>(dst + (size_t)key * ld + d0) = val;
      }
    }
  }
}

// =============================================================================================
// launcher
// =============================================================================================
template <bool D_, bool M_, int A_ = 0>
static int launch_bwd3(const AttnArgs& a, hipStream_t st) {
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd3_kernel<D_, M_, A_>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, B2Lds::bytes) == hipSuccess;
  BB_REQUIRE(ok, "attention bwd (7+1 waves, gen 3): cannot raise the dynamic LDS limit to %d bytes", B2Lds::bytes);
  hipLaunchKernelGGL((attn_bwd3_kernel<D_, M_, A_>), dim3((unsigned)a.B * a.nh), dim3(512), B2Lds::bytes, st, a);
  BB_CHECK_LAUNCH("attn_bwd(7+1 waves, gen 3)");
  return BB_OK;
}

bool attn_bwd3_supported(const AttnArgs& a) {
  return a.bias == nullptr && a.Lk > 256 && a.Lk <= B2_NK && (a.drop_p <= 0.f || a.drop_bits_b != nullptr);
}

int attn_bwd3(const AttnArgs& a, hipStream_t st) {
  BB_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.bsq % 8 == 0 && a.bsk % 8 == 0 &&
                 a.bsv % 8 == 0 && a.bso % 8 == 0 && ((uintptr_t)a.q % 16) == 0 && ((uintptr_t)a.k % 16) == 0 &&
                 ((uintptr_t)a.v % 16) == 0 && ((uintptr_t)a.o % 16) == 0 && ((uintptr_t)a.dout % 16) == 0 &&
                 ((uintptr_t)a.dq % 16) == 0 && ((uintptr_t)a.dk % 16) == 0 && ((uintptr_t)a.dv % 16) == 0,
             "attention bwd (MFMA path): pointers must be 16-byte aligned and strides multiples of 8 elements");
  const bool hd = a.drop_p > 0.f, km = a.key_mask != nullptr;
  static const int abl = [] { const char* v = getenv("BEVBERT_B3_ABL"); return v ? atoi(v) : 0; }();
  if (abl && hd && !km) {
    switch (abl) {
      case 1: return launch_bwd3<true, false, 1>(a, st);
      case 2: return launch_bwd3<true, false, 2>(a, st);
      case 4: return launch_bwd3<true, false, 4>(a, st);
      case 8: return launch_bwd3<true, false, 8>(a, st);
      case 10: return launch_bwd3<true, false, 10>(a, st);
      case 11: return launch_bwd3<true, false, 11>(a, st);
      case 15: return launch_bwd3<true, false, 15>(a, st);
      default: break;
    }
  }
  if (hd) return km ? launch_bwd3<true, true>(a, st) : launch_bwd3<true, false>(a, st);
  return km ? launch_bwd3<false, true>(a, st) : launch_bwd3<false, false>(a, st);
}
