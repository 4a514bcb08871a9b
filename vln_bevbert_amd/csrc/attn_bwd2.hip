// K2 backward in one pass, second generation (bf16 in / fp32 accumulate): 7 key waves + 1 dQ wave per (batch, head).
//
// The round-2 single-pass kernel (attn_bwd1.hip) gives each of 4 waves 112 keys: 224 accumulator registers per wave,
// one wave per SIMD, every dependent chain (matrix result -> softmax arithmetic -> matrix operand) and every LDS round
// trip fully exposed, and two workgroup barriers around a serial dQ phase per query tile: 17 % of the MFMA peak, 37 %
// of the wave time parked (profiles/r02f_pmc_attn_sq_counters.txt).  Here a workgroup has EIGHT waves, two per SIMD:
//
//   * waves 0..6 ("key waves") own 64 keys each (4 tiles of 16): dK^T / dV^T accumulators [64 d][64 keys] = 128
//     registers, V fragments 32 registers; everything else streams from LDS just in time, so a wave stays at <= 256
//     registers and two waves share a SIMD;
//   * the loop runs over 32-query steps.  Per step a key wave computes S and dP against its keys (Q / dO rows are read
//     once per step and reused over the four key tiles), the softmax backward on the accumulators, dK^T += Q^T dS,
//     dV^T += dO^T P (P / dS never leave their lanes), and writes dS (bf16) into a [key][query] LDS image;
//   * wave 7 ("dQ wave") owns no keys.  During step k it (1) starts the global loads of query tile k+1, (2) computes
//     dQ^T[64 d][32 q] of step k-1 over ALL keys from the previous dS image (A = K^T, B = dS^T, both by transposing LDS
//     reads) and stores it, (3) writes tile k+1 (Q, dO rows, delta = rowsum(dO o O), lse) to LDS.  The dS image and the
//     Q / dO tiles are double buffered, so ONE barrier per step is all the synchronisation there is, and the dQ
//     contraction (1/5 of the flops) runs beside the key waves' vector arithmetic instead of after it;
//   * dropout: keep bits in the lanes <-> keys layout written by attn_drop_bits_kernel, fetched with scalar loads and
//     applied with ONE v_cndmask per element (dS / ks = P keep dP - P delta / ks, ks folded into the output scalings);
//     without dropout -delta enters as the C operand of the dP product and dS = P * dP' is a single multiply;
//   * without an additive key mask the scores need no mask term at all (keys beyond Lk are zero rows of the K image).
//
// Covers 256 < Lk <= 448 without a per-element bias (the BEV encoder's 441 cells); everything else stays on attn_bwd1.
#include "attn_bwd7p1.h"

// BEVBERT_B2_TRACE=1 (bench_attn_shape.py passes a scratch buffer as dbias): workgroup 0 stamps s_memtime per wave, step
// and phase into it -- the phase timeline of profiles/r03*_bwd2_timeline*.txt.  Compiled out otherwise.
#define B2_STAMP(step, slot)                                                                         \
  do {                                                                                               \
    if (TRACE && bh == 0 && lane == 0 && (step) < 16)                                                \
      reinterpret_cast<unsigned long long*>(a.dbias)[(w * 16 + (step)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

template <bool DROP, bool KMASK, bool TRACE>
__global__ __launch_bounds__(512, 2) void attn_bwd2_kernel(AttnArgs a) {
  typedef B2Lds L;
  extern __shared__ __attribute__((aligned(16))) unsigned char b2_smem[];
  bf16_raw* const s_k = reinterpret_cast<bf16_raw*>(b2_smem + L::k_off);
  bf16_raw* const s_ds = reinterpret_cast<bf16_raw*>(b2_smem + L::ds_off);
  bf16_raw* const s_q = reinterpret_cast<bf16_raw*>(b2_smem + L::q_off);
  bf16_raw* const s_do = reinterpret_cast<bf16_raw*>(b2_smem + L::do_off);
  float* const s_stat = reinterpret_cast<float*>(b2_smem + L::stat_off);

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
  const bool is_dq_wave = (w == B2_NKEYW);

  // ---- prologue, all waves: K -> LDS row-major, zero rows beyond Lk
  {
    constexpr int KCH = B2_NK * 8 / 512;          // 16-byte chunks of K per thread (7)
    uint4 kbuf[KCH];
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int c16 = tid + i * 512, row = c16 >> 3, ch = c16 & 7;
      kbuf[i] = make_uint4(0, 0, 0, 0);
      if (row < a.Lk) kbuf[i] = ld_frag_global(kp, a.ldk, row, ch * 8);
    }
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int c16 = tid + i * 512, row = c16 >> 3, ch = c16 & 7;
      *reinterpret_cast<uint4*>(s_k + row * LDT + ch * 8) = kbuf[i];
    }
  }

  if (is_dq_wave) {
    // =====================================================================================================
    // dQ wave: staging of the 32-query tiles and dQ^T = K^T dS^T one step behind the key waves
    // =====================================================================================================
    // lane owns 16-byte chunks ch = lane + 64 i (i = 0..3): row ch >> 3, dims 8 (ch & 7) .. +7; a row is covered by 8 lanes
    uint4 qreg[4], doreg[4], oreg[4];
    uint32_t okm[4];
    float lreg = INFINITY;
    const int row0 = lane >> 3, dcol = (lane & 7) * 8;      // chunk i of this lane: row row0 + 8 i, dims dcol .. dcol + 7
    const bf16_raw* qlane = qp + dcol;
    const bf16_raw* dolane = dop + dcol;
    const bf16_raw* olane = op + dcol;
    const int ldq = (int)a.ldq, ldo = (int)a.ldo;           // 32-bit element offsets: a (batch, head) slice is far below 2^31 elements
    auto tile_issue = [&](int q0) __attribute__((always_inline)) {      // loads only: nothing here may consume a loaded value (that would wait for all of them)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int row = q0 + row0 + 8 * i;
        const bool ok = row < a.Lq;
        row = ok ? row : a.Lq - 1;                          // rows past the end: clamped load, zeroed below
        okm[i] = ok ? 0xffffffffu : 0u;                     // applied when the tile is committed (touching a loaded value
        qreg[i] = *reinterpret_cast<const uint4*>(qlane + row * ldq);      // here would wait for the loads right away)
        doreg[i] = *reinterpret_cast<const uint4*>(dolane + row * ldo);
        oreg[i] = *reinterpret_cast<const uint4*>(olane + row * ldo);
      }
      lreg = INFINITY;                            // padding rows: p = exp2(-inf) = 0
      if (lane < 32 && q0 + lane < a.Lq) lreg = a.lse[((size_t)b * a.nh + h) * a.Lq + q0 + lane];
    };
    auto tile_commit = [&](int buf) __attribute__((always_inline)) {
      bf16_raw* tq = s_q + buf * (32 * LDT);
      bf16_raw* tdo = s_do + buf * (32 * LDT);
      float* st = s_stat + buf * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = lane + i * 64, row = ch >> 3, d0 = (ch & 7) * 8;
        const uint32_t m = okm[i];                          // rows past the end are zero filled
        const uint4 qv = make_uint4(qreg[i].x & m, qreg[i].y & m, qreg[i].z & m, qreg[i].w & m);
        const uint4 dv = make_uint4(doreg[i].x & m, doreg[i].y & m, doreg[i].z & m, doreg[i].w & m);
        *reinterpret_cast<uint4*>(tq + row * LDT + d0) = qv;
        *reinterpret_cast<uint4*>(tdo + row * LDT + d0) = dv;
        const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
        const uint32_t ow[4] = {oreg[i].x, oreg[i].y, oreg[i].z, oreg[i].w};
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          dsum += __uint_as_float(dw[j] << 16) * __uint_as_float(ow[j] << 16) +
                  __uint_as_float(dw[j] & 0xffff0000u) * __uint_as_float(ow[j] & 0xffff0000u);
        dsum = sum8(dsum);
        if ((lane & 7) == 0) st[32 + row] = dsum * inv_ks;
      }
      if (lane < 32) st[lane] = lreg * LOG2E;
    };
    // dQ^T[64 d][32 q] += K^T dS^T over all 448 key rows of the images (rows beyond Lk are zero in both), 14 k-steps of
    // 32 keys.  One wave has nobody to hide its LDS round trips behind and gets a fraction of the LDS issue rate
    // (r03e timeline: 168 transposing reads per step = 4 300 cycles), so (1) the K^T fragments of the upper seven k-steps
    // live in registers for the whole kernel (112 VGPRs: the wave owns no accumulators), (2) the reads of k-step ks + 1
    // are issued before the matrix instructions of k-step ks (explicit two-deep software pipeline).
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
    auto dq_step = [&](int q0, int buf) __attribute__((always_inline)) {
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

    tile_issue(0);
    tile_commit(0);
    __syncthreads();                                   // K image and tile 0 visible
#pragma unroll
    for (int i = 0; i < NRES; ++i)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int ks = NKS - NRES + i;
        kres[i][dt] = lds_frag_tr(s_k, LDT, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * dt, lane);
      }
    for (int k = 0; k < nsteps; ++k) {
      const bool more = k + 1 < nsteps;
      B2_STAMP(k, 0);
      if (more) tile_issue(32 * (k + 1));
      B2_STAMP(k, 1);
      if (k > 0) dq_step(32 * (k - 1), (k - 1) & 1);
      B2_STAMP(k, 2);
      if (more) tile_commit((k + 1) & 1);
      B2_STAMP(k, 3);
      __syncthreads();                                 // closes step k
      B2_STAMP(k, 4);
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
  // keep-bit words travel one key tile ahead of their use (every word is read once: each scalar load is a cache miss
  // with an L2 / HBM round trip; issued right in front of its first use that latency was fully exposed, 1 100 cycles per
  // key tile in the r03h timeline)
  uint64_t bw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bw[i] = (DROP && has_keys) ? wbits[(i >> 2) * 16 + (i & 3)] : 0;
  __syncthreads();                                       // K image and tile 0 visible
  for (int k = 0; k < nsteps; ++k) {
    const int buf = k & 1;
    const bf16_raw* tq = s_q + buf * (32 * LDT);
    const bf16_raw* tdo = s_do + buf * (32 * LDT);
    const float* st = s_stat + buf * 64;
    bf16_raw* img = s_ds + buf * (B2_NK * B2_LDS_DS);
    B2_STAMP(k, 0);
    if (has_keys) {
      // A operands of S / dP: the rows of both query tiles, read once per step
      bf16x8 qa[2][2], da[2][2];
      float nl[2][4];
      f32x4 ndl[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          qa[tt][ks] = lds_frag_rows(tq, tt, ks, lane);
          da[tt][ks] = lds_frag_rows(tdo, tt, ks, lane);
        }
        const float4 l4 = *reinterpret_cast<const float4*>(&st[tt * 16 + g * 4]);
        const float4 d4 = *reinterpret_cast<const float4*>(&st[32 + tt * 16 + g * 4]);
        nl[tt][0] = -l4.x; nl[tt][1] = -l4.y; nl[tt][2] = -l4.z; nl[tt][3] = -l4.w;      // -lse (log2 domain)
        ndl[tt] = (f32x4){-d4.x, -d4.y, -d4.z, -d4.w};                                     // -delta / ks
      }
      B2_STAMP(k, 1);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const bf16x8 vf0 = vf[kt][0], vf1 = vf[kt][1];
        uint64_t bwn[8];
        if (DROP) {     // next key tile's words (tile 0 of the next step after tile 3; the workspace has nq16 / 2 >= nsteps blocks)
          const int kn = kt == 3 ? (k + 1 < nsteps ? k + 1 : k) : k, ktn = (kt + 1) & 3;
#pragma unroll
          for (int i = 0; i < 8; ++i) bwn[i] = wbits[(size_t)kn * bits_step + (i >> 2) * 16 + ktn * 4 + (i & 3)];
          __builtin_amdgcn_sched_barrier(0);      // the loads stay HERE, ahead of this tile's arithmetic
        }
        const bf16x8 kf0 = lds_frag_rows(s_k, w * 4 + kt, 0, lane), kf1 = lds_frag_rows(s_k, w * 4 + kt, 1, lane);
        // lane (key = key0 + 16 kt + c) holds S[q = 32 k + 16 tt + 4 g + r][key], r = 0..3; the two query tiles are two
        // independent chains (matrix -> exp -> select -> pack) in flight together
        f32x4 sacc[2], dpacc[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          sacc[tt] = mfma16(qa[tt][0], kf0, (f32x4){0.f, 0.f, 0.f, 0.f});
          sacc[tt] = mfma16(qa[tt][1], kf1, sacc[tt]);
          dpacc[tt] = mfma16(da[tt][0], vf0, DROP ? (f32x4){0.f, 0.f, 0.f, 0.f} : ndl[tt]);
          dpacc[tt] = mfma16(da[tt][1], vf1, dpacc[tt]);
        }
        if (kt == 0) B2_STAMP(k, 5);
        uint2 dsu[2], pdu[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          float dsv[4], pdv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float off = KMASK ? mask2[kt] + nl[tt][r] : nl[tt][r];
            const float p = fast_exp2(fmaf(sacc[tt][r], sc2, off));
            if (DROP) {
              const float pd = keep_select(p, bw[tt * 4 + r]);
              dsv[r] = fmaf(pd, dpacc[tt][r], p * ndl[tt][r]);            // P keep dP - P delta / ks
              pdv[r] = pd;
            } else {
              dsv[r] = p * dpacc[tt][r];                                  // dP' = dP - delta came out of the matrix unit
              pdv[r] = p;
            }
          }
          dsu[tt] = make_uint2(pack_bf16x2(dsv[0], dsv[1]), pack_bf16x2(dsv[2], dsv[3]));
          pdu[tt] = make_uint2(pack_bf16x2(pdv[0], pdv[1]), pack_bf16x2(pdv[2], pdv[3]));
          // dS image [key][query]: this lane's 4 consecutive queries of tile tt at row key
          *reinterpret_cast<uint2*>(img + (key0 + kt * 16 + c) * B2_LDS_DS + 16 * tt + 4 * g) = dsu[tt];
        }
        if (kt == 0) B2_STAMP(k, 6);
        // B operands of the "contract over queries" products: k-slot (g, j) <-> query 16 (j >> 2) + 4 g + (j & 3)
        const bf16x8 dsb = as_bf16x8(make_uint4(dsu[0].x, dsu[0].y, dsu[1].x, dsu[1].y));
        const bf16x8 pdb = as_bf16x8(make_uint4(pdu[0].x, pdu[0].y, pdu[1].x, pdu[1].y));
        // dK^T += Q^T dS, dV^T += dO^T P right away (A operands: rows d, k-slots = the 32 queries, by transposing reads):
        // these eight matrix instructions run under the next key tile's dependent chain
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 qtf = lds_frag_tr(tq, LDT, 4 * g, 16 + 4 * g, dt * 16, lane);
          const bf16x8 dotf = lds_frag_tr(tdo, LDT, 4 * g, 16 + 4 * g, dt * 16, lane);
          dkacc[kt][dt] = mfma16(qtf, dsb, dkacc[kt][dt]);
          dvacc[kt][dt] = mfma16(dotf, pdb, dvacc[kt][dt]);
        }
        if (DROP) {
#pragma unroll
          for (int i = 0; i < 8; ++i) bw[i] = bwn[i];
        }
        if (kt == 0) B2_STAMP(k, 7);
        if (kt == 1) B2_STAMP(k, 2);
      }
    }
    B2_STAMP(k, 3);
    __syncthreads();                                     // closes step k
    B2_STAMP(k, 4);
  }

  // ---- epilogue: dK = scale * ks * dK^T, dV = ks * dV^T
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int key = key0 + kt * 16 + c;
    if (key < a.Lk) {
      bf16_raw* dkp = (bf16_raw*)a.dk + (size_t)b * a.bsk + (size_t)key * a.ldk + h * ATTN_D;
      bf16_raw* dvp = (bf16_raw*)a.dv + (size_t)b * a.bsv + (size_t)key * a.ldv + h * ATTN_D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        st4<bf16_raw>(dkp + dt * 16 + g * 4,
                      make_float4(dkacc[kt][dt][0] * dq_scale, dkacc[kt][dt][1] * dq_scale, dkacc[kt][dt][2] * dq_scale,
                                  dkacc[kt][dt][3] * dq_scale));
        st4<bf16_raw>(dvp + dt * 16 + g * 4, make_float4(dvacc[kt][dt][0] * out_ks, dvacc[kt][dt][1] * out_ks,
                                                         dvacc[kt][dt][2] * out_ks, dvacc[kt][dt][3] * out_ks));
      }
    }
  }
}

// =============================================================================================
// launcher
// =============================================================================================
template <bool D_, bool M_, bool T_ = false>
static int launch_bwd2(const AttnArgs& a, hipStream_t st) {
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd2_kernel<D_, M_, T_>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, B2Lds::bytes) == hipSuccess;
  BB_REQUIRE(ok, "attention bwd (7+1 waves): cannot raise the dynamic LDS limit to %d bytes", B2Lds::bytes);
  hipLaunchKernelGGL((attn_bwd2_kernel<D_, M_, T_>), dim3((unsigned)a.B * a.nh), dim3(512), B2Lds::bytes, st, a);
  BB_CHECK_LAUNCH("attn_bwd(7+1 waves)");
  return BB_OK;
}

bool attn_bwd2_supported(const AttnArgs& a) {
  return a.bias == nullptr && a.Lk > 256 && a.Lk <= B2_NK && (a.drop_p <= 0.f || a.drop_bits_b != nullptr);
}

int attn_bwd2(const AttnArgs& a, hipStream_t st) {
  BB_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.bsq % 8 == 0 && a.bsk % 8 == 0 &&
                 a.bsv % 8 == 0 && a.bso % 8 == 0 && ((uintptr_t)a.q % 16) == 0 && ((uintptr_t)a.k % 16) == 0 &&
                 ((uintptr_t)a.v % 16) == 0 && ((uintptr_t)a.o % 16) == 0 && ((uintptr_t)a.dout % 16) == 0 &&
                 ((uintptr_t)a.dq % 16) == 0 && ((uintptr_t)a.dk % 16) == 0 && ((uintptr_t)a.dv % 16) == 0,
             "attention bwd (MFMA path): pointers must be 16-byte aligned and strides multiples of 8 elements");
  const bool hd = a.drop_p > 0.f, km = a.key_mask != nullptr;
  static const bool trace = [] { const char* v = getenv("BEVBERT_B2_TRACE"); return v && v[0] == '1'; }();
  if (trace && a.dbias != nullptr && hd && !km) return launch_bwd2<true, false, true>(a, st);
  if (hd) return km ? launch_bwd2<true, true>(a, st) : launch_bwd2<true, false>(a, st);
  return km ? launch_bwd2<false, true>(a, st) : launch_bwd2<false, false>(a, st);
}
