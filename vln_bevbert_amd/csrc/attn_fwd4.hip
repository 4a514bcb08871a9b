// K2 forward, third generation (bf16 in / fp32 accumulate), for the long self-attention of the path: no graph bias,
// 256 < Lk <= 448 keys, more than 256 queries (the 21 x 21 BEV: 441 x 441).  Same arithmetic as attn_fwd3_kernel
// (transposed formulation S^T = K Q^T, O^T = V^T P^T, lazy running maximum, mask as the C operand), different shape.
//
// What the shape answers (measured on the MI355X, this round):
//   * scripts/probes/mfma_valu_overlap.hip (profiles/r05_mfma_valu_overlap_probe.txt): two waves of a gfx950 SIMD that both
//     interleave matrix and PLAIN vector instructions hide the matrix instructions almost completely (8 v_mfma + 40 vector
//     instructions: 234 - 247 cycles against 205 - 225 for the vector instructions alone), while packed fp32 instructions
//     (v_pk_fma_f32 ...) do not mix with v_mfma at all (worse than one after the other).  The ceiling of this kernel is its vector
//     work, ~280 cycles per (16 queries x 32 keys); its loop runs at ~380 (the dependent chains inside a unit, the LDS reads), and
//     what a kernel can win beyond that is everything that is NOT arithmetic.  This file is compiled without the SLP vectoriser
//     and keeps its row sums in scalar chains so that no packed fp32 instruction sits in the loop (measured: no difference,
//     77.6 against 77.7 us -- they were not what keeps the loop from the probe's overlap);
//   * the s_memtime timeline of one workgroup of the non-persistent version (BEVBERT_FWD4_ABL=32): 22 % of a workgroup's life
//     was its prologue -- 256 workgroups start a round together and ask for their first 110 KB at once, 28 MB at HBM speed --
//     and the units in which a wave fetched fragments from LDS took twice the time of the others.
// Hence:
//   * PERSISTENT workgroups, one per CU, each walking (batch, head, 448-query block) items.  K and V of an item are staged
//     into LDS ONCE (2 x 448 x 144 B = 126 KB; the 4-wave kernel staged them once per 128 queries) by a PRODUCER wave --
//     the 8th wave, in the SIMD slot that 7 x 64 = 448 queries leave empty -- which keeps two 64-key tiles in flight in its
//     registers and runs ahead across item boundaries: tile 0 of the next item is in its registers before the current item
//     ends, so an item's prologue is an LDS store and a barrier, and the global loads of a round no longer arrive as one burst.
//     LDS regions are recycled item to item: region t is rewritten after the barrier that closed tile t + 1 of the previous
//     item (every wave has issued its last read of tile t by then).  One bare s_barrier per tile ("tile t has landed"),
//     nothing else: no fences, no end-of-item barrier (compute waves walk the items in order);
//   * 7 compute waves x 64 queries (4 query tiles of 16): a K / V^T fragment fetched from LDS feeds 4 matrix instructions;
//   * the unit of work is (32 keys, 16 queries): S two units ahead, P V one unit behind, the fragment reads of the next
//     iteration issued right after the last matrix instruction on the old contents -- order pinned with a sched_barrier
//     (see the loop);
//   * dropout: one 32-bit word per lane and iteration, written by attn_drop_bits_kernel as a third layout and fetched with
//     an ordinary coalesced global load two iterations ahead -- scalar loads share the LDS wait counter (every LDS wait
//     would wait for them), and 64 wave masks per key tile would not fit the scalar registers.  Element e = 8 qt + 4 tt + r of
//     the lane sits at bit e >> 1 of half (e & 1): the mask of a packed bf16 pair is two packed 16-bit shifts of the
//     word, the drop one AND on the pair.
//
// ABL (diagnostics, BEVBERT_FWD4_ABL; results WRONG): 1 = no softmax arithmetic, 2 = no P V products, 4 = no Q K^T products;
// 32 = s_memtime stamps of one workgroup's second item -> $BEVBERT_FWD4_DBG (results correct)
#include <type_traits>

#include "attn_mfma_common.h"

#define F4_LD 72                       // row stride (bf16) of the K / V images: see F2_LD in attn_fwd2.hip
#define F4_NK 448                      // keys resident per workgroup
#define F4_NQ 448                      // queries per item: 7 compute waves x 64
#define F4_THR 5.0f

struct F4Lds {
  static constexpr int k_off = 0;
  static constexpr int v_off = F4_NK * F4_LD * 2;
  static constexpr int mask_off = 2 * F4_NK * F4_LD * 2;       // two mask rows (item parity)
  static constexpr int bytes = mask_off + 2 * F4_NK * 4;
};

__device__ __forceinline__ float f4_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <bool DROP, int ABL>
__global__ __launch_bounds__(512, 2) void attn_fwd4_kernel(AttnArgs a, const uint32_t* __restrict__ bits_l, int nitems,
                                                           unsigned long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char f4_smem[];
  bf16_raw* s_k = reinterpret_cast<bf16_raw*>(f4_smem + F4Lds::k_off);
  bf16_raw* s_v = reinterpret_cast<bf16_raw*>(f4_smem + F4Lds::v_off);
  float* s_mask = reinterpret_cast<float*>(f4_smem + F4Lds::mask_off);
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = (a.Lk + 63) >> 6, nsteps = 2 * ntiles;
  const int my_items = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // items blockIdx.x + i * gridDim.x
  // item -> (query block, head, batch): blocks of one (batch, head) are neighbours, heads of one sample too
  auto decode = [&](int item, int& blk, int& h, int& b) {
    blk = item % a.nblk;
    const int t = item / a.nblk;
    h = t % a.nh;
    b = t / a.nh;
  };
  bool dbg_on = false;
  auto stamp = [&](int slot) {
    if ((ABL & 32) && dbg_on && lane == 0) dbg[w * 64 + slot] = __builtin_amdgcn_s_memtime();
  };

  if (w == 7) {
    // ================= producer wave: the tile stream (item, tile) of this workgroup, two tiles in flight =================
    uint4 kr[2][8], vr[2][8];
    float mr[2][7];                    // mask row of an item (raw-score units, -inf beyond Lk), fetched with its tile 0
    int ld_it = 0, ld_t = 0;           // next tile to LOAD
    auto load = [&](auto slot_tag) {
      constexpr int slot = decltype(slot_tag)::value;
      int blk, h, b;
      decode((int)blockIdx.x + ld_it * (int)gridDim.x, blk, h, b);
      const bf16_raw* kp = (const bf16_raw*)a.k + (size_t)b * a.bsk + h * ATTN_D;
      const bf16_raw* vp = (const bf16_raw*)a.v + (size_t)b * a.bsv + h * ATTN_D;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = ld_t * 64 + (lane >> 3) + 8 * i, ch = lane & 7;
        kr[slot][i] = vr[slot][i] = make_uint4(0, 0, 0, 0);          // rows past the end are zero filled
        if (row < a.Lk) {
          kr[slot][i] = ld_frag_global(kp, a.ldk, row, ch * 8);
          vr[slot][i] = ld_frag_global(vp, a.ldv, row, ch * 8);
        }
      }
      if (ld_t == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const int key = lane + 64 * i;
          float m = -INFINITY;
          if (key < a.Lk) m = a.key_mask ? a.key_mask[(size_t)b * a.Lk + key] * (1.0f / a.scale) : 0.f;
          mr[slot][i] = m;
        }
      }
      if (++ld_t == ntiles) { ld_t = 0; ++ld_it; }
    };
    auto store = [&](int it, int t, auto slot_tag) {
      constexpr int slot = decltype(slot_tag)::value;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = t * 64 + (lane >> 3) + 8 * i, ch = lane & 7;
        *reinterpret_cast<uint4*>(s_k + row * F4_LD + ch * 8) = kr[slot][i];
        *reinterpret_cast<uint4*>(s_v + row * F4_LD + ch * 8) = vr[slot][i];
      }
      if (t == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) s_mask[(it & 1) * F4_NK + lane + 64 * i] = mr[slot][i];
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int total = my_items * ntiles;
    load(S0{});
    if (total > 1) load(S1{});
    int n = 0;                                         // stream index of the tile that the next barrier announces
    for (int it = 0; it < my_items; ++it) {
      for (int t = 0; t < ntiles; ++t, ++n) {
        // region t was last read for tile t of the previous item.  This wave has passed the barrier of tile t - 1 of THIS
        // item (t = 0: of the last tile of the previous item), so has every compute wave, and a compute wave at that barrier
        // has issued its last reads of region t long ago (it walks the items in order): no "item done" barrier is needed
        if (n & 1) {
          store(it, t, S1{});
          if (n + 2 < total) load(S1{});
        } else {
          store(it, t, S0{});
          if (n + 2 < total) load(S0{});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
        __builtin_amdgcn_s_barrier();                  // "tile t of item it has landed"
      }
    }
    return;
  }

  // ================= compute waves: 64 queries of every item =================
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const float sc2 = a.scale * LOG2E;
  bf16x8 qf[4][2];
  int qrow[4], hb_h = 0, hb_b = 0;                     // of the item whose Q fragments are in qf
  const uint32_t* wl = nullptr;                        // this lane's keep words of that item: 64 words per iteration
  auto fetch_q = [&](int it) {
    int blk;
    decode((int)blockIdx.x + it * (int)gridDim.x, blk, hb_h, hb_b);
    const int qbase = blk * F4_NQ + w * 64;
    const bf16_raw* qp = (const bf16_raw*)a.q + (size_t)hb_b * a.bsq + hb_h * ATTN_D;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      qrow[qt] = qbase + qt * 16 + c;
      const int r = qrow[qt] < a.Lq ? qrow[qt] : a.Lq - 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = as_bf16x8(ld_frag_global(qp, a.ldq, r, ks * 32 + g * 8));
    }
    if (DROP) {
      const int nq64 = a.nq16 >> 2;
      int q64 = qbase >> 6;
      q64 = q64 < nq64 ? q64 : nq64 - 1;
      wl = bits_l + ((size_t)(hb_b * a.nh + hb_h) * nq64 + q64) * (size_t)(a.nk64 * 2 * 64) + lane;
    }
  };
  fetch_q(0);
  uint32_t wd0 = 0, wd1 = 0, wd2 = 0;
  if (DROP) {
    wd0 = wl[0];
    wd1 = wl[64];                                      // nsteps >= 2 always
  }

  f32x4 oacc[4][4];
  float m_run[4], nm[4], thr[4];
  f32x2 l_run[4];
  // ONE register set each for the K / mask / V fragments.  Inside a unit the order is pinned (sched_barrier in the middle):
  //   first half : P V of the previous unit   | V reads of this iteration (unit 0, after its P V)  | exponentials 0..3
  //   second half: exponentials 4..7, packing | S two units ahead | K / mask reads of the next iteration (unit 1, after its S)
  // so every LDS read has most of a unit between its issue and the first matrix instruction that consumes it (left to itself
  // the compiler puts all eight matrix instructions at the top of the block and waits for the reads there: 664 against 328
  // cycles for such a unit on a wave that has its SIMD to itself).
  bf16x8 kf[2][2], vf[4];
  float4 mk[2];
  const float* cmask = s_mask;
  auto load_k = [&](int step) {
    const bf16_raw* kb = s_k + step * 32 * F4_LD;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      mk[tt] = *reinterpret_cast<const float4*>(&cmask[step * 32 + tt * 16 + g * 4]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        kf[tt][ks] = as_bf16x8(*reinterpret_cast<const uint4*>(kb + (tt * 16 + c) * F4_LD + (ks * 4 + g) * 8));
    }
  };
  auto load_v = [&](int step) {     // k-slot (g, j) <-> key 16 (j >> 2) + 4 g + (j & 3) of the iteration
    const bf16_raw* vb = s_v + step * 32 * F4_LD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vf[dt] = lds_frag_tr(vb, F4_LD, 4 * g, 16 + 4 * g, dt * 16, lane);
  };
  f32x4 sa[4][2];                   // [query tile][tt]: S of a unit is issued TWO units ahead of its softmax
  bf16x8 pbq[2];                    // [unit parity]
  auto issue_S = [&](f32x4 (&dst)[2], const bf16x8 (&q)[2]) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      dst[tt] = (f32x4){mk[tt].x, mk[tt].y, mk[tt].z, mk[tt].w};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) if (!(ABL & 4)) dst[tt] = mfma16(kf[tt][ks], q[ks], dst[tt]);
    }
  };
  auto issue_PV = [&](f32x4 (&o)[4], const bf16x8& pb) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) if (!(ABL & 2)) o[dt] = mfma16(vf[dt], pb, o[dt]);
  };

  for (int it = 0; it < my_items; ++it) {
    dbg_on = (ABL & 32) && blockIdx.x == 100 && it == 1;
    stamp(0);
    const int h = hb_h, b = hb_b;                      // qf / qrow / wl belong to this item; they move on in the last iteration
    int orow[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      orow[qt] = qrow[qt];
      m_run[qt] = -INFINITY;     // running maximum (log2 domain) shared by the four lanes of a query
      nm[qt] = 0.f;              // -(m_run), 0 while m_run is still -inf
      thr[qt] = -INFINITY;       // rescale when the unit's maximum exceeds this: any finite score while there is no maximum yet
      l_run[qt] = (f32x2){0.f, 0.f};   // this lane's share of the row sum (packed adds: even / odd elements)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    pbq[0] = pbq[1] = as_bf16x8(make_uint4(0, 0, 0, 0));
    cmask = s_mask + (it & 1) * F4_NK;
    __builtin_amdgcn_s_barrier();                      // tile 0 and the mask row of this item are in LDS
    asm volatile("" : : : "memory");
    stamp(1);
    load_k(0);
    load_v(0);
    issue_S(sa[0], qf[0]);
    issue_S(sa[1], qf[1]);

    // Unit (s, qt) = 32 keys of iteration s x the 16 queries of tile qt.  In program order a unit issues the 4 matrix
    // instructions of S two units ahead and the 4 of P V of the previous unit between the vector instructions of its softmax:
    //   S(s, qt + 2)  [qt >= 2: S(s + 1, qt - 2)]      softmax(s, qt)      P V (s, qt - 1)  [qt = 0: P V (s - 1, 3)]
    // (two ahead: with S of the NEXT unit the chain matrix result -> scale -> maximum -> vote -> branch sat exposed at the
    // end of every unit).  The rescale of a query tile (rare, wave-uniform) never meets a product in flight for the same
    // tile: P V (s - 1, qt) was issued three units ago.  LAST = the last iteration of the item: no S beyond the item, and
    // once S(s, 3) has been issued the Q registers take the NEXT item's fragments.
    auto iteration = [&](int s, auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      if (DROP) {
        // keep words two iterations ahead; the last iteration of an item fetches the first two words of the NEXT item
        // (unit 1, where wl moves on)
        if (!LAST) wd2 = wl[(s + 2 < nsteps ? s + 2 : nsteps - 1) * 64];
      }
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        const int N = (qt & 1) ^ 1, P = qt & 1;
        stamp(2 + 4 * s + qt);
        // tile (s + 1) / 2 must have landed before the K fragments of iteration s + 1 are fetched (end of this unit)
        if (qt == 1 && (s & 1) && !LAST) {
          __builtin_amdgcn_s_barrier();
          asm volatile("" : : : "memory");
        }
        // ---- t = s * scale * log2e - m_stale, the maximum of the unit, the (wave-uniform) rescale decision
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sa[qt][tt][r] = fmaf(sa[qt][tt][r], sc2, nm[qt]);
        const f32x4 s0 = sa[qt][0], s1 = sa[qt][1];
        const float lm = f4_max3(f4_max3(s0[0], s0[1], s0[2]), f4_max3(s0[3], s1[0], s1[1]), fmaxf(s1[2], s1[3]));
        if (__any(lm > thr[qt])) {
          const float m_new = fmaxf(m_run[qt], quad_max(lm) - nm[qt]);          // lm is relative to the stale maximum
          const float nm_new = (m_new == -INFINITY) ? 0.f : -m_new;
          const float alpha = fast_exp2(m_run[qt] + nm_new);  // exp2(m_old - m_new); first update: exp2(-inf) = 0, O = l = 0
          const float shift = nm_new - nm[qt];
          m_run[qt] = m_new;
          nm[qt] = nm_new;
          thr[qt] = (m_new == -INFINITY) ? -INFINITY : F4_THR;
          l_run[qt] *= alpha;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] *= alpha;
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sa[qt][tt][r] += shift;
        }
        // P = exp2(t); row sums before dropout (packed adds); dropout on the packed bf16 pairs: pair j of the iteration
        // (elements 2 j, 2 j + 1 in (qt, tt, r) order) has its keep bits at bit j of the two halves of the word, so
        // mask = each half shifted left by 15 - j, then arithmetically right by 15: two packed 16-bit shifts + one AND per pair
        float ps0 = 0.f, ps1 = 0.f;     // two scalar chains: packed fp32 adds would not overlap with the other wave's matrix work
        uint32_t pk[4];
        auto soft_pair = [&](int i) {
          const int tt = i >> 1, r0 = 2 * (i & 1);
          const float p0 = (ABL & 1) ? sa[qt][tt][r0] : fast_exp2(sa[qt][tt][r0]);
          const float p1 = (ABL & 1) ? sa[qt][tt][r0 + 1] : fast_exp2(sa[qt][tt][r0 + 1]);
          ps0 += p0;
          ps1 += p1;
          pk[i] = pack_bf16x2(p0, p1);
          if (DROP && !(ABL & 1)) {
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            const s16x2 wv = __builtin_bit_cast(s16x2, wd0);
            const s16x2 m = (s16x2)(wv << (short)(15 - 4 * qt - i)) >> (short)15;
            pk[i] &= __builtin_bit_cast(uint32_t, m);
          }
        };
        // ---- first half: P V of the previous unit (qt = 0: vf still holds iteration s - 1; s = 0: P = 0), then the V reads
        issue_PV(oacc[(qt + 3) & 3], pbq[N]);
        if (qt == 0) load_v(s);
        soft_pair(0);
        soft_pair(1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- second half: S two units ahead (qt >= 2: kf / mk already hold iteration s + 1), then the K / mask reads
        soft_pair(2);
        soft_pair(3);
        if (!LAST || qt < 2) issue_S(sa[(qt + 2) & 3], qf[(qt + 2) & 3]);
        if (qt == 1 && !LAST) load_k(s + 1);
        if (qt == 1 && LAST && it + 1 < my_items) {
          // the Q registers are free: fragments, row numbers and keep words of the next item (wd0 still holds this iteration's)
          fetch_q(it + 1);
          if (DROP) {
            wd1 = wl[0];
            wd2 = wl[64];
          }
        }
        l_run[qt][0] += ps0;
        l_run[qt][1] += ps1;
        pbq[P] = as_bf16x8(make_uint4(pk[0], pk[1], pk[2], pk[3]));
      }
      wd0 = wd1;
      wd1 = wd2;
    };
    for (int s = 0; s < nsteps - 1; ++s) iteration(s, std::false_type{});
    iteration(nsteps - 1, std::true_type{});
    issue_PV(oacc[3], pbq[1]);
    stamp(60);

    // ---- epilogue: normalise (dropout scaling folded in) and store O as whole 128-byte rows, 16 bytes per lane, through an
    //      LDS region nobody reads any more (the accumulator layout would give sixteen 32-byte pieces per store instruction).
    //      Free regions: this wave has passed the barrier of the last tile, so every wave is done with tiles 0 .. ntiles - 2;
    //      region 0 is the producer's (next item's tile 0, stored right after that barrier), regions >= ntiles are never
    //      written; the producer touches region t >= 1 only after the barrier of tile t - 1 of the NEXT item, i.e. after
    //      every wave has left this epilogue.  Five such K regions (waves 0 .. 4) and two V regions (waves 5, 6).
    const float ks = DROP ? a.keep_scale : 1.0f;
    const int ri = w < 5 ? w : w - 5;
    const int region = ri + 1 < ntiles - 1 ? ri + 1 : ri + 2;
    bf16_raw* stg = (w < 5 ? s_k : s_v) + region * 64 * F4_LD;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const float l = quad_sum(l_run[qt][0] + l_run[qt][1]);
      const float inv = ks / l;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<uint2*>(stg + (16 * qt + c) * F4_LD + 16 * dt + 4 * g) =
            make_uint2(pack_bf16x2(oacc[qt][dt][0] * inv, oacc[qt][dt][1] * inv),
                       pack_bf16x2(oacc[qt][dt][2] * inv, oacc[qt][dt][3] * inv));
      if (orow[qt] < a.Lq && a.lse && g == 0) a.lse[((size_t)b * a.nh + h) * a.Lq + orow[qt]] = (log2f(l) - nm[qt]) * LN2;
    }
    {
      const int q0 = orow[0] - c;                      // first query of this wave
      bf16_raw* op = (bf16_raw*)a.o + (size_t)b * a.bso + h * ATTN_D;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = (lane >> 3) + 8 * i, ch = lane & 7;
        const uint4 val = *reinterpret_cast<const uint4*>(stg + row * F4_LD + ch * 8);
        if (q0 + row < a.Lq) *reinterpret_cast<uint4*>(op + (size_t)(q0 + row) * a.ldo + ch * 8) = val;
      }
    }
    stamp(62);
  }
}

// =============================================================================================
// launcher
// =============================================================================================
static int f4_workgroups() {
  static const int ncu = [] {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const char* v = getenv("BEVBERT_FWD4_WGS");        // read per call: tests walk several items per workgroup on small batches
  return (v && atoi(v) > 0) ? atoi(v) : ncu;
}

template <bool D_, int A_ = 0>
static int launch_fwd4(const AttnArgs& a_in, const uint32_t* bits_l, hipStream_t st) {
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd4_kernel<D_, A_>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, F4Lds::bytes) == hipSuccess;
  BB_REQUIRE(ok, "attention fwd (gen 4): cannot raise the dynamic LDS limit to %d bytes", F4Lds::bytes);
  // one workgroup per CU (its LDS fills the CU), each walking items blockIdx.x, blockIdx.x + gridDim.x, ...
  const int ncu = f4_workgroups();
  AttnArgs a = a_in;
  a.nblk = (a.Lq + F4_NQ - 1) / F4_NQ;
  const int nitems = a.nblk * a.nh * a.B;
  const int grid = nitems < ncu ? nitems : ncu;
  unsigned long long* dbg = nullptr;
  if (A_ & 32) {                                       // diagnostics: stamps of one workgroup -> $BEVBERT_FWD4_DBG (text)
    static unsigned long long* buf = [] { void* p = nullptr; (void)hipMalloc(&p, 8 * 64 * 8); return (unsigned long long*)p; }();
    dbg = buf;
    (void)hipMemsetAsync(dbg, 0, 8 * 64 * 8, st);
  }
  hipLaunchKernelGGL((attn_fwd4_kernel<D_, A_>), dim3((unsigned)grid), dim3(512), F4Lds::bytes, st, a, bits_l, nitems, dbg);
  BB_CHECK_LAUNCH("attn_fwd(mfma, gen 4)");
  if (A_ & 32) {
    unsigned long long host[8 * 64];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(host, dbg, sizeof(host), hipMemcpyDeviceToHost);
    const char* path = getenv("BEVBERT_FWD4_DBG");
    if (FILE* f = fopen(path ? path : "/tmp/fwd4_dbg.txt", "w")) {
      for (int w = 0; w < 8; ++w) {
        for (int i = 0; i < 64; ++i) fprintf(f, "%llu ", host[w * 64 + i] ? host[w * 64 + i] - host[0] : 0ull);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
  return BB_OK;
}

// BEVBERT_ATTN_FWD4=0: these shapes go to the 4-wave kernels of attn_fwd2.hip (A/B measurements, on-GPU cross-check)
// An item fills a CU, so the kernel pays for whole rounds of items: it takes the call when the rounds are at least 85 % full
// and there are at least two of them (B = 64 x 12 heads = 768 items = 3.0 rounds of 256 CUs: 78.7 against 89.1 us; B = 32:
// 1.5 rounds, 51.4 against 47.8 us for the 4-wave kernel with its four workgroups per CU; B = 16: 28.6 against 28.0 us).
// BEVBERT_ATTN_FWD4=1 forces it for every supported shape (tests), =0 switches it off.
bool attn_fwd4_supported(const AttnArgs& a, const uint32_t* bits_l) {
  const char* env = getenv("BEVBERT_ATTN_FWD4");      // read per call
  const int mode = env ? atoi(env) : -1;
  if (mode == 0) return false;
  if (!(a.bias == nullptr && a.Lk > 256 && a.Lk <= F4_NK && a.Lq > 256 && (a.drop_p <= 0.f || bits_l != nullptr))) return false;
  if (mode == 1) return true;
  const int nitems = (a.Lq + F4_NQ - 1) / F4_NQ * a.nh * a.B, ncu = f4_workgroups();
  const int rounds = (nitems + ncu - 1) / ncu;
  return rounds >= 2 && nitems * 100 >= rounds * ncu * 85;
}

int attn_fwd4(const AttnArgs& a, const uint32_t* bits_l, hipStream_t st) {
  BB_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.bsq % 8 == 0 && a.bsk % 8 == 0 &&
                 a.bsv % 8 == 0 && a.bso % 8 == 0 && ((uintptr_t)a.q % 16) == 0 && ((uintptr_t)a.k % 16) == 0 &&
                 ((uintptr_t)a.v % 16) == 0 && ((uintptr_t)a.o % 16) == 0,
             "attention (MFMA path): pointers must be 16-byte aligned and strides multiples of 8 elements");
  static const int abl = [] { const char* v = getenv("BEVBERT_FWD4_ABL"); return v ? atoi(v) : 0; }();
  if (abl && a.drop_p > 0.f) {
    switch (abl) {
      case 1: return launch_fwd4<true, 1>(a, bits_l, st);
      case 2: return launch_fwd4<true, 2>(a, bits_l, st);
      case 4: return launch_fwd4<true, 4>(a, bits_l, st);
      case 6: return launch_fwd4<true, 6>(a, bits_l, st);
      case 7: return launch_fwd4<true, 7>(a, bits_l, st);
      case 32: return launch_fwd4<true, 32>(a, bits_l, st);
      default: break;
    }
  }
  if (a.drop_p > 0.f) return launch_fwd4<true>(a, bits_l, st);
  return launch_fwd4<false>(a, bits_l, st);
}
