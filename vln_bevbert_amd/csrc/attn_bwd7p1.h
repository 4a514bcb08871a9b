// Shared by the two generations of the 7 + 1-wave single-pass backward (attn_bwd2.hip: round 3, attn_bwd3.hip: round 5):
// LDS map, keep-bit select, the DPP row sum.
#pragma once
#include "attn_mfma_common.h"

#define B2_LDS_DS 36                      // row stride (bf16) of the dS image: 32 queries + 4 (conflict-free 8-byte writes)
#define B2_NKEYW 7                        // key waves
#define B2_NK (64 * B2_NKEYW)             // keys covered: 448

struct B2Lds {
  static constexpr int k_off = 0;                                   // [448][LDT] bf16   K, row-major, whole kernel
  static constexpr int ds_off = k_off + B2_NK * LDT * 2;            // [2][448][36] bf16 dS of a 32-query step
  static constexpr int q_off = ds_off + 2 * B2_NK * B2_LDS_DS * 2;  // [2][32][LDT] bf16 Q tile
  static constexpr int do_off = q_off + 2 * 32 * LDT * 2;           // [2][32][LDT] bf16 dO tile
  static constexpr int stat_off = do_off + 2 * 32 * LDT * 2;        // [2][2][32] float  lse (log2 domain), delta / ks
  static constexpr int bytes = stat_off + 2 * 2 * 32 * 4;
};

typedef const __attribute__((address_space(4))) uint64_t* bb_cu64p;
__device__ __forceinline__ float keep_select(float p, uint64_t lane_mask) {
  return __builtin_amdgcn_inverse_ballot_w64(lane_mask) ? p : 0.f;     // one v_cndmask_b32 with an SGPR-pair condition
}

// sum over the 8 lanes that share a tile row (lanes 8 j .. 8 j + 7) on the DPP network: no LDS round trip
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum8(float v) {
  v += dpp_f32<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);      // quad_perm [2,3,0,1]
  return v + dpp_f32<0x141>(v);   // row_half_mirror: lane i <-> 7 - i of its half row
}


// a.lo * b.lo + a.hi * b.hi + acc on two packed bf16 pairs (v_dot2c_f32_bf16): products exact in fp32, fp32 accumulation
typedef __bf16 bb_bf16pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bb_bf16pair, a), __builtin_bit_cast(bb_bf16pair, b), acc, false);
}
