// Building blocks shared by the MFMA attention kernels (attn_mfma.hip: forward + two-kernel backward;
// attn_bwd1.hip: single-pass backward): fragment loads, tile staging, the hardware-transposing LDS read.
#pragma once
#include <stdlib.h>

#include "attn_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TK 64   // keys (or queries, in dKV) per LDS tile
#define LDT 80  // LDS row stride in bf16 elements: 160 B keeps the 16-byte row reads of 16 consecutive rows AND the
                // transposing 8-byte reads (rows 4 g ..) conflict-free; 144 B was 2-way on both (scripts/lds_bank_sim.py)

#define DKV_TILE_BYTES (2 * 4 * TK * LDT * 2)                 // dK/dV kernel: two buffers of four bf16 tiles
#define DKV_SMEM_BYTES (DKV_TILE_BYTES + 2 * 2 * TK * 4)     // + lse / delta rows

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 as_bf16x8(uint4 u) { return __builtin_bit_cast(bf16x8, u); }

// 16-byte fragment (8 consecutive head-dim elements) of row `row` of a (rows, 64) head slice in global memory
__device__ __forceinline__ uint4 ld_frag_global(const bf16_raw* base, int64_t ld, int row, int d0) {
  return *reinterpret_cast<const uint4*>(base + (size_t)row * ld + d0);
}

// A [TK rows][64] tile travels global -> registers -> LDS in two steps so that the global loads of tile j+1 are in
// flight while tile j is being computed.  Thread t owns 16-byte chunks c = t and t + 256: row c >> 3, dims 8 (c & 7)..+7.
struct TileRegs { uint4 v[2]; };
__device__ __forceinline__ void tile_load(TileRegs& r, const bf16_raw* src, int64_t ld, int row0, int nrows, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256, row = c >> 3, ch = c & 7;
    r.v[i] = make_uint4(0, 0, 0, 0);                       // rows past the end are zero filled
    if (row0 + row < nrows) r.v[i] = ld_frag_global(src, ld, row0 + row, ch * 8);
  }
}
// row-major image: dst[row][d], stride LDT
__device__ __forceinline__ void tile_store_rows(bf16_raw* dst, const TileRegs& r, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256, row = c >> 3, ch = c & 7;
    *reinterpret_cast<uint4*>(dst + row * LDT + ch * 8) = r.v[i];
  }
}
// transposed image: dst[d][row ^ swz(d)], swz(d) = 8 * ((d >> 3) & 7).  Without the XOR the 64 lanes of a wave
// (8 rows x 8 chunks) would hit 4 banks (16-way conflict); with it they cover 64 consecutive columns of one d-row.
__device__ __forceinline__ int swz_cols(int d) { return ((d >> 3) & 7) << 3; }
__device__ __forceinline__ void tile_store_cols(bf16_raw* dst, const TileRegs& r, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256, row = c >> 3, ch = c & 7;
    const int col = row ^ (ch << 3);                        // swz_cols(ch*8 + e) == ch << 3 for e in 0..7
    const uint32_t w[4] = {r.v[i].x, r.v[i].y, r.v[i].z, r.v[i].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dst[(ch * 8 + 2 * j) * LDT + col] = (bf16_raw)(w[j] & 0xffffu);
      dst[(ch * 8 + 2 * j + 1) * LDT + col] = (bf16_raw)(w[j] >> 16);
    }
  }
}
// A fragment from a row-major tile: row (t*16 + (l&15)), k-slots = head-dim ks*32 + g*8 .. +7
__device__ __forceinline__ bf16x8 lds_frag_rows(const bf16_raw* tile, int t, int ks, int lane) {
  return as_bf16x8(*reinterpret_cast<const uint4*>(tile + (t * 16 + (lane & 15)) * LDT + ks * 32 + (lane >> 4) * 8));
}
// A fragment from a transposed tile: row d = dt*16 + (l&15), k-slots (g, j) <-> tile column 32 m + 16 (j>>2) + 4 g + (j&3)
__device__ __forceinline__ bf16x8 lds_frag_cols(const bf16_raw* tileT, int dt, int m, int lane) {
  const int d = dt * 16 + (lane & 15);
  const bf16_raw* row = tileT + d * LDT;
  const int c0 = (m * 32 + (lane >> 4) * 4) ^ swz_cols(d);         // XOR by a multiple of 8 keeps 4-runs contiguous
  const int c1 = (m * 32 + 16 + (lane >> 4) * 4) ^ swz_cols(d);
  const uint2 lo = *reinterpret_cast<const uint2*>(row + c0);
  const uint2 hi = *reinterpret_cast<const uint2*>(row + c1);
  return as_bf16x8(make_uint4(lo.x, lo.y, hi.x, hi.y));
}
// 2^x on the transcendental unit (v_exp_f32); arguments here are <= 0 and results below 2^-126 may flush to 0
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// pack accumulators of key/query tiles (2m, 2m+1) into the B operand of the second product
__device__ __forceinline__ bf16x8 pack_pair(const f32x4& a, const f32x4& b) {
  return as_bf16x8(make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]),
                              pack_bf16x2(b[2], b[3])));
}
// reduce over the four lanes that own the same query/key column (l&15)
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}


// ---- ds_read_b64_tr_b16: the hardware-transposing LDS read of gfx950 ---------------------------------------------
// Within each group of 16 lanes, lane i passes the address of 4 consecutive bf16 (8-byte aligned): lane i stands for
// row i >> 2, column chunk i & 3 of a [4 rows][16 columns] block whose row addresses are free.  Lane n of the group
// receives column n of the block: element j = row j.  (Verified on the MI355X: scripts/probes/tr_probe.hip.)
// An MFMA operand whose k-slots are ROWS of a row-major LDS image therefore needs no transposed copy of the image.
typedef __bf16 bb_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr16(const bf16_raw* p) {
  typedef __attribute__((address_space(3))) bb_bf16x4* lds_ptr;
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p)));
}
// Fragment (A or B operand of v_mfma_f32_16x16x32_bf16) whose eight k-slots of lane group g are the image rows
// row_lo + (0..3) and row_hi + (0..3) (callers fold g into both), at the 16 columns col0 .. col0+15 (lane & 15).
__device__ __forceinline__ bf16x8 lds_frag_tr(const bf16_raw* img, int stride, int row_lo, int row_hi, int col0,
                                              int lane) {
  const int i = lane & 15;
  const int off = (i >> 2) * stride + col0 + 4 * (i & 3);
  const uint2 lo = lds_tr16(img + row_lo * stride + off);
  const uint2 hi = lds_tr16(img + row_hi * stride + off);
  return as_bf16x8(make_uint4(lo.x, lo.y, hi.x, hi.y));
}
