// Argument block shared by the attention kernels (K2).
//
// One kernel family covers every attention in the path: BertSelfAttention (vilmodel.py:79-141),
// BertOutAttention (vilmodel.py:301-352) and nn.MultiheadAttention inside the pano encoder
// (transformer.py:138,174-178).  Q/K/V are read in place from the projection GEMM outputs --
// (batch, seq, heads*64) with arbitrary row/batch strides, so a packed QKV buffer needs no split
// or permute -- and O is written merged-head (batch, seq, heads*64), i.e. no transpose_for_scores
// / permute / contiguous copies ever touch HBM.  The (N,12,Lq,Lk) score tensor is never written.
//
//   S = (Q K^T) * scale + key_mask[b, k] + bias[b, q, k]        (mask/bias additive fp32; -inf allowed)
//   P = softmax_k(S) ; O = dropout(P) V ; lse = logsumexp_k(S)
#pragma once
#include "common.h"

#define ATTN_D 64  // head dim (hidden 768 / 12 heads: configs/r2r_model.json)

struct AttnArgs {
  const void *q, *k, *v;
  void* o;
  float* lse;              // (B, nh, Lq) natural-log logsumexp; may be null for inference-only forward
  const float* key_mask;   // (B, Lk) additive or null
  const float* bias;       // (B, Lq, Lk) additive, shared by heads, or null
  int64_t ldq, ldk, ldv, ldo;  // row strides (elements)
  int64_t bsq, bsk, bsv, bso;  // batch strides (elements)
  int B, nh, Lq, Lk;
  float scale;
  float drop_p;
  float keep_scale;    // 1 / (1 - drop_p)
  uint32_t drop_thr;   // 16-bit threshold
  uint32_t drop_key;   // bb_site_key(seed, offset); kernels use bb_salted(drop_key, salt)
  const uint32_t* salt; // per-step salt word in device memory (bevbert_set_step_salt) or null
  int nblk;            // workgroups per (batch, head) along the tiled sequence axis (set by the launcher)
  int Lk2;             // Lk rounded up to even: dropout element index = ((b*nh + h)*Lq + q)*Lk2 + k
  // Keep-bit matrix of the dropout mask (MFMA path, optional).  The forward kernel has every keep decision in a wave-
  // wide compare mask anyway: it stores those masks (one 64-bit word per 16 queries x 4 keys), and the backward kernels
  // read one bit per score element instead of hashing again.  Word / bit of element (q, k) of batch-head bh:
  //   word = ((((bh * nq16 + (q >> 4)) * nk64 + (k >> 6)) * 4 + ((k >> 4) & 3)) * 4 + (k & 3)),  bit = 16 * ((k >> 2) & 3) + (q & 15)
  uint64_t* drop_bits; // B * nh * nq16 * nk64 * 16 words, or null (backward then regenerates the mask from the hash)
  // The same bits with lanes <-> keys (attn_bwd2.hip): word ((((bh * (nq16/2) + (q >> 5)) * nk64 + (k >> 6)) * 2 + ((q >> 4) & 1)) * 16
  // + 4 * ((k >> 4) & 3) + (q & 3)), bit = 16 * ((q >> 2) & 3) + (k & 15).  Both are written by attn_drop_bits_kernel (attn_fwd2.hip).
  uint64_t* drop_bits_b;
  int nq16, nk64;      // nq16 = 8 * ceil(Lq / 128), nk64 = ceil(Lk / 64)
  // backward only
  const void* dout;        // (B, Lq, nh*64), strides ldo/bso
  const float* delta;      // (B, nh, Lq) rowsum(dO * O)
  void *dq, *dk, *dv;      // same strides as q/k/v
  float* dbias;            // (B, nh, Lq, Lk) fp32 per-head bias gradients (every element written once, no atomics; the
                           // caller folds the heads in a fixed order), or null
  int dbg;                 // timing ablations of attn_short.hip (BEVBERT_SHORT_DBG; results WRONG): 1 no output stores, 2 no O loads
};

// XCD-aware decode of a 1-D grid: hardware places workgroup id on XCD id % 8, each XCD has a private L2.  Work items
// are numbered (block-in-sequence fastest, then head, then batch) and every XCD takes one contiguous run of them, so
// the nblk workgroups that share one (batch, head)'s K/V (or Q/dO) tiles hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ void attn_decode_block(const AttnArgs& a, int& blk, int& h, int& b) {
  const int nwg = gridDim.x, id = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, slot = id >> 3;
  const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  blk = w % a.nblk;
  const int t = w / a.nblk;
  h = t % a.nh;
  b = t / a.nh;
}

__device__ __forceinline__ size_t attn_bits_word(const AttnArgs& a, int bh, int q16, int k64, int t, int r) {
  return ((((size_t)bh * a.nq16 + q16) * a.nk64 + k64) * 4 + t) * 4 + r;
}

__device__ __forceinline__ uint32_t attn_row_base(const AttnArgs& a, int b, int h, int q) {
  return (uint32_t)((((uint32_t)b * a.nh + h) * a.Lq + q) * (uint32_t)a.Lk2);
}
__device__ __forceinline__ uint32_t attn_elem(const AttnArgs& a, int b, int h, int q, int k) {
  return attn_row_base(a, b, h, q) + (uint32_t)k;
}
