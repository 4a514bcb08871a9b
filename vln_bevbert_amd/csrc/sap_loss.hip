// Fused tail of the single-action-prediction task: pretrain_src/model/pretrain_cmt.py:225-275 (forward_sap) after the
// three prediction heads --
//   fuse weight      fw = sigmoid(fuse_raw)            (0.5 when the model has no sap_fuse_linear)
//   global logits    gl = global_raw * fw,        -inf on visited nodes and beyond gmap_lens
//   local logits     ll = local_raw * (1 - fw),   -inf where the candidate's BEV cell is not navigable
//   fused logits     fu = gl + [ll | sum of ll over visited candidates | 0][src]     (vilmodel-side vpid matching: src)
//   loss             CE(gl, global label) + CE(ll, local label) + CE(fu, global label)
// In PyTorch this is ~35 elementwise / reduction launches forward and as many backward, each on a (B, <= 64) tensor.
// One wave per sample does all of it: lane j owns global-map node j and BEV candidate j.  The kernel also leaves the
// gradients w.r.t. the three head outputs for a unit upstream gradient; sap_loss_grad scales them by dloss[b].
#include "common.h"

template <typename T>
__global__ __launch_bounds__(64) void sap_loss_kernel(const T* __restrict__ graw, const T* __restrict__ lraw,
                                                      const T* __restrict__ fraw, const uint8_t* __restrict__ visited,
                                                      const int64_t* __restrict__ gmap_lens,
                                                      const uint8_t* __restrict__ nav_masks,
                                                      const int64_t* __restrict__ cand_idxs, const int64_t* __restrict__ src,
                                                      const uint8_t* __restrict__ vis_c, const int64_t* __restrict__ glabel,
                                                      const int64_t* __restrict__ llabel, float* __restrict__ loss,
                                                      float* __restrict__ dG, float* __restrict__ dL, float* __restrict__ dF,
                                                      int G, int K, int P) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float fw = fraw ? 1.0f / (1.0f + __expf(-io<T>::ld(fraw + b))) : 0.5f;
  const int gt = (int)glabel[b], lt = (int)llabel[b];
  // global branch
  const bool gv = lane < G;
  const float gr = gv ? io<T>::ld(graw + (size_t)b * G + lane) : 0.f;
  const bool gmasked = !gv || visited[(size_t)b * G + lane] != 0 || lane >= (int)gmap_lens[b];
  const float gl = gmasked ? -INFINITY : gr * fw;
  // local branch
  const bool lv = lane < K;
  const float lr = lv ? io<T>::ld(lraw + (size_t)b * K + lane) : 0.f;
  const bool lmasked = !lv || nav_masks[(size_t)b * P + cand_idxs[(size_t)b * K + lane]] == 0;
  const float ll = lmasked ? -INFINITY : lr * (1.0f - fw);
  const bool vc = lv && vis_c[(size_t)b * K + lane] != 0;
  const float bw = wave_sum(vc ? ll : 0.f);
  // fused logits: ext = [ll (K) | bw | 0]
  const int s = gv ? (int)src[(size_t)b * G + lane] : K + 1;
  const float from_l = __shfl(ll, s < K ? s : 0, 64);
  const float fu = gv ? gl + (s < K ? from_l : (s == K ? bw : 0.f)) : -INFINITY;
  // three log-sum-exps
  const float mg = wave_max(gl), ml = wave_max(ll), mf = wave_max(fu);
  const float eg = __expf(gl - mg), el = __expf(ll - ml), ef = __expf(fu - mf);      // exp(-inf - m) = 0
  const float sg = wave_sum(eg), sl = wave_sum(el), sf = wave_sum(ef);
  const float lse_g = mg + __logf(sg), lse_l = ml + __logf(sl), lse_f = mf + __logf(sf);
  const float gl_t = __shfl(gl, gt, 64), ll_t = __shfl(ll, lt, 64), fu_t = __shfl(fu, gt, 64);
  if (lane == 0) loss[b] = (lse_g - gl_t) + (lse_l - ll_t) + (lse_f - fu_t);
  // gradients for dloss[b] = 1
  const float pg = eg / sg, pl = el / sl, pf = ef / sf;
  const float dfu = gv ? pf - (lane == gt ? 1.f : 0.f) : 0.f;
  const float dgl = gv ? (pg - (lane == gt ? 1.f : 0.f)) + dfu : 0.f;
  float dext = 0.f, dext_bw = 0.f;            // gather's backward: lane k sums the fused gradients routed to candidate k
  for (int j = 0; j < G; ++j) {               // in node order: the summation order is fixed
    const int sj = __shfl(s, j, 64);
    const float dj = __shfl(dfu, j, 64);
    if (sj == lane && lane < K) dext += dj;
    if (sj == K) dext_bw += dj;
  }
  const float dll = lv ? (pl - (lane == lt ? 1.f : 0.f)) + dext + (vc ? dext_bw : 0.f) : 0.f;
  const float dgr = gmasked ? 0.f : dgl * fw;
  const float dlr = lmasked ? 0.f : dll * (1.0f - fw);
  if (gv) dG[(size_t)b * G + lane] = dgr;
  if (lv) dL[(size_t)b * K + lane] = dlr;
  const float dfw = wave_sum(gmasked ? 0.f : dgl * gr) - wave_sum(lmasked ? 0.f : dll * lr);
  if (lane == 0) dF[b] = dfw * fw * (1.0f - fw);
}

template <typename T>
__global__ __launch_bounds__(64) void sap_loss_grad_kernel(const float* __restrict__ dG, const float* __restrict__ dL,
                                                           const float* __restrict__ dF, const float* __restrict__ dloss,
                                                           T* __restrict__ dgraw, T* __restrict__ dlraw, T* __restrict__ dfraw,
                                                           int G, int K) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float g = dloss[b];
  if (lane < G) io<T>::st(dgraw + (size_t)b * G + lane, dG[(size_t)b * G + lane] * g);
  if (lane < K) io<T>::st(dlraw + (size_t)b * K + lane, dL[(size_t)b * K + lane] * g);
  if (lane == 0 && dfraw) io<T>::st(dfraw + b, dF[b] * g);
}

BEVBERT_API int bevbert_sap_loss_fwd(const void* global_raw, const void* local_raw, const void* fuse_raw,
                                     const uint8_t* visited, const int64_t* gmap_lens, const uint8_t* nav_masks,
                                     const int64_t* cand_idxs, const int64_t* src, const uint8_t* vis_c,
                                     const int64_t* global_labels, const int64_t* local_labels, float* loss, float* dG,
                                     float* dL, float* dF, int B, int G, int K, int P, int dtype, hipStream_t stream) {
  BB_REQUIRE(G >= 1 && G <= 64 && K >= 1 && K <= 62, "sap_loss: G=%d (<= 64) / K=%d (<= 62) out of range", G, K);
  if (B <= 0) return BB_OK;
#define GO(T)                                                                                                         \
  hipLaunchKernelGGL(sap_loss_kernel<T>, dim3(B), dim3(64), 0, stream, (const T*)global_raw, (const T*)local_raw,      \
                     (const T*)fuse_raw, visited, gmap_lens, nav_masks, cand_idxs, src, vis_c, global_labels,         \
                     local_labels, loss, dG, dL, dF, G, K, P)
  if (dtype == BB_F32) GO(float);
  else if (dtype == BB_BF16) GO(bf16_raw);
  else {
    bb_set_error("sap_loss: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
#undef GO
  BB_CHECK_LAUNCH("sap_loss_fwd");
  return BB_OK;
}

BEVBERT_API int bevbert_sap_loss_bwd(const float* dG, const float* dL, const float* dF, const float* dloss,
                                     void* d_global_raw, void* d_local_raw, void* d_fuse_raw, int B, int G, int K,
                                     int dtype, hipStream_t stream) {
  BB_REQUIRE(G >= 1 && G <= 64 && K >= 1 && K <= 62, "sap_loss: G=%d (<= 64) / K=%d (<= 62) out of range", G, K);
  if (B <= 0) return BB_OK;
#define GO(T)                                                                                                     \
  hipLaunchKernelGGL(sap_loss_grad_kernel<T>, dim3(B), dim3(64), 0, stream, dG, dL, dF, dloss, (T*)d_global_raw, \
                     (T*)d_local_raw, (T*)d_fuse_raw, G, K)
  if (dtype == BB_F32) GO(float);
  else if (dtype == BB_BF16) GO(bf16_raw);
  else {
    bb_set_error("sap_loss: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
#undef GO
  BB_CHECK_LAUNCH("sap_loss_bwd");
  return BB_OK;
}

// =============================================================================================
// Row-wise cross-entropy on wide rows (the MLM head: rows = masked tokens, C = vocabulary; pretrain_cmt.py:262-266
// F.cross_entropy(scores, labels, reduction="none") on fp32 copies of the logits).  Reads the head's logits in their
// own dtype, accumulates in fp32: forward = one pass for the maximum, one for the sum (the row stays in L2), loss and
// log-sum-exp out; backward writes d logits = (softmax - onehot) * dloss[row] in the logits' dtype.
// =============================================================================================
// A row starts at element row * C: the 4-wide loads begin at the first element whose index is a multiple of 4 (the
// vocabularies of the shipped configurations, 30 522 and 250 002, are not), scalars cover the ragged ends.
struct CeSpan { int head, nvec, tail0; };
__device__ __forceinline__ CeSpan ce_span(int row, int C) {
  const int head = min(C, (int)((4 - (((size_t)row * C) & 3)) & 3));
  const int nvec = (C - head) / 4;
  return {head, nvec, head + 4 * nvec};
}

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                     float* __restrict__ loss, float* __restrict__ lse, int C) {
  const int row = blockIdx.x, tid = threadIdx.x;
  const T* x = logits + (size_t)row * C;
  const CeSpan sp = ce_span(row, C);
  __shared__ float sh[4];
  float m = -INFINITY;
  for (int v = tid; v < sp.nvec; v += 256) {
    const float4 q = ld4<T>(x + sp.head + 4 * v);
    m = fmaxf(fmaxf(m, fmaxf(q.x, q.y)), fmaxf(q.z, q.w));
  }
  for (int c = tid; c < sp.head; c += 256) m = fmaxf(m, io<T>::ld(x + c));
  for (int c = sp.tail0 + tid; c < C; c += 256) m = fmaxf(m, io<T>::ld(x + c));
  m = wave_max(m);
  if ((tid & 63) == 0) sh[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  float s = 0.f;
  for (int v = tid; v < sp.nvec; v += 256) {
    const float4 q = ld4<T>(x + sp.head + 4 * v);
    s += (__expf(q.x - m) + __expf(q.y - m)) + (__expf(q.z - m) + __expf(q.w - m));
  }
  for (int c = tid; c < sp.head; c += 256) s += __expf(io<T>::ld(x + c) - m);
  for (int c = sp.tail0 + tid; c < C; c += 256) s += __expf(io<T>::ld(x + c) - m);
  s = wave_sum(s);
  if ((tid & 63) == 0) sh[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    const float l = m + __logf((sh[0] + sh[1]) + (sh[2] + sh[3]));
    lse[row] = l;
    loss[row] = l - io<T>::ld(x + target[row]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                     const float* __restrict__ lse, const float* __restrict__ dloss,
                                                     T* __restrict__ dlogits, int C) {
  const int row = blockIdx.x, tid = threadIdx.x;
  const T* x = logits + (size_t)row * C;
  T* d = dlogits + (size_t)row * C;
  const CeSpan sp = ce_span(row, C);
  const float l = lse[row], g = dloss[row];
  const int t = (int)target[row];
  for (int v = tid; v < sp.nvec; v += 256) {
    const int c = sp.head + 4 * v;
    const float4 q = ld4<T>(x + c);
    float4 o = make_float4(__expf(q.x - l) * g, __expf(q.y - l) * g, __expf(q.z - l) * g, __expf(q.w - l) * g);
    if (t == c) o.x -= g;
    if (t == c + 1) o.y -= g;
    if (t == c + 2) o.z -= g;
    if (t == c + 3) o.w -= g;
    st4<T>(d + c, o);
  }
  for (int c = tid; c < sp.head; c += 256) io<T>::st(d + c, (__expf(io<T>::ld(x + c) - l) - (c == t ? 1.f : 0.f)) * g);
  for (int c = sp.tail0 + tid; c < C; c += 256)
    io<T>::st(d + c, (__expf(io<T>::ld(x + c) - l) - (c == t ? 1.f : 0.f)) * g);
}

BEVBERT_API int bevbert_cross_entropy_fwd(const void* logits, const int64_t* target, float* loss, float* lse, int rows,
                                          int C, int dtype, hipStream_t stream) {
  BB_REQUIRE(C >= 1, "cross_entropy: C=%d", C);
  if (rows <= 0) return BB_OK;
  BB_REQUIRE(((uintptr_t)logits % 16) == 0, "cross_entropy: logits must be 16-byte aligned%s", "");
  if (dtype == BB_F32)
    hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3(rows), dim3(256), 0, stream, (const float*)logits, target, loss, lse, C);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL(ce_fwd_kernel<bf16_raw>, dim3(rows), dim3(256), 0, stream, (const bf16_raw*)logits, target, loss,
                       lse, C);
  else {
    bb_set_error("cross_entropy: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("cross_entropy_fwd");
  return BB_OK;
}

BEVBERT_API int bevbert_cross_entropy_bwd(const void* logits, const int64_t* target, const float* lse, const float* dloss,
                                          void* dlogits, int rows, int C, int dtype, hipStream_t stream) {
  if (rows <= 0) return BB_OK;
  BB_REQUIRE(((uintptr_t)logits % 16) == 0 && ((uintptr_t)dlogits % 16) == 0,
             "cross_entropy: logits / dlogits must be 16-byte aligned%s", "");
  if (dtype == BB_F32)
    hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(rows), dim3(256), 0, stream, (const float*)logits, target, lse, dloss,
                       (float*)dlogits, C);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL(ce_bwd_kernel<bf16_raw>, dim3(rows), dim3(256), 0, stream, (const bf16_raw*)logits, target, lse,
                       dloss, (bf16_raw*)dlogits, C);
  else {
    bb_set_error("cross_entropy: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("cross_entropy_bwd");
  return BB_OK;
}
