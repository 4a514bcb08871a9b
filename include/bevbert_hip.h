/* libbevbert_hip.so -- C ABI of the MI355X-native BEVBert cross-modal hot path (gfx950 only).
 *
 * The reference (MarSaKi/VLN-BEVBert) has no native/plugin layer: its hot path is Python nn.Modules over ATen,
 * torch_scatter and nn.MultiheadAttention.  This header is the boundary a reference maintainer binds with ctypes
 * (INTEGRATION.md shows the stubs); each entry names the reference code it replaces (paths under /root/reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host"; the caller owns all memory, nothing is allocated,
 *     retained or freed here.  Process-wide state, all of it: (1) bevbert_gemm's mutex-guarded plan cache (library
 *     handle + per-shape descriptors), (2) the step-salt POINTER registered by bevbert_set_step_salt (one per process:
 *     two training loops in one process share the salt word -- their masks still differ through their seeds), (3) the
 *     environment switches read by capi.hip (A/B knobs).  Entries may be called concurrently from several threads on
 *     different streams; bevbert_set_step_salt must not race with launches;
 *   - `stream` is a hipStream_t (PyTorch-ROCm: torch.cuda.current_stream().cuda_stream); launches are asynchronous;
 *   - return value 0 = ok, <0 = error (-1 invalid argument, -2 launch failure, -3 unsupported); the message is
 *     available from bevbert_last_error() (thread-local);
 *   - dtype codes: 0 = float32, 1 = bfloat16, 2 = float16; parameters (bias, gamma, beta) and statistics are always
 *     float32; arithmetic is float32 everywhere except the MFMA operands of the bf16 attention path;
 *   - dropout masks are a pure function of (seed, offset, flat element index) -- a 32-bit site key from (seed, offset),
 *     16 random bits per element, two elements per 32-bit mix: backward entries regenerate the forward's mask from
 *     the same (seed, offset) instead of storing it.  Attention indexes its elements as
 *     ((b*nh + h)*Lq + q) * Lk2 + k with Lk2 = Lk rounded up to even.
 */
#ifndef BEVBERT_HIP_H
#define BEVBERT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

#define BEVBERT_F32 0
#define BEVBERT_BF16 1
#define BEVBERT_F16 2

const char* bevbert_last_error(void);
int bevbert_version(void); /* major*10000 + minor*100 + patch */
/* drain the HIP runtime's sticky last-error slot after a FAILED stream capture (the stale code would otherwise be
 * reported by the next launch check of any library in the process); returns the number of errors dropped. */
int bevbert_hip_error_reset(void);
const char* bevbert_arch(void);
/* name of the kernel the calling thread's last bevbert_attn_fwd (backward = 0) / bevbert_attn_bwd (backward = 1) call was
 * dispatched to ("attn_fwd4", "attn_bwd3", "attn_short_fwd", ...; "" before the first call).  The choice depends on
 * shape, dtype, dropout and the BEVBERT_ATTN_* environment knobs; tests and bench.py use this to prove which path ran. */
const char* bevbert_attn_last_path(int backward);

/* ---------------------------------------------------------------------------------------------------------------
 * K1  lift + BEV binning + deterministic scatter-mean.
 * Replaces PointCloud.forward (pretrain_src/model/bev_utils.py:349-378), the world->ego transform of
 * GlocalTextPathCMTPreTraining.lift_splat (pretrain_src/model/pretrain_cmt.py:124-137), and
 * PointCloud.project_bev + torch_scatter.scatter_mean (bev_utils.py:381-430; fine-tune twin
 * map_nav_src/models/bev_utils.py:381-417, map_nav_src/r2r/agent.py:143-192).
 *
 * bevbert_bev_lift_bin: depths (B,V,hw,hw) stored /depth_scale, T_c2w (B,V,4,4), T_w2c (B,4,4), S_w2c (B,3),
 *   pix_scale (hw) = ((u + .5 - c)/f) fp32.  Outputs: cell (B,P) int32 (cell id = dim*z + x, -1 = dropped),
 *   order (B,P) int32 (point ids sorted by (cell, id)), cell_start (B, dim*dim+1) int32.  P = V*hw*hw <= 24576.
 * bevbert_bev_bin_points: same outputs from ready-made ego-frame points (B,P,3) + drop mask (B,P) uint8.
 * bevbert_bev_splat_mean: out[b,cell,:] = mean of feat[b,p,:] over the cell's points (0 if empty);
 *   semantics: sem_ids (B,P) uint8 class ids  XOR  sem_dense (B,P,S) float64 one-hot (the reference's format);
 *   out_sem (B,K,S) uint8 {0,1}, out_sem_mask (B,K) uint8; pass out_sem = NULL to skip semantics.
 *   sample_rows (B, rows_per_sample) int32 or NULL: feat / sem_ids are a device-resident store of per-viewpoint grid
 *   features ((N, P/rows_per_sample, C) / (N, P/rows_per_sample); dataset.py:110-118) read in place -- no batch copy:
 *   sample b's points p*P0 .. (p+1)*P0-1 are store row sample_rows[b][p].  rows_per_sample = 1 in pre-training; the
 *   fine-tuning agent splats the current viewpoint together with its visited neighbours (GraphMap.gather_node_pc,
 *   map_nav_src/models/graph_utils.py:129-144; agent.py:282-296), rows_per_sample = the batch maximum, short lists
 *   padded with any valid row whose depths are 0 (those points are dropped by the binning). */
int bevbert_bev_lift_bin(const float* depths, const float* T_c2w, const float* T_w2c, const float* S_w2c,
                         const float* pix_scale, int B, int V, int hw, float depth_scale, int dim, float res,
                         float y_clip, int* cell, int* order, int* cell_start, hipStream_t stream);
int bevbert_bev_bin_points(const float* points, const uint8_t* drop_mask, int B, int P, int dim, float res,
                           float y_clip, int* cell, int* order, int* cell_start, hipStream_t stream);
int bevbert_bev_splat_mean(const void* feat, int feat_dtype, const int* order, const int* cell_start, void* out,
                           int out_dtype, int B, int P, int K, int C, const uint8_t* sem_ids, const double* sem_dense,
                           int S, uint8_t* out_sem, uint8_t* out_sem_mask, const int* sample_rows, int rows_per_sample,
                           hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K2  fused multi-head attention (head_dim 64).
 * Replaces BertSelfAttention.forward (pretrain_src/model/vilmodel.py:103-141), BertOutAttention.forward
 * (vilmodel.py:325-352) and nn.MultiheadAttention inside TransformerEncoderLayer.forward_pre
 * (pretrain_src/model/transformer.py:170-182): softmax(Q K^T * scale + key_mask[b,k] + bias[b,q,k]) V with dropout
 * on the probabilities.  q/k/v/o are (B, L, nh*64) views with row/batch strides (elements):
 *   strides[8] (host) = {ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso}; dq/dk/dv use the q/k/v strides, dout the o strides.
 * key_mask (B,Lk) and bias (B,Lq,Lk) are additive fp32 (NULL = none; -inf allowed); lse (B,nh,Lq) fp32.
 * impl: 0 = auto (bf16 -> MFMA kernels, f32 -> exact fp32 kernels), 1 = exact kernels, 2 = MFMA kernels.
 * bwd: delta_ws is a (B,nh,Lq) fp32 scratch; dbias (B,nh,Lq,Lk) fp32 receives the PER-HEAD bias gradients (written,
 * not accumulated: the bias is shared by the heads and the caller sums dim 1 in a fixed order -- no atomics), NULL to
 * skip. */
int bevbert_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* key_mask,
                     const float* bias, const int64_t* strides, int B, int nh, int Lq, int Lk, int head_dim,
                     float scale, int dtype, int impl, float drop_p, uint64_t seed, uint64_t offset,
                     uint64_t* drop_bits, int bits_ready, hipStream_t stream);
int bevbert_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                     float* delta_ws, void* dq, void* dk, void* dv, float* dbias, const float* key_mask,
                     const float* bias, const int64_t* strides, int B, int nh, int Lq, int Lk, int head_dim,
                     float scale, int dtype, int impl, float drop_p, uint64_t seed, uint64_t offset,
                     const uint64_t* drop_bits, hipStream_t stream);
/* drop_bits (may be NULL): keep-bit workspace of the dropout mask on the attention probabilities (nn.Dropout,
 * vilmodel.py:134), bevbert_attn_drop_bits_words(B, nh, Lq, Lk) 64-bit words: the mask as one bit per score element,
 * once in the lane layout of the forward's accumulators and once in the backward's.  The bf16 kernels read the bits with
 * scalar loads and drop with one select per element instead of hashing per element.  bevbert_attn_drop_bits fills the
 * workspace (any stream, any time before the forward: the mask is a pure function of seed, offset, step salt and
 * element index); bevbert_attn_fwd does that itself first when bits_ready == 0.  NULL: both directions derive the mask
 * from (seed, offset) inline (round-2 kernels) -- same mask, more arithmetic. */
int64_t bevbert_attn_drop_bits_words(int B, int nh, int Lq, int Lk);
int bevbert_attn_drop_bits(uint64_t* drop_bits, int B, int nh, int Lq, int Lk, float drop_p, uint64_t seed,
                           uint64_t offset, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K3  y = LayerNorm(dropout(x + bias) + residual).
 * Replaces BertSelfOutput.forward / BertOutput.forward (vilmodel.py:150-154,189-193) and every bare LayerNorm of the
 * path (bias/residual NULL).  z_out (optional) receives the pre-norm sum for the backward; mean/rstd optional.
 * bwd: dz = grad wrt the pre-norm sum (= grad of residual), dx = grad wrt the dense output (dz through the dropout
 * mask; pass NULL when p == 0 and use dz); dgamma/dbeta/dbias (H) fp32 are written or accumulated (accumulate != 0);
 * workspace: bevbert_colsum_workspace_floats(3*H) floats. */
int bevbert_bias_dropout_residual_layernorm_fwd(const void* x, const float* bias, const void* residual,
                                                const float* gamma, const float* beta, void* y, void* z_out,
                                                float* mean, float* rstd, int rows, int H, float eps, int dtype,
                                                float drop_p, uint64_t seed, uint64_t offset, hipStream_t stream);
/* y = LayerNorm(x + bias) + post1 + post2 (post terms optional, same dtype / shape as y, added in that order after the
 * affine): the sums that follow a LayerNorm in the embedding compositions -- ImageEmbeddings (vilmodel.py:494-532:
 * LN(img) + LN(loc) + nav_type ...) and bev_input_embedding (:589-593) -- ride on its store instead of being separate
 * element-wise launches.  Backward: bevbert_layernorm_bwd with dy (the post terms receive dy unchanged). */
int bevbert_layernorm_post_fwd(const void* x, const float* bias, const float* gamma, const float* beta,
                               const void* post1, const void* post2, void* y, void* z_out, float* mean, float* rstd,
                               int rows, int H, float eps, int dtype, hipStream_t stream);
int bevbert_layernorm_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                          void* dz, void* dx, float* dgamma, float* dbeta, float* dbias, float* workspace, int rows,
                          int H, int dtype, float drop_p, uint64_t seed, uint64_t offset, int accumulate,
                          hipStream_t stream);
/* bevbert_layernorm_bwd with an addend: dz (and dx behind the dropout mask) = LayerNorm's input gradient + dz_add, the
 * gradient that reaches z through its second consumer when z is also a module output (pre-norm residual stream of the
 * panorama encoder, pretrain_src/model/transformer.py:170-182). */
int bevbert_layernorm_bwd_add(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                              void* dz, void* dx, const void* dz_add, float* dgamma, float* dbeta, float* dbias,
                              float* workspace, int rows, int H, int dtype, float drop_p, uint64_t seed, uint64_t offset,
                              int accumulate, hipStream_t stream);
/* The LayerNorm pair with an fp32 RESIDUAL STREAM around bf16 matrix operands -- what torch.autocast does to these modules
 * (pretrain_src/train_r2r.py:256-258: LayerNorm outputs and residual sums stay fp32, only GEMM operands are rounded).
 * fwd: x bf16 (dense output), residual fp32 or bf16 (residual_dtype: 0 / 1) or NULL; y16 = bf16(y) for the next GEMMs,
 * y32 = y (fp32) for the next residual add, z32 = the pre-norm sum in fp32 (y32 / z32 / mean / rstd may be NULL).
 * bwd: the output gradient is dy16 (bf16, from the GEMMs that read y16) + dy32 (fp32, from the residual add that read
 * y32), either may be NULL; dz32 (fp32) = gradient w.r.t. the residual, dx16 (bf16) = gradient w.r.t. the dense output
 * (dz through the dropout mask); parameter gradients and workspace as in bevbert_layernorm_bwd. */
int bevbert_layernorm_res32_fwd(const void* x, const float* bias, const void* residual, int residual_dtype,
                                const float* gamma, const float* beta, void* y16, float* y32, float* z32, float* mean,
                                float* rstd, int rows, int H, float eps, float drop_p, uint64_t seed, uint64_t offset,
                                hipStream_t stream);
int bevbert_layernorm_res32_bwd(const void* dy16, const float* dy32, const float* z32, const float* mean,
                                const float* rstd, const float* gamma, void* dz, void* dx16, float* dgamma,
                                float* dbeta, float* dbias, float* workspace, int rows, int H, float drop_p, uint64_t seed,
                                uint64_t offset, int accumulate, int dz_dtype,
                                hipStream_t stream);
int64_t bevbert_colsum_workspace_floats(int total_cols);
/* Split form of the parameter-gradient reductions: bevbert_layernorm_bwd / bevbert_bias_gelu_bwd called with NULL
 * dgamma/dbeta/dbias leave per-block partial sums [bevbert_colsum_partial_rows(rows)][nwhich][C] (nwhich = 3 for
 * LayerNorm: dgamma, dbeta, dbias; 1 for GELU) in `workspace`; bevbert_colsum_finalize folds them into the outputs
 * (written, or accumulated when accumulate != 0; NULL outputs are skipped) -- typically on another stream. */
int bevbert_colsum_partial_rows(int rows);
int bevbert_colsum_finalize(const float* partials, int nblocks, int nwhich, int C, float* out0, float* out1,
                            float* out2, int accumulate, hipStream_t stream);
/* Batched second stage.  bevbert_colsum_partials = the first stage of bevbert_colsum alone.  bevbert_multi_finalize runs
 * `ntasks` 64-column finalize tasks in ONE launch; `tasks` is a device array of 40-byte records
 * {u64 partials, u64 out, i32 nblocks, i32 row_stride, i32 col0, i32 ncols, i32 accumulate, i32 pad}:
 * out[c] (+)= sum_b partials[b * row_stride + col0 + c], c < ncols <= 64, fixed summation order (deterministic). */
int bevbert_colsum_partials(const void* dy, float* partials, int rows, int C, int dtype, hipStream_t stream);
int bevbert_multi_finalize(const void* tasks, int ntasks, hipStream_t stream);
/* Batched bevbert_accum_partials: `tasks` = device array of 40-byte records
 * {u64 partials, u64 sink, u64 n4_total, u32 off4, u32 n4, i32 S, i32 dtype}: for the n4 float4 groups starting at
 * off4, sink[i] += sum_{s < S} partials[s * n4_total + off4 + i] (sink already points at the range's first element). */
int bevbert_multi_accum(const void* tasks, int ntasks, hipStream_t stream);

/* K5  BertEmbeddings.forward (vilmodel.py:62-77): y = LayerNorm(word[ids] + pos[row % L] + type_row) (+dropout). */
int bevbert_embed_sum_layernorm_fwd(const int64_t* ids, const void* word, const void* pos, const void* type_row,
                                    const float* gamma, const float* beta, void* y, void* z_out, float* mean,
                                    float* rstd, int rows, int L, int H, float eps, int dtype, float drop_p,
                                    uint64_t seed, uint64_t offset, hipStream_t stream);

/* backward of the word-embedding gather of BertEmbeddings (vilmodel.py:50,67): table_grad[t, :] += sum of d[r, :] over
 * the rows with ids[r] == t, rows with ids[r] == padding_idx skipped (nn.Embedding(padding_idx=0); -1: none).  No
 * atomics: the first row of every id sums its id's rows in ascending row order (four waves, one contiguous quarter of
 * the row range each, folded pairwise), so the result is a pure function of the inputs.  H % 4 == 0, H <= 1024. */
int bevbert_embedding_grad(const int64_t* ids, const void* d, float* table_grad, int rows, int H, int padding_idx,
                           int dtype, hipStream_t stream);

/* Feature projections with a tiny inner dimension fused into their LayerNorm (pretrain_src/model/vilmodel.py:507-518
 * loc_layer_norm(loc_linear(loc_fts)), :577-583 bev_pos_embeddings, :589-593 gmap_pos_embeddings):
 *   y = (LayerNorm(feat W^T + bias) + post1) + table[idx]      feat (rows, K) fp32, K <= 16; weight (H, K) fp32 (nn.Linear
 * layout); post1 (rows, H) and table (T, H) of ``dtype`` or NULL; idx (rows) int64 (with table).  mean / rstd (rows) are left
 * for the backward, which RECOMPUTES feat W^T (no z tensor is stored, no dz tensor written): it adds the gradients of
 * weight (H, K), bias, gamma, beta into the given fp32 buffers (NULL: skipped) through per-workgroup partial sums in
 * ``workspace`` (bevbert_smallk_workspace_floats(rows, K, H) floats), folded in a fixed order.  d post1 = d table rows = dy.
 * H in {256, 512, 768, 1024}; (K + 8) * H * 4 bytes must fit 64 KB of LDS. */
int64_t bevbert_smallk_workspace_floats(int rows, int K, int H);
int bevbert_smallk_linear_layernorm_fwd(const float* feat, const float* weight, const float* bias, const float* gamma,
                                        const float* beta, const void* post1, const void* table, const int64_t* idx,
                                        void* y, float* mean, float* rstd, int rows, int K, int H, float eps, int dtype,
                                        hipStream_t stream);
int bevbert_smallk_linear_layernorm_bwd(const void* dy, const float* feat, const float* weight, const float* bias,
                                        const float* mean, const float* rstd, const float* gamma, float* dweight,
                                        float* dbias, float* dgamma, float* dbeta, float* workspace, int rows, int K,
                                        int H, int dtype, hipStream_t stream);

/* column sums for any column count (C % 4 != 0: the 30 522-wide vocabulary bias, the 1-wide prediction heads):
 * out[c] (+)= sum_r dy[r][c], fixed summation order. */
int bevbert_colsum_any(const void* dy, float* out, int rows, int C, int dtype, int accumulate, hipStream_t stream);

/* loss.mean() of the step (pretrain_src/train_r2r.py:263) over a static batch's rows: *out = sum_i w[i] x[i] / denom (w NULL:
 * ones -- zero for the padding rows of a static batch; denom_dev, if given, is read from device memory: the row count of
 * the batch that currently sits in the buffers); bwd: dx[i] = w[i] * *dout / denom. */
int bevbert_weighted_mean_fwd(const float* x, const float* w, const float* denom_dev, float denom, int n, float* out,
                              hipStream_t stream);
int bevbert_weighted_mean_bwd(const float* dout, const float* w, const float* denom_dev, float denom, int n, float* dx,
                              hipStream_t stream);

/* Semantic head's loss (pretrain_src/model/pretrain_cmt.py:391-441).  sem_select: row numbers of the cells with mask1 (and
 * mask2, if given) set, compacted in ascending order into idx (cap entries; padding: row 0), valid[i] = 1 for the real ones,
 * *denom = count x classes -- one workgroup, no host sync for the data-dependent count.  bce_rows_fwd: out[r] = sum over the
 * C classes of binary_cross_entropy_with_logits(logits[r, c], labels[idx[r], c]) (labels uint8 {0, 1}; idx NULL: row r);
 * bce_rows_bwd: dlogits[r, c] = (sigmoid(logits[r, c]) - label) * g[r]. */
int bevbert_sem_select(const uint8_t* mask1, const uint8_t* mask2, int n, int cap, int classes, int64_t* idx, float* valid,
                       float* denom, hipStream_t stream);
int bevbert_bce_rows_fwd(const void* logits, const uint8_t* labels, const int64_t* idx, float* out, int rows, int C, int dtype,
                         hipStream_t stream);
int bevbert_bce_rows_bwd(const void* logits, const uint8_t* labels, const int64_t* idx, const float* g, void* dlogits, int rows,
                         int C, int dtype, hipStream_t stream);

/* Graph-aware attention bias of the global-map encoder (pretrain_src/model/vilmodel.py:543-546,575-577: sprel_linear =
 * nn.Linear(1, 1) over the pairwise node distances).  fwd: out[i] = dists[i] * *w + *b (w, b: the parameters in device
 * memory).  bwd: *dw += sum dbias * dists, *db += sum dbias over the (layers, B, nh, G, G) per-head bias gradients the
 * attention backward leaves for every layer that used the bias (dists: (B, G, G)); two stages, fixed summation order. */
int bevbert_graph_bias_fwd(const float* dists, const float* w, const float* b, float* out, int64_t n, hipStream_t stream);
int bevbert_graph_bias_bwd(const float* dbias, const float* dists, int layers, int B, int nh, int G, float* dw, float* db,
                           float* workspace /* 1024 floats */, hipStream_t stream);

/* Row selection of activations and its backward (the reference indexes with boolean masks / index tensors:
 * pretrain_src/model/pretrain_cmt.py:254-256 masked tokens of the MLM head, :321-326 candidate cells of the SAP head, :403-410
 * supervised cells of the semantic head).  rows_gather: out[i, :] = src[ids[i], :].  rows_scatter: out[t, :] (+)= sum of
 * d[r, :] over the rows with ids[r] == t, for the ids that occur -- the caller zeroes ``out`` first (bevbert_zero);
 * accumulate = 1 adds to what the row holds (a second selection from the same tensor).  Duplicate ids are summed by the
 * first row of the id in ascending row order (no atomics, as bevbert_embedding_grad).  H % 4 == 0, H <= 1024. */
int bevbert_rows_gather(const void* src, const int64_t* ids, void* out, int rows, int H, int dtype, hipStream_t stream);
int bevbert_rows_scatter(const int64_t* ids, const void* d, void* out, int rows, int H, int dtype, int accumulate,
                         hipStream_t stream);

/* backward of nn.Embedding lookups on SMALL tables (vilmodel.py:452 nav_type_embedding, :567 gmap_step_embedding, the
 * token-type row): partials[s][t][:] = sum of d[r, :] over the rows r of slice s (rows_per_slice rows each) with
 * ids[r] == t; ceil(rows / rows_per_slice) slices of table_rows * H floats, folded into the table gradient by the
 * second stage of the column reductions (bevbert_multi_finalize with row_stride = table_rows * H).  No atomics: the
 * summation order is fixed. */
int bevbert_embedding_grad_sliced(const int64_t* ids, const void* d, float* partials, int rows, int H, int table_rows,
                                  int rows_per_slice, int dtype, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K4  y = gelu_erf(x + bias)  -- BertIntermediate.forward + gelu (vilmodel.py:31-37,177-180); F.gelu in the pano
 * encoder (transformer.py:178).  bwd: dx = dy * gelu'(x + bias) (dx may alias dy), dbias (C) written/accumulated. */
int bevbert_bias_gelu_fwd(const void* x, const float* bias, void* y, int rows, int C, int dtype, hipStream_t stream);
int bevbert_bias_gelu_bwd(const void* dy, const void* x, const float* bias, void* dx, float* dbias, float* workspace,
                          int rows, int C, int dtype, int accumulate, hipStream_t stream);
/* the same pair with ReLU: the prediction heads' Linear -> ReLU -> LayerNorm -> Linear (pretrain_src/model/pretrain_cmt.py:34-71;
 * the Linear's bias rides on the activation kernel, its gradient on the activation's backward) */
int bevbert_bias_relu_fwd(const void* x, const float* bias, void* y, int rows, int C, int dtype, hipStream_t stream);
int bevbert_bias_relu_bwd(const void* dy, const void* x, const float* bias, void* dx, float* dbias, float* workspace,
                          int rows, int C, int dtype, int accumulate, hipStream_t stream);
/* out[c] (+)= sum_r dy[r,c]: bias gradients of the projection GEMMs (autograd's grad.sum(0) in the reference). */
int bevbert_colsum(const void* dy, float* out, float* workspace, int rows, int C, int dtype, int accumulate,
                   hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K6  out[r] = sum_e w[e] * src[idx[e]], e in [rowptr[r], rowptr[r+1])  -- GlobalMapEncoder._aggregate_gmap_features
 * (vilmodel.py:632-666) with the visited/unvisited bookkeeping turned into a CSR on the host; the backward is the
 * same call on the transposed CSR. */
int bevbert_segment_wsum(const void* src, const int* rowptr, const int* idx, const float* w, void* out, int out_rows,
                         int H, int dtype, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K7  flat-arena optimiser.  All parameters / gradients / Adam moments live in contiguous fp32 arenas whose tensors
 * start on 1024-element boundaries; chunk_flags[n/1024]: bit0 = weight decay applies, bit1 = tensor has had a grad.
 * bevbert_grad_norm_clip: scalars[0] = ||pre_scale * g||_2, scalars[1] = pre_scale * min(1, max_norm/(norm+1e-6))
 *   (torch.nn.utils.clip_grad_norm_ as called at pretrain_src/train_r2r.py:295-306; pre_scale folds DDP's 1/world).
 *   partials: 1024 floats scratch.  No host sync: adamw_step reads the multiplier from device memory.
 * bevbert_adamw_step: AdamW.step (pretrain_src/optim/adamw.py:53-112): bias-corrected Adam, decoupled decay applied
 *   after the update; optionally refreshes the bf16 shadow copy of the parameters in the same pass.  chunk_steps
 *   (int32 per 1024-element chunk, zero-initialised by the caller) is the reference's per-parameter state["step"]:
 *   it advances only for chunks whose flag bit1 is set, so a parameter first used at step 6 starts at t = 1.
 *   lr_dev (may be NULL): learning rate read from device memory instead of the `lr` argument, so that a step captured
 *   in a hipGraph follows the schedule (optim/sched.py:24-30) without being re-captured. */
int bevbert_grad_norm_clip(const float* grads, int64_t n, float pre_scale, float max_norm, float* partials,
                           float* scalars, hipStream_t stream);
int bevbert_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* params_bf16,
                       const uint8_t* chunk_flags, int* chunk_steps, int64_t n, const float* grad_scale_dev,
                       const float* lr_dev, float lr, float beta1, float beta2, float eps, float weight_decay,
                       hipStream_t stream);
int bevbert_cast_f32(const float* src, void* dst, int64_t n, int dst_dtype, hipStream_t stream);
/* zero ``bytes`` bytes on the stream with a memset command (optimizer.zero_grad() of the flat gradient arena,
 * pretrain_src/train_r2r.py:313; a memset node when the step is captured) */
int bevbert_zero(void* p, int64_t bytes, hipStream_t stream);
/* sink[i] += sum_{s<S} partials[s*n + i]: reduction of the host-side split-K weight-gradient GEMMs (library batched
 * GEMM over S chunks of the token axis), fused with the accumulation into the fp32 gradient arena. */
int bevbert_accum_partials(const void* partials, float* sink, int S, int64_t n, int dtype, hipStream_t stream);

/* ---- library GEMM (hipBLASLt) with cached per-shape plans ---------------------------------------------------------
 * Replaces every nn.Linear matmul of the path (pretrain_src/model/vilmodel.py:92-94,314-316 query/key/value,
 * :146,:171,:185,:247,:261,:281 dense/decoder layers, :469-475,:542,:577-581,:621 input projections;
 * pretrain_cmt.py:38-68 prediction heads) and their two backward GEMMs, which PyTorch dispatches to the same
 * library at ~4x the host cost per call.
 *   C[M x N] = alpha * op(A)[M x K] . op(B)[K x N] (+ bias[N]),  row-major, optionally strided-batched.
 *   opA == 0: A stored M x K with row stride lda;  opA == 1: A stored K x M (row stride lda), used transposed.
 *   opB == 0: B stored K x N with row stride ldb;  opB == 1: B stored N x K (row stride ldb), used transposed.
 * in_dtype in {F32, BF16}; out_dtype == in_dtype or F32; bias (may be NULL) has dtype bias_dtype (F32 or out_dtype).
 * autotune > 1: on the first call of a problem, time the library's top `autotune` (<= 64) heuristic candidates on the
 * given operands (the product is recomputed, beta == 0) and keep the fastest; a problem listed in an imported tuning
 * table starts with the recorded choice instead.  Returns BB_EUNSUPPORTED (-3) when the
 * library has no kernel for the problem.
 * bevbert_gemm_plan / bevbert_gemm_run split the same call in two so that the per-call path carries 8 arguments:
 * plan returns an id >= 0 (cached by problem; bias_dtype < 0 = no bias; accumulate != 0: C += product, used by the
 * weight-gradient GEMMs that add into the fp32 gradient arena) or a negative error; run executes it (and autotunes
 * on its first call).  `workspace_bytes` at run time must be >= the value the plan was made with. */
int bevbert_gemm(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int opA, int opB,
                 int64_t lda, int64_t ldb, int64_t ldc, int batch, int64_t stride_a, int64_t stride_b,
                 int64_t stride_c, int in_dtype, int out_dtype, int bias_dtype, float alpha, void* workspace,
                 int64_t workspace_bytes, int autotune, hipStream_t stream);
int bevbert_gemm_plan(int M, int N, int K, int opA, int opB, int64_t lda, int64_t ldb, int64_t ldc, int batch,
                      int64_t stride_a, int64_t stride_b, int64_t stride_c, int in_dtype, int out_dtype, int bias_dtype,
                      int accumulate, int64_t workspace_bytes, int autotune);
int bevbert_gemm_run(int plan, const void* A, const void* B, void* C, const void* bias, void* workspace,
                     int64_t workspace_bytes, hipStream_t stream);
/* D = op(A) . op(B) + Cin through an ACCUMULATING plan (beta = 1) with a separate addend of D's layout. */
int bevbert_gemm_run_add(int plan, const void* A, const void* B, const void* Cin, void* D, const void* bias,
                         void* workspace, int64_t workspace_bytes, hipStream_t stream);
int bevbert_gemm_plan_count(void);
/* library algorithms dropped so far because two launches on the same operands left different output bits (split-K /
 * stream-K reductions through atomics): plans only ever use algorithms whose results repeat (BEVBERT_GEMM_DETERMINISTIC=0
 * turns the screening off). */
int bevbert_gemm_rejected_count(void);
/* Tuning table (text): one line "<problem key> <choice> <ncand>" per autotuned plan after a header naming the hipBLASLt
 * version.  export returns the bytes the text needs (incl. the final 0) and fills buf when cap suffices; import makes
 * later plans reuse the recorded choice instead of timing candidates (returns the number of rows; -3 when the table
 * belongs to another library version).  Shipping a table makes runs start tuned and removes run-to-run variation of
 * the picks (and timing candidates under a profiler is unreliable). */
int64_t bevbert_gemm_tuning_export(char* buf, int64_t cap);
int bevbert_gemm_tuning_import(const char* text);

/* y = residual + dropout(x) over n (multiple of 4) elements; residual may be NULL, y may alias x when the dtypes match.
 * Replaces the bare nn.Dropout sites of the path: feature dropout of the loader's fp32 features fused with their cast
 * to the compute dtype (pretrain_cmt.py:102-106 drop_feats; map_nav_src/models/model.py:30-36), the embedding dropout
 * (vilmodel.py:76, :494-496) and the two residual dropouts of TransformerEncoderLayer.forward_pre
 * (transformer.py:170-182).  The backward of x is the same call on dy with residual = NULL. */
int bevbert_dropout_add(const void* x, const void* residual, void* y, int64_t n, int in_dtype, int out_dtype,
                        float drop_p, uint64_t seed, uint64_t offset, hipStream_t stream);

/* Tail of forward_sap (pretrain_src/model/pretrain_cmt.py:225-275) behind the three prediction heads, one launch:
 * fw = sigmoid(fuse_raw) (fuse_raw NULL: 0.5); global logits = global_raw * fw with -inf on visited nodes and beyond
 * gmap_lens; local logits = local_raw * (1 - fw) with -inf where nav_masks[b, cand_idxs[b, k]] == 0; fused logits =
 * global + [local | sum of local over vis_c | 0][src]; loss[b] = CE(global) + CE(local) + CE(fused) (labels:
 * global_labels twice, local_labels).  dG (B,G) / dL (B,K) / dF (B): fp32 gradients w.r.t. global_raw / local_raw /
 * fuse_raw for dloss[b] = 1.  bool tensors are bytes.  G <= 64, K <= 62.  bevbert_sap_loss_bwd: d_*_raw = d* * dloss[b]
 * in the heads' dtype (d_fuse_raw may be NULL). */
int bevbert_sap_loss_fwd(const void* global_raw, const void* local_raw, const void* fuse_raw, const uint8_t* visited,
                         const int64_t* gmap_lens, const uint8_t* nav_masks, const int64_t* cand_idxs, const int64_t* src,
                         const uint8_t* vis_c, const int64_t* global_labels, const int64_t* local_labels, float* loss,
                         float* dG, float* dL, float* dF, int B, int G, int K, int P, int dtype, hipStream_t stream);
int bevbert_sap_loss_bwd(const float* dG, const float* dL, const float* dF, const float* dloss, void* d_global_raw,
                         void* d_local_raw, void* d_fuse_raw, int B, int G, int K, int dtype, hipStream_t stream);

/* F.cross_entropy(logits.float(), target, reduction="none") on wide rows -- the MLM head (pretrain_cmt.py:262-266,
 * rows = masked tokens, C = vocabulary): logits are read in their own dtype (bf16 / fp32, rows x C contiguous, base
 * 16-byte aligned, any C), statistics in fp32; loss[r] and lse[r] out.  bwd: dlogits = (softmax - onehot) * dloss[r]
 * in the logits' dtype (dlogits may alias logits). */
int bevbert_cross_entropy_fwd(const void* logits, const int64_t* target, float* loss, float* lse, int rows, int C,
                              int dtype, hipStream_t stream);
int bevbert_cross_entropy_bwd(const void* logits, const int64_t* target, const float* lse, const float* dloss,
                              void* dlogits, int rows, int C, int dtype, hipStream_t stream);

/* Per-step dropout salt.  Every dropout site derives its mask from hash(seed, offset) -- launch arguments, frozen in a
 * captured hipGraph.  After bevbert_set_step_salt(word) (word: 4 bytes of device memory owned by the caller, NULL to
 * clear) every dropout kernel of the library uses hash(site_key ^ *word): the host rewrites the word once per training
 * step and a replayed graph draws fresh masks; forward and backward of one step read the same word, so they agree.
 * Process-wide setting (the one piece of state besides the hipBLASLt plan cache); set it once, before any launch. */
int bevbert_set_step_salt(const void* device_word);

/* test hook: keep-mask (uint8) the kernels derive for n consecutive elements starting at `offset` */
int bevbert_dropout_keep_mask(uint8_t* out, int64_t n, float drop_p, uint64_t seed, uint64_t offset,
                              hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Fine-tune rollout bookkeeping on the device: the B topological maps of a rollout as dense
 * device arrays over a node capacity N.  Replaces map_nav_src/models/graph_utils.py:44-94 (FloydGraph.add_edge / update /
 * path lengths), :96-189 (GraphMap.update_graph, get_pos_fts, gather_node_pc's node choice) and the numeric half of
 * map_nav_src/r2r/agent.py:194-276,326-331 (_nav_gmap_variable, the start-viewpoint position features).  The host keeps
 * only the viewpoint-id -> node-index dictionaries and the presentation order of the nodes.
 * All pointers are device pointers except the state block itself (a host struct of device pointers).  f64 add / compare
 * follow the reference's Python floats bit for bit. */
typedef struct bevbert_gm_state {
  double* pos;        /* (B, N, 3)  node positions */
  double* dis;        /* (B, N, N)  shortest known distances, 95959595 = none (graph_utils.py:46) */
  int* point;         /* (B, N, N)  next hop of the shortest path, -1 = direct edge */
  int* hops;          /* (B, N, N)  len(FloydGraph.path(x, y)), written by bevbert_gm_update */
  uint8_t* visited;   /* (B, N) */
  int* step_ids;      /* (B, N)     agent.py:471-474 */
  int* pc_list;       /* (B, N)     visited nodes in first-visit order (GraphMap.node_pc's dict order) */
  int* npc;           /* (B) */
  int* node_row;      /* (B, N)     feature-store row of a visited node */
  float* node_T;      /* (B, N, V, 16) its V camera-to-world matrices */
  int B, N, V, pad;
} bevbert_gm_state;

/* One navigation step of every episode: live_graph[b] -> register the edges cur[b] -- cand[b, 0..ncand[b]) (positions
 * cur_pos (B,3), cand_pos (B,C,3) and edge lengths cand_dist (B,C), f64, computed by the host with the reference's own
 * arithmetic), relax all pairs through cur[b], mark it visited, rebuild the hop counts of the
 * first n_nodes[b] nodes; live_step[b] -> step_ids[b, cur[b]] = step_id (if > 0) and, when row != NULL and row[b] >= 0,
 * remember the node's feature-store row and poses T (B, V*16). */
int bevbert_gm_update(const bevbert_gm_state* st, const uint8_t* live_graph, const uint8_t* live_step, const int* cur,
                      const int* ncand, const int* cand, const double* cur_pos, const double* cand_pos,
                      const double* cand_dist, const int* n_nodes, int C, int step_id, const int* row, const float* T,
                      hipStream_t stream);

/* The tensors of agent.py:194-276 for the node order node (B, G-1) (cnt[b] real entries per sample, chosen by the host:
 * visited nodes first): step_ids (B,G) i64, visited / masks (B,G) bool bytes, pair (B,G,G) f32 = dis / 30, pos_fts
 * (B,G,7) f32 (row 0 = [stop]); gpos (B,7) or NULL: position features of start[b] seen from cur[b]. */
int bevbert_gm_nav_vars(const bevbert_gm_state* st, const int* node, const int* cnt, const int* cur, const int* start,
                        const double* heading, const double* elevation, int G, int enc_full_graph, int act_visited,
                        int64_t* step_ids, uint8_t* visited, uint8_t* masks, float* pair, float* pos_fts, float* gpos,
                        hipStream_t stream);

/* graph_utils.py:129-144: per sample the visited nodes within `order` hops of cur[b], in first-visit order, as
 * feature-store rows (B,R) (padding repeats the first row with live = 0) and their poses T_c2w (B,R,V*16);
 * *overflow = 1 if a sample has more than R such nodes. */
int bevbert_gm_bev_select(const bevbert_gm_state* st, const int* cur, int order, int R, int* rows, uint8_t* live,
                          float* T_c2w, int* overflow, hipStream_t stream);

/* agent.py:150-156,168-176: the stored views of the selected nodes, out[i] = live[i] ? store[rows[i]] : 0 for i < n_out;
 * store / out rows are row_bytes bytes (V x h x w depths of any element type). */
int bevbert_gm_gather_views(const void* store, const int* rows, const uint8_t* live, void* out, int n_out, int64_t row_bytes,
                            hipStream_t stream);

/* agent.py:485-494 without gradients: embed_sum (B,N,H) / embed_cnt (B,N) running sums of the node embeddings (slot = node
 * index): the current viewpoint's slot is rewritten with avg (B,H), every not-yet-visited candidate j < ncand[b] accumulates
 * pano (B,V,H)[b, j].  live (B) bytes; cand (B,C) node indices (-1 = none). */
int bevbert_gm_embed_update(const bevbert_gm_state* st, void* embed_sum, float* embed_cnt, const void* avg, const void* pano,
                            const uint8_t* live, const int* cur, const int* ncand, const int* cand, int C, int V, int H,
                            int dtype, hipStream_t stream);
/* graph_utils.py:146-147 for the listed nodes: out (B,G,H), row 0 = [stop] = 0, row j = sum / count of node[b, j-1]. */
int bevbert_gm_node_embeds(const bevbert_gm_state* st, const void* embed_sum, const float* embed_cnt, const int* node,
                           const int* cnt, int G, int H, int dtype, void* out, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BEVBERT_HIP_H */
