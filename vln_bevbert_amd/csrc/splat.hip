// K1: lift (depth -> ego-frame points) + BEV cell binning + deterministic scatter-mean ("splat").
//
// Replaces, for a whole batch in two launches and with no host sync:
//   PointCloud.forward / pixel_to_world_mapping      pretrain_src/model/bev_utils.py:349-378,200-248
//   world->ego transform in lift_splat               pretrain_src/model/pretrain_cmt.py:124-137
//   PointCloud.project_bev + torch_scatter.scatter_mean  bev_utils.py:381-430 (B-iteration Python loop,
//                                                    3 boolean compactions = 3 D2H syncs per sample)
//
// HBM layout: features stay where the loader put them, (B, P, C) row-major; a cell's mean is a GATHER
// over the cell's point list (ascending point id), so every feature row is read exactly once, fully
// coalesced (C*4 = 3 KiB per row), the output row is written once, and there are no float atomics:
// the sum order is fixed => bit-reproducible and bit-equal to the CPU oracle (index_add in point order).
#include "common.h"

#define MAX_CELLS 1024
#define MAX_POINTS 24576   // 10 panoramas of 12 x 14 x 14 points (fine-tune: current viewpoint + visited neighbours)

// ---------------------------------------------------------------------------------------------
// Kernel A: one workgroup per sample: cell id per point, then a stable counting sort by cell.
// All arithmetic that feeds round() is written with explicitly rounded mul/add in the oracle's
// order (oracle/bevbert_ref.py lift_points): a fused multiply-add would move points across cells.
// ---------------------------------------------------------------------------------------------
struct LiftArgs {
  const float* depths;   // (B, V, hw, hw) stored /depth_scale   [mode 0]
  const float* T_c2w;    // (B, V, 4, 4)
  const float* T_w2c;    // (B, 4, 4)
  const float* S_w2c;    // (B, 3)
  const float* pix;      // (hw) = ((u + .5 - c) / f) in fp32
  const float* points;   // (B, P, 3) ego-frame points           [mode 1]
  const uint8_t* pmask;  // (B, P) 1 = drop                      [mode 1]
  int V, hw;
  float depth_scale;
  int P, dim;
  float res, half, y_clip;
  int* cell;        // (B, P) out: cell id or -1
  int* order;       // (B, P) out: point ids sorted by (cell, point id)
  int* cell_start;  // (B, dim*dim + 1) out
};

__device__ __forceinline__ int cell_of(float ex, float ey, float ez, bool dropped, int dim, float res, float half,
                                       float y_clip) {
  const float fx = rintf(__fadd_rn(__fdiv_rn(ex, res), half));  // half-to-even, as torch.round
  const float fz = rintf(__fadd_rn(__fdiv_rn(ez, res), half));
  const float fd = (float)dim;
  const bool outside = (fx >= fd) | (fz >= fd) | (fx < 0.f) | (fz < 0.f);
  if (dropped | outside | (ey > y_clip)) return -1;
  return dim * (int)fz + (int)fx;
}

template <int MODE>
__global__ __launch_bounds__(256) void bev_bin_sort_kernel(LiftArgs a) {
  __shared__ short s_cell[MAX_POINTS];
  __shared__ int s_cnt[MAX_CELLS + 1];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int K = a.dim * a.dim;
  for (int c = tid; c <= K; c += 256) s_cnt[c] = 0;
  __syncthreads();

  for (int p = tid; p < a.P; p += 256) {
    int cid;
    if (MODE == 0) {
      const int hw2 = a.hw * a.hw;
      const int v = p / hw2, r = (p - v * hw2) / a.hw, u = p % a.hw;
      const float z = __fmul_rn(a.depths[(size_t)b * a.P + p], a.depth_scale);
      const float x = __fmul_rn(z, a.pix[u]);
      const float y = __fmul_rn(z, a.pix[r]);
      const float* T = a.T_c2w + ((size_t)b * a.V + v) * 16;
      float w[3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        w[i] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[i * 4 + 0], x), __fmul_rn(T[i * 4 + 1], y)),
                                   __fmul_rn(T[i * 4 + 2], z)),
                         T[i * 4 + 3]);
      const float* S = a.S_w2c + (size_t)b * 3;
      const float p0 = __fsub_rn(w[0], S[0]), p1 = __fsub_rn(w[1], S[1]), p2 = __fsub_rn(w[2], S[2]);
      const float* R = a.T_w2c + (size_t)b * 16;
      float e[3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        e[i] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(R[i * 4 + 0], p0), __fmul_rn(R[i * 4 + 1], p1)),
                                   __fmul_rn(R[i * 4 + 2], p2)),
                         R[i * 4 + 3]);
      cid = cell_of(e[0], e[1], e[2], z == 0.f, a.dim, a.res, a.half, a.y_clip);
    } else {
      const float* q = a.points + ((size_t)b * a.P + p) * 3;
      cid = cell_of(q[0], q[1], q[2], a.pmask[(size_t)b * a.P + p] != 0, a.dim, a.res, a.half, a.y_clip);
    }
    s_cell[p] = (short)cid;
    a.cell[(size_t)b * a.P + p] = cid;
    if (cid >= 0) atomicAdd(&s_cnt[cid], 1);  // integer LDS atomic: result is order independent
  }
  __syncthreads();

  // exclusive scan of the K counts by wave 0 (each lane owns a contiguous run of cells)
  if (tid < 64) {
    const int per = (K + 63) / 64;
    const int c0 = tid * per;
    int sum = 0;
    for (int i = 0; i < per; ++i)
      if (c0 + i < K) sum += s_cnt[c0 + i];
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o, 64);
      if (tid >= o) incl += t;
    }
    int run = incl - sum;
    for (int i = 0; i < per; ++i)
      if (c0 + i < K) {
        const int n = s_cnt[c0 + i];
        s_cnt[c0 + i] = run;
        run += n;
      }
    if (tid == 63) s_cnt[K] = incl;  // total kept points
  }
  __syncthreads();
  for (int c = tid; c <= K; c += 256) a.cell_start[(size_t)b * (K + 1) + c] = s_cnt[c];

  // Stable placement (point ids ascending inside every cell), in parallel.  Round 2 let the thread that owns a cell
  // walk all P points (2 352 serial LDS reads per thread: 153 us for 64 workgroups, 1.5x the splat it feeds).  Now each
  // of the four waves owns a contiguous quarter of the points: (A) per-wave cell counts, (B) per-cell exclusive prefix
  // over the waves on top of the cell's start, (C) every wave walks its quarter 64 points at a time and ranks the lanes
  // that share a cell with a compare mask (__ballot) -- lanes in ascending point order, waves in ascending ranges, so the
  // order is exactly that of the serial walk.
  __shared__ int s_wbase[4][MAX_CELLS + 1];
  const int lane = tid & 63, wv = tid >> 6;
  for (int c = tid; c < 4 * (MAX_CELLS + 1); c += 256) (&s_wbase[0][0])[c] = 0;
  __syncthreads();
  const int per_wave = ((a.P + 255) / 256) * 64;              // points per wave, a multiple of 64
  const int pw0 = wv * per_wave;
  for (int p = pw0 + lane; p < pw0 + per_wave && p < a.P; p += 64) {
    const int cid = s_cell[p];
    if (cid >= 0) atomicAdd(&s_wbase[wv][cid], 1);
  }
  __syncthreads();
  for (int c = tid; c < K; c += 256) {
    int base = s_cnt[c];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int n = s_wbase[w][c];
      s_wbase[w][c] = base;
      base += n;
    }
  }
  __syncthreads();
  int* dst = a.order + (size_t)b * a.P;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));     // lanes below this one
  for (int p0 = pw0; p0 < pw0 + per_wave && p0 < a.P; p0 += 64) {
    const int p = p0 + lane;
    const int cid = p < a.P ? (int)s_cell[p] : -1;
    unsigned long long todo = __ballot(cid >= 0);
    while (todo) {                                              // one round per distinct cell among the 64 points
      const int leader = __builtin_ctzll(todo);
      const int c0 = __shfl(cid, leader, 64);
      const unsigned long long same = __ballot(cid == c0);
      const int base = s_wbase[wv][c0];                         // (broadcast read)
      if (cid == c0) dst[base + __builtin_popcountll(same & lt)] = p;
      if (lane == leader) s_wbase[wv][c0] = base + __builtin_popcountll(same);
      todo &= ~same;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Kernel B: one workgroup per (sample, cell): mean of the cell's feature rows + semantic pooling.
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void bev_splat_mean_kernel(const TI* __restrict__ feat, const int* __restrict__ order,
                                                             const int* __restrict__ cell_start, TO* __restrict__ out,
                                                             int P, int K, int C,
                                                             const uint8_t* __restrict__ sem_ids,   // (B,P) or null
                                                             const double* __restrict__ sem_dense,  // (B,P,S) or null
                                                             int S, uint8_t* __restrict__ out_sem,  // (B,K,S)
                                                             uint8_t* __restrict__ out_sem_mask,    // (B,K)
                                                             const int* __restrict__ sample_rows,   // (B,R) or null
                                                             int R) {
  const int cellg = blockIdx.x;  // b*K + cell
  const int b = cellg / K, cell = cellg - b * K;
  // Where sample b's points live in the feature / semantic-id arrays: row b itself, or -- zero-copy batches drawn from
  // a device-resident feature store -- R store rows of P0 = P / R points each (pre-training: the sample's viewpoint;
  // fine-tuning: the current viewpoint and its visited neighbours, in the order the reference concatenates them)
  const int P0 = P / R;
  const int* rows = sample_rows != nullptr ? sample_rows + (size_t)b * R : nullptr;
  auto src_point = [&](int p) -> size_t {      // global point index of sample-local point p
    if (rows == nullptr) return (size_t)b * P + p;
    if (R == 1) return (size_t)rows[0] * P0 + p;
    const int r = p / P0;
    return (size_t)rows[r] * P0 + (p - r * P0);
  };
  const int s0 = cell_start[(size_t)b * (K + 1) + cell];
  const int n = cell_start[(size_t)b * (K + 1) + cell + 1] - s0;
  const int* ord = order + (size_t)b * P + s0;
  const float inv_denominator = (float)(n > 1 ? n : 1);

  for (int c4 = threadIdx.x; c4 * 4 < C; c4 += blockDim.x) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = 0;
    for (; i + 4 <= n; i += 4) {  // 4 independent row loads in flight, added in point order
      const int p0 = ord[i], p1 = ord[i + 1], p2 = ord[i + 2], p3 = ord[i + 3];
      const float4 v0 = ld4<TI>(feat + src_point(p0) * C + c4 * 4);
      const float4 v1 = ld4<TI>(feat + src_point(p1) * C + c4 * 4);
      const float4 v2 = ld4<TI>(feat + src_point(p2) * C + c4 * 4);
      const float4 v3 = ld4<TI>(feat + src_point(p3) * C + c4 * 4);
      acc.x = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(acc.x, v0.x), v1.x), v2.x), v3.x);
      acc.y = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(acc.y, v0.y), v1.y), v2.y), v3.y);
      acc.z = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(acc.z, v0.z), v1.z), v2.z), v3.z);
      acc.w = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(acc.w, v0.w), v1.w), v2.w), v3.w);
    }
    for (; i < n; ++i) {
      const float4 v = ld4<TI>(feat + src_point(ord[i]) * C + c4 * 4);
      acc.x = __fadd_rn(acc.x, v.x); acc.y = __fadd_rn(acc.y, v.y);
      acc.z = __fadd_rn(acc.z, v.z); acc.w = __fadd_rn(acc.w, v.w);
    }
    acc.x = __fdiv_rn(acc.x, inv_denominator); acc.y = __fdiv_rn(acc.y, inv_denominator);
    acc.z = __fdiv_rn(acc.z, inv_denominator); acc.w = __fdiv_rn(acc.w, inv_denominator);
    st4<TO>(out + (size_t)cellg * C + c4 * 4, acc);
  }

  if (out_sem != nullptr) {
    int any = 0;
    for (int c = threadIdx.x; c < S; c += blockDim.x) {
      uint8_t flag = 0;
      if (sem_ids != nullptr) {
        for (int i = 0; i < n; ++i) flag |= (uint8_t)(sem_ids[src_point(ord[i])] == (uint8_t)c);
      } else {
        // dense (one-hot, fp64 in the reference): mean > 0 -> 1 (bev_utils.py:417-422)
        const double* sb = sem_dense + (size_t)b * P * S;
        double acc = 0.0;
        for (int i = 0; i < n; ++i) acc += sb[(size_t)ord[i] * S + c];
        flag = (uint8_t)(acc / (double)(n > 1 ? n : 1) > 0.0);
      }
      out_sem[(size_t)cellg * S + c] = flag;
      any |= flag;
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) out_sem_mask[cellg] = (uint8_t)(any != 0);
  }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
BEVBERT_API int bevbert_bev_lift_bin(const float* depths, const float* T_c2w, const float* T_w2c, const float* S_w2c,
                                     const float* pix_scale, int B, int V, int hw, float depth_scale, int dim,
                                     float res, float y_clip, int* cell, int* order, int* cell_start,
                                     hipStream_t stream) {
  const int P = V * hw * hw;
  BB_REQUIRE(B > 0 && P > 0 && P <= MAX_POINTS, "bev_lift_bin: P=%d out of range (max %d)", P, MAX_POINTS);
  BB_REQUIRE(dim > 0 && dim * dim <= MAX_CELLS, "bev_lift_bin: dim=%d too large", dim);
  LiftArgs a;
  memset(&a, 0, sizeof(a));
  a.depths = depths; a.T_c2w = T_c2w; a.T_w2c = T_w2c; a.S_w2c = S_w2c; a.pix = pix_scale;
  a.V = V; a.hw = hw; a.depth_scale = depth_scale; a.P = P; a.dim = dim; a.res = res;
  a.half = (float)(dim - 1) / 2.0f; a.y_clip = y_clip;
  a.cell = cell; a.order = order; a.cell_start = cell_start;
  hipLaunchKernelGGL(bev_bin_sort_kernel<0>, dim3(B), dim3(256), 0, stream, a);
  BB_CHECK_LAUNCH("bev_lift_bin");
  return BB_OK;
}

BEVBERT_API int bevbert_bev_bin_points(const float* points, const uint8_t* drop_mask, int B, int P, int dim, float res,
                                       float y_clip, int* cell, int* order, int* cell_start, hipStream_t stream) {
  BB_REQUIRE(B > 0 && P > 0 && P <= MAX_POINTS, "bev_bin_points: P=%d out of range (max %d)", P, MAX_POINTS);
  BB_REQUIRE(dim > 0 && dim * dim <= MAX_CELLS, "bev_bin_points: dim=%d too large", dim);
  LiftArgs a;
  memset(&a, 0, sizeof(a));
  a.points = points; a.pmask = drop_mask; a.P = P; a.dim = dim; a.res = res;
  a.half = (float)(dim - 1) / 2.0f; a.y_clip = y_clip;
  a.cell = cell; a.order = order; a.cell_start = cell_start;
  hipLaunchKernelGGL(bev_bin_sort_kernel<1>, dim3(B), dim3(256), 0, stream, a);
  BB_CHECK_LAUNCH("bev_bin_points");
  return BB_OK;
}

template <typename TI, typename TO>
static int launch_splat(const void* feat, const int* order, const int* cell_start, void* out, int B, int P, int K,
                        int C, const uint8_t* sem_ids, const double* sem_dense, int S, uint8_t* out_sem,
                        uint8_t* out_sem_mask, const int* sample_rows, int R, hipStream_t stream) {
  const int threads = (C / 4 >= 192) ? 192 : ((C / 4 + 63) / 64) * 64;
  hipLaunchKernelGGL((bev_splat_mean_kernel<TI, TO>), dim3(B * K), dim3(threads < 64 ? 64 : threads), 0, stream,
                     (const TI*)feat, order, cell_start, (TO*)out, P, K, C, sem_ids, sem_dense, S, out_sem,
                     out_sem_mask, sample_rows, R);
  BB_CHECK_LAUNCH("bev_splat_mean");
  return BB_OK;
}

BEVBERT_API int bevbert_bev_splat_mean(const void* feat, int feat_dtype, const int* order, const int* cell_start,
                                       void* out, int out_dtype, int B, int P, int K, int C, const uint8_t* sem_ids,
                                       const double* sem_dense, int S, uint8_t* out_sem, uint8_t* out_sem_mask,
                                       const int* sample_rows, int rows_per_sample, hipStream_t stream) {
  BB_REQUIRE(C % 4 == 0, "bev_splat_mean: C=%d must be a multiple of 4", C);
  const int R = rows_per_sample < 1 ? 1 : rows_per_sample;
  BB_REQUIRE(R == 1 || sample_rows != nullptr, "bev_splat_mean: rows_per_sample=%d needs sample_rows", R);
  BB_REQUIRE(P % R == 0, "bev_splat_mean: P=%d is not a multiple of rows_per_sample=%d", P, R);
  BB_REQUIRE(sample_rows == nullptr || sem_dense == nullptr,
             "bev_splat_mean: sample_rows indexes feat / sem_ids stores; dense semantics are per batch");
  BB_REQUIRE(out_sem == nullptr || (sem_ids != nullptr) != (sem_dense != nullptr),
             "bev_splat_mean: exactly one of sem_ids / sem_dense when semantics are requested");
#define GO(TI, TO) \
  return launch_splat<TI, TO>(feat, order, cell_start, out, B, P, K, C, sem_ids, sem_dense, S, out_sem, out_sem_mask, sample_rows, R, stream)
  if (feat_dtype == BB_F32 && out_dtype == BB_F32) GO(float, float);
  if (feat_dtype == BB_F32 && out_dtype == BB_BF16) GO(float, bf16_raw);
  if (feat_dtype == BB_BF16 && out_dtype == BB_F32) GO(bf16_raw, float);
  if (feat_dtype == BB_BF16 && out_dtype == BB_BF16) GO(bf16_raw, bf16_raw);
  if (feat_dtype == BB_F16 && out_dtype == BB_F32) GO(_Float16, float);
  if (feat_dtype == BB_F16 && out_dtype == BB_BF16) GO(_Float16, bf16_raw);
  if (feat_dtype == BB_F16 && out_dtype == BB_F16) GO(_Float16, _Float16);
#undef GO
  bb_set_error("bev_splat_mean: unsupported dtype pair (%d -> %d)", feat_dtype, out_dtype);
  return BB_EUNSUPPORTED;
}
