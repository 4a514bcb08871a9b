// Fine-tune rollout bookkeeping on the device.
//
// The reference keeps one GraphMap per episode -- Python dicts and a dict-of-dicts Floyd graph
// (map_nav_src/models/graph_utils.py:44-94,96-189) -- and rebuilds the navigation inputs of every step with nested Python
// loops (map_nav_src/r2r/agent.py:194-337).  Here the B maps of a rollout are dense device arrays over a fixed node
// capacity N, one workgroup per episode:
//
//   pos (B,N,3) f64 | dis (B,N,N) f64 | point (B,N,N) i32 (next hop, -1 = direct edge) | hops (B,N,N) i32
//   visited (B,N) u8 | step_ids (B,N) i32 | pc_list (B,N) i32 + npc (B): visited nodes in first-visit order
//   node_row (B,N) i32: feature-store row of a visited node | node_T (B,N,V,16) f32: its camera poses
//
// What the host still does is what is host data: viewpoint-id strings -> node indices, and the order in which the nodes of
// a map are presented (it needs the id lists of that order anyway).  Everything numeric -- edge distances, the min-plus
// relaxation through the current viewpoint, hop counts, pair distances, position features, the choice of the visited
// 1-hop neighbours whose grid features feed the BEV -- happens here.  Integer and f64 add / compare / sqrt results are
// bit-equal to the reference's Python floats: edge lengths arrive from the host (numpy float64, the reference's own
// arithmetic), the device only adds and compares them.  The position features (sqrt / asin / sin / cos of the device math
// library, no FMA contraction) are within 2 ulp of numpy's.
#include "common.h"

#define GM_INF 95959595.0      // graph_utils.py:46

struct GmState {               // mirrors bevbert_gm_state (include/bevbert_hip.h)
  double* pos;
  double* dis;
  int* point;
  int* hops;
  uint8_t* visited;
  int* step_ids;
  int* pc_list;
  int* npc;
  int* node_row;
  float* node_T;
  int B, N, V, pad;
};

// GraphMap.update_graph (graph_utils.py:109-115) + FloydGraph.add_edge / update (:55-72) for every live episode,
// agent.py:471-474 (step ids), GraphMap.update_node_pc's bookkeeping (which store row / poses a visited node has), and the
// hop counts len(FloydGraph.path(x, y)) of the new graph.
__global__ __launch_bounds__(256) void gm_update_kernel(GmState s, const uint8_t* __restrict__ live_graph,
                                                        const uint8_t* __restrict__ live_step, const int* __restrict__ cur,
                                                        const int* __restrict__ ncand, const int* __restrict__ cand,
                                                        const double* __restrict__ cur_pos, const double* __restrict__ cand_pos,
                                                        const double* __restrict__ cand_dist,
                                                        const int* __restrict__ n_nodes, int C, int step_id,
                                                        const int* __restrict__ row, const float* __restrict__ T) {
  const int b = blockIdx.x, tid = threadIdx.x, N = s.N;
  const int k = cur[b], nb = n_nodes[b];
  double* dis = s.dis + (size_t)b * N * N;
  int* point = s.point + (size_t)b * N * N;
  int* hops = s.hops + (size_t)b * N * N;
  double* pos = s.pos + (size_t)b * N * 3;
  if (live_graph[b]) {
    if (tid < 3) pos[k * 3 + tid] = cur_pos[b * 3 + tid];
    for (int j = tid; j < ncand[b]; j += 256) {
      const int jn = cand[b * C + j];
      const double* p = cand_pos + ((size_t)b * C + j) * 3;
      pos[jn * 3] = p[0]; pos[jn * 3 + 1] = p[1]; pos[jn * 3 + 2] = p[2];
      // edge length: computed by the host in float64 exactly as the reference does (graph_utils.py:8-13) -- the device
      // math library's f64 sqrt is not guaranteed to round like numpy's, and every later distance is a SUM of these
      const double d = cand_dist[(size_t)b * C + j];
      if (d < dis[k * N + jn]) {
        dis[k * N + jn] = d; dis[jn * N + k] = d;
        point[k * N + jn] = -1; point[jn * N + k] = -1;
      }
    }
    __syncthreads();
    // relax every pair through k.  Row / column k cannot change (dis[k][k] stays "infinite"), so the pairs are independent
    for (int idx = tid; idx < nb * nb; idx += 256) {
      const int i = idx / nb, j = idx - i * nb;
      if (i == j) continue;
      const double via = dis[i * N + k] + dis[k * N + j];
      if (via < dis[i * N + j]) {
        dis[i * N + j] = via;
        point[i * N + j] = k;
      }
    }
    if (tid == 0) s.visited[(size_t)b * N + k] = 1;
  }
  if (live_step[b]) {
    if (tid == 0 && step_id > 0) s.step_ids[(size_t)b * N + k] = step_id;
    if (row != nullptr && row[b] >= 0) {         // remember_views: store row + camera poses; first visit fixes the order
      if (tid == 0) {
        s.node_row[(size_t)b * N + k] = row[b];
        const int np = s.npc[b];
        bool seen = false;
        for (int i = 0; i < np; ++i) seen |= s.pc_list[(size_t)b * N + i] == k;
        if (!seen) {
          s.pc_list[(size_t)b * N + np] = k;
          s.npc[b] = np + 1;
        }
      }
      for (int i = tid; i < s.V * 16; i += 256) s.node_T[((size_t)b * N + k) * s.V * 16 + i] = T[(size_t)b * s.V * 16 + i];
    }
  }
  if (!live_graph[b]) return;
  __syncthreads();
  // hop counts from the next-hop table the way the reference's recursion reads it: path(x, y) = path(x, k) + path(k, y)
  // with k = point[x][y]; a direct edge (or no known path) is one hop.  Bottom-up to the unique fixpoint.
  for (int idx = tid; idx < nb * nb; idx += 256) {
    const int i = idx / nb, j = idx - i * nb;
    hops[i * N + j] = i == j ? 0 : (point[i * N + j] < 0 ? 1 : -1);
  }
  __syncthreads();
  for (int round = 0; round <= nb; ++round) {
    int pending = 0;
    for (int idx = tid; idx < nb * nb; idx += 256) {
      const int i = idx / nb, j = idx - i * nb;
      if (hops[i * N + j] >= 0) continue;
      const int m = point[i * N + j];
      const int a = ((volatile int*)hops)[i * N + m], c = ((volatile int*)hops)[m * N + j];
      if (a >= 0 && c >= 0) hops[i * N + j] = a + c;
      else pending = 1;
    }
    if (!__syncthreads_or(pending)) break;
  }
}

// graph_utils.py:16-42,149-172 (calculate_vp_rel_pos_fts, get_angle_fts, get_pos_fts) for one (origin, target) pair
__device__ __forceinline__ void gm_pos_fts(const double* __restrict__ pos, const double* __restrict__ dis,
                                           const int* __restrict__ hops, int N, int cur, int tgt, double heading,
                                           double elevation, float* __restrict__ out) {
#pragma clang fp contract(off)
  const double ax = pos[cur * 3], ay = pos[cur * 3 + 1], az = pos[cur * 3 + 2];
  const double bx = pos[tgt * 3], by = pos[tgt * 3 + 1], bz = pos[tgt * 3 + 2];
  const double dx = bx - ax, dy = by - ay, dz = bz - az;
  const double xx = dx * dx, yy = dy * dy, zz = dz * dz;
  const double sxy = xx + yy;
  const double xy = fmax(__dsqrt_rn(sxy), 1e-8), xyz = fmax(__dsqrt_rn(sxy + zz), 1e-8);
  double h = asin(dx / xy);
  if (by < ay) h = 3.141592653589793 - h;
  h = h - heading;
  const double e = asin(dz / xyz) - elevation;
  const float hf = (float)h, ef = (float)e;
  out[0] = sinf(hf); out[1] = cosf(hf); out[2] = sinf(ef); out[3] = cosf(ef);
  const double gd = cur == tgt ? 0.0 : dis[cur * N + tgt];
  out[4] = (float)(xyz / 30.0);
  out[5] = (float)(gd / 30.0);
  out[6] = (float)((double)hops[cur * N + tgt] / 10.0);
}

// agent.py:194-276 (_nav_gmap_variable) for the node order the host chose: row 0 is [stop], rows 1..cnt[b] the map nodes
// node[b, 0..cnt[b]), the rest padding.  Also the position features of the start viewpoint (agent.py:326-331).
__global__ __launch_bounds__(256) void gm_nav_vars_kernel(GmState s, const int* __restrict__ node, const int* __restrict__ cnt,
                                                          const int* __restrict__ cur, const int* __restrict__ start,
                                                          const double* __restrict__ heading,
                                                          const double* __restrict__ elevation, int G, int enc_full_graph,
                                                          int act_visited, int64_t* __restrict__ step_ids,
                                                          uint8_t* __restrict__ visited, uint8_t* __restrict__ masks,
                                                          float* __restrict__ pair, float* __restrict__ posf,
                                                          float* __restrict__ gpos) {
  const int b = blockIdx.x, tid = threadIdx.x, N = s.N, G1 = G - 1;
  const double* dis = s.dis + (size_t)b * N * N;
  const int* hops = s.hops + (size_t)b * N * N;
  const double* pos = s.pos + (size_t)b * N * 3;
  const int c = cnt[b], k = cur[b];
  const int* nd = node + (size_t)b * G1;
  for (int j = tid; j < G; j += 256) {
    const bool real = j >= 1 && j - 1 < c;
    const int m = real ? nd[j - 1] : 0;
    masks[(size_t)b * G + j] = j < c + 1;
    const bool vis = act_visited ? m == k : s.visited[(size_t)b * N + m] != 0;
    visited[(size_t)b * G + j] = real && enc_full_graph && vis;
    step_ids[(size_t)b * G + j] = real ? s.step_ids[(size_t)b * N + m] : 0;
    float* o = posf + ((size_t)b * G + j) * 7;
    if (real) gm_pos_fts(pos, dis, hops, N, k, m, heading[b], elevation[b], o);
    else {
      const float on = j < c + 1 ? 1.f : 0.f;       // [stop]: angle features of (0, 0); padding: zeros
      o[0] = 0.f; o[1] = on; o[2] = 0.f; o[3] = on; o[4] = 0.f; o[5] = 0.f; o[6] = 0.f;
    }
  }
  for (int idx = tid; idx < G * G; idx += 256) {
    const int i = idx / G, j = idx - i * G;
    float v = 0.f;
    if (i >= 1 && j >= 1 && i - 1 < c && j - 1 < c && i != j) v = (float)(dis[nd[i - 1] * N + nd[j - 1]] / 30.0);
    pair[(size_t)b * G * G + idx] = v;
  }
  if (tid == 0 && gpos != nullptr) gm_pos_fts(pos, dis, hops, N, k, start[b], heading[b], elevation[b], gpos + (size_t)b * 7);
}

// GraphMap.gather_node_pc's node selection (graph_utils.py:129-144): the visited nodes within `order` hops of the current
// viewpoint, in first-visit order; rows padded to R by repeating the first row with live = 0.
__global__ __launch_bounds__(64) void gm_bev_select_kernel(GmState s, const int* __restrict__ cur, int order, int R,
                                                           int* __restrict__ rows, uint8_t* __restrict__ live,
                                                           float* __restrict__ T_c2w, int* __restrict__ overflow) {
  const int b = blockIdx.x, lane = threadIdx.x, N = s.N, k = cur[b], TV = s.V * 16;
  __shared__ int sel[64], s_n;
  const int* hops = s.hops + (size_t)b * N * N;
  const int np = s.npc[b];
  int n = 0;
  for (int base = 0; base < np; base += 64) {
    const int i = base + lane;
    const int c = i < np ? s.pc_list[(size_t)b * N + i] : -1;
    const bool take = c >= 0 && (order == 0 ? c == k : hops[k * N + c] <= order);
    const unsigned long long bal = __ballot(take);
    const int at = n + __popcll(bal & ((1ull << lane) - 1ull));
    if (take && at < 64 && at < R) sel[at] = c;
    n += __popcll(bal);
  }
  if (lane == 0) {
    if (n > R || n > 64) *overflow = 1;
    s_n = min(n, min(R, 64));
  }
  __syncthreads();
  n = s_n;
  const int first = n > 0 ? s.node_row[(size_t)b * N + sel[0]] : 0;
  for (int r = lane; r < R; r += 64) {
    rows[(size_t)b * R + r] = r < n ? s.node_row[(size_t)b * N + sel[r]] : first;
    live[(size_t)b * R + r] = r < n;
  }
  for (int r = 0; r < R; ++r) {
    const float* src = r < n ? s.node_T + ((size_t)b * N + sel[r]) * TV : nullptr;
    for (int i = lane; i < TV; i += 64) T_c2w[((size_t)b * R + r) * TV + i] = src ? src[i] : 0.f;
  }
}

// agent.py:485-494 for inference rollouts: the current viewpoint's embedding slot is REWRITTEN with the panorama mean, every
// not-yet-visited candidate ACCUMULATES the embedding of the view it was seen in (running mean = sum / count).  Slot = node
// index.  grid (B, 1 + C): y = 0 -> the current viewpoint, y = j + 1 -> candidate j.  Sums are kept in the tensor's dtype
// (bf16 sums round after every addition, exactly like the torch index_put path used when gradients are required).
template <typename T>
__global__ __launch_bounds__(256) void gm_embed_update_kernel(GmState s, T* __restrict__ embed_sum, float* __restrict__ embed_cnt,
                                                              const T* __restrict__ avg, const T* __restrict__ pano,
                                                              const uint8_t* __restrict__ live, const int* __restrict__ cur,
                                                              const int* __restrict__ ncand, const int* __restrict__ cand,
                                                              int C, int V, int H) {
  const int b = blockIdx.x, j = (int)blockIdx.y - 1, N = s.N;
  if (!live[b]) return;
  if (j < 0) {
    const int k = cur[b];
    T* dst = embed_sum + ((size_t)b * N + k) * H;
    for (int c = threadIdx.x * 4; c < H; c += 1024) st4<T>(dst + c, ld4<T>(avg + (size_t)b * H + c));
    if (threadIdx.x == 0) embed_cnt[(size_t)b * N + k] = 1.f;
    return;
  }
  if (j >= ncand[b] || j >= V) return;
  const int m = cand[b * C + j];
  if (m < 0 || s.visited[(size_t)b * N + m]) return;
  T* dst = embed_sum + ((size_t)b * N + m) * H;
  const T* src = pano + ((size_t)b * V + j) * H;
  for (int c = threadIdx.x * 4; c < H; c += 1024) {
    float4 a = ld4<T>(dst + c);
    const float4 v = ld4<T>(src + c);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    st4<T>(dst + c, a);
  }
  if (threadIdx.x == 0) embed_cnt[(size_t)b * N + m] += 1.f;
}

// GraphMap.get_node_embed (graph_utils.py:146-147) for the listed nodes: out[b, 0] = 0 ([stop]), out[b, j] = sum / count of
// node[b, j - 1] for j - 1 < cnt[b] (0 where nothing has been stored), padding rows 0.
template <typename T>
__global__ __launch_bounds__(256) void gm_node_embeds_kernel(GmState s, const T* __restrict__ embed_sum,
                                                             const float* __restrict__ embed_cnt, const int* __restrict__ node,
                                                             const int* __restrict__ cnt, int G, int H, T* __restrict__ out) {
  const int b = blockIdx.x, j = blockIdx.y, N = s.N;
  T* dst = out + ((size_t)b * G + j) * H;
  float n = 0.f;
  int m = 0;
  if (j >= 1 && j - 1 < cnt[b]) {
    m = node[(size_t)b * (G - 1) + j - 1];
    n = embed_cnt[(size_t)b * N + m];
  }
  const T* src = embed_sum + ((size_t)b * N + m) * H;
  for (int c = threadIdx.x * 4; c < H; c += 1024) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n > 0.f) {
      a = ld4<T>(src + c);
      // the torch path divides in the tensor's dtype by the count cast to that dtype: one rounding of the quotient
      a.x = a.x / n; a.y = a.y / n; a.z = a.z / n; a.w = a.w / n;
    }
    st4<T>(dst + c, a);
  }
}

static int gm_check(const GmState* st, const char* what) {
  BB_REQUIRE(st != nullptr && st->B > 0 && st->N > 0 && st->V > 0, "%s: empty graph-map state", what);
  BB_REQUIRE(st->pos && st->dis && st->point && st->hops && st->visited && st->step_ids, "%s: null state array", what);
  return BB_OK;
}

BEVBERT_API int bevbert_gm_update(const GmState* st, const uint8_t* live_graph, const uint8_t* live_step, const int* cur,
                                  const int* ncand, const int* cand, const double* cur_pos, const double* cand_pos,
                                  const double* cand_dist, const int* n_nodes, int C, int step_id, const int* row,
                                  const float* T, hipStream_t stream) {
  if (int rc = gm_check(st, "gm_update")) return rc;
  BB_REQUIRE(row == nullptr || (T != nullptr && st->pc_list && st->npc && st->node_row && st->node_T),
             "gm_update: store rows need the pose table and the visit list");
  hipLaunchKernelGGL(gm_update_kernel, dim3(st->B), dim3(256), 0, stream, *st, live_graph, live_step, cur, ncand, cand,
                     cur_pos, cand_pos, cand_dist, n_nodes, C, step_id, row, T);
  BB_CHECK_LAUNCH("gm_update");
  return BB_OK;
}

BEVBERT_API int bevbert_gm_nav_vars(const GmState* st, const int* node, const int* cnt, const int* cur, const int* start,
                                    const double* heading, const double* elevation, int G, int enc_full_graph,
                                    int act_visited, int64_t* step_ids, uint8_t* visited, uint8_t* masks, float* pair,
                                    float* pos_fts, float* gpos, hipStream_t stream) {
  if (int rc = gm_check(st, "gm_nav_vars")) return rc;
  BB_REQUIRE(G >= 1, "gm_nav_vars: G=%d", G);
  hipLaunchKernelGGL(gm_nav_vars_kernel, dim3(st->B), dim3(256), 0, stream, *st, node, cnt, cur, start, heading, elevation,
                     G, enc_full_graph, act_visited, step_ids, visited, masks, pair, pos_fts, gpos);
  BB_CHECK_LAUNCH("gm_nav_vars");
  return BB_OK;
}

BEVBERT_API int bevbert_gm_bev_select(const GmState* st, const int* cur, int order, int R, int* rows, uint8_t* live,
                                      float* T_c2w, int* overflow, hipStream_t stream) {
  if (int rc = gm_check(st, "gm_bev_select")) return rc;
  BB_REQUIRE(R >= 1 && R <= 64, "gm_bev_select: R=%d outside 1..64", R);
  BB_REQUIRE(st->pc_list && st->npc && st->node_row && st->node_T, "gm_bev_select: no visit list in the state");
  hipLaunchKernelGGL(gm_bev_select_kernel, dim3(st->B), dim3(64), 0, stream, *st, cur, order, R, rows, live, T_c2w, overflow);
  BB_CHECK_LAUNCH("gm_bev_select");
  return BB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// views of the selected nodes out of the feature store: out[i] = live[i] ? store[rows[i]] : 0, rows of `row_bytes` bytes
// (agent.py:150-156 pads the point clouds of a batch with empty views; one launch for index_select + mask)
template <typename W>
__global__ __launch_bounds__(256) void gm_gather_views_kernel(const W* __restrict__ store, const int* __restrict__ rows,
                                                              const uint8_t* __restrict__ live, W* __restrict__ out,
                                                              int words, int chunks) {
  const int o = blockIdx.x / chunks, c = blockIdx.x % chunks;
  const bool ok = live[o] != 0;
  const W* s = store + (size_t)rows[o] * words;
  W* d = out + (size_t)o * words;
  const int per = (words + chunks - 1) / chunks, lo = c * per, hi = min(words, lo + per);
  W zero;
  __builtin_memset(&zero, 0, sizeof(W));
  for (int i = lo + threadIdx.x; i < hi; i += 256) d[i] = ok ? s[i] : zero;
}

BEVBERT_API int bevbert_gm_gather_views(const void* store, const int* rows, const uint8_t* live, void* out, int n_out,
                                        int64_t row_bytes, hipStream_t stream) {
  BB_REQUIRE(store && rows && live && out, "gm_gather_views: null pointer");
  BB_REQUIRE(n_out >= 0 && row_bytes > 0, "gm_gather_views: n_out=%d row_bytes=%lld", n_out, (long long)row_bytes);
  if (n_out == 0) return BB_OK;
  const int chunks = (int)std::min<int64_t>(8, (row_bytes + 4095) / 4096);      // >= 4 KiB per workgroup
  const uintptr_t bits = (uintptr_t)store | (uintptr_t)out | (uintptr_t)row_bytes;   // widest word that divides all three
#define GATHER(W)                                                                                                        \
  hipLaunchKernelGGL(gm_gather_views_kernel<W>, dim3(n_out * chunks), dim3(256), 0, stream, (const W*)store, rows, live, \
                     (W*)out, (int)(row_bytes / sizeof(W)), chunks)
  if (bits % 16 == 0) GATHER(uint4);
  else if (bits % 4 == 0) GATHER(uint32_t);
  else if (bits % 2 == 0) GATHER(uint16_t);
  else GATHER(uint8_t);
#undef GATHER
  BB_CHECK_LAUNCH("gm_gather_views");
  return BB_OK;
}

BEVBERT_API int bevbert_gm_embed_update(const GmState* st, void* embed_sum, float* embed_cnt, const void* avg,
                                        const void* pano, const uint8_t* live, const int* cur, const int* ncand,
                                        const int* cand, int C, int V, int H, int dtype, hipStream_t stream) {
  if (int rc = gm_check(st, "gm_embed_update")) return rc;
  BB_REQUIRE(H % 4 == 0 && C >= 1 && V >= 1, "gm_embed_update: H=%d C=%d V=%d", H, C, V);
  const dim3 grid(st->B, 1 + C);
  if (dtype == BB_F32)
    hipLaunchKernelGGL(gm_embed_update_kernel<float>, grid, dim3(256), 0, stream, *st, (float*)embed_sum, embed_cnt,
                       (const float*)avg, (const float*)pano, live, cur, ncand, cand, C, V, H);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL(gm_embed_update_kernel<bf16_raw>, grid, dim3(256), 0, stream, *st, (bf16_raw*)embed_sum, embed_cnt,
                       (const bf16_raw*)avg, (const bf16_raw*)pano, live, cur, ncand, cand, C, V, H);
  else {
    bb_set_error("gm_embed_update: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("gm_embed_update");
  return BB_OK;
}

BEVBERT_API int bevbert_gm_node_embeds(const GmState* st, const void* embed_sum, const float* embed_cnt, const int* node,
                                       const int* cnt, int G, int H, int dtype, void* out, hipStream_t stream) {
  if (int rc = gm_check(st, "gm_node_embeds")) return rc;
  BB_REQUIRE(H % 4 == 0 && G >= 1, "gm_node_embeds: H=%d G=%d", H, G);
  const dim3 grid(st->B, G);
  if (dtype == BB_F32)
    hipLaunchKernelGGL(gm_node_embeds_kernel<float>, grid, dim3(256), 0, stream, *st, (const float*)embed_sum, embed_cnt, node,
                       cnt, G, H, (float*)out);
  else if (dtype == BB_BF16)
    hipLaunchKernelGGL(gm_node_embeds_kernel<bf16_raw>, grid, dim3(256), 0, stream, *st, (const bf16_raw*)embed_sum, embed_cnt,
                       node, cnt, G, H, (bf16_raw*)out);
  else {
    bb_set_error("gm_node_embeds: dtype %d unsupported", dtype);
    return BB_EUNSUPPORTED;
  }
  BB_CHECK_LAUNCH("gm_node_embeds");
  return BB_OK;
}
