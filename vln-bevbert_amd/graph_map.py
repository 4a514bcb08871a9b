"""Fine-tune rollout bookkeeping, batched and device-resident.

The reference keeps one ``GraphMap`` per episode (map_nav_src/models/graph_utils.py:96-189): Python dicts of per-node
torch tensors (running sums of panorama embeddings, world-frame point clouds with their 2352 x 768 features), a
dict-of-dicts Floyd graph, and rebuilds the navigation inputs of every step with nested Python loops
(map_nav_src/r2r/agent.py:194-337).  Between the three model calls of a step this is what the fine-tune path waits on.

Here:
  * ``FloydGraph``: the same incremental all-pairs relaxation on dense numpy matrices (one vectorised update per visited
    node instead of a V^2 Python loop); distances, next-hop table and ``path`` reproduce the reference exactly.
  * ``GraphMapBatch``: all B episodes of a rollout.  Node embeddings live in ONE (B, cap, H) device tensor of running
    sums + a count tensor, updated functionally with two ``index_put`` per step (autograd flows through them across
    steps like it does through the reference's stored tensors); the per-step navigation inputs
    (``nav_gmap_variable``) are assembled with numpy fancy indexing and one device gather.
  * Point clouds are never stored: a node remembers its row in the device-resident ``feature_store.GridFeatureStore``
    and its camera poses; ``bev_inputs`` returns, per sample, the store rows of the current viewpoint and its visited
    neighbours (``pc_order`` hops) in the order the reference concatenates them, the depths gathered from the store and
    the per-view poses -- exactly what ``ops.bev_lift_bin`` + ``ops.bev_splat_mean(rows=...)`` consume.
Host logic is plain numpy / torch indexing (device agnostic); the kernels it feeds are the C-ABI ones.
"""
import numpy as np
import torch

from .synthetic import pose_matrix

MAX_DIST = 30       # graph_utils.py:5-6
MAX_STEP = 10
_INF = 95959595     # graph_utils.py:46: the reference's "no path yet"


def rel_pos_fts(a, b, base_heading=0.0, base_elevation=0.0):
    """graph_utils.py:16-34 calculate_vp_rel_pos_fts for many (origin, target) pairs at once: ``a`` (3,) or (n, 3)
    origins, ``b`` (n, 3) targets, scalar or (n,) base angles."""
    b = np.asarray(b, dtype=np.float64).reshape(-1, 3)
    a = np.broadcast_to(np.asarray(a, dtype=np.float64).reshape(-1, 3), b.shape)
    d = b - a
    xy = np.maximum(np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2), 1e-8)
    xyz = np.maximum(np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2 + d[:, 2] ** 2), 1e-8)
    heading = np.arcsin(d[:, 0] / xy)
    heading = np.where(b[:, 1] < a[:, 1], np.pi - heading, heading) - base_heading
    elevation = np.arcsin(d[:, 2] / xyz) - base_elevation
    return heading, elevation, xyz


def angle_fts(headings, elevations, angle_feat_size=4):
    """graph_utils.py:36-42 get_angle_fts."""
    f = np.stack([np.sin(headings), np.cos(headings), np.sin(elevations), np.cos(elevations)], 1).astype(np.float32)
    return np.concatenate([f] * (angle_feat_size // 4), 1) if angle_feat_size // 4 > 1 else f


class FloydGraph:
    """graph_utils.py:44-94 on dense matrices.  Nodes are registered on their first edge (like the reference's
    defaultdict keys) and keep their insertion order."""

    def __init__(self, capacity=32):
        self.index = {}
        self.names = []
        self._dis = np.full((capacity, capacity), float(_INF))
        self._point = np.full((capacity, capacity), -1, dtype=np.int32)     # -1: direct edge ("" in the reference)
        self._visited = set()

    def _idx(self, x):
        i = self.index.get(x)
        if i is None:
            i = self.index[x] = len(self.names)
            self.names.append(x)
            if i >= self._dis.shape[0]:
                n = 2 * self._dis.shape[0]
                dis = np.full((n, n), float(_INF))
                pt = np.full((n, n), -1, dtype=np.int32)
                dis[:i, :i], pt[:i, :i] = self._dis[:i, :i], self._point[:i, :i]
                self._dis, self._point = dis, pt
        return i

    def __len__(self):
        return len(self.names)

    def distance(self, x, y):
        if x == y:
            return 0
        i, j = self.index.get(x), self.index.get(y)
        return float(_INF) if i is None or j is None else float(self._dis[i, j])

    def add_edge(self, x, y, dis):
        i, j = self._idx(x), self._idx(y)
        if dis < self._dis[i, j]:
            self._dis[i, j] = self._dis[j, i] = dis
            self._point[i, j] = self._point[j, i] = -1

    def update(self, k):
        """Relax every pair through k (graph_utils.py:63-72).  Row / column k cannot change during the reference's
        double loop (the diagonal stays 'infinite'), so one vectorised comparison is the same sequence of updates."""
        kk = self._idx(k)
        n = len(self.names)
        d = self._dis[:n, :n]
        via = d[:, kk][:, None] + d[kk, :][None, :]
        better = via < d
        np.fill_diagonal(better, False)
        d[better] = via[better]
        self._point[:n, :n][better] = kk
        self._visited.add(k)

    def visited(self, k):
        return k in self._visited

    def path(self, x, y):
        if x == y:
            return []
        i, j = self.index[x], self.index[y]
        k = self._point[i, j]
        if k < 0:
            return [y]
        kn = self.names[k]
        return self.path(x, kn) + self.path(kn, y)

    def submatrix(self, names):
        """Distances between the named nodes, (len, len) float64; 0 on the diagonal like ``distance``."""
        idx = np.asarray([self.index[v] for v in names], dtype=np.int64)
        m = self._dis[np.ix_(idx, idx)].copy()
        np.fill_diagonal(m, 0.0)
        return m


class _Episode:
    def __init__(self, start_vp):
        self.start_vp = start_vp
        self.node_positions = {}        # insertion order = the reference's node_positions order
        self.graph = FloydGraph()
        self.node_slot = {}             # vp -> column of the batch's embedding buffers
        self.node_step_ids = {}
        self.pc_nodes = {}              # vp -> (store row, T_c2w (V,4,4) float32): visited nodes, in visit order


class GraphMapBatch:
    def __init__(self, start_vps, hidden_size, device, dtype=torch.float32, capacity=64):
        self.eps = [_Episode(vp) for vp in start_vps]
        self.B, self.H, self.device, self.dtype = len(start_vps), hidden_size, torch.device(device), dtype
        self.cap = capacity
        self.embed_sum = torch.zeros(self.B, capacity, hidden_size, dtype=dtype, device=self.device)
        self.embed_cnt = torch.zeros(self.B, capacity, dtype=torch.float32, device=self.device)

    # -- graph structure (host) ----------------------------------------------------------------------------------
    def update_graph(self, obs, ended=None):
        """GraphMap.update_graph for every live episode (graph_utils.py:109-115; agent.py:447-449,556-559)."""
        for i, ob in enumerate(obs):
            if ended is not None and ended[i]:
                continue
            ep = self.eps[i]
            ep.node_positions[ob["viewpoint"]] = ob["position"]
            for cc in ob["candidate"]:
                ep.node_positions[cc["viewpointId"]] = cc["position"]
                a, b = ob["position"], cc["position"]
                dist = np.sqrt((b[0] - a[0]) ** 2 + (b[1] - a[1]) ** 2 + (b[2] - a[2]) ** 2)
                ep.graph.add_edge(ob["viewpoint"], cc["viewpointId"], dist)
            ep.graph.update(ob["viewpoint"])

    def set_step_ids(self, obs, t, ended=None):
        """agent.py:471-474."""
        for i, ob in enumerate(obs):
            if ended is None or not ended[i]:
                self.eps[i].node_step_ids[ob["viewpoint"]] = t + 1

    def _slot(self, i, vp):
        ep = self.eps[i]
        s = ep.node_slot.get(vp)
        if s is None:
            s = ep.node_slot[vp] = len(ep.node_slot)
            if s >= self.cap:
                grow = self.cap
                self.embed_sum = torch.cat([self.embed_sum, self.embed_sum.new_zeros(self.B, grow, self.H)], 1)
                self.embed_cnt = torch.cat([self.embed_cnt, self.embed_cnt.new_zeros(self.B, grow)], 1)
                self.cap += grow
        return s

    # -- node embeddings (device) --------------------------------------------------------------------------------
    def update_node_embeds(self, obs, cand_vpids, avg_pano_embeds, pano_embeds, ended=None):
        """agent.py:485-494 for the whole batch: the current viewpoint's embedding is REWRITTEN with the panorama mean,
        every not-yet-visited candidate ACCUMULATES the embedding of the view it was seen in (running mean).
        Two functional index_put per step; gradients reach avg_pano_embeds / pano_embeds like in the reference."""
        rb, rs, ab, as_, aj = [], [], [], [], []
        for i, ob in enumerate(obs):
            if ended is not None and ended[i]:
                continue
            ep = self.eps[i]
            rb.append(i)
            rs.append(self._slot(i, ob["viewpoint"]))
            for j, vp in enumerate(cand_vpids[i]):
                if not ep.graph.visited(vp):
                    ab.append(i)
                    as_.append(self._slot(i, vp))
                    aj.append(j)
        if not rb:
            return
        dev = self.device
        t = lambda v: torch.tensor(v, dtype=torch.long, device=dev)
        rb_t, rs_t = t(rb), t(rs)
        self.embed_sum = self.embed_sum.index_put((rb_t, rs_t), avg_pano_embeds[rb_t].to(self.dtype))
        self.embed_cnt = self.embed_cnt.index_put((rb_t, rs_t), torch.ones(len(rb), device=dev))
        if ab:
            ab_t, as_t, aj_t = t(ab), t(as_), t(aj)
            self.embed_sum = self.embed_sum.index_put((ab_t, as_t), pano_embeds[ab_t, aj_t].to(self.dtype),
                                                      accumulate=True)
            self.embed_cnt = self.embed_cnt.index_put((ab_t, as_t), torch.ones(len(ab), device=dev), accumulate=True)

    def node_embed(self, i, vp):
        """GraphMap.get_node_embed (graph_utils.py:146-147)."""
        s = self.eps[i].node_slot[vp]
        return self.embed_sum[i, s] / self.embed_cnt[i, s]

    # -- per-step navigation inputs ------------------------------------------------------------------------------
    def pos_fts(self, i, cur_vp, vpids, cur_heading, cur_elevation, angle_feat_size=4):
        """GraphMap.get_pos_fts (graph_utils.py:149-172), vectorised over the nodes; None = the [stop] token."""
        ep = self.eps[i]
        out = np.zeros((len(vpids), angle_feat_size + 3), dtype=np.float32)
        real = [k for k, vp in enumerate(vpids) if vp is not None]
        ang = np.zeros((len(vpids), 2), dtype=np.float32)
        dists = np.zeros((len(vpids), 3), dtype=np.float32)
        if real:
            pos = np.asarray([ep.node_positions[vpids[k]] for k in real], dtype=np.float64)
            h, e, d = rel_pos_fts(ep.node_positions[cur_vp], pos, cur_heading, cur_elevation)
            ang[real, 0], ang[real, 1] = h, e
            dists[real, 0] = d / MAX_DIST
            dists[real, 1] = [ep.graph.distance(cur_vp, vpids[k]) / MAX_DIST for k in real]
            dists[real, 2] = [len(ep.graph.path(cur_vp, vpids[k])) / MAX_STEP for k in real]
        out[:, :angle_feat_size] = angle_fts(ang[:, 0], ang[:, 1], angle_feat_size)
        out[:, angle_feat_size:] = dists
        return out

    def pos_fts_batch(self, obs, vpid_lists, angle_feat_size=4):
        """pos_fts for every sample with ONE pass of numpy over all (sample, node) pairs: per-sample Python only walks
        the graphs (distances are matrix rows, hop counts come from the next-hop table)."""
        org, tgt, bh, be, gd, hops, where = [], [], [], [], [], [], []
        for i, (ob, g) in enumerate(zip(obs, vpid_lists)):
            ep = self.eps[i]
            cur = ob["viewpoint"]
            a = ep.node_positions[cur]
            for k, vp in enumerate(g):
                if vp is None:
                    continue
                org.append(a)
                tgt.append(ep.node_positions[vp])
                bh.append(ob["heading"])
                be.append(ob["elevation"])
                gd.append(ep.graph.distance(cur, vp))
                hops.append(len(ep.graph.path(cur, vp)))
                where.append((i, k))
        stop = angle_fts(np.zeros(1, np.float32), np.zeros(1, np.float32), angle_feat_size)   # angles (0, 0) -> (0, 1, 0, 1)
        outs = [np.zeros((len(g), angle_feat_size + 3), dtype=np.float32) for g in vpid_lists]
        for o in outs:
            o[:, :angle_feat_size] = stop
        if where:
            h, e, d = rel_pos_fts(np.asarray(org), np.asarray(tgt), np.asarray(bh), np.asarray(be))
            ang = angle_fts(h.astype(np.float32), e.astype(np.float32), angle_feat_size)
            dist = np.stack([d / MAX_DIST, np.asarray(gd) / MAX_DIST, np.asarray(hops) / MAX_STEP], 1).astype(np.float32)
            for r, (i, k) in enumerate(where):
                outs[i][k, :angle_feat_size] = ang[r]
                outs[i][k, angle_feat_size:] = dist[r]
        return outs

    def nav_gmap_variable(self, obs, enc_full_graph=True, act_visited_nodes=False, angle_feat_size=4):
        """agent.py:194-276 (_nav_gmap_variable): [stop] + map nodes per sample, padded to the batch maximum."""
        B = self.B
        vpids, visited, step_ids, pos, pair, slots, no_left = [], [], [], [], [], [], []
        for i, ob in enumerate(obs):
            ep = self.eps[i]
            vis, unvis = [], []
            for k in ep.node_positions.keys():
                is_vis = (k == ob["viewpoint"]) if act_visited_nodes else ep.graph.visited(k)
                (vis if is_vis else unvis).append(k)
            no_left.append(len(unvis) == 0)
            if enc_full_graph:
                g = [None] + vis + unvis
                m = [0] + [1] * len(vis) + [0] * len(unvis)
            else:
                g = [None] + unvis
                m = [0] * len(g)
            vpids.append(g)
            visited.append(m)
            step_ids.append([ep.node_step_ids.get(vp, 0) for vp in g])
            pd = np.zeros((len(g), len(g)), dtype=np.float32)
            if len(g) > 1:
                pd[1:, 1:] = (ep.graph.submatrix(g[1:]) / MAX_DIST).astype(np.float32)
            pair.append(pd)
            slots.append([-1] + [ep.node_slot[vp] for vp in g[1:]])
        pos = self.pos_fts_batch(obs, vpids, angle_feat_size)
        lens = np.asarray([len(g) for g in vpids])
        G = int(lens.max())
        pad = lambda rows, fill, dt: np.stack([np.concatenate([np.asarray(r, dtype=dt),
                                                               np.full((G - len(r),) + np.asarray(r).shape[1:], fill, dtype=dt)])
                                               for r in rows])
        slot_np = pad(slots, -1, np.int64)
        pair_np = np.zeros((B, G, G), dtype=np.float32)
        for i, pd in enumerate(pair):
            pair_np[i, :len(pd), :len(pd)] = pd
        dev = self.device
        slot_t = torch.from_numpy(slot_np).to(dev)
        valid = slot_t >= 0
        bi = torch.arange(B, device=dev)[:, None].expand(-1, G)
        si = slot_t.clamp(min=0)
        cnt = self.embed_cnt[bi, si].clamp(min=1.0).to(self.dtype)
        embeds = (self.embed_sum[bi, si] / cnt[..., None]) * valid[..., None].to(self.dtype)    # [stop] / padding = 0
        return {
            "gmap_vpids": vpids, "gmap_img_embeds": embeds,
            "gmap_step_ids": torch.from_numpy(pad(step_ids, 0, np.int64)).to(dev),
            "gmap_pos_fts": torch.from_numpy(pad(pos, 0, np.float32)).to(dev),
            "gmap_visited_masks": torch.from_numpy(pad(visited, 0, np.int64).astype(bool)).to(dev),
            "gmap_visited_masks_cpu": torch.from_numpy(pad(visited, 0, np.int64).astype(bool)),
            "gmap_pair_dists": torch.from_numpy(pair_np).to(dev),
            "gmap_masks": torch.from_numpy(np.arange(G)[None] < lens[:, None]).to(dev),
            "no_vp_left": no_left,
        }

    # -- BEV inputs: store rows instead of stored point clouds ---------------------------------------------------------
    def remember_views(self, obs, store_keys, store, ended=None, views=12):
        """The reference stores every visited node's world-frame point cloud and 2352 x 768 features
        (GraphMap.update_node_pc, agent.py:488).  Here a node keeps its feature-store row and its 12 camera poses
        (agent.py:114-126: position (x, z, -y), heading -(k * 30 deg + ob heading), elevation pi)."""
        live = [i for i in range(len(obs)) if ended is None or not ended[i]]
        if not live:
            return
        xyzhe = np.zeros((len(live), views, 5))             # float64 like the agent's; the matrices are cast to fp32
        for r, i in enumerate(live):
            x, y, z = obs[i]["position"]
            xyzhe[r, :, 0], xyzhe[r, :, 1], xyzhe[r, :, 2] = x, z, -y
            xyzhe[r, :, 3] = -(np.arange(views) * np.radians(30) + obs[i]["heading"])
        xyzhe[:, :, 4] = np.pi
        T = pose_matrix(xyzhe.reshape(-1, 5)).reshape(len(live), views, 4, 4)
        for r, i in enumerate(live):
            self.eps[i].pc_nodes[obs[i]["viewpoint"]] = (store.row[store_keys[i]], T[r])

    def gather_nodes(self, i, vp, order):
        """GraphMap.gather_node_pc's node selection (graph_utils.py:129-144): visited nodes within `order` hops of vp,
        in the order they were first stored (dict order) -- the concatenation order of the reference's point cloud."""
        ep = self.eps[i]
        if order == 0:
            return [vp]
        return [c for c in ep.pc_nodes.keys() if len(ep.graph.path(vp, c)) <= order]

    def bev_inputs(self, obs, store, pc_order=1, bev_dim=21, bev_res=0.5):
        """agent.py:143-192,282-337 (splat + _nav_bev_variable) as inputs of the fused kernels: per sample the R store
        rows (current viewpoint + visited neighbours, padded by repeating the first row with zero depths), the matching
        depths and per-view poses, the world->ego transform of the current pose, candidate cells and nav masks."""
        B = self.B
        nodes = [self.gather_nodes(i, ob["viewpoint"], pc_order) for i, ob in enumerate(obs)]
        R = max(len(n) for n in nodes)
        V = store.V
        rows = np.zeros((B, R), dtype=np.int32)
        T_c2w = np.zeros((B, R * V, 4, 4), dtype=np.float32)
        live = np.zeros((B, R), dtype=bool)
        for i, ns in enumerate(nodes):
            for r, vp in enumerate(ns):
                rows[i, r], T_c2w[i, r * V:(r + 1) * V] = self.eps[i].pc_nodes[vp]
                live[i, r] = True
            rows[i, len(ns):] = rows[i, 0]
        dev = self.device
        rows_t = torch.from_numpy(rows).to(dev)
        depths = store.depths.index_select(0, rows_t.reshape(-1).long()).reshape(B, R * V, store.hw, store.hw)
        depths = depths * torch.from_numpy(live).to(dev).repeat_interleave(V, 1)[..., None, None]   # padding: no depth
        S = np.asarray([[ob["position"][0], ob["position"][2], -ob["position"][1]] for ob in obs], dtype=np.float32)
        xyzhe = np.zeros((B, 5))
        xyzhe[:, 3] = [ob["heading"] for ob in obs]
        K = bev_dim * bev_dim
        cand_vpids = [[None] + [c["viewpointId"] for c in ob["candidate"]] for ob in obs]
        cells = self.cand_cells_batch(obs, bev_dim, bev_res)
        C = 1 + max(len(c) for c in cells)
        cand_np = np.zeros((B, C), dtype=np.int64)
        nav_masks = np.zeros((B, K), dtype=bool)
        for i, c in enumerate(cells):
            cand_np[i, 0] = (K - 1) // 2                      # [stop]: the centre cell (agent.py:318)
            cand_np[i, 1:1 + len(c)] = c
            nav_masks[i, cand_np[i, :1 + len(c)]] = True
        gpos = [g[0] for g in self.pos_fts_batch(obs, [[ep.start_vp] for ep in self.eps])]
        return {
            "grid_rows": rows_t, "depths": depths, "T_c2w": torch.from_numpy(T_c2w).to(dev),
            "T_w2c": torch.from_numpy(pose_matrix(xyzhe)).to(dev)[:, None], "S_w2c": torch.from_numpy(S).to(dev)[:, None],
            "bev_nav_masks": torch.from_numpy(nav_masks).to(dev), "bev_cand_idxs": torch.from_numpy(cand_np).to(dev),
            "bev_cand_vpids": cand_vpids, "bev_gpos_fts": torch.from_numpy(np.stack(gpos)).to(dev)[:, None],
        }

    @staticmethod
    def cand_cells_batch(obs, bev_dim, bev_res):
        """cand_cells for every sample: the poses of the whole batch come from one pose_matrix call; the 4-term
        products stay per-sample numpy matmuls (same BLAS path, hence the same roundings, as the reference's np.dot)."""
        flip = np.array([1, 1, -1], dtype=np.float32)
        xyzhe = np.zeros((len(obs), 5))
        xyzhe[:, 3] = [-ob["heading"] for ob in obs]
        T = pose_matrix(xyzhe)
        out = []
        for i, ob in enumerate(obs):
            if not ob["candidate"]:
                out.append(np.zeros(0, dtype=np.int64))
                continue
            S = np.asarray(ob["position"], dtype=np.float32)[[0, 2, 1]] * flip
            p = np.asarray([c["position"] for c in ob["candidate"]], dtype=np.float32)[:, [0, 2, 1]] * flip - S
            p1 = np.concatenate([p, np.ones((p.shape[0], 1), dtype=np.float32)], -1) @ T[i]    # see cand_cells
            c = np.clip(np.round(p1[:, [0, 2]] / bev_res) + (bev_dim - 1) // 2, 0, bev_dim - 1).astype(np.int64)
            out.append(c[:, 1] * bev_dim + c[:, 0])
        return out

    @staticmethod
    def cand_cells(ob, bev_dim, bev_res):
        """agent.py:278-300 (_map_cand_to_bev): BEV cell index of every candidate viewpoint, clamped to the grid."""
        S = np.asarray(ob["position"], dtype=np.float32)[None][:, [0, 2, 1]] * np.array([1, 1, -1], dtype=np.float32)
        xyzhe = np.zeros((1, 5))
        xyzhe[:, 3] = -ob["heading"]
        T = pose_matrix(xyzhe)[0]
        if not ob["candidate"]:
            return np.zeros(0, dtype=np.int64)
        p = np.asarray([c["position"] for c in ob["candidate"]], dtype=np.float32)[:, [0, 2, 1]] \
            * np.array([1, 1, -1], dtype=np.float32)
        p = p - S
        # the reference writes np.dot(cand_pos1, T.transpose(0, 1)): on a 2-D numpy array transpose(0, 1) is the identity
        # permutation, so the product is with T itself (not its transpose) -- reproduced as is
        p1 = np.concatenate([p, np.ones((p.shape[0], 1), dtype=np.float32)], -1) @ T
        c = np.round(p1[:, [0, 2]] / bev_res) + (bev_dim - 1) // 2
        c = np.clip(c, 0, bev_dim - 1).astype(np.int64)
        return c[:, 1] * bev_dim + c[:, 0]
