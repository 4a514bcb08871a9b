"""Fine-tune rollout bookkeeping, batched and device-resident.

The reference keeps one ``GraphMap`` per episode (map_nav_src/models/graph_utils.py:96-189): Python dicts of per-node
torch tensors (running sums of panorama embeddings, world-frame point clouds with their 2352 x 768 features), a
dict-of-dicts Floyd graph, and rebuilds the navigation inputs of every step with nested Python loops
(map_nav_src/r2r/agent.py:194-337).  Between the three model calls of a step this is what the fine-tune path waits on.

Here:
  * ``GraphMapBatch``: all B episodes of a rollout as dense BATCHED arrays -- node positions (B, cap, 3), the Floyd
    distance and next-hop matrices (B, cap, cap), visited flags, step ids, embedding slots.  The incremental all-pairs
    relaxation is one masked minimum for the new edges of the whole batch plus ONE batched comparison through the
    current viewpoints per step; hop counts come from the next-hop tables bottom-up (no recursion); the per-step
    navigation inputs (``nav_gmap_variable``: visited-first ordering by a stable argsort, pair distances, position
    features) are numpy over (B, nodes).  Python only resolves viewpoint names to node indices.  Distances, next-hop
    tables, paths, orderings and features reproduce the reference bit for bit (golden ``graph_nav.npz``).
  * Node embeddings live in ONE (B, cap, H) device tensor of running sums + a count tensor, updated functionally with
    two ``index_put`` per step (autograd flows through them across steps like it does through the reference's stored
    tensors) and read with one device gather.
  * ``FloydGraph`` is the single-episode form of the same relaxation (kept for callers that want one graph).
  * Point clouds are never stored: a node remembers its row in the device-resident ``feature_store.GridFeatureStore``
    and its camera poses; ``bev_inputs`` returns, per sample, the store rows of the current viewpoint and its visited
    neighbours (``pc_order`` hops) in the order the reference concatenates them, the depths gathered from the store and
    the per-view poses -- exactly what ``ops.bev_lift_bin`` + ``ops.bev_splat_mean(rows=...)`` consume.
Host logic is plain numpy / torch indexing (device agnostic); the kernels it feeds are the C-ABI ones.
"""
import threading

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


class HostFeed:
    """One host -> device copy per call instead of one per array.

    ``torch.from_numpy(a).to(device)`` from pageable memory is a blocking copy ordered behind everything already queued
    on the stream: with ~20 small arrays per navigation step the host ends up waiting for the GPU twenty times a step.
    Here the arrays of a call are packed (16-byte aligned) into a slot of a small ring of PINNED buffers and shipped
    with a single non-blocking copy; the results are typed views of one device buffer.  A slot is reused only after the
    copy that read it has completed (event), i.e. the host can run ``slots`` calls ahead of the GPU."""

    def __init__(self, device, slots=8):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.ring = [{"buf": None, "ev": None} for _ in range(slots)]
        self.i = 0
        self._lock = threading.Lock()       # the per-device instance is shared: slot selection + fill are one critical section

    _shared = {}

    @classmethod
    def shared(cls, device):
        """One ring per device for the whole process: a GraphMapBatch lives for one episode batch, and pinning memory
        costs ~10 ms per buffer."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        f = cls._shared.get(device)
        if f is None:
            f = cls._shared[device] = cls(device)
        return f

    def __call__(self, arrays):
        """{name: device tensor} of the packed arrays (typed views of one device buffer)."""
        sh = self.ship(arrays)
        return {k: sh[k] for k in arrays}

    def ship(self, arrays):
        """As ``__call__`` but returns a ``Shipped``: device addresses for kernel arguments (``ptr(name)``) without a
        tensor view per array -- three torch calls each, the larger part of a call with a dozen small arrays -- and
        tensor views made on first use (``[name]``) for the arrays a caller hands on."""
        with self._lock:
            return self._ship(arrays)

    def _ship(self, arrays):
        meta, total = {}, 0
        for k, v in arrays.items():
            if not v.flags.c_contiguous:
                v = np.ascontiguousarray(v)
            meta[k] = (total, v.nbytes, v.dtype, v.shape, v)
            total += (v.nbytes + 15) // 16 * 16
        total = max(total, 16)
        slot = self.ring[self.i]
        self.i = (self.i + 1) % len(self.ring)
        if slot["ev"] is not None:
            slot["ev"].synchronize()
        if slot["buf"] is None or slot["buf"].numel() < total:
            buf = torch.empty(max(2 * total, 1 << 20), dtype=torch.uint8)
            slot["buf"] = buf.pin_memory() if self.cuda else buf
            slot["np"] = slot["buf"].numpy()
        host = slot["np"]
        for off, n, _, _, v in meta.values():
            if n:
                host[off:off + n] = v.reshape(-1).view(np.uint8)
        devbuf = torch.empty(total, dtype=torch.uint8, device=self.device)
        devbuf.copy_(slot["buf"][:total], non_blocking=True)
        if self.cuda:
            if slot["ev"] is None:
                slot["ev"] = torch.cuda.Event()
            slot["ev"].record(torch.cuda.current_stream(self.device))
        return Shipped(devbuf, meta)


_TORCH_DTYPES = {}


def _torch_dtype(np_dtype):
    t = _TORCH_DTYPES.get(np_dtype)
    if t is None:
        t = _TORCH_DTYPES[np_dtype] = torch.from_numpy(np.empty(0, dtype=np_dtype)).dtype
    return t


class Shipped:
    """The arrays of one HostFeed call, on the device in one buffer."""
    __slots__ = ("buf", "base", "meta", "_views")

    def __init__(self, buf, meta):
        self.buf, self.base, self.meta, self._views = buf, buf.data_ptr(), meta, {}

    def ptr(self, name):
        return self.base + self.meta[name][0]

    def __contains__(self, name):
        return name in self.meta

    def __getitem__(self, name):
        v = self._views.get(name)
        if v is None:
            off, n, dt, shape, _ = self.meta[name]
            v = self._views[name] = self.buf[off:off + n].view(_torch_dtype(dt)).view(shape)
        return v


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


class _EpisodeView:
    """Per-episode window onto the batched state (what callers that think in episodes see: tests, ``gather_nodes``)."""

    def __init__(self, owner, b, start_vp):
        self._o, self._b = owner, b
        self.start_vp = start_vp
        self.index = {}                 # vp -> node index: registration order = the reference's node_positions order
        self.names = []
        self.node_slot = {}             # vp -> column of the batch's embedding buffers
        self.pc_nodes = {}              # vp -> (store row, T_c2w (V,4,4) float32): visited nodes, in visit order
        self.graph = self

    # FloydGraph's read interface (graph_utils.py:44-94) on the batched matrices
    def visited(self, k):
        i = self.index.get(k)
        return i is not None and bool(self._o.visited[self._b, i])

    def distance(self, x, y):
        if x == y:
            return 0
        i, j = self.index.get(x), self.index.get(y)
        return float(_INF) if i is None or j is None else float(self._o.dis[self._b, i, j])

    def path(self, x, y):
        if x == y:
            return []
        i, j = self.index[x], self.index[y]
        k = int(self._o.point[self._b, i, j])
        if k < 0:
            return [y]
        kn = self.names[k]
        return self.path(x, kn) + self.path(kn, y)

    @property
    def node_positions(self):
        return {vp: tuple(self._o.pos[self._b, i]) for i, vp in enumerate(self.names)}

    @property
    def node_step_ids(self):
        st = self._o.step_ids[self._b]
        return {vp: int(st[i]) for i, vp in enumerate(self.names) if st[i] != 0}


class GraphMapBatch:
    """All B episodes of a rollout as dense batched arrays: positions (B, cap, 3), Floyd distance / next-hop matrices
    (B, cap, cap), visited flags, step ids, embedding slots -- every per-step builder is numpy over (B, nodes); Python
    only resolves viewpoint names to node indices (a few dict lookups per sample)."""

    def __init__(self, start_vps, hidden_size, device, dtype=torch.float32, capacity=64, node_capacity=32):
        self.B, self.H, self.device, self.dtype = len(start_vps), hidden_size, torch.device(device), dtype
        self.eps = [_EpisodeView(self, b, vp) for b, vp in enumerate(start_vps)]
        self.cap = capacity
        self.embed_sum = torch.zeros(self.B, capacity, hidden_size, dtype=dtype, device=self.device)
        self.embed_cnt = torch.zeros(self.B, capacity, dtype=torch.float32, device=self.device)
        self.ncap = 0
        self.n = np.zeros(self.B, dtype=np.int64)
        self._alloc(node_capacity)
        self._hops = None               # hop counts of the current graphs (rebuilt after update_graph)

    @property
    def feed(self):
        return HostFeed.shared(self.device)

    def _alloc(self, ncap):
        B, old = self.B, self.ncap
        pos = np.zeros((B, ncap, 3))
        dis = np.full((B, ncap, ncap), float(_INF))
        point = np.full((B, ncap, ncap), -1, dtype=np.int32)             # -1: direct edge ("" in the reference)
        visited = np.zeros((B, ncap), dtype=bool)
        step_ids = np.zeros((B, ncap), dtype=np.int64)
        slot = np.full((B, ncap), -1, dtype=np.int64)
        if old:
            pos[:, :old], visited[:, :old], step_ids[:, :old], slot[:, :old] = self.pos, self.visited, self.step_ids, self.slot
            dis[:, :old, :old], point[:, :old, :old] = self.dis, self.point
        self.pos, self.dis, self.point, self.visited, self.step_ids, self.slot = pos, dis, point, visited, step_ids, slot
        self.ncap = ncap

    def _node(self, b, vp):
        ep = self.eps[b]
        i = ep.index.get(vp)
        if i is None:
            i = ep.index[vp] = len(ep.names)
            ep.names.append(vp)
            self.n[b] = i + 1
            if i >= self.ncap:
                self._alloc(2 * self.ncap)
        return i

    # -- graph structure (host) ----------------------------------------------------------------------------------
    def update_graph(self, obs, ended=None):
        """GraphMap.update_graph for every live episode (graph_utils.py:109-115; agent.py:447-449,556-559): edges of the
        whole batch in one masked minimum, then ONE batched relaxation through the current viewpoints."""
        eb, ei, ej, pa, pb_, lb, lk = [], [], [], [], [], [], []
        for b, ob in enumerate(obs):
            if ended is not None and ended[b]:
                continue
            cur = self._node(b, ob["viewpoint"])
            self.pos[b, cur] = ob["position"]
            for cc in ob["candidate"]:
                j = self._node(b, cc["viewpointId"])
                self.pos[b, j] = cc["position"]
                eb.append(b)
                ei.append(cur)
                ej.append(j)
                pa.append(ob["position"])
                pb_.append(cc["position"])
            lb.append(b)
            lk.append(cur)
        if not lb:
            return
        self._hops = None
        if eb:
            eb, ei, ej = np.asarray(eb), np.asarray(ei), np.asarray(ej)
            d = np.asarray(pb_, dtype=np.float64) - np.asarray(pa, dtype=np.float64)
            dist = np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2 + d[:, 2] ** 2)
            better = dist < self.dis[eb, ei, ej]
            eb, ei, ej, dist = eb[better], ei[better], ej[better], dist[better]
            self.dis[eb, ei, ej] = dist
            self.dis[eb, ej, ei] = dist
            self.point[eb, ei, ej] = -1
            self.point[eb, ej, ei] = -1
        # relax every pair through k = the current viewpoint (graph_utils.py:63-72).  Row / column k cannot change
        # during the reference's double loop (the diagonal stays 'infinite'), so one comparison is the same sequence
        lb, lk = np.asarray(lb), np.asarray(lk)
        nm = int(self.n[lb].max())
        d = self.dis[lb, :nm, :nm]
        ar = np.arange(len(lb))
        via = d[ar, :, lk][:, :, None] + d[ar, lk, :][:, None, :]
        better = via < d
        better[:, np.arange(nm), np.arange(nm)] = False
        d[better] = via[better]
        pt = self.point[lb, :nm, :nm]
        pt[better] = np.broadcast_to(lk[:, None, None], better.shape)[better]
        self.dis[lb, :nm, :nm] = d
        self.point[lb, :nm, :nm] = pt
        self.visited[lb, lk] = True

    def set_step_ids(self, obs, t, ended=None):
        """agent.py:471-474."""
        for b, ob in enumerate(obs):
            if ended is None or not ended[b]:
                self.step_ids[b, self._node(b, ob["viewpoint"])] = t + 1

    def _slot(self, b, vp):
        ep = self.eps[b]
        s = ep.node_slot.get(vp)
        if s is None:
            s = ep.node_slot[vp] = len(ep.node_slot)
            self.slot[b, self._node(b, vp)] = s
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
                if not ep.visited(vp):
                    ab.append(i)
                    as_.append(self._slot(i, vp))
                    aj.append(j)
        if not rb:
            return
        dev = self.device
        ix = self.feed({"rb": np.asarray(rb, dtype=np.int64), "rs": np.asarray(rs, dtype=np.int64),
                        "ab": np.asarray(ab, dtype=np.int64), "as": np.asarray(as_, dtype=np.int64),
                        "aj": np.asarray(aj, dtype=np.int64)})
        rb_t, rs_t = ix["rb"], ix["rs"]
        self.embed_sum = self.embed_sum.index_put((rb_t, rs_t), avg_pano_embeds[rb_t].to(self.dtype))
        self.embed_cnt = self.embed_cnt.index_put((rb_t, rs_t), torch.ones(len(rb), device=dev))
        if ab:
            ab_t, as_t, aj_t = ix["ab"], ix["as"], ix["aj"]
            self.embed_sum = self.embed_sum.index_put((ab_t, as_t), pano_embeds[ab_t, aj_t].to(self.dtype),
                                                      accumulate=True)
            self.embed_cnt = self.embed_cnt.index_put((ab_t, as_t), torch.ones(len(ab), device=dev), accumulate=True)

    def node_embed(self, i, vp):
        """GraphMap.get_node_embed (graph_utils.py:146-147)."""
        s = self.eps[i].node_slot[vp]
        return self.embed_sum[i, s] / self.embed_cnt[i, s]

    # -- hop counts -------------------------------------------------------------------------------------------------
    def hops(self):
        """(B, nmax, nmax) int: len(FloydGraph.path(x, y)) for every pair, from the next-hop tables the way the
        reference's recursion reads them (path(x, y) = path(x, k) + path(k, y), k = point[x][y]; a direct edge -- or no
        known path -- is one hop; x == y is none).  Resolved bottom-up: an entry is known once both halves are."""
        if self._hops is not None:
            return self._hops
        nm = max(1, int(self.n.max()))
        P = self.point[:, :nm, :nm]
        L = np.where(P < 0, 1, -1).astype(np.int64)
        L[:, np.arange(nm), np.arange(nm)] = 0
        Pc = np.clip(P, 0, None).astype(np.int64)
        for _ in range(nm + 1):
            todo = L < 0
            if not todo.any():
                break
            lik = np.take_along_axis(L, Pc, axis=2)             # L[b, i, k(b, i, j)]
            lkj = np.take_along_axis(L, Pc, axis=1)             # L[b, k(b, i, j), j]
            ok = todo & (lik >= 0) & (lkj >= 0)
            L[ok] = (lik + lkj)[ok]
        self._hops = L
        return L

    # -- per-step navigation inputs ------------------------------------------------------------------------------
    def _pos_fts_rows(self, b, cur, tgt, heading, elevation, angle_feat_size):
        """Rows of get_pos_fts (graph_utils.py:149-172) for flat index arrays: sample b, origin node cur, target node
        tgt (all (n,)), the agent's heading / elevation per row."""
        h, e, d = rel_pos_fts(self.pos[b, cur], self.pos[b, tgt], heading, elevation)
        ang = angle_fts(h.astype(np.float32), e.astype(np.float32), angle_feat_size)
        same = cur == tgt
        gd = np.where(same, 0.0, self.dis[b, cur, tgt])
        hp = self.hops()[b, cur, tgt]
        dist = np.stack([d / MAX_DIST, gd / MAX_DIST, hp / MAX_STEP], 1).astype(np.float32)
        return np.concatenate([ang, dist], 1)

    def pos_fts(self, i, cur_vp, vpids, cur_heading, cur_elevation, angle_feat_size=4):
        """GraphMap.get_pos_fts for one sample; None = the [stop] token."""
        ep = self.eps[i]
        out = np.zeros((len(vpids), angle_feat_size + 3), dtype=np.float32)
        out[:, :angle_feat_size] = angle_fts(np.zeros(1, np.float32), np.zeros(1, np.float32), angle_feat_size)
        real = [k for k, vp in enumerate(vpids) if vp is not None]
        if real:
            tgt = np.asarray([ep.index[vpids[k]] for k in real])
            cur = np.full(len(real), ep.index[cur_vp])
            out[real] = self._pos_fts_rows(np.full(len(real), i), cur, tgt, cur_heading, cur_elevation, angle_feat_size)
        return out

    def nav_gmap_variable(self, obs, enc_full_graph=True, act_visited_nodes=False, angle_feat_size=4):
        """agent.py:194-276 (_nav_gmap_variable): [stop] + map nodes per sample (visited nodes first, each group in
        registration order), padded to the batch maximum."""
        B = self.B
        n = self.n
        nm = max(1, int(n.max()))
        cur = np.asarray([self.eps[b].index[ob["viewpoint"]] for b, ob in enumerate(obs)])
        heading = np.asarray([ob["heading"] for ob in obs], dtype=np.float64)
        elevation = np.asarray([ob["elevation"] for ob in obs], dtype=np.float64)
        col = np.arange(nm)
        valid = col[None] < n[:, None]
        vis = (col[None] == cur[:, None]) if act_visited_nodes else self.visited[:, :nm]
        vis = vis & valid
        key = np.where(valid, np.where(vis, 0, 1), 2) * nm + col[None]
        order = np.argsort(key, axis=1, kind="stable")                      # visited | unvisited | unused, each in order
        nvis = vis.sum(1)
        first = np.zeros(B, dtype=np.int64) if enc_full_graph else nvis
        cnt = n - first
        G = 1 + int(cnt.max())
        j = np.arange(G - 1)
        real = j[None] < cnt[:, None]                                       # (B, G-1)
        node = np.take_along_axis(order, np.minimum(first[:, None] + j[None], nm - 1), axis=1)
        node = np.where(real, node, 0)
        bi = np.arange(B)[:, None]
        lens = cnt + 1
        masks = np.arange(G)[None] < lens[:, None]
        visited = np.zeros((B, G), dtype=bool)
        if enc_full_graph:
            visited[:, 1:] = vis[bi, node] & real
        step_ids = np.zeros((B, G), dtype=np.int64)
        step_ids[:, 1:] = np.where(real, self.step_ids[bi, node], 0)
        slot_np = np.full((B, G), -1, dtype=np.int64)
        slot_np[:, 1:] = np.where(real, self.slot[bi, node], -1)
        pair_np = np.zeros((B, G, G), dtype=np.float32)
        sub = self.dis[bi[:, :, None], node[:, :, None], node[:, None, :]] / MAX_DIST
        sub[:, np.arange(G - 1), np.arange(G - 1)] = 0.0
        pair_np[:, 1:, 1:] = np.where(real[:, :, None] & real[:, None, :], sub, 0.0).astype(np.float32)
        pos = np.zeros((B, G, angle_feat_size + 3), dtype=np.float32)
        pos[:, :, :angle_feat_size] = np.where(masks[..., None],
                                               angle_fts(np.zeros(1, np.float32), np.zeros(1, np.float32), angle_feat_size), 0)
        rb, rj = np.nonzero(real)
        if len(rb):
            pos[rb, rj + 1] = self._pos_fts_rows(rb, cur[rb], node[rb, rj], heading[rb], elevation[rb], angle_feat_size)
        vpids = []
        for b in range(B):
            names = self.eps[b].names
            vpids.append([None] + [names[k] for k in node[b, :cnt[b]]])
        dev = self.device
        up = self.feed({"slot": slot_np, "step_ids": step_ids, "pos": pos, "visited": visited, "pair": pair_np,
                        "masks": masks})
        slot_t = up["slot"]
        ok = slot_t >= 0
        bt = torch.arange(B, device=dev)[:, None].expand(-1, G)
        si = slot_t.clamp(min=0)
        c = self.embed_cnt[bt, si].clamp(min=1.0).to(self.dtype)
        embeds = (self.embed_sum[bt, si] / c[..., None]) * ok[..., None].to(self.dtype)    # [stop] / padding = 0
        return {
            "gmap_vpids": vpids, "gmap_img_embeds": embeds,
            "gmap_step_ids": up["step_ids"],
            "gmap_pos_fts": up["pos"],
            "gmap_visited_masks": up["visited"],
            "gmap_visited_masks_cpu": torch.from_numpy(visited),
            "gmap_pair_dists": up["pair"],
            "gmap_masks": up["masks"],
            "no_vp_left": [bool(x) for x in (n - nvis) == 0],
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
        p = np.asarray([obs[i]["position"] for i in live], dtype=np.float64)
        xyzhe[:, :, 0], xyzhe[:, :, 1], xyzhe[:, :, 2] = p[:, None, 0], p[:, None, 2], -p[:, None, 1]
        hd = np.asarray([obs[i]["heading"] for i in live], dtype=np.float64)
        xyzhe[:, :, 3] = -(np.arange(views)[None] * np.radians(30) + hd[:, None])
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
        hp = self.hops()[i, ep.index[vp]]
        return [c for c in ep.pc_nodes.keys() if hp[ep.index[c]] <= order]

    def bev_inputs(self, obs, store, pc_order=1, bev_dim=21, bev_res=0.5):
        """agent.py:143-192,282-337 (splat + _nav_bev_variable) as inputs of the fused kernels: per sample the R store
        rows (current viewpoint + visited neighbours, padded by repeating the first row with zero depths), the matching
        depths and per-view poses, the world->ego transform of the current pose, candidate cells and nav masks."""
        B = self.B
        nodes = [self.gather_nodes(i, ob["viewpoint"], pc_order) for i, ob in enumerate(obs)]
        R = max(len(n) for n in nodes)
        V = store.V
        rows = np.zeros((B, R), dtype=np.int32)
        T_c2w = np.zeros((B, R, V, 4, 4), dtype=np.float32)
        live = np.zeros((B, R), dtype=bool)
        for i, ns in enumerate(nodes):
            pc = self.eps[i].pc_nodes
            for r, vp in enumerate(ns):
                rows[i, r], T_c2w[i, r] = pc[vp]
            live[i, :len(ns)] = True
            rows[i, len(ns):] = rows[i, 0]
        P = np.asarray([ob["position"] for ob in obs], dtype=np.float32)
        S = np.stack([P[:, 0], P[:, 2], -P[:, 1]], 1)
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
        ar = np.arange(B)
        cur = np.asarray([self.eps[b].index[ob["viewpoint"]] for b, ob in enumerate(obs)])
        start = np.asarray([self.eps[b].index[self.eps[b].start_vp] for b in range(B)])
        gpos = self._pos_fts_rows(ar, cur, start, np.asarray([ob["heading"] for ob in obs], dtype=np.float64),
                                  np.asarray([ob["elevation"] for ob in obs], dtype=np.float64), 4)
        up = self.feed({"rows": rows, "live": live, "T_c2w": T_c2w.reshape(B, R * V, 4, 4), "T_w2c": pose_matrix(xyzhe),
                        "S": S, "nav_masks": nav_masks, "cand": cand_np, "gpos": gpos})
        rows_t = up["rows"]
        depths = store.depths.index_select(0, rows_t.reshape(-1).long()).reshape(B, R * V, store.hw, store.hw)
        depths = depths * up["live"].repeat_interleave(V, 1)[..., None, None]                        # padding: no depth
        return {
            "grid_rows": rows_t, "depths": depths, "T_c2w": up["T_c2w"],
            "T_w2c": up["T_w2c"][:, None], "S_w2c": up["S"][:, None],
            "bev_nav_masks": up["nav_masks"], "bev_cand_idxs": up["cand"],
            "bev_cand_vpids": cand_vpids, "bev_gpos_fts": up["gpos"][:, None],
        }

    @staticmethod
    def cand_cells_batch(obs, bev_dim, bev_res):
        """cand_cells for every sample: one pass of numpy over all candidates of the batch; only the 4-term products stay
        per-sample numpy matmuls (same BLAS path, hence the same roundings, as the reference's np.dot)."""
        counts = np.fromiter((len(ob["candidate"]) for ob in obs), dtype=np.int64, count=len(obs))
        if counts.sum() == 0:
            return [np.zeros(0, dtype=np.int64) for _ in obs]
        cells = GraphMapBatch.cand_cells_flat(
            np.array([ob["position"] for ob in obs], dtype=np.float64).reshape(len(obs), 3),
            np.array([ob["heading"] for ob in obs], dtype=np.float64),
            np.array([c["position"] for ob in obs for c in ob["candidate"]], dtype=np.float64),
            counts, np.repeat(np.arange(len(obs)), counts), bev_dim, bev_res)
        ends = np.cumsum(counts)
        return [cells[ends[i] - n:ends[i]] for i, n in enumerate(counts)]

    @staticmethod
    def cand_cells_flat(pos, heading, cand_pos, counts, sample, bev_dim, bev_res, T=None):
        """cand_cells for the candidates of a whole batch, flattened: ``pos`` (B, 3) / ``heading`` (B,) of the agents,
        ``cand_pos`` (sum(counts), 3), ``sample`` = the batch index of each candidate (all float64, as the simulator's)."""
        if T is None:                        # ``T``: pose_matrix of heading -h per sample, when the caller has it already
            xyzhe = np.zeros((len(pos), 5))
            xyzhe[:, 3] = -heading
            T = pose_matrix(xyzhe)
        flip = np.array([1, 1, -1], dtype=np.float32)
        S = pos.astype(np.float32)[:, [0, 2, 1]] * flip
        P = cand_pos.astype(np.float32)[:, [0, 2, 1]] * flip
        P = P - S[sample]
        p1 = np.concatenate([P, np.ones((len(P), 1), dtype=np.float32)], -1)
        # one stacked (1 x 4) @ (4 x 4) product per candidate: the same BLAS path per row as the reference's per-sample
        # np.dot, hence the same roundings (checked bit for bit against the per-sample loop: tests/test_host_logic.py)
        q = np.matmul(p1[:, None, :], T[sample])[:, 0]                                            # see cand_cells
        c = np.clip(np.round(q[:, [0, 2]] / bev_res) + (bev_dim - 1) // 2, 0, bev_dim - 1).astype(np.int64)
        return c[:, 1] * bev_dim + c[:, 0]

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
