"""Synthetic R2R-shaped batches in the reference's collate-output schema.

The reference's data layer needs Matterport HDF5 stores and the MatterSim
simulator (pretrain_src/data/dataset.py:16-17), neither of which exists here, so
the bench, the tests and the golden-vector generator all draw batches from this
seeded generator instead.  Field names, dtypes, padding rules and the nesting of
the Python id lists follow the collate functions
(pretrain_src/data/tasks.py:116-163 mlm_collate, :361-411 sap_collate,
:669-720 masksem_collate) and the per-sample builders
(pretrain_src/data/dataset.py:157-235 get_input, :327-360 get_gmap_inputs,
:397-440 get_bev_inputs, :254-325 get_traj_pano_fts).
"""
import math

import numpy as np
import torch

N_VIEWS = 36


def pose_matrix(xyzhe: np.ndarray) -> np.ndarray:
    """(N,5) x,y,z,heading,elevation -> (N,4,4) fp32 rigid transforms.

    Same matrix as the reference's ``transfrom3D`` (pretrain_src/model/bev_utils.py:7-36):
    R = Ry(heading) * Rx(elevation) with the translation in the last column.
    """
    x, y, z, h, e = (xyzhe[:, i] for i in range(5))       # trig and products stay in the input dtype
    ce, se, ch, sh = np.cos(e), np.sin(e), np.cos(h), np.sin(h)
    T = np.zeros((xyzhe.shape[0], 4, 4), dtype=np.float64)
    T[:, 0, 0], T[:, 0, 1], T[:, 0, 2], T[:, 0, 3] = ch, se * sh, ce * sh, x
    T[:, 1, 1], T[:, 1, 2], T[:, 1, 3] = ce, -se, y
    T[:, 2, 0], T[:, 2, 1], T[:, 2, 2], T[:, 2, 3] = -sh, ch * se, ch * ce, z
    T[:, 3, 3] = 1.0
    return T.astype(np.float32)


def _vocab_hi(cfg):
    """upper end of the synthetic token ids: BERT-sized vocabularies keep the historical 29 000 (the committed golden
    vectors depend on it); larger ones (xlm-roberta, 250 002) use their whole range so that high ids are exercised"""
    return min(cfg.vocab_size - 1, 29000) if cfg.vocab_size <= 30522 else cfg.vocab_size - 1


def _mask_tokens(rng, ids, vocab_lo, vocab_hi, mask_id):
    """BERT 15 % masking, at least one label (pretrain_src/data/tasks.py:14-55)."""
    out, lab = list(ids), [-1] * len(ids)
    for k, tok in enumerate(ids):
        p = rng.random()
        if p < 0.15:
            p /= 0.15
            if p < 0.8:
                out[k] = mask_id
            elif p < 0.9:
                out[k] = int(rng.integers(vocab_lo, vocab_hi))
            lab[k] = tok
    if all(v == -1 for v in lab):
        lab[0] = ids[0]
        out[0] = mask_id
    return out, lab


def make_sample(rng, i, cfg, n_steps, txt_len, ragged_views=False, grid=True):
    """One sample in the schema of ``get_input`` + the task Dataset ``__getitem__``.  ``grid=False``: without the grid
    features / depths / class ids (samples whose grid features live in a device-resident feature_store.GridFeatureStore)."""
    bev_dim = cfg.bev_dim
    n_cells = bev_dim * bev_dim
    hw = cfg.grid_hw
    s = {}

    # ---- text
    hi = min(cfg.vocab_size - 1, 29000)
    lo = min(1000, hi - 1)
    ids = [101 % cfg.vocab_size] + [int(v) for v in rng.integers(lo, hi, size=txt_len - 2)] + [102 % cfg.vocab_size]
    s["txt_ids"] = ids

    # ---- trajectory: path p{i}_{t}; candidates = [prev] + [next] + fresh (+ a shared one)
    path = [f"p{i}_{t}" for t in range(n_steps)]
    cands = []
    for t in range(n_steps):
        c = []
        if t > 0:
            c.append(path[t - 1])            # already visited -> backtrack logit path
        if t < n_steps - 1:
            c.append(path[t + 1])            # unvisited now, visited later
        c += [f"c{i}_{t}_{k}" for k in range(2)]
        if t % 2 == 0:
            c.append(f"s{i}")                # seen from several steps -> averaged
        cands.append(c)
    view_fts, loc_fts, nav_types, view_lens, obj_fts, obj_lens, dep_fts = [], [], [], [], [], [], []
    has_obj = getattr(cfg, "obj_feat_size", 0) > 0
    dep_size = getattr(cfg, "depth_feat_size", 0)
    loc_size = getattr(cfg, "loc_feat_size", None) or cfg.angle_feat_size + 3
    for t in range(n_steps):
        nv = N_VIEWS + (int(rng.integers(0, 3)) if ragged_views else 0)
        view_fts.append(rng.standard_normal((nv, cfg.image_feat_size)).astype(np.float32))
        ang = rng.uniform(-math.pi, math.pi, size=(nv, 2))
        loc = np.concatenate(
            [np.sin(ang[:, :1]), np.cos(ang[:, :1]), np.sin(ang[:, 1:]), np.cos(ang[:, 1:]),
             np.ones((nv, 3))], 1).astype(np.float32)[:, :loc_size]
        if dep_size:        # CE fork: one DD-PPO depth feature per view
            dep_fts.append(rng.standard_normal((nv, dep_size)).astype(np.float32))
        nt = [1] * len(cands[t]) + [0] * (nv - len(cands[t]))
        if has_obj:     # REVERIE: object tokens follow the views (dataset.py:283-321): loc = [angle(4), box(3)], type 2
            no = int(rng.integers(0, 5)) if ragged_views else 3
            if t == n_steps - 1 and no == 0 and i % 2 == 0:
                no = 2          # most samples end on a viewpoint with objects; odd ones may have none
            obj_fts.append(rng.standard_normal((no, cfg.obj_feat_size)).astype(np.float32))
            oang = rng.uniform(-math.pi, math.pi, size=(no, 2))
            oloc = np.concatenate(
                [np.sin(oang[:, :1]), np.cos(oang[:, :1]), np.sin(oang[:, 1:]), np.cos(oang[:, 1:]),
                 rng.uniform(0, 1, size=(no, 3))], 1).astype(np.float32)
            loc = np.concatenate([loc, oloc], 0)
            nt = nt + [2] * no
            obj_lens.append(no)
        loc_fts.append(loc)
        nav_types.append(nt)
        view_lens.append(nv)
    s.update(traj_view_img_fts=view_fts, traj_loc_fts=loc_fts, traj_nav_types=nav_types,
             traj_vp_view_lens=view_lens, traj_vpids=path, traj_cand_vpids=cands)
    if dep_size:
        s["traj_view_dep_fts"] = dep_fts
    if has_obj:
        no = obj_lens[-1]
        logits = rng.standard_normal((no, cfg.obj_prob_size))
        e = np.exp(logits - logits.max(1, keepdims=True)) if no else logits
        s.update(traj_obj_img_fts=obj_fts, traj_vp_obj_lens=obj_lens,
                 vp_obj_probs=(e / e.sum(1, keepdims=True)).astype(np.float32) if no else logits.astype(np.float32),
                 obj_labels=int(rng.integers(0, no)) if no else -100)
        mrc_o = rng.random(no) < 0.15
        if no and not mrc_o.any():
            mrc_o[int(rng.integers(0, no))] = True
        s["vp_obj_mrc_masks"] = mrc_o

    # ---- global map (get_gmap_inputs): visited in path order, then still-unvisited candidates
    visited, unvisited = {}, {}
    for t, vp in enumerate(path):
        visited[vp] = t + 1
        unvisited.pop(vp, None)
        for nvp in cands[t]:
            if nvp not in visited:
                unvisited[nvp] = 0
    gmap_vpids = [None] + list(visited) + list(unvisited)
    G = len(gmap_vpids)
    s["gmap_vpids"] = gmap_vpids
    s["gmap_step_ids"] = [0] + list(visited.values()) + list(unvisited.values())
    s["gmap_visited_masks"] = [False] + [True] * len(visited) + [False] * len(unvisited)
    pos = rng.standard_normal((G, 7)).astype(np.float32)
    pos[0] = 0
    s["gmap_pos_fts"] = pos
    d = rng.uniform(0, 1, size=(G, G)).astype(np.float32)
    d = np.triu(d, 1)
    d = d + d.T
    d[0, :] = 0
    d[:, 0] = 0
    s["gmap_pair_dists"] = d

    # ---- local metric map inputs (get_bev_inputs)
    V = cfg.grid_views
    P = V * hw * hw
    if grid:
        s["rgbs"] = rng.standard_normal((V, hw, hw, cfg.grid_feat_size)).astype(np.float32)
        dep = rng.uniform(0, 0.6, size=(V, 1, hw, hw)).astype(np.float32)
        dep[rng.random(dep.shape) < 0.05] = 0.0
        s["depths"] = dep
        s["sem_ids"] = rng.integers(0, max(1, cfg.sem_classes), size=(P,)).astype(np.int64)
    xyz = rng.uniform(-5, 5, size=3)
    xyzhe = np.zeros((V, 5), dtype=np.float32)
    xyzhe[:, 0], xyzhe[:, 1], xyzhe[:, 2] = xyz
    xyzhe[:, 3] = -np.arange(V) * np.radians(30)
    xyzhe[:, 4] = np.pi
    s["T_c2w"] = pose_matrix(xyzhe)
    s["S_w2c"] = xyzhe[:1, :3].copy()
    h = np.zeros((1, 5), dtype=np.float32)
    h[:, 3] = rng.uniform(0, 2 * math.pi)
    s["T_w2c"] = pose_matrix(h)
    K = 1 + len(cands[-1])
    idx = rng.choice(n_cells, size=K - 1, replace=False)
    s["bev_cand_idxs"] = np.concatenate([[(n_cells - 1) // 2], idx]).astype(np.int64)
    s["bev_gpos_fts"] = rng.standard_normal((1, 7)).astype(np.float32)
    mrc = rng.random(n_cells) < 0.15
    if not mrc.any():
        mrc[int(rng.integers(0, n_cells))] = True
    s["bev_mrc_masks"] = mrc

    # ---- action labels: a non-visited gmap node / a local candidate
    choices = [k for k in range(G) if k == 0 or not s["gmap_visited_masks"][k]]
    s["global_act_labels"] = int(choices[int(rng.integers(0, len(choices)))])
    s["local_act_labels"] = int(rng.integers(0, K))
    return s


def _pad_stack(arrs, pad_value=0):
    """pad_tensors (pretrain_src/data/common.py): zero-pad dim 0 to the max, stack."""
    n = max(a.shape[0] for a in arrs)
    out = np.full((len(arrs), n) + arrs[0].shape[1:], pad_value, dtype=arrs[0].dtype)
    for k, a in enumerate(arrs):
        out[k, : a.shape[0]] = a
    return out


def collate(samples, cfg, task, rng, sems_as="onehot64"):
    """Batch the samples the way mlm/sap/masksem_collate do.

    sems_as: "onehot64" -> (B,P,40) float64 one-hot as the reference ships it
             (pretrain_src/data/dataset.py:402); "ids" -> (B,P) uint8 class ids,
             the compact form this package's splat kernel also accepts.
    """
    B = len(samples)
    b = {}
    txt, labs = [], []
    for s in samples:
        ids = s["txt_ids"]
        if task.startswith("mlm"):
            hi = _vocab_hi(cfg)
            ids, lab = _mask_tokens(rng, ids, min(1000, hi - 1), hi, 103 % cfg.vocab_size)
            labs.append(np.asarray(lab, dtype=np.int64))
        txt.append(np.asarray(ids, dtype=np.int64))
    b["txt_lens"] = torch.tensor([len(t) for t in txt], dtype=torch.long)
    b["txt_ids"] = torch.from_numpy(_pad_stack(txt, 0))
    if labs:
        b["txt_labels"] = torch.from_numpy(_pad_stack(labs, -1))

    b["traj_step_lens"] = [len(s["traj_view_img_fts"]) for s in samples]
    flat = lambda key: [a for s in samples for a in s[key]]
    b["traj_vp_view_lens"] = torch.tensor(flat("traj_vp_view_lens"), dtype=torch.long)
    b["traj_view_img_fts"] = torch.from_numpy(_pad_stack(flat("traj_view_img_fts")))
    b["traj_loc_fts"] = torch.from_numpy(_pad_stack(flat("traj_loc_fts")))
    if "traj_view_dep_fts" in samples[0]:
        b["traj_view_dep_fts"] = torch.from_numpy(_pad_stack(flat("traj_view_dep_fts")))
    b["traj_nav_types"] = torch.from_numpy(
        _pad_stack([np.asarray(v, dtype=np.int64) for v in flat("traj_nav_types")]))
    b["traj_vpids"] = [s["traj_vpids"] for s in samples]
    b["traj_cand_vpids"] = [s["traj_cand_vpids"] for s in samples]
    if "traj_obj_img_fts" in samples[0]:            # mrc_collate / og_collate (tasks.py:291-297,471-475)
        objs = [[o.copy() for o in s["traj_obj_img_fts"]] for s in samples]
        if task.startswith("mrc"):                  # masked objects of the LAST viewpoint are zeroed (tasks.py:240-242)
            for o, s in zip(objs, samples):
                o[-1][s["vp_obj_mrc_masks"]] = 0
        b["traj_vp_obj_lens"] = torch.tensor(flat("traj_vp_obj_lens"), dtype=torch.long)
        b["traj_obj_img_fts"] = torch.from_numpy(_pad_stack([o for so in objs for o in so]))
        if task.startswith("mrc"):
            b["vp_obj_mrc_masks"] = torch.from_numpy(_pad_stack([s["vp_obj_mrc_masks"] for s in samples], False))
            b["vp_obj_probs"] = torch.from_numpy(_pad_stack([s["vp_obj_probs"] for s in samples]))
        if task.startswith("og"):
            b["obj_labels"] = torch.tensor([s["obj_labels"] for s in samples], dtype=torch.long)

    b["gmap_vpids"] = [s["gmap_vpids"] for s in samples]
    b["gmap_lens"] = torch.tensor([len(s["gmap_step_ids"]) for s in samples], dtype=torch.long)
    b["gmap_step_ids"] = torch.from_numpy(
        _pad_stack([np.asarray(s["gmap_step_ids"], dtype=np.int64) for s in samples]))
    b["gmap_visited_masks"] = torch.from_numpy(
        _pad_stack([np.asarray(s["gmap_visited_masks"], dtype=bool) for s in samples], False))
    b["gmap_pos_fts"] = torch.from_numpy(_pad_stack([s["gmap_pos_fts"] for s in samples]))
    G = int(b["gmap_lens"].max())
    pd = np.zeros((B, G, G), dtype=np.float32)
    for k, s in enumerate(samples):
        g = s["gmap_pair_dists"].shape[0]
        pd[k, :g, :g] = s["gmap_pair_dists"]
    b["gmap_pair_dists"] = torch.from_numpy(pd)

    n_cells = cfg.bev_dim * cfg.bev_dim
    if "rgbs" in samples[0]:            # absent: the grid features are rows of a device-resident store
        b["rgbs"] = torch.from_numpy(np.stack([s["rgbs"] for s in samples]))
        b["depths"] = torch.from_numpy(np.stack([s["depths"] for s in samples]))
        sem_ids = np.stack([s["sem_ids"] for s in samples])
        if cfg.sem_classes <= 0:            # continuous-environment fork: no semantic maps in the batch
            pass
        elif sems_as == "onehot64":
            b["sems"] = torch.from_numpy(np.eye(cfg.sem_classes)[sem_ids])          # float64
        else:
            b["sems"] = torch.from_numpy(sem_ids.astype(np.uint8))
    b["T_c2w"] = torch.from_numpy(np.stack([s["T_c2w"] for s in samples]))
    b["T_w2c"] = torch.from_numpy(np.stack([s["T_w2c"] for s in samples]))
    b["S_w2c"] = torch.from_numpy(np.stack([s["S_w2c"] for s in samples]))
    b["bev_masks"] = torch.ones(B, n_cells, dtype=torch.bool)
    cand = _pad_stack([s["bev_cand_idxs"] for s in samples])
    nav = np.zeros((B, n_cells), dtype=bool)
    for k, s in enumerate(samples):
        nav[k, s["bev_cand_idxs"]] = True
    b["bev_nav_masks"] = torch.from_numpy(nav)
    b["bev_cand_idxs"] = torch.from_numpy(cand)
    b["bev_gpos_fts"] = torch.from_numpy(np.stack([s["bev_gpos_fts"] for s in samples]))
    if task.startswith("sap"):
        b["global_act_labels"] = torch.tensor([s["global_act_labels"] for s in samples], dtype=torch.long)
        b["local_act_labels"] = torch.tensor([s["local_act_labels"] for s in samples], dtype=torch.long)
    if task.startswith("masksem"):
        b["bev_mrc_masks"] = torch.from_numpy(np.stack([s["bev_mrc_masks"] for s in samples]))
    return b


def make_batch(cfg, task, batch_size, seed=1000, txt_len=80, n_steps=5, ragged=False,
               sems_as="onehot64"):
    """Seeded batch. ``ragged``: T_i in [1,7], text lens in [L/2, L], 36..38 views."""
    rng = np.random.default_rng(seed)
    samples = []
    for i in range(batch_size):
        T = int(rng.integers(1, 8)) if ragged else n_steps
        L = int(rng.integers(txt_len // 2, txt_len + 1)) if ragged else txt_len
        samples.append(make_sample(rng, i, cfg, T, L, ragged_views=ragged))
    return collate(samples, cfg, task, rng, sems_as=sems_as)


HOST_COPIES = ("gmap_visited_masks", "txt_labels", "traj_vp_view_lens", "traj_vp_obj_lens", "vp_obj_mrc_masks")


def batch_to(batch, device, non_blocking=True):
    """move_to_cuda (pretrain_src/data/loader.py:78-120 PrefetchLoader): tensors move, lists stay.

    The three small tensors the host-side index building needs (SAP fusion sets, MLM positions, gmap CSR) also keep
    a '<key>_cpu' copy, so the forward never has to read them back from the device."""
    out = {}
    for k, v in batch.items():
        out[k] = v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v
        if k in HOST_COPIES and torch.is_tensor(v):
            out[k + "_cpu"] = v
    if torch.device(device).type == "cuda" and "gmap_vpids" in batch and "traj_loc_fts" in batch:
        out["gmap_csr"] = gmap_csr(batch, device)
    return out


def gmap_csr(batch, device):
    """Host-side index building of the global-map aggregation (vilmodel.py:632-666 walks the same string ids inside
    the model forward): done here, with the other host->device copies of the loader, so that the training step itself
    carries no Python loops over viewpoint ids.  Returns (SegmentCSR, G) for GlocalTextPathCMT's ``gmap_csr``."""
    from .vilmodel import build_gmap_csr
    lens = batch["traj_vp_view_lens"]
    if batch.get("traj_vp_obj_lens") is not None:
        lens = lens + batch["traj_vp_obj_lens"]
    return build_gmap_csr(list(batch["traj_step_lens"]), lens.tolist(), batch["traj_vpids"], batch["traj_cand_vpids"],
                          batch["gmap_vpids"], batch["traj_loc_fts"].shape[1], device)


# ----------------------------------------------------------------------------- fine-tune rollouts (graph bookkeeping)
def make_nav_episodes(B, T, seed, n_nodes=14, degree=3):
    """Synthetic observation streams for the fine-tune bookkeeping (map_nav_src/r2r/env.py _get_obs fields the agent's
    graph code reads): per episode a random connectivity graph of viewpoints with 3-D positions and a T-step walk that
    revisits nodes now and then.  Returns obs[t][i] = {'scan', 'viewpoint', 'position', 'heading', 'elevation',
    'candidate': [{'viewpointId', 'position', 'pointId'}]} and ended[t] (B,) flags (episodes stop at random steps)."""
    rng = np.random.default_rng(seed)
    graphs = []
    for i in range(B):
        pos = np.concatenate([rng.uniform(-8, 8, size=(n_nodes, 2)), rng.uniform(-1, 1, size=(n_nodes, 1))], 1)
        d = np.linalg.norm(pos[:, None] - pos[None], axis=-1)
        adj = [set() for _ in range(n_nodes)]
        for a in range(n_nodes):
            for b in np.argsort(d[a])[1:1 + degree]:
                adj[a].add(int(b))
                adj[int(b)].add(a)
        graphs.append((pos, [sorted(s) for s in adj]))
    cur = [int(rng.integers(0, n_nodes)) for _ in range(B)]
    stop_at = [int(rng.integers(max(2, T - 2), T + 1)) for _ in range(B)]
    obs, ended = [], []
    for t in range(T):
        step = []
        for i in range(B):
            pos, adj = graphs[i]
            c = cur[i]
            step.append({
                "scan": f"scan{i}", "viewpoint": f"e{i}_v{c}", "position": tuple(float(x) for x in pos[c]),
                "heading": float(rng.uniform(0, 2 * math.pi)), "elevation": 0.0,
                "candidate": [{"viewpointId": f"e{i}_v{n}", "position": tuple(float(x) for x in pos[n]),
                               "pointId": int(k)} for k, n in enumerate(adj[c])],
            })
        obs.append(step)
        ended.append(np.asarray([t >= stop_at[i] for i in range(B)]))
        for i in range(B):
            if t + 1 < stop_at[i]:
                cur[i] = int(rng.choice(graphs[i][1][cur[i]]))
    return obs, ended
