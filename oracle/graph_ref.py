"""CPU restatement of the fine-tune agent's per-step map inputs (test infrastructure, like bevbert_ref.py).

``nav_gmap_variable`` follows map_nav_src/r2r/agent.py:194-276 (_nav_gmap_variable) line by line over GraphMap-like
objects -- in tests/golden/make_golden.py those are the REFERENCE's own GraphMap / FloydGraph (map_nav_src/models/
graph_utils.py, imported there), so the golden vectors pin the product's batched implementation
(vln_bevbert_amd/graph_map.py) to the reference's classes plus this thin, loop-for-loop restatement of the agent method
(the agent module itself cannot be imported: it needs MatterSim).  ``map_cand_to_bev`` follows agent.py:278-300.
Only tests/ and the golden generator may import this file.
"""
import numpy as np
import torch

MAX_DIST = 30


def nav_gmap_variable(obs, gmaps, enc_full_graph=True, act_visited_nodes=False):
    """Per-sample map inputs of one navigation step, unpadded (the collation that follows in the reference only pads).
    Node order: [stop], then the visited nodes, then the unvisited ones, each group in the map's insertion order."""
    keys = ("gmap_vpids", "gmap_lens", "gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_pair_dists",
            "gmap_visited_masks", "no_vp_left")
    out = {k: [] for k in keys}
    for ob, gmap in zip(obs, gmaps):
        here = ob["viewpoint"]
        is_visited = (lambda vp: vp == here) if act_visited_nodes else gmap.graph.visited      # agent.py:205-216
        seen = [vp for vp in gmap.node_positions if is_visited(vp)]
        frontier = [vp for vp in gmap.node_positions if not is_visited(vp)]
        if enc_full_graph:                                                                      # agent.py:218-223
            nodes, flags = [None] + seen + frontier, [0] + [1] * len(seen) + [0] * len(frontier)
        else:
            nodes, flags = [None] + frontier, [0] * (1 + len(frontier))
        n = len(nodes)
        feats = [gmap.get_node_embed(vp) for vp in nodes[1:]]
        pair = np.zeros((n, n), dtype=np.float32)                                               # agent.py:236-240
        for a in range(1, n):
            for b in range(a + 1, n):
                pair[a, b] = pair[b, a] = gmap.graph.distance(nodes[a], nodes[b]) / MAX_DIST
        out["gmap_vpids"].append(nodes)
        out["gmap_lens"].append(n)
        out["gmap_img_embeds"].append(torch.stack([torch.zeros_like(feats[0])] + feats, 0))     # [stop] = zeros
        out["gmap_step_ids"].append([gmap.node_step_ids.get(vp, 0) for vp in nodes])
        out["gmap_pos_fts"].append(gmap.get_pos_fts(here, nodes, ob["heading"], ob["elevation"]))
        out["gmap_pair_dists"].append(pair)
        out["gmap_visited_masks"].append(flags)
        out["no_vp_left"].append(len(frontier) == 0)
    return out


def map_cand_to_bev(ob, bev_dim, bev_res, transfrom3D):
    """agent.py:278-300: BEV (x, z) cell of every candidate viewpoint of `ob`, clamped to the grid.  Simulator positions
    (x, y, z) become (x, z, -y); they are taken relative to the agent and multiplied by the pose matrix of heading
    -ob.heading -- by the matrix ITSELF: the reference calls ``T.transpose(0, 1)`` on a 2-D numpy array, which numpy
    defines as the identity permutation."""
    to_cam = lambda p: np.asarray(p, dtype=np.float32).reshape(-1, 3)[:, [0, 2, 1]] * np.float32([1, 1, -1])
    pose = np.zeros([1, 5])
    pose[0, 3] = -ob["heading"]
    T = transfrom3D(pose)[0]
    rel = to_cam([c["position"] for c in ob["candidate"]]) - to_cam(ob["position"])
    hom = np.concatenate([rel, np.ones([rel.shape[0], 1], dtype=np.float32)], axis=-1)
    ego = np.dot(hom, T)
    cell = (ego[:, [0, 2]] / bev_res).round() + (bev_dim - 1) // 2
    return np.clip(cell, 0, bev_dim - 1).astype(np.int64)
