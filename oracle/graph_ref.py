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
    out = {"gmap_vpids": [], "gmap_lens": [], "gmap_img_embeds": [], "gmap_step_ids": [], "gmap_pos_fts": [],
           "gmap_pair_dists": [], "gmap_visited_masks": [], "no_vp_left": []}
    for i, gmap in enumerate(gmaps):
        visited_vpids, unvisited_vpids = [], []
        for k in gmap.node_positions.keys():                      # agent.py:205-216
            if act_visited_nodes:
                (visited_vpids if k == obs[i]["viewpoint"] else unvisited_vpids).append(k)
            else:
                (visited_vpids if gmap.graph.visited(k) else unvisited_vpids).append(k)
        out["no_vp_left"].append(len(unvisited_vpids) == 0)
        if enc_full_graph:                                        # agent.py:218-223
            gmap_vpids = [None] + visited_vpids + unvisited_vpids
            gmap_visited_masks = [0] + [1] * len(visited_vpids) + [0] * len(unvisited_vpids)
        else:
            gmap_vpids = [None] + unvisited_vpids
            gmap_visited_masks = [0] * len(gmap_vpids)
        gmap_step_ids = [gmap.node_step_ids.get(vp, 0) for vp in gmap_vpids]
        embeds = [gmap.get_node_embed(vp) for vp in gmap_vpids[1:]]
        embeds = torch.stack([torch.zeros_like(embeds[0])] + embeds, 0)
        pos_fts = gmap.get_pos_fts(obs[i]["viewpoint"], gmap_vpids, obs[i]["heading"], obs[i]["elevation"])
        pair = np.zeros((len(gmap_vpids), len(gmap_vpids)), dtype=np.float32)
        for a in range(1, len(gmap_vpids)):                       # agent.py:236-240
            for b in range(a + 1, len(gmap_vpids)):
                pair[a, b] = pair[b, a] = gmap.graph.distance(gmap_vpids[a], gmap_vpids[b]) / MAX_DIST
        out["gmap_vpids"].append(gmap_vpids)
        out["gmap_lens"].append(len(gmap_vpids))
        out["gmap_img_embeds"].append(embeds)
        out["gmap_step_ids"].append(gmap_step_ids)
        out["gmap_pos_fts"].append(pos_fts)
        out["gmap_pair_dists"].append(pair)
        out["gmap_visited_masks"].append(gmap_visited_masks)
    return out


def map_cand_to_bev(ob, bev_dim, bev_res, transfrom3D):
    """agent.py:278-300 with the reference's transfrom3D passed in."""
    S = np.array(ob["position"])[None, :].astype(np.float32)
    S = S[:, [0, 2, 1]] * np.array([1, 1, -1], dtype=np.float32)
    xyzhe = np.zeros([1, 5])
    xyzhe[:, 3] = -ob["heading"]
    T = transfrom3D(xyzhe)[0, :, :]
    cand_pos = np.array([c["position"] for c in ob["candidate"]]).astype(np.float32)
    cand_pos = cand_pos[:, [0, 2, 1]] * np.array([1, 1, -1], dtype=np.float32)
    cand_pos = cand_pos - S
    ones = np.ones([cand_pos.shape[0], 1]).astype(np.float32)
    cand_pos1 = np.concatenate([cand_pos, ones], axis=-1)
    cand_pos1 = np.dot(cand_pos1, T.transpose(0, 1))
    cand_pos = cand_pos1[:, :3]
    cand_pos = (cand_pos[:, [0, 2]] / bev_res).round() + (bev_dim - 1) // 2
    cand_pos[cand_pos < 0] = 0
    cand_pos[cand_pos >= bev_dim] = bev_dim - 1
    return cand_pos.astype(np.int64)
