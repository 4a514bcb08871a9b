"""CPU ORACLE for the BEVBert cross-modal hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch fp32 on the CPU, what the reference
(MarSaKi/VLN-BEVBert, /root/reference) computes on the path named by
BASELINE.json's north_star.  It is written functionally over a flat
``state_dict`` (``sd``: {reference key -> tensor}); it does not import, subclass or
copy the reference's modules.  Every function cites the reference file:line it
follows.

Allowed users: ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg -- as the checker / timed CPU baseline, never as the product
path.  Nothing under ``vln_bevbert_amd/`` imports this module.

Pinning: the reference has no tests of its own, so this oracle is
pinned against golden vectors produced by importing the reference in the build
container: ``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``, checked by
``tests/test_oracle_golden.py``.  Third-party piece restated from its published
behaviour: ``torch_scatter.scatter_mean`` 2.0.9 (sum / clamp(count, min=1)); the
reference wheel is not available offline, so that single primitive is "parity
unpinned" beyond the stub used when generating the golden vectors.

Dropout: all functions are the eval()/p=0 forward.  The reference's dropout
stream cannot be reproduced, so training-mode kernels are tested
against this oracle with explicit keep-masks exported by the kernels.
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-12      # configs/r2r_model.json "layer_norm_eps"; vilmodel.py:59,147,186,470-484
PANO_LN_EPS = 1e-5  # nn.LayerNorm default inside TransformerEncoderLayer: transformer.py:144-145


# ----------------------------------------------------------------------------- primitives
def linear(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def layer_norm(sd, p, x, eps=LN_EPS):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def gelu_erf(x):
    """vilmodel.py:31-37 (x * 0.5 * (1 + erf(x / sqrt(2))))."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def seq_mask(lens, max_len=None):
    """ops.py:36-44 gen_seq_masks."""
    if max_len is None:
        max_len = int(max(int(v) for v in lens))
    return torch.arange(max_len, device=lens.device)[None, :] < lens[:, None]


def neg_mask(mask):
    """ops.py:25-34 extend_neg_masks: (N,L) bool -> (N,1,1,L) additive -10000 fp32."""
    return (1.0 - mask[:, None, None, :].to(torch.float32)) * -10000.0


def _heads(x, nh):
    n, l, h = x.shape
    return x.view(n, l, nh, h // nh).permute(0, 2, 1, 3)


def attention(sd, p, hidden, context, add_mask, nh):
    """vilmodel.py:103-141 (BertSelfAttention) / :325-352 (BertOutAttention).

    softmax(Q K^T / sqrt(d) + add_mask) V with separate query/key/value Linears.
    """
    q = _heads(linear(sd, p + ".query", hidden), nh)
    k = _heads(linear(sd, p + ".key", context), nh)
    v = _heads(linear(sd, p + ".value", context), nh)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if add_mask is not None:
        s = s + add_mask
    pr = torch.softmax(s, dim=-1)
    o = torch.matmul(pr, v).permute(0, 2, 1, 3).contiguous()
    return o.view(o.shape[0], o.shape[1], -1)


def dense_res_ln(sd, p, h, inp):
    """vilmodel.py:143-154 BertSelfOutput / :182-193 BertOutput: LN(dense(h) + inp)."""
    return layer_norm(sd, p + ".LayerNorm", linear(sd, p + ".dense", h) + inp)


def bert_attention(sd, p, x, add_mask, nh):
    """vilmodel.py:156-166."""
    return dense_res_ln(sd, p + ".output", attention(sd, p + ".self", x, x, add_mask, nh), x)


def bert_xattention(sd, p, x, ctx, add_mask, nh):
    """vilmodel.py:354-363."""
    return dense_res_ln(sd, p + ".output", attention(sd, p + ".att", x, ctx, add_mask, nh), x)


def ffn(sd, p_inter, p_out, x):
    """vilmodel.py:168-193 BertIntermediate + BertOutput."""
    return dense_res_ln(sd, p_out, gelu_erf(linear(sd, p_inter + ".dense", x)), x)


def bert_layer(sd, p, x, add_mask, nh):
    """vilmodel.py:195-208."""
    a = bert_attention(sd, p + ".attention", x, add_mask, nh)
    return ffn(sd, p + ".intermediate", p + ".output", a)


# ----------------------------------------------------------------------------- encoders
def text_embeddings(sd, p, ids):
    """vilmodel.py:48-77 BertEmbeddings (token_type 0, positions 0..L-1)."""
    L = ids.shape[1]
    e = sd[p + ".word_embeddings.weight"][ids] \
        + sd[p + ".position_embeddings.weight"][:L][None] \
        + sd[p + ".token_type_embeddings.weight"][0][None, None]
    return layer_norm(sd, p + ".LayerNorm", e)


def lang_encoder(sd, p, cfg, x, txt_masks):
    """vilmodel.py:424-444."""
    m = neg_mask(txt_masks)
    for i in range(cfg.num_l_layers):
        x = bert_layer(sd, f"{p}.layer.{i}", x, m, cfg.num_attention_heads)
    return x


def pano_layer(sd, p, x, key_pad, nh):
    """transformer.py:170-182 forward_pre with nn.MultiheadAttention (packed in_proj)."""
    h = layer_norm(sd, p + ".norm1", x, PANO_LN_EPS)
    w, b = sd[p + ".self_attn.in_proj_weight"], sd[p + ".self_attn.in_proj_bias"]
    H = x.shape[-1]
    q = _heads(F.linear(h, w[:H], b[:H]), nh)
    k = _heads(F.linear(h, w[H:2 * H], b[H:2 * H]), nh)
    v = _heads(F.linear(h, w[2 * H:], b[2 * H:]), nh)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    s = s.masked_fill(key_pad[:, None, None, :], float("-inf"))
    o = torch.matmul(torch.softmax(s, -1), v).permute(0, 2, 1, 3).contiguous()
    o = linear(sd, p + ".self_attn.out_proj", o.view(o.shape[0], o.shape[1], -1))
    x = x + o
    h = layer_norm(sd, p + ".norm2", x, PANO_LN_EPS)
    h = linear(sd, p + ".linear2", F.gelu(linear(sd, p + ".linear1", h)))
    return x + h


def image_embeddings(sd, p, cfg, view_fts, loc_fts, nav_types, view_lens, type_emb_row1, obj_fts=None, obj_lens=None,
                     dep_fts=None):
    """vilmodel.py:494-532 ImageEmbeddings.forward incl. the object branch (:502-516) and the pano encoder; with
    ``dep_fts`` the continuous-environment fork's depth-feature term (bevbert_ce/pretrain/pretrain_src/model/
    vilmodel.py:507-509).

    Returns (sum_T, L, H) embeddings, the (sum_T, L) validity mask and the token counts (views + objects).
    """
    img = layer_norm(sd, p + ".img_layer_norm", linear(sd, p + ".img_linear", view_fts))
    if (p + ".dep_linear.weight") in sd:
        img = img + layer_norm(sd, p + ".dep_layer_norm", linear(sd, p + ".dep_linear", dep_fts))
    lens = view_lens
    if obj_fts is not None:
        if (p + ".obj_linear.weight") in sd:
            obj = layer_norm(sd, p + ".obj_layer_norm", linear(sd, p + ".obj_linear", obj_fts))
        else:
            obj = layer_norm(sd, p + ".img_layer_norm", linear(sd, p + ".img_linear", obj_fts))
        lens = view_lens + obj_lens
        L = int(lens.max())
        rows = []
        for k in range(img.shape[0]):           # cat(view[:view_len], obj[:obj_len]) then zero-pad (:507-515)
            r = torch.cat([img[k, :int(view_lens[k])], obj[k, :int(obj_lens[k])]], 0)
            rows.append(torch.cat([r, r.new_zeros(L - r.shape[0], r.shape[1])], 0))
        img = torch.stack(rows, 0)
    e = img \
        + layer_norm(sd, p + ".loc_layer_norm", linear(sd, p + ".loc_linear", loc_fts)) \
        + sd[p + ".nav_type_embedding.weight"][nav_types] \
        + type_emb_row1[None, None]
    e = layer_norm(sd, p + ".layer_norm", e)
    masks = seq_mask(lens, img.shape[1])
    if cfg.num_pano_layers > 0:
        for i in range(cfg.num_pano_layers):
            e = pano_layer(sd, f"{p}.pano_encoder.layers.{i}", e, ~masks, cfg.num_attention_heads)
        e = layer_norm(sd, p + ".pano_encoder.norm", e)          # ops.py:19-20, eps 1e-12
    return e, masks, lens


def aggregate_gmap(traj_embeds, traj_masks, view_lens, step_lens, traj_vpids, traj_cand_vpids, gmap_vpids):
    """vilmodel.py:632-666 _aggregate_gmap_features.

    visited node  = masked sum over its panorama tokens / view_len;
    candidate seen while still unvisited -> its token joins that node's list;
    node resolved at the END: visited mean if it was ever visited, else mean of the list;
    rows padded to the batch max and a zero [stop] row prepended.
    """
    H = traj_embeds.shape[-1]
    out, off = [], 0
    for i, T in enumerate(step_lens):
        emb = traj_embeds[off:off + T] * traj_masks[off:off + T, :, None]
        lens = view_lens[off:off + T]
        off += T
        visited, unvisited = {}, {}
        for t in range(T):
            visited[traj_vpids[i][t]] = emb[t].sum(0) / lens[t]
            for j, vp in enumerate(traj_cand_vpids[i][t]):
                if vp not in visited:
                    unvisited.setdefault(vp, []).append(emb[t, j])
        rows = []
        for vp in gmap_vpids[i][1:]:
            rows.append(visited[vp] if vp in visited else torch.stack(unvisited[vp], 0).mean(0))
        out.append(torch.stack(rows, 0) if rows else traj_embeds.new_zeros(0, H))
    G = max(r.shape[0] for r in out)
    pad = traj_embeds.new_zeros(len(out), G + 1, H)
    for i, r in enumerate(out):
        pad[i, 1:1 + r.shape[0]] = r
    return pad


def gmap_input_embedding(sd, p, gmap_img, gmap_step_ids, gmap_pos_fts):
    """vilmodel.py:668-679."""
    return gmap_img + sd[p + ".gmap_step_embeddings.weight"][gmap_step_ids] \
        + layer_norm(sd, p + ".gmap_pos_embeddings.1", linear(sd, p + ".gmap_pos_embeddings.0", gmap_pos_fts))


def bev_input_embedding(sd, p, bev_fts, bev_pos_fts, bev_nav_masks):
    """vilmodel.py:589-593."""
    return layer_norm(sd, p + ".bev_fts_embeddings.1", linear(sd, p + ".bev_fts_embeddings.0", bev_fts)) \
        + layer_norm(sd, p + ".bev_pos_embeddings.1", linear(sd, p + ".bev_pos_embeddings.0", bev_pos_fts)) \
        + sd[p + ".nav_type_embedding.weight"][bev_nav_masks.long()]


def x_layer_visn(sd, p, nh, lang, lang_m, visn, visn_m, sprels=None):
    """vilmodel.py:383-398 GraphLXRTXLayer.forward."""
    a = bert_xattention(sd, p + ".visual_attention", visn, lang, lang_m, nh)
    m = visn_m if sprels is None else visn_m + sprels
    a = bert_attention(sd, p + ".visn_self_att", a, m, nh)
    return ffn(sd, p + ".visn_inter", p + ".visn_output", a)


def x_layer_lang2visn(sd, p, nh, lang, lang_m, visn, visn_m):
    """vilmodel.py:400-411 forward_lang2visn."""
    a = bert_xattention(sd, p + ".visual_attention", lang, visn, visn_m, nh)
    a = bert_attention(sd, p + ".lang_self_att", a, lang_m, nh)
    return ffn(sd, p + ".lang_inter", p + ".lang_output", a)


def x_layer_visn2visn(sd, p, nh, visn, visn_m):
    """vilmodel.py:413-421 forward_visn2visn."""
    a = bert_attention(sd, p + ".visn_self_att", visn, visn_m, nh)
    return ffn(sd, p + ".visn_inter", p + ".visn_output", a)


def crossmodal_encoder(sd, p, cfg, txt, txt_masks, img, img_masks, sprels=None):
    """vilmodel.py:446-463."""
    tm, im = neg_mask(txt_masks), neg_mask(img_masks)
    for i in range(cfg.num_x_layers):
        img = x_layer_visn(sd, f"{p}.x_layers.{i}", cfg.num_attention_heads, txt, tm, img, im, sprels)
    return img


def sprel_bias(sd, p, pair_dists):
    """vilmodel.py:691-692: Linear(1,1) on (B,G,G) -> (B,1,G,G)."""
    return (pair_dists * sd[p + ".sprel_linear.weight"].view(()) + sd[p + ".sprel_linear.bias"].view(()))[:, None]


def _common(sd, cfg, b, pfx):
    txt_masks = seq_mask(b["txt_lens"], b["txt_ids"].shape[1])
    txt = text_embeddings(sd, pfx + "embeddings", b["txt_ids"])
    txt = lang_encoder(sd, pfx + "lang_encoder", cfg, txt, txt_masks)
    traj, traj_masks, lens = image_embeddings(
        sd, pfx + "img_embeddings", cfg, b["traj_view_img_fts"], b["traj_loc_fts"], b["traj_nav_types"],
        b["traj_vp_view_lens"], sd[pfx + "embeddings.token_type_embeddings.weight"][1],
        b.get("traj_obj_img_fts"), b.get("traj_vp_obj_lens"), b.get("traj_view_dep_fts"))
    b["_traj_lens"] = lens
    return txt, txt_masks, traj, traj_masks


def _obj_tokens(b, traj):
    """vilmodel.py:748-756: object tokens of every sample's LAST panorama, zero-padded, + validity mask."""
    if b.get("traj_obj_img_fts") is None:
        return None, None
    ends = torch.cumsum(torch.tensor(b["traj_step_lens"]), 0) - 1
    vl, ol = b["traj_vp_view_lens"][ends], b["traj_vp_obj_lens"][ends]
    O = int(ol.max())
    out = traj.new_zeros(len(ends), O, traj.shape[-1])
    for i, e in enumerate(ends.tolist()):
        out[i, :int(ol[i])] = traj[e, int(vl[i]):int(vl[i]) + int(ol[i])]
    return out, seq_mask(ol, O)


def _local(sd, cfg, pfx, txt, txt_masks, b, obj, obj_masks):
    """vilmodel.py:595-615 LocalBEVEncoder.forward with the optional object tokens appended to the BEV cells."""
    bev_in = bev_input_embedding(sd, pfx + "local_encoder", b["bev_fts"], b["bev_pos_fts"], b["bev_nav_masks"])
    masks = b["bev_masks"]
    if obj is not None:
        bev_in = torch.cat([bev_in, obj], 1)
        masks = torch.cat([masks, obj_masks], 1)
    out = crossmodal_encoder(sd, pfx + "local_encoder.encoder", cfg, txt, txt_masks, bev_in, masks)
    K = cfg.bev_dim * cfg.bev_dim
    return out[:, :K], (out[:, K:] if obj is not None else None)


def _gmap_inputs(sd, cfg, b, pfx, traj, traj_masks):
    img = aggregate_gmap(traj, traj_masks, b["_traj_lens"], b["traj_step_lens"],
                         b["traj_vpids"], b["traj_cand_vpids"], b["gmap_vpids"])
    emb = gmap_input_embedding(sd, pfx + "global_encoder", img, b["gmap_step_ids"], b["gmap_pos_fts"])
    return emb, seq_mask(b["gmap_lens"], emb.shape[1])


def cmt_forward(sd, cfg, b, pfx="bert.", return_gmap_embeds=True, with_objs=False):
    """vilmodel.py:717-765 GlocalTextPathCMT.forward.  Returns (gmap, bev) or, with_objs, (gmap, bev, obj, obj_masks)."""
    b = dict(b)
    txt, txt_masks, traj, traj_masks = _common(sd, cfg, b, pfx)
    gmap = None
    if return_gmap_embeds:
        g_in, g_masks = _gmap_inputs(sd, cfg, b, pfx, traj, traj_masks)
        sp = sprel_bias(sd, pfx + "global_encoder", b["gmap_pair_dists"]) if cfg.graph_sprels else None
        gmap = crossmodal_encoder(sd, pfx + "global_encoder.encoder", cfg, txt, txt_masks, g_in, g_masks, sp)
    obj, obj_masks = _obj_tokens(b, traj)
    bev, obj_out = _local(sd, cfg, pfx, txt, txt_masks, b, obj, obj_masks)
    if with_objs:
        return gmap, bev, obj_out, obj_masks
    return gmap, bev


def cmt_forward_mlm(sd, cfg, b, pfx="bert."):
    """vilmodel.py:768-830 forward_mlm: text is the query stream of both map encoders."""
    nh = cfg.num_attention_heads
    b = dict(b)
    txt, txt_masks, traj, traj_masks = _common(sd, cfg, b, pfx)
    tm = neg_mask(txt_masks)
    g_in, g_masks = _gmap_inputs(sd, cfg, b, pfx, traj, traj_masks)
    gm = neg_mask(g_masks)
    g_txt = txt
    for i in range(cfg.num_x_layers):
        g_txt = x_layer_lang2visn(sd, f"{pfx}global_encoder.encoder.x_layers.{i}", nh, g_txt, tm, g_in, gm)
    bev_in = bev_input_embedding(sd, pfx + "local_encoder", b["bev_fts"], b["bev_pos_fts"], b["bev_nav_masks"])
    obj, obj_masks = _obj_tokens(b, traj)
    bev_masks = b["bev_masks"]
    if obj is not None:                                               # vilmodel.py:814-816
        bev_in = torch.cat([bev_in, obj], 1)
        bev_masks = torch.cat([bev_masks, obj_masks], 1)
    bm = neg_mask(bev_masks)
    b_txt = txt
    for i in range(cfg.num_x_layers):
        b_txt = x_layer_lang2visn(sd, f"{pfx}local_encoder.encoder.x_layers.{i}", nh, b_txt, tm, bev_in, bm)
    return g_txt + b_txt


def cmt_forward_sem(sd, cfg, b, sem_pred_token, pfx="bert."):
    """vilmodel.py:833-883 forward_sem."""
    if sem_pred_token == "cattn":
        b = dict(b)
        txt, txt_masks, traj, _ = _common(sd, cfg, b, pfx)
        obj, obj_masks = _obj_tokens(b, traj)
        return _local(sd, cfg, pfx, txt, txt_masks, b, obj, obj_masks)[0]
    bev = bev_input_embedding(sd, pfx + "local_encoder", b["bev_fts"], b["bev_pos_fts"], b["bev_nav_masks"])
    if sem_pred_token == "sattn":
        bm = neg_mask(b["bev_masks"])
        for i in range(cfg.num_x_layers):
            bev = x_layer_visn2visn(sd, f"{pfx}local_encoder.encoder.x_layers.{i}", cfg.num_attention_heads, bev, bm)
        return bev
    if sem_pred_token == "embed":
        return bev
    raise NotImplementedError


# ----------------------------------------------------------------------------- lift + splat
def bevpos_polar(dim):
    """bev_utils.py:39-58: (dim,dim,3) = (cos, sin, dist/(dim/2)) of the cell centres, y flipped."""
    c = torch.linspace(0.5, dim - 0.5, dim, dtype=torch.float32)
    ry, rx = torch.meshgrid(c, c, indexing="ij")
    ry = -(ry - dim / 2)
    rx = rx - dim / 2
    dis = (ry ** 2 + rx ** 2) ** 0.5
    cos = rx / dis
    sin = ry / dis
    cos[dis == 0] = 0
    sin[dis == 0] = 0
    return torch.stack([cos, sin, dis / (dim / 2)], -1)


def pixel_scale(hw=14, vfov=math.radians(90)):
    """bev_utils.py:91-137: ((u + .5 - cx) / fx) for u in 0..hw-1, fp32 (fx = fy = 7, cx = cy = 7 @ 14x14, 90 deg)."""
    f = torch.tensor(hw / (2.0 * math.tan(vfov / 2.0)), dtype=torch.float32)
    c = torch.tensor(hw / 2.0, dtype=torch.float32)
    return (torch.arange(hw, dtype=torch.float32) + 0.5 - c) / f


def lift_points(depths, T_c2w, T_w2c, S_w2c, hw=14):
    """pretrain_cmt.py:124-137 + bev_utils.py:139-172,200-248,349-378.

    depths (B,V,1,hw,hw) stored /10; returns ego-frame points (B,P,3) and no-depth mask (B,P).
    The 4x4 products are written out as explicit fp32 mul/add chains in k-order
    (((t0*x + t1*y) + t2*z) + t3*1) so that the HIP kernel, compiled with
    -ffp-contract=off, reproduces them bit for bit (cell indices are discontinuous
    in these values).
    """
    B, V = depths.shape[:2]
    z = (depths.reshape(B, V, hw, hw) * 10).to(torch.float32)
    sc = pixel_scale(hw)
    x = z * sc[None, None, None, :]
    y = z * sc[None, None, :, None]
    T = T_c2w.reshape(B, V, 4, 4)[:, :, :, None, None, :]          # (B,V,4,1,1,4)
    w = [((T[:, :, r, :, :, 0] * x + T[:, :, r, :, :, 1] * y) + T[:, :, r, :, :, 2] * z) + T[:, :, r, :, :, 3]
         for r in range(3)]
    pc = torch.stack(w, -1).reshape(B, -1, 3)                       # world frame
    pc = pc - S_w2c.reshape(B, 1, 3)
    R = T_w2c.reshape(B, 4, 4)[:, None]                             # (B,1,4,4)
    e = [((R[:, :, r, 0] * pc[..., 0] + R[:, :, r, 1] * pc[..., 1]) + R[:, :, r, 2] * pc[..., 2]) + R[:, :, r, 3]
         for r in range(3)]
    return torch.stack(e, -1), (z == 0).reshape(B, -1)


def scatter_mean(src, index, dim_size):
    """torch_scatter 2.0.9 scatter_mean for floating src (published behaviour):
    out = zeros; out.index_add(index, src); count.clamp_(min=1); out / count."""
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    out.index_add_(0, index, src)
    cnt = src.new_zeros(dim_size)
    cnt.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
    cnt.clamp_(min=1)
    return out / cnt.view(-1, *([1] * (src.dim() - 1)))


def cell_index(pc, no_depth, dim=21, res=0.5):
    """bev_utils.py:393-406: cell id (dim*z + x) per point, -1 where masked."""
    xz = (pc[..., [0, 2]] / res + (dim - 1) / 2).round()
    outside = (xz[..., 0] >= dim) | (xz[..., 1] >= dim) | (xz[..., 0] < 0) | (xz[..., 1] < 0)
    bad = no_depth | outside | (pc[..., 1] > 0.5)
    idx = (dim * xz[..., 1] + xz[..., 0]).long()
    return torch.where(bad, torch.full_like(idx, -1), idx)


def project_bev(pc, no_depth, feat, sem, dim=21, res=0.5):
    """bev_utils.py:381-430 project_bev (per-sample loop); sem=None: the continuous-environment fork's variant without
    semantic maps (bevbert_ce/pretrain/pretrain_src/model/bev_utils.py:382-417)."""
    bevs, sems, sem_masks = [], [], []
    for i in range(pc.shape[0]):
        idx = cell_index(pc[i], no_depth[i], dim, res)
        keep = idx >= 0
        bevs.append(scatter_mean(feat[i][keep], idx[keep], dim * dim))
        if sem is None:
            continue
        s = scatter_mean(sem[i][keep], idx[keep], dim * dim)
        s[s > 0] = 1
        sems.append(s)
        sem_masks.append(s.sum(1) > 0)
    if sem is None:
        return torch.stack(bevs), None, None
    return torch.stack(bevs), torch.stack(sems), torch.stack(sem_masks)


def lift_splat(cfg, batch):
    """pretrain_cmt.py:114-167: returns the five tensors the reference adds to the batch."""
    B = batch["rgbs"].shape[0]
    dim = cfg.bev_dim
    pc, nod = lift_points(batch["depths"], batch["T_c2w"], batch["T_w2c"], batch["S_w2c"], cfg.grid_hw)
    feat = batch["rgbs"].reshape(B, -1, cfg.grid_feat_size).to(torch.float32)
    sem = batch.get("sems")
    if sem is not None:
        if sem.dim() == 2:                               # compact class ids -> one-hot
            sem = F.one_hot(sem.long(), cfg.sem_classes).to(torch.float64)
        sem = sem.reshape(B, -1, cfg.sem_classes)
    bev, bsem, bsem_m = project_bev(pc, nod, feat, sem, dim, cfg.bev_res)
    pos = bevpos_polar(dim).reshape(1, dim * dim, 3).expand(B, -1, -1)
    pos = torch.cat([batch["bev_gpos_fts"].expand(-1, dim * dim, -1), pos], -1)
    return dict(bev_fts=bev, bev_masks=torch.ones(B, dim * dim, dtype=torch.bool), bev_pos_fts=pos,
                bev_sems=bsem, bev_sem_masks=bsem_m)


# ----------------------------------------------------------------------------- heads / tasks
def cls_head(sd, p, x):
    """pretrain_cmt.py:47-71 ClsPrediction / MulClsPrediction: Linear-ReLU-LN-Linear."""
    h = layer_norm(sd, p + ".net.2", torch.relu(linear(sd, p + ".net.0", x)))
    return linear(sd, p + ".net.3", h)


def mlm_head(sd, p, x):
    """vilmodel.py:258-299 BertOnlyMLMHead (decoder tied to the word embeddings)."""
    h = layer_norm(sd, p + ".predictions.transform.LayerNorm",
                   gelu_erf(linear(sd, p + ".predictions.transform.dense", x)))
    return F.linear(h, sd[p + ".predictions.decoder.weight"]) + sd[p + ".predictions.bias"]


def fuse_sap_logits(global_logits, local_logits, gmap_vpids, gmap_visited_masks, cand_vpids):
    """pretrain_cmt.py:338-356 (fine-tune twin map_nav_src/models/vilmodel.py:852-871).

    cand_vpids[i]: local candidate ids INCLUDING the leading None ([stop])."""
    fused = global_logits.clone()
    fused[:, 0] += local_logits[:, 0]
    for i in range(global_logits.shape[0]):
        visited = {vp for vp, m in zip(gmap_vpids[i], gmap_visited_masks[i].tolist()) if m}
        tmp, bw = {}, 0
        for j, vp in enumerate(cand_vpids[i]):
            if j > 0:
                if vp in visited:
                    bw = bw + local_logits[i, j]
                else:
                    tmp[vp] = local_logits[i, j]
        for j, vp in enumerate(gmap_vpids[i]):
            if j > 0 and vp not in visited:
                fused[i, j] += tmp[vp] if vp in tmp else bw
    return fused


def sap_logits(sd, cfg, b, gmap, bev, pfx=""):
    """pretrain_cmt.py:322-356 (pfx='') / map_nav_src/models/vilmodel.py:826-871 (pfx='bert.'... none)."""
    if cfg.glocal_fuse:
        center = (cfg.bev_dim * cfg.bev_dim - 1) // 2
        fw = torch.sigmoid(cls_head(sd, pfx + "sap_fuse_linear", torch.cat([gmap[:, 0], bev[:, center]], 1)))
    else:
        fw = 0.5
    g = cls_head(sd, pfx + "global_sap_head", gmap).squeeze(2) * fw
    g = g.masked_fill(b["gmap_visited_masks"], float("-inf"))
    gm = b["gmap_masks"] if "gmap_masks" in b else seq_mask(b["gmap_lens"], g.shape[1])
    g = g.masked_fill(~gm, float("-inf"))
    bi = torch.arange(bev.shape[0])[:, None]
    cand = bev[bi, b["bev_cand_idxs"]]
    cmask = b["bev_nav_masks"][bi, b["bev_cand_idxs"]]
    l = cls_head(sd, pfx + "local_sap_head", cand).squeeze(2) * (1 - fw)
    l = l.masked_fill(~cmask, float("-inf"))
    if "bev_cand_vpids" in b:
        cvp = b["bev_cand_vpids"]
    else:
        cvp = [[None] + c[-1] for c in b["traj_cand_vpids"]]
    return g, l, fuse_sap_logits(g, l, b["gmap_vpids"], b["gmap_visited_masks"], cvp)


def pretrain_forward(sd, cfg, batch, task, compute_loss=True):
    """pretrain_cmt.py:169-238 GlocalTextPathCMTPreTraining.forward (eval mode: dropouts identity).

    Does not mutate ``batch``; the lifted tensors are merged into a copy."""
    b = dict(batch)
    b.update(lift_splat(cfg, b))
    if task.startswith("mlm"):
        txt = cmt_forward_mlm(sd, cfg, b)
        sel = b["txt_labels"] != -1
        scores = mlm_head(sd, "mlm_head", txt[sel])                     # pretrain_cmt.py:254-256
        if compute_loss:
            return F.cross_entropy(scores, b["txt_labels"][sel], reduction="none")
        return scores
    if task.startswith("sap"):
        gmap, bev = cmt_forward(sd, cfg, b)
        g, l, f = sap_logits(sd, cfg, b, gmap, bev)
        if compute_loss:
            return F.cross_entropy(g, b["global_act_labels"], reduction="none") \
                + F.cross_entropy(l, b["local_act_labels"], reduction="none") \
                + F.cross_entropy(f, b["global_act_labels"], reduction="none")
        return g, l, f, b["global_act_labels"], b["local_act_labels"]
    if task.startswith("mrc"):                                              # pretrain_cmt.py:272-297
        _, _, obj, _ = cmt_forward(sd, cfg, b, return_gmap_embeds=False, with_objs=True)
        sel = b["vp_obj_mrc_masks"]
        pred = cls_head(sd, "obj_classifier", obj[sel])
        target = b["vp_obj_probs"][sel]
        if compute_loss:
            return F.kl_div(F.log_softmax(pred, -1), target, reduction="none").sum(1)
        return pred, target
    if task.startswith("og"):                                               # pretrain_cmt.py:367-389
        _, _, obj, obj_masks = cmt_forward(sd, cfg, b, return_gmap_embeds=False, with_objs=True)
        logits = cls_head(sd, "og_head", obj).squeeze(2).masked_fill(~obj_masks, float("-inf"))
        if compute_loss:
            return F.cross_entropy(logits, b["obj_labels"], reduction="none")
        return logits
    if task.startswith("sem") or task.startswith("masksem"):
        sel = b["bev_sem_masks"]
        if task.startswith("masksem"):                                  # pretrain_cmt.py:423-435
            b["bev_fts"] = b["bev_fts"].masked_fill(b["bev_mrc_masks"][..., None], 0)
            sel = sel & b["bev_mrc_masks"]
        bev = cmt_forward_sem(sd, cfg, b, cfg.sem_pred_token)
        logits = cls_head(sd, "local_sem_head", bev[sel])
        labels = b["bev_sems"][sel].float()
        if compute_loss:
            return F.binary_cross_entropy_with_logits(logits, labels, reduction="none")
        return logits, labels
    raise ValueError("invalid task")


# ----------------------------------------------------------------------------- fine-tune API
def nav_forward(sd, cfg, mode, b):
    """map_nav_src/models/vilmodel.py:889-911 GlocalTextPathNavCMT.forward (keys without 'bert.' prefix
    inside VLNBert.vln_bert; this oracle takes the inner module's keys)."""
    if mode == "language":                                              # :744-748
        txt = text_embeddings(sd, "embeddings", b["txt_ids"])
        return lang_encoder(sd, "lang_encoder", cfg, txt, b["txt_masks"])
    if mode == "panorama":                                              # :750-795
        e, m, _ = image_embeddings(sd, "img_embeddings", cfg, b["view_img_fts"], b["loc_fts"], b["nav_types"],
                                   b["view_lens"], sd["embeddings.token_type_embeddings.weight"][1],
                                   b.get("obj_img_fts"), b.get("obj_lens"))
        return e, m
    if mode == "navigation":                                            # :803-887
        g_in = gmap_input_embedding(sd, "global_encoder", b["gmap_img_embeds"], b["gmap_step_ids"], b["gmap_pos_fts"])
        sp = sprel_bias(sd, "global_encoder", b["gmap_pair_dists"]) if cfg.graph_sprels else None
        gmap = crossmodal_encoder(sd, "global_encoder.encoder", cfg, b["txt_embeds"], b["txt_masks"],
                                  g_in, b["gmap_masks"], sp)
        bev_in = bev_input_embedding(sd, "local_encoder", b["bev_fts"], b["bev_pos_fts"], b["bev_nav_masks"])
        bev = crossmodal_encoder(sd, "local_encoder.encoder", cfg, b["txt_embeds"], b["txt_masks"],
                                 bev_in, b["bev_masks"])
        g, l, f = sap_logits(sd, cfg, b, gmap, bev)
        return dict(gmap_embeds=gmap, global_logits=g, local_logits=l, fused_logits=f, obj_logits=None)
    raise ValueError(mode)


# ----------------------------------------------------------------------------- optimiser
def warmup_linear_lr(step, base_lr, warmup_steps, total_steps):
    """optim/sched.py:17-30 get_lr_sched."""
    f = step / warmup_steps if step < warmup_steps else max(0, (total_steps - step) / (total_steps - warmup_steps))
    lr = base_lr * f
    return lr if lr > 0 else 1e-8


def no_decay_key(name):
    """optim/misc.py:12-22: weight decay is switched off by SUBSTRING match."""
    return any(nd in name for nd in ("bias", "LayerNorm.bias", "LayerNorm.weight"))


def adamw_step(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.98, eps=1e-6):
    """optim/adamw.py:53-112, in place on (p, m, v); ``step`` is the 1-based count after increment."""
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if wd > 0.0:
        p.add_(p, alpha=-lr * wd)
    return p
