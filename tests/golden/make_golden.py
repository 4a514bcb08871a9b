#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference; never on the GPU box,
and nothing in tests/, smoke() or bench.py imports this file).  It commits data
only: inputs are re-derivable from seeds (vln_bevbert_amd.synthetic /
vln_bevbert_amd.weights), outputs are what the reference computed on the CPU.

Shims -- installed before the reference is imported:
  * torch_scatter (pinned 2.0.9, not installed): stub scatter_mean = index_add sum /
    clamp(count, 1).  This stub IS the definition of that primitive here.
  * cv2: empty module (only used under ``viz = False``).
  * hard .cuda() calls: Tensor.cuda -> identity; build_projector rebuilt on the CPU
    from the reference's own PointCloud / bevpos_polar.
  * transformers 5.x: init_weights()/tie_weights() are replaced; weights come from
    vln_bevbert_amd.weights (key-name seeded) and the MLM decoder is tied by hand.

Usage:  python tests/golden/make_golden.py           (writes tests/golden/*.npz)
        python tests/golden/make_golden.py --graph   (fine-tune GraphMap bookkeeping only)
        python tests/golden/make_golden.py --rxr     (xlm-roberta vocabulary only)
        python tests/golden/make_golden.py --modules --autocast --configs   (per-module vectors / the reference's own
                                                      autocast-bf16 noise / the model configuration key-values)
        python tests/golden/make_golden.py --ce      (continuous-environment fork only: its modules are also called
                                                      ``model.*``, so it needs a process of its own)
"""
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from vln_bevbert_amd import synthetic, weights  # noqa: E402
from vln_bevbert_amd.config import BevBertConfig  # noqa: E402


# ----------------------------------------------------------------------------- shims
def _install_shims():
    ts = types.ModuleType("torch_scatter")

    def scatter_mean(src, index, dim=0, dim_size=None):
        assert dim == 0
        out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
        out.index_add_(0, index, src)
        cnt = torch.zeros(dim_size, dtype=src.dtype)
        cnt.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        cnt.clamp_(min=1)
        return out / cnt.view(-1, *([1] * (src.dim() - 1)))

    ts.scatter_mean = scatter_mean
    ts.scatter_max = None
    sys.modules["torch_scatter"] = ts
    sys.modules["cv2"] = types.ModuleType("cv2")
    torch.Tensor.cuda = lambda self, *a, **k: self


def _ref_config(cfg: BevBertConfig):
    from transformers import PretrainedConfig
    d = cfg.to_dict()
    d["pretrain_tasks"] = set(d["pretrain_tasks"])
    pc = PretrainedConfig()
    for k, v in d.items():
        setattr(pc, k, v)
    pc.output_hidden_states = False
    return pc


def build_ref_pretrain(cfg, src="pretrain_src"):
    sys.path.insert(0, os.path.join(REF, src))
    from model import bev_utils, pretrain_cmt, vilmodel

    def build_projector():
        p = bev_utils.PointCloud(math.radians(90), 1, feature_map_height=14, feature_map_width=14,
                                 map_dim=cfg.bev_dim, map_res=cfg.bev_res,
                                 world_shift_origin=torch.zeros(3), z_clip_threshold=0.5,
                                 device=torch.device("cpu"))
        bp = bev_utils.bevpos_polar(cfg.bev_dim).reshape(cfg.bev_dim ** 2, 3)[None]
        return p, bp

    pretrain_cmt.build_projector = build_projector
    for cls in (vilmodel.GlocalTextPathCMT, pretrain_cmt.GlocalTextPathCMTPreTraining):
        cls.init_weights = lambda self: None
        cls.tie_weights = lambda self, *a, **k: None
    m = pretrain_cmt.GlocalTextPathCMTPreTraining(_ref_config(cfg))
    load_rule_weights(m)
    if hasattr(m, "mlm_head"):
        m.mlm_head.predictions.decoder.weight = m.bert.embeddings.word_embeddings.weight
    return m.eval()


def load_rule_weights(m):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = weights.fill_state_dict(shapes)
    missing = m.load_state_dict(sd, strict=True)
    return shapes


def build_ref_nav(cfg):
    sys.path.insert(0, os.path.join(REF, "map_nav_src"))
    from models import vilmodel as nav_vilmodel
    nav_vilmodel.GlocalTextPathNavCMT.init_weights = lambda self: None
    m = nav_vilmodel.GlocalTextPathNavCMT(_ref_config(cfg))
    load_rule_weights(m)
    return m.eval()


# ----------------------------------------------------------------------------- helpers
def npy(t):
    return t.detach().cpu().numpy()


def sub(t, step):
    """Strided subsample over the flattened tensor (keeps fixtures small)."""
    return npy(t).reshape(-1)[::step].copy()


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {os.path.relpath(path, ROOT)}  {os.path.getsize(path) / 1024:.0f} KiB")


def edge_case_points():
    """Hand-built ego-frame points hitting every branch of project_bev (bev_utils.py:393-403):
    exact .5 rounding (half-to-even), borders, outside, above-threshold, no-depth, shared and empty cells."""
    pts = [
        (0.0, 0.0, 0.0), (0.25, 0.0, 0.25), (-0.25, 0.0, -0.25),       # x/0.5+10 = 10.5 / 9.5 -> 10 / 10 (even)
        (0.75, 0.0, 0.75), (-0.75, 0.0, -0.75),                        # 11.5 -> 12 ; 8.5 -> 8
        (5.0, 0.0, 5.0), (5.24, 0.0, -5.0), (5.25, 0.0, 0.0),          # 20 (border) ; 20.48 -> 20 ; 20.5 -> 20 (even)
        (5.26, 0.0, 0.0), (-5.25, 0.0, 0.0), (-5.26, 0.0, 0.0),        # 20.52 -> 21 outside ; -0.5 -> -0 (inside!) ; -0.52 -> -1 outside
        (0.0, 0.5, 1.0), (0.0, 0.5000001, 1.0), (0.0, 0.6, 1.0),       # y == 0.5 kept ; just above dropped
        (1.0, -1.0, 1.0), (1.0, -0.2, 1.0), (1.1, 0.1, 0.9),           # three points sharing cell (12,12)
        (2.0, 0.0, -3.0), (100.0, 0.0, 0.0), (0.0, 0.0, -100.0),
    ]
    return torch.tensor(pts, dtype=torch.float32)


# ----------------------------------------------------------------------------- generators
def gen_splat(ref, cfg):
    print("splat / lift_splat")
    b = synthetic.make_batch(cfg, "sap", 2, seed=11, ragged=True)
    depths_var = (b["depths"] * 10).reshape(-1, 1, 14, 14)
    pc_w, nod = ref.projector.forward(depths_var, b["T_c2w"].reshape(-1, 4, 4))
    rb = dict(b)
    out = ref.lift_splat(rb)
    # the ego-frame points exactly as the reference computes them (pretrain_cmt.py:127-137)
    pc = pc_w.reshape(2, -1, 3) - b["S_w2c"]
    pc1 = torch.cat([pc, torch.ones(2, pc.shape[1], 1)], -1)
    pc = torch.matmul(pc1, b["T_w2c"].squeeze(1).transpose(1, 2))[:, :, :3]
    save("splat_b2",
         seed=np.int64(11), pc_ego=npy(pc), no_depth=npy(nod.reshape(2, -1)),
         bev_fts_sub=sub(out["bev_fts"], 7), bev_fts_sum=npy(out["bev_fts"].double().sum((1, 2))),
         bev_fts_cell_l1=npy(out["bev_fts"].abs().sum(-1)),
         bev_pos_fts=npy(out["bev_pos_fts"]), bev_masks=npy(out["bev_masks"]),
         bev_sems=npy(out["bev_sems"]).astype(np.uint8), bev_sem_masks=npy(out["bev_sem_masks"]))

    # hand-built edge cases straight into project_bev
    pts = edge_case_points()
    n = pts.shape[0]
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(1, n, 768, generator=g)
    sem_ids = torch.randint(0, 40, (1, n), generator=g)
    sem = torch.nn.functional.one_hot(sem_ids, 40).double()
    nod = torch.zeros(1, n, dtype=torch.bool)
    nod[0, 0] = True                                                  # the first point has no depth
    bev, obm, bsem, bsm = ref.projector.project_bev(pts[None], nod, feat, sem)
    save("splat_edge", pts=npy(pts), no_depth=npy(nod), feat=npy(feat), sem_ids=npy(sem_ids),
         bev=npy(bev.reshape(1, -1, 768)), ob_mask=npy(obm.reshape(1, -1)),
         bev_sems=npy(bsem.reshape(1, -1, 40)).astype(np.uint8), bev_sem_masks=npy(bsm.reshape(1, -1)))

    from model import bev_utils
    save("bevpos_polar", **{f"d{d}": npy(bev_utils.bevpos_polar(d)) for d in (11, 14, 21)})
    xyzhe = np.array([[1, 2, 3, 0.3, np.pi], [0, 0, 0, -1.2, 0], [-4, 1, 2, 2.5, 0.1]], dtype=np.float32)
    save("pose_matrix", xyzhe=xyzhe, T=bev_utils.transfrom3D(xyzhe))


def gen_tasks(ref, cfg, tag, B, seed, ragged, with_grads):
    print(f"tasks [{tag}]")
    arrs = {"seed": np.int64(seed), "B": np.int64(B), "ragged": np.bool_(ragged)}
    for task in ("mlm", "sap", "masksem"):
        b = synthetic.make_batch(cfg, task, B, seed=seed, ragged=ragged)
        with torch.no_grad():
            loss = ref(dict(b), task, True)
            outs = ref(dict(b), task, False)
        arrs[f"{task}_loss"] = npy(loss)
        if task == "mlm":
            arrs["mlm_scores_sub"] = sub(outs, 13)
            arrs["mlm_scores_rowmax"] = npy(outs.max(1).values)
        elif task == "sap":
            g, l, f = outs[:3]
            arrs.update(sap_global=npy(g), sap_local=npy(l), sap_fused=npy(f))
            rb = ref.lift_splat(dict(b))
            with torch.no_grad():
                gm, bev, _, _ = ref.bert(*[rb.get(k) for k in CMT_ARGS])
            arrs.update(gmap_embeds=npy(gm), bev_embeds_sub=sub(bev, 11),
                        bev_embeds_center=npy(bev[:, 220]))
        else:
            arrs.update(masksem_logits=npy(outs[0]), masksem_labels=npy(outs[1]).astype(np.uint8))
    # the other two sem_pred_token modes
    b = synthetic.make_batch(cfg, "sem", B, seed=seed, ragged=ragged)
    for tok in ("sattn", "embed"):
        ref.sem_pred_token = tok
        with torch.no_grad():
            lg, _ = ref(dict(b), "sem", False)
        arrs[f"sem_{tok}_logits_sub"] = sub(lg, 3)
    ref.sem_pred_token = cfg.sem_pred_token
    with torch.no_grad():
        lg, lb = ref(dict(b), "sem", False)
    arrs["sem_cattn_logits_sub"] = sub(lg, 3)
    arrs["sem_n_rows"] = np.int64(lg.shape[0])

    if with_grads:
        for task in ("mlm", "sap", "masksem"):
            ref.zero_grad(set_to_none=True)
            b = synthetic.make_batch(cfg, task, B, seed=seed, ragged=ragged)
            ref(dict(b), task, True).mean().backward()          # train_r2r.py:262-273
            n_with = 0
            for k, p in ref.named_parameters():
                if p.grad is None:
                    continue
                n_with += 1
                if k in GRAD_KEYS:
                    arrs[f"{task}_grad::{k}"] = sub(p.grad, 97 if p.numel() > 4096 else 1)
                arrs.setdefault(f"{task}_gradnorm_keys", [])
            arrs[f"{task}_n_params_with_grad"] = np.int64(n_with)
            arrs[f"{task}_grad_sqnorm"] = np.float64(
                sum(float((p.grad.double() ** 2).sum()) for p in ref.parameters() if p.grad is not None))
            arrs.pop(f"{task}_gradnorm_keys")
        ref.zero_grad(set_to_none=True)
    save(f"tasks_{tag}", **arrs)


CMT_ARGS = ["txt_ids", "txt_lens", "traj_view_img_fts", "traj_obj_img_fts", "traj_loc_fts", "traj_nav_types",
            "traj_step_lens", "traj_vp_view_lens", "traj_vp_obj_lens", "traj_vpids", "traj_cand_vpids",
            "gmap_lens", "gmap_step_ids", "gmap_pos_fts", "gmap_pair_dists", "gmap_vpids",
            "bev_fts", "bev_pos_fts", "bev_masks", "bev_nav_masks"]

GRAD_KEYS = {
    "bert.embeddings.word_embeddings.weight",
    "bert.embeddings.token_type_embeddings.weight",
    "bert.embeddings.LayerNorm.weight",
    "bert.lang_encoder.layer.0.attention.self.query.weight",
    "bert.lang_encoder.layer.1.output.dense.bias",
    "bert.img_embeddings.img_linear.weight",
    "bert.img_embeddings.pano_encoder.layers.0.self_attn.in_proj_weight",
    "bert.img_embeddings.pano_encoder.layers.0.norm1.weight",
    "bert.local_encoder.bev_fts_embeddings.0.weight",
    "bert.local_encoder.encoder.x_layers.0.visual_attention.att.key.weight",
    "bert.local_encoder.encoder.x_layers.1.visn_self_att.self.value.bias",
    "bert.local_encoder.encoder.x_layers.0.lang_inter.dense.weight",
    "bert.global_encoder.sprel_linear.weight",
    "bert.global_encoder.sprel_linear.bias",
    "bert.global_encoder.gmap_step_embeddings.weight",
    "bert.global_encoder.encoder.x_layers.1.visn_output.LayerNorm.weight",
    "mlm_head.predictions.bias",
    "global_sap_head.net.0.weight",
    "sap_fuse_linear.net.3.weight",
    "local_sem_head.net.3.bias",
}


def gen_configs():
    """Key/value content of the reference's model configuration files: what PretrainedConfig.from_json_file hands the
    model constructors (pretrain_src/train_r2r.py:102-113).  Data, not source: the product's boundary test rebuilds
    the attribute bag from it and checks that nothing the JSON lacks (BEV resolution, grid shape, semantic classes ...)
    is needed from the caller."""
    import json
    out = {"_comment": "key/value content of the reference's configs/{r2r,rxr,rvr}_model.json, written by "
                       "tests/golden/make_golden.py --configs"}
    for tag in ("r2r", "rxr", "rvr"):
        with open(os.path.join(REF, "configs", f"{tag}_model.json")) as f:
            out[tag] = json.load(f)
    with open(os.path.join(OUT, "model_configs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("  wrote tests/golden/model_configs.json")


def gen_modules(ref, cfg):
    """Per-module vectors: one BertLayer, one GraphLXRTXLayer in each of its three forwards
    (with / without graph_sprels), the panorama encoder alone with ragged view counts, and ImageEmbeddings' embedding
    sum.  Inputs are re-derivable: torch.randn from the stated generator seeds, in the order written here."""
    print("per-module vectors")
    sys.path.insert(0, os.path.join(REF, "pretrain_src"))
    from model.ops import extend_neg_masks
    g = torch.Generator().manual_seed(4242)
    arrs = {"seed": np.int64(4242)}
    with torch.no_grad():
        # BertLayer (vilmodel.py:195-208): text layer 0, lens (11, 6)
        x = torch.randn(2, 11, 768, generator=g)
        m = torch.arange(11)[None] < torch.tensor([11, 6])[:, None]
        arrs["bert_layer"] = npy(ref.bert.lang_encoder.layer[0](x, extend_neg_masks(m))[0])
        # GraphLXRTXLayer (vilmodel.py:365-421): global-map encoder layer 0
        layer = ref.bert.global_encoder.encoder.x_layers[0]
        lang = torch.randn(2, 9, 768, generator=g)
        visn = torch.randn(2, 7, 768, generator=g)
        spr = torch.randn(2, 7, 7, generator=g)
        lm = torch.arange(9)[None] < torch.tensor([9, 4])[:, None]
        vm = torch.arange(7)[None] < torch.tensor([5, 7])[:, None]
        el, ev = extend_neg_masks(lm), extend_neg_masks(vm)
        arrs["x_visn_sprels"] = npy(layer(lang, el, visn, ev, graph_sprels=spr[:, None]))
        arrs["x_visn"] = npy(layer(lang, el, visn, ev, graph_sprels=None))
        arrs["x_lang2visn"] = npy(layer.forward_lang2visn(lang, el, visn, ev))
        arrs["x_visn2visn"] = npy(layer.forward_visn2visn(visn, ev))
        # panorama encoder alone (transformer.py:133-182, pre-norm; vilmodel.py:527-532), view counts (36, 20, 5)
        pano = torch.randn(3, 36, 768, generator=g)
        pm = torch.arange(36)[None] < torch.tensor([36, 20, 5])[:, None]
        out = ref.bert.img_embeddings.pano_encoder(pano, src_key_padding_mask=pm.logical_not())
        arrs["pano_encoder"] = npy(out)
        arrs["pano_valid"] = npy(pm)
        # ImageEmbeddings without the pano encoder: the embedding sum + LayerNorm (vilmodel.py:494-524)
        ie = ref.bert.img_embeddings
        vf = torch.randn(3, 36, cfg.image_feat_size, generator=g)
        lf = torch.randn(3, 36, 7, generator=g)
        nt = torch.randint(0, 3, (3, 36), generator=g)
        te = ref.bert.embeddings.token_type_embeddings
        e = ie.img_layer_norm(ie.img_linear(vf)) + ie.loc_layer_norm(ie.loc_linear(lf)) + ie.nav_type_embedding(nt) \
            + te(torch.ones(1, 1).long())
        arrs["img_embed_sum_ln"] = npy(ie.layer_norm(e))
        # LocalBEVEncoder's input embedding (vilmodel.py:585-593)
        le = ref.bert.local_encoder
        bf = torch.randn(2, 441, 768, generator=g)
        bp = torch.randn(2, 441, 10, generator=g)
        bn = torch.rand(2, 441, generator=g) < 0.1
        be = le.bev_fts_embeddings(bf) + le.bev_pos_embeddings(bp) + le.nav_type_embedding(bn.long())
        arrs["bev_input_embedding_sub"] = sub(be, 5)
    save("modules_tiny", **arrs)


def gen_autocast(ref, cfg):
    """How far the REFERENCE's own autocast forward sits from its fp32 forward (the yardstick for 'within 1e-2 bf16'):
    torch.autocast('cpu', bfloat16) around the reference model, same batches as tasks_tiny_b3_ragged."""
    print("reference autocast-bf16 vs fp32 (tiny, B=3 ragged)")
    arrs = {}
    for task in ("mlm", "sap", "masksem"):
        b = synthetic.make_batch(cfg, task, 3, seed=7, ragged=True)
        with torch.no_grad():
            want = ref(dict(b), task, False)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                got = ref(dict(b), task, False)
        want = want if torch.is_tensor(want) else want[2] if task == "sap" else want[0]
        got = got if torch.is_tensor(got) else got[2] if task == "sap" else got[0]
        w, gt = npy(want.float()).astype(np.float64), npy(got.float()).astype(np.float64)
        if w.shape != gt.shape:
            # the lift runs under autocast too (train_r2r.py:256-258 wraps the whole forward): bf16 point coordinates
            # move points across cell borders, so even the SET of supervised cells differs from the fp32 run
            arrs[f"{task}_rows_fp32"], arrs[f"{task}_rows_bf16"] = np.int64(w.shape[0]), np.int64(gt.shape[0])
            print(f"   {task}: row count differs under autocast ({w.shape[0]} fp32 vs {gt.shape[0]} bf16) -- not comparable")
            continue
        fin = np.isfinite(w) & np.isfinite(gt)
        scale = np.abs(w[fin]).max()
        err = np.abs(w[fin] - gt[fin])
        arrs[f"{task}_max_rel"] = np.float64(err.max() / scale)
        arrs[f"{task}_mean_rel"] = np.float64(err.mean() / scale)
        print(f"   {task}: max-abs / absmax = {err.max() / scale:.3e}   mean-abs / absmax = {err.mean() / scale:.3e}")
    save("ref_autocast_noise", **arrs)



def _task_outputs(ref, cfg, B, seed, ragged):
    """The forward tensors test_gpu_model._check_tasks compares, under whatever autocast state the caller set."""
    out = {}
    mk = lambda task: synthetic.make_batch(cfg, task, B, seed=seed, ragged=ragged)
    with torch.no_grad():
        out["mlm_loss"] = ref(dict(mk("mlm")), "mlm", True)
        out["mlm_scores_sub"] = ref(dict(mk("mlm")), "mlm", False)
        out["sap_loss"] = ref(dict(mk("sap")), "sap", True)
        g, l, f = ref(dict(mk("sap")), "sap", False)[:3]
        out.update(sap_global=g, sap_local=l, sap_fused=f)
        rb = ref.lift_splat(dict(mk("sap")))
        gm, bev, _, _ = ref.bert(*[rb.get(k) for k in CMT_ARGS])
        out.update(gmap_embeds=gm, bev_embeds_sub=bev)
        out["masksem_loss"] = ref(dict(mk("masksem")), "masksem", True)
        out["masksem_logits"] = ref(dict(mk("masksem")), "masksem", False)[0]
        for tok in ("sattn", "embed", "cattn"):
            ref.sem_pred_token = tok
            out[f"sem_{tok}_logits_sub"] = ref(dict(mk("sem")), "sem", False)[0]
        ref.sem_pred_token = cfg.sem_pred_token
    return {k: v.float() for k, v in out.items()}


def _task_grads(ref, cfg, B, seed, ragged):
    out = {}
    for task in ("mlm", "sap", "masksem"):
        ref.zero_grad(set_to_none=True)
        b = synthetic.make_batch(cfg, task, B, seed=seed, ragged=ragged)
        ref(dict(b), task, True).mean().backward()
        for k, p in ref.named_parameters():
            if p.grad is not None and k in GRAD_KEYS:
                out[f"{task}_grad::{k}"] = sub(p.grad.float(), 97 if p.numel() > 4096 else 1)
        out[f"{task}_grad_sqnorm"] = np.float64(sum(float((p.grad.double() ** 2).sum()) for p in ref.parameters()
                                                    if p.grad is not None))
    ref.zero_grad(set_to_none=True)
    return out


def _obj_outputs(ref, cfg, tasks, B, seed):
    out = {}
    with torch.no_grad():
        for task in tasks:
            b = synthetic.make_batch(cfg, task, B, seed=seed, ragged=True)
            out[f"{task}_loss"] = ref(dict(b), task, True)
            outs = ref(dict(b), task, False)
            if task == "og":
                out["og_logits"] = outs
            elif task == "mrc":
                out["mrc_pred"] = outs[0]
            elif task == "sap":
                out["sap_fused"] = outs[2]
    return {k: v.float() for k, v in out.items()}


def _obj_grads(ref, cfg, tasks, B, seed):
    out = {}
    for task in tasks:
        ref.zero_grad(set_to_none=True)
        b = synthetic.make_batch(cfg, task, B, seed=seed, ragged=True)
        ref(dict(b), task, True).mean().backward()
        for k, p in ref.named_parameters():
            if p.grad is not None and k in OBJ_GRAD_KEYS:
                out[f"{task}_grad::{k}"] = sub(p.grad.float(), 97 if p.numel() > 4096 else 1)
        out[f"{task}_grad_sqnorm"] = np.float64(sum(float((p.grad.double() ** 2).sum()) for p in ref.parameters()
                                                    if p.grad is not None))
    ref.zero_grad(set_to_none=True)
    return out


def merge_autocast_errors(ref, tag, outputs, grads=None):
    """gen_autocast_errors for the configurations that are generated by their own functions / processes (RxR vocabulary,
    CE fork): ``outputs()`` / ``grads()`` return {name: tensor} / {name: sub-sampled array}; the per-tensor errors of the
    reference's autocast-bf16 run against its fp32 run are MERGED into tests/golden/ref_autocast_errors.npz."""
    print(f"reference autocast-bf16 vs fp32, per tensor [{tag}]")
    path = os.path.join(OUT, "ref_autocast_errors.npz")
    arrs = dict(np.load(path)) if os.path.exists(path) else {}
    orig = ref.lift_splat

    def lift_fp32(batch):
        with torch.autocast("cpu", enabled=False):
            return orig(batch)
    want = outputs()
    want_g = grads() if grads else {}
    ref.lift_splat = lift_fp32
    try:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            got = outputs()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            got_g = grads() if grads else {}
    finally:
        del ref.lift_splat
    for k, w in want.items():
        w, gt = npy(w.float()).astype(np.float64), npy(got[k].float()).astype(np.float64)
        fin = np.isfinite(w) & np.isfinite(gt)
        scale = max(1e-6, np.abs(w[fin]).max())
        err = np.abs(w[fin] - gt[fin])
        arrs[f"{tag}::{k}::max_rel"] = np.float64(err.max() / scale)
        arrs[f"{tag}::{k}::mean_rel"] = np.float64(err.mean() / scale)
        print(f"   {k:28s} max-abs/absmax {err.max() / scale:.3e}  mean-abs/absmax {err.mean() / scale:.3e}")
    for k, w in want_g.items():
        w, gt = np.asarray(w, dtype=np.float64), np.asarray(got_g[k], dtype=np.float64)
        l2 = np.linalg.norm(gt - w) / max(1e-12, np.linalg.norm(w))
        arrs[f"{tag}::{k}::rel_l2"] = np.float64(l2)
        print(f"   {k:60s} rel-L2 {l2:.3e}")
    save("ref_autocast_errors", **arrs)


def gen_autocast_errors(ref, cfg, tag, B, seed, ragged, arrs, obj_tasks=None):
    """The yardstick of the bf16 parity gates: how far the REFERENCE's own autocast-bf16 run
    (train_r2r.py:256-258 torch.cuda.amp.autocast; bf16 per BASELINE.json) sits from its fp32 run, per compared tensor --
    forward outputs as max-abs / absmax and mean-abs / absmax, gradients of the named parameters as relative L2 over the
    same sub-sampled entries the tests read.  The lift + splat is kept in fp32 on both sides (the product lifts in fp32;
    under a whole-forward autocast the reference's bf16 point coordinates move points across cell borders and even the
    set of supervised cells changes: that figure stays in ref_autocast_noise.npz)."""
    print(f"reference autocast-bf16 vs fp32, per tensor [{tag}]")
    orig = ref.lift_splat

    def lift_fp32(batch):
        with torch.autocast("cpu", enabled=False):
            return orig(batch)
    outputs = (lambda: _obj_outputs(ref, cfg, obj_tasks, B, seed)) if obj_tasks else \
        (lambda: _task_outputs(ref, cfg, B, seed, ragged))
    grads = (lambda: _obj_grads(ref, cfg, obj_tasks, B, seed)) if obj_tasks else \
        (lambda: _task_grads(ref, cfg, B, seed, ragged))
    want = outputs()
    want_g = grads()
    ref.lift_splat = lift_fp32
    try:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            got = outputs()
        # a context of its own: the weight casts cached by the no_grad forwards above carry no autograd history
        with torch.autocast("cpu", dtype=torch.bfloat16):
            got_g = grads()
    finally:
        del ref.lift_splat
    for k, w in want.items():
        w, gt = npy(w).astype(np.float64), npy(got[k]).astype(np.float64)
        assert w.shape == gt.shape, (k, w.shape, gt.shape)
        fin = np.isfinite(w) & np.isfinite(gt)
        scale = max(1e-6, np.abs(w[fin]).max())
        err = np.abs(w[fin] - gt[fin])
        arrs[f"{tag}::{k}::max_rel"] = np.float64(err.max() / scale)
        arrs[f"{tag}::{k}::mean_rel"] = np.float64(err.mean() / scale)
        print(f"   {k:28s} max-abs/absmax {err.max() / scale:.3e}  mean-abs/absmax {err.mean() / scale:.3e}")
    for k, w in want_g.items():
        if k.endswith("_grad_sqnorm"):
            arrs[f"{tag}::{k}::rel"] = np.float64(abs(got_g[k] - w) / w)
            print(f"   {k:60s} rel {abs(got_g[k] - w) / w:.3e}")
            continue
        w, gt = np.asarray(w, dtype=np.float64), np.asarray(got_g[k], dtype=np.float64)
        l2 = np.linalg.norm(gt - w) / max(1e-12, np.linalg.norm(w))
        arrs[f"{tag}::{k}::rel_l2"] = np.float64(l2)
        print(f"   {k:60s} rel-L2 {l2:.3e}")


FULLSIZE = dict(fwd_batch=64, fwd_seed=3000, bwd_batch=16, bwd_seed=3100, txt_len=80,
                grad_keys=("bert.local_encoder.encoder.x_layers.0.visn_inter.dense.weight",
                           "bert.lang_encoder.layer.0.attention.self.query.weight",
                           "bert.local_encoder.bev_fts_embeddings.0.weight", "global_sap_head.net.0.weight",
                           "bert.embeddings.word_embeddings.weight",
                           "bert.global_encoder.encoder.x_layers.1.visual_attention.att.key.weight"))


def gen_fullsize():
    """BASELINE.json configs[1] at its real size through the REFERENCE: full R2R model, batch 64,
    80 tokens -- per-sample SAP and MLM losses (forward, eval); batch 16 -- gradient norm and sub-sampled named
    gradients; each in fp32 and under the reference's own autocast-bf16 (lift + splat in fp32 on both sides, as in
    gen_autocast_errors), so that the bf16 gates of tests/test_gpu_model.py::test_full_size_parity_vs_oracle are held to the
    reference's own error AT THIS SIZE.  Slow (minutes on 8 cores): run on its own with --fullsize."""
    full = BevBertConfig()
    ref = build_ref_pretrain(full)
    F_ = FULLSIZE
    arrs = {"fwd_batch": np.int64(F_["fwd_batch"]), "fwd_seed": np.int64(F_["fwd_seed"]), "bwd_batch": np.int64(F_["bwd_batch"]),
            "bwd_seed": np.int64(F_["bwd_seed"]), "txt_len": np.int64(F_["txt_len"])}
    orig = ref.lift_splat

    def lift_fp32(batch):
        with torch.autocast("cpu", enabled=False):
            return orig(batch)

    def forward_losses():
        out = {}
        with torch.no_grad():
            for task in ("sap", "mlm"):
                b = synthetic.make_batch(full, task, F_["fwd_batch"], seed=F_["fwd_seed"], txt_len=F_["txt_len"])
                out[f"{task}_loss"] = ref(dict(b), task, True).float()
        return out

    def grads():
        out = {}
        for task in ("sap", "mlm"):
            ref.zero_grad(set_to_none=True)
            b = synthetic.make_batch(full, task, F_["bwd_batch"], seed=F_["bwd_seed"], txt_len=F_["txt_len"])
            ref(dict(b), task, True).mean().backward()
            for k, p_ in ref.named_parameters():
                if k in F_["grad_keys"]:
                    out[f"{task}_grad::{k}"] = None if p_.grad is None else sub(p_.grad.float(), 97 if p_.numel() > 4096 else 1)
            out[f"{task}_grad_sqnorm"] = np.float64(sum(float((p_.grad.double() ** 2).sum()) for p_ in ref.parameters()
                                                        if p_.grad is not None))
        ref.zero_grad(set_to_none=True)
        return out

    import time
    t0 = time.perf_counter()
    want = forward_losses()
    print(f"  reference fp32 forward at batch {F_['fwd_batch']}: {time.perf_counter() - t0:.1f} s")
    t0 = time.perf_counter()
    want_g = grads()
    print(f"  reference fp32 forward + backward at batch {F_['bwd_batch']}: {time.perf_counter() - t0:.1f} s")
    ref.lift_splat = lift_fp32
    try:
        t0 = time.perf_counter()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            got = forward_losses()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            got_g = grads()
        print(f"  the same under autocast-bf16: {time.perf_counter() - t0:.1f} s")
    finally:
        del ref.lift_splat
    for k, w in want.items():
        arrs[k] = npy(w)
        w64, g64 = npy(w).astype(np.float64), npy(got[k]).astype(np.float64)
        scale = max(1e-6, np.abs(w64).max())
        arrs[f"ref_autocast::{k}::max_rel"] = np.float64(np.abs(w64 - g64).max() / scale)
        arrs[f"ref_autocast::{k}::mean_rel"] = np.float64(np.abs(w64 - g64).mean() / scale)
        print(f"   {k:12s} n={w64.size:5d} reference autocast max-abs/absmax {np.abs(w64 - g64).max() / scale:.3e}")
    for k, w in want_g.items():
        if w is None:
            continue
        if k.endswith("_grad_sqnorm"):
            arrs[k] = w
            arrs[f"ref_autocast::{k}::rel"] = np.float64(abs(got_g[k] - w) / w)
            continue
        arrs[k] = np.asarray(w, dtype=np.float32)
        w64, g64 = np.asarray(w, dtype=np.float64), np.asarray(got_g[k], dtype=np.float64)
        l2 = np.linalg.norm(g64 - w64) / max(1e-12, np.linalg.norm(w64))
        arrs[f"ref_autocast::{k}::rel_l2"] = np.float64(l2)
        print(f"   {k:80s} reference autocast rel-L2 {l2:.3e}")
    save("tasks_r2r_fullsize", **arrs)


CURVE = dict(n_steps=100, batch=2, lr=1e-4, warmup=10, total=200, wd=0.01, betas=(0.9, 0.98), clip=5.0,
             ratio="mlm.5.sap.5.masksem.1", sampler_seed=1, batch_seed0=50)


def gen_curve():
    """north_star: "loss curves overlapping for 100 steps".  The REFERENCE trains here: its model
    (pretrain_src/model), its AdamW (optim/adamw.py) behind build_optimizer's two parameter groups
    (optim/misc.py:12-37), its schedule (optim/sched.py:24-30) and its loop body (train_r2r.py:256-313: loss.mean(),
    clip_grad_norm_ 5.0, optimizer.step, optimizer.zero_grad) on 100 synthetic batches, dropout disabled (eval mode:
    the model has no other train/eval difference).  zero_grad is called with set_to_none=False, the default of the
    reference's pinned torch 1.9.1 (environment.yaml:245): a parameter that has had a gradient once keeps being
    stepped with a zero gradient on steps whose task does not use it."""
    print("100-step training curve from the reference loop [tiny, 1+1 layers]")
    from vln_bevbert_amd.train import TaskSampler
    cfg = BevBertConfig.tiny(num_l_layers=1, num_x_layers=1, vocab_size=600)
    ref = build_ref_pretrain(cfg)
    sys.path.insert(0, os.path.join(REF, "pretrain_src"))
    from optim.misc import build_optimizer
    from optim.sched import get_lr_sched
    opts = types.SimpleNamespace(weight_decay=CURVE["wd"], optim="adamw", learning_rate=CURVE["lr"],
                                 betas=CURVE["betas"], warmup_steps=CURVE["warmup"], num_train_steps=CURVE["total"])
    for p in ref.parameters():
        p.requires_grad_(True)
    optimizer = build_optimizer(ref, opts)
    optimizer.zero_grad()
    optimizer.step()                                        # train_r2r.py:244-246 (no-op: no gradients yet)
    sampler = TaskSampler(CURVE["ratio"], seed=CURVE["sampler_seed"])
    names = ["mlm", "sap", "masksem"]
    losses, norms, tasks = [], [], []
    for i in range(CURVE["n_steps"]):
        t = sampler.next()
        b = synthetic.make_batch(cfg, t, CURVE["batch"], seed=CURVE["batch_seed0"] + i, ragged=True)
        loss = ref(dict(b), t, True).mean()
        loss.backward()
        lr = get_lr_sched(i + 1, opts)
        for g in optimizer.param_groups:
            g["lr"] = lr
        gn = torch.nn.utils.clip_grad_norm_(ref.parameters(), CURVE["clip"])
        optimizer.step()
        optimizer.zero_grad(set_to_none=False)
        losses.append(float(loss.detach()))
        norms.append(float(gn))
        tasks.append(names.index(t))
        if i % 10 == 0:
            print(f"   step {i:3d} {t:8s} loss {losses[-1]:.5f} |g| {norms[-1]:.4f}")
    save("train_curve_tiny", losses=np.asarray(losses, dtype=np.float64), grad_norms=np.asarray(norms, dtype=np.float64),
         tasks=np.asarray(tasks, dtype=np.int64), **{k: np.asarray(v) for k, v in CURVE.items() if k != "ratio"})



def gen_objects(cfg, tag, tasks, seed):
    """REVERIE-style object tokens (configs/rvr_model.json): MLM / MRC / SAP / OG with objects appended to the
    panoramas and to the BEV cells; `tag` selects shared (obj_feat == image_feat) or separate obj_linear weights."""
    print(f"object tokens [{tag}]")
    ref = build_ref_pretrain(cfg)
    with open(os.path.join(OUT, f"pretrain_state_dict_keys_{tag}.txt"), "w") as f:
        for k, v in ref.state_dict().items():
            f.write(f"{k} {tuple(v.shape)}\n")
    B = 4
    arrs = {"seed": np.int64(seed), "B": np.int64(B)}
    for task in tasks:
        b = synthetic.make_batch(cfg, task, B, seed=seed, ragged=True)
        with torch.no_grad():
            arrs[f"{task}_loss"] = npy(ref(dict(b), task, True))
            outs = ref(dict(b), task, False)
        if task == "mlm":
            arrs["mlm_scores_sub"] = sub(outs, 13)
        elif task == "sap":
            arrs.update(sap_global=npy(outs[0]), sap_local=npy(outs[1]), sap_fused=npy(outs[2]))
        elif task == "mrc":
            arrs.update(mrc_pred_sub=sub(outs[0], 7), mrc_n=np.int64(outs[0].shape[0]))
        elif task == "og":
            arrs["og_logits"] = npy(outs)
        ref.zero_grad(set_to_none=True)
        ref(dict(b), task, True).mean().backward()
        arrs[f"{task}_grad_sqnorm"] = np.float64(
            sum(float((p.grad.double() ** 2).sum()) for p in ref.parameters() if p.grad is not None))
        arrs[f"{task}_n_params_with_grad"] = np.int64(sum(p.grad is not None for p in ref.parameters()))
        for k, p in ref.named_parameters():
            if p.grad is not None and k in OBJ_GRAD_KEYS:
                arrs[f"{task}_grad::{k}"] = sub(p.grad, 97 if p.numel() > 4096 else 1)
    save(f"tasks_{tag}", **arrs)


OBJ_GRAD_KEYS = {
    "bert.img_embeddings.img_linear.weight", "bert.img_embeddings.obj_linear.weight",
    "bert.img_embeddings.obj_layer_norm.weight", "bert.img_embeddings.nav_type_embedding.weight",
    "obj_classifier.net.3.weight", "og_head.net.0.weight", "og_head.net.3.bias",
    "bert.local_encoder.encoder.x_layers.0.visn_self_att.self.key.weight",
    "bert.embeddings.word_embeddings.weight",
}


def gen_nav(cfg):
    print("fine-tune API (GlocalTextPathNavCMT)")
    nav = build_ref_nav(cfg)
    keys = sorted(nav.state_dict().keys())
    B = 3
    pb = synthetic.make_batch(cfg, "sap", B, seed=23, ragged=True)
    arrs = {"seed": np.int64(23), "n_keys": np.int64(len(keys))}
    with torch.no_grad():
        txt_masks = torch.arange(pb["txt_ids"].shape[1])[None] < pb["txt_lens"][:, None]
        txt = nav("language", {"txt_ids": pb["txt_ids"], "txt_masks": txt_masks})
        arrs["txt_embeds_sub"] = sub(txt, 7)
        # panorama: the LAST step of every sample
        ends = np.cumsum(pb["traj_step_lens"]) - 1
        pano_in = {"view_img_fts": pb["traj_view_img_fts"][ends], "obj_img_fts": None,
                   "loc_fts": pb["traj_loc_fts"][ends], "nav_types": pb["traj_nav_types"][ends],
                   "view_lens": pb["traj_vp_view_lens"][ends], "obj_lens": None}
        pano, pmask = nav("panorama", pano_in)
        arrs["pano_embeds_sub"] = sub(pano, 5)
        arrs["pano_masks"] = npy(pmask)
        # navigation: gmap_img_embeds are given (GraphMap averages them on the host in the agent)
        G = int(pb["gmap_lens"].max())
        g = torch.Generator().manual_seed(99)
        gimg = torch.randn(B, G, 768, generator=g)
        gimg[:, 0] = 0
        gmasks = torch.arange(G)[None] < pb["gmap_lens"][:, None]
        lifted = build_ref_pretrain_cached["m"].lift_splat(dict(pb))
        nav_in = {
            "txt_embeds": txt, "txt_masks": txt_masks, "gmap_img_embeds": gimg,
            "gmap_step_ids": pb["gmap_step_ids"], "gmap_pos_fts": pb["gmap_pos_fts"], "gmap_masks": gmasks,
            "gmap_pair_dists": pb["gmap_pair_dists"], "gmap_visited_masks": pb["gmap_visited_masks"],
            "gmap_vpids": pb["gmap_vpids"],
            "bev_fts": lifted["bev_fts"], "bev_pos_fts": lifted["bev_pos_fts"], "bev_masks": lifted["bev_masks"],
            "bev_nav_masks": pb["bev_nav_masks"], "bev_cand_idxs": pb["bev_cand_idxs"],
            "bev_cand_vpids": [[None] + c[-1] for c in pb["traj_cand_vpids"]],
            "obj_embeds": None, "obj_masks": None,
        }
        out = nav("navigation", nav_in)
        arrs.update(nav_gmap_embeds=npy(out["gmap_embeds"]), nav_global=npy(out["global_logits"]),
                    nav_local=npy(out["local_logits"]), nav_fused=npy(out["fused_logits"]))
    save("nav_tiny", **arrs)
    with open(os.path.join(OUT, "nav_state_dict_keys.txt"), "w") as f:
        for k in keys:
            f.write(f"{k} {tuple(nav.state_dict()[k].shape)}\n")


def gen_adamw():
    print("AdamW trajectory (optim/adamw.py) + lr schedule (optim/sched.py)")
    sys.path.insert(0, os.path.join(REF, "pretrain_src"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_adamw", os.path.join(REF, "pretrain_src/optim/adamw.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    spec2 = importlib.util.spec_from_file_location("ref_sched", os.path.join(REF, "pretrain_src/optim/sched.py"))
    sched = importlib.util.module_from_spec(spec2)
    spec2.loader.exec_module(sched)
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(257, generator=g)
    grads = [torch.randn(257, generator=g) * (10.0 ** (k - 1)) for k in range(4)]
    traj = {}
    for wd in (0.01, 0.0):
        p = torch.nn.Parameter(p0.clone())
        opt = mod.AdamW([{"params": [p], "weight_decay": wd}], lr=5e-5, betas=(0.9, 0.98))
        steps = []
        for k, gr in enumerate(grads):
            for grp in opt.param_groups:
                grp["lr"] = 5e-5 * (k + 1) / 4
            p.grad = gr.clone()
            opt.step()
            steps.append(npy(p).copy())
        traj[f"wd{wd}"] = np.stack(steps)

    class O:
        learning_rate, warmup_steps, num_train_steps = 5e-5, 10000, 100000
    lr_steps = np.array([0, 1, 5000, 9999, 10000, 10001, 55000, 99999, 100000, 100001])
    lrs = np.array([sched.get_lr_sched(int(s), O) for s in lr_steps])
    save("adamw", p0=npy(p0), grads=np.stack([npy(x) for x in grads]), lr_steps=lr_steps, lrs=lrs, **traj)


def gen_keys(ref, tag):
    with open(os.path.join(OUT, f"pretrain_state_dict_keys_{tag}.txt"), "w") as f:
        for k, v in ref.state_dict().items():
            f.write(f"{k} {tuple(v.shape)}\n")


def gen_ce():
    """bevbert_ce/pretrain/pretrain_src: 11x11 BEV @ 1 m, depth-feature branch, 4-d location features, 2 nav types,
    tasks mlm + sap, no semantic maps (run_pt/r2r_model_config_dep.json; its pretrain_cmt.py:16-17,100-135)."""
    print("continuous-environment fork [tiny_ce]")
    cfg = BevBertConfig.ce(num_l_layers=2, num_x_layers=2, num_pano_layers=1, vocab_size=1200,
                           max_position_embeddings=128)
    ref = build_ref_pretrain(cfg, src="bevbert_ce/pretrain/pretrain_src")
    gen_keys(ref, "tiny_ce")
    B, seed = 3, 41
    arrs = {"seed": np.int64(seed), "B": np.int64(B)}
    for task in ("mlm", "sap"):
        b = synthetic.make_batch(cfg, task, B, seed=seed, ragged=True)
        with torch.no_grad():
            arrs[f"{task}_loss"] = npy(ref(dict(b), task, True))
            outs = ref(dict(b), task, False)
        if task == "mlm":
            arrs["mlm_scores_sub"] = sub(outs, 13)
        else:
            arrs.update(sap_global=npy(outs[0]), sap_local=npy(outs[1]), sap_fused=npy(outs[2]))
        ref.zero_grad(set_to_none=True)
        ref(dict(b), task, True).mean().backward()
        arrs[f"{task}_grad_sqnorm"] = np.float64(
            sum(float((p.grad.double() ** 2).sum()) for p in ref.parameters() if p.grad is not None))
        arrs[f"{task}_n_params_with_grad"] = np.int64(sum(p.grad is not None for p in ref.parameters()))
        for k, p in ref.named_parameters():
            if p.grad is not None and k in CE_GRAD_KEYS:
                arrs[f"{task}_grad::{k}"] = sub(p.grad, 97 if p.numel() > 4096 else 1)
    save("tasks_tiny_ce", **arrs)

    def outputs():
        out = {}
        with torch.no_grad():
            for task in ("mlm", "sap"):
                b = synthetic.make_batch(cfg, task, B, seed=seed, ragged=True)
                out[f"{task}_loss"] = ref(dict(b), task, True)
                o = ref(dict(b), task, False)
                if task == "sap":
                    out["sap_fused"] = o[2]
        return out

    def grads():
        out = {}
        for task in ("mlm", "sap"):
            ref.zero_grad(set_to_none=True)
            b = synthetic.make_batch(cfg, task, B, seed=seed, ragged=True)
            ref(dict(b), task, True).mean().backward()
            for k, p in ref.named_parameters():
                if p.grad is not None and k in CE_GRAD_KEYS:
                    out[f"{task}_grad::{k}"] = sub(p.grad.float(), 97 if p.numel() > 4096 else 1)
        ref.zero_grad(set_to_none=True)
        return out
    merge_autocast_errors(ref, "tiny_ce", outputs, grads)


CE_GRAD_KEYS = {
    "bert.img_embeddings.dep_linear.weight", "bert.img_embeddings.dep_layer_norm.bias",
    "bert.img_embeddings.loc_linear.weight", "bert.img_embeddings.nav_type_embedding.weight",
    "bert.local_encoder.bev_fts_embeddings.0.weight", "bert.embeddings.word_embeddings.weight",
    "sap_fuse_linear.net.3.weight",
}


def gen_rxr():
    """BASELINE.json configs[3]: RxR pre-training -- the xlm-roberta vocabulary (250 002 tokens, 514 positions,
    configs/rxr_model.json) and 160-token instructions; one layer of each kind keeps the fixture generation short."""
    print("RxR vocabulary [tiny_rxr]")
    cfg = BevBertConfig.rxr(num_l_layers=1, num_x_layers=1, num_pano_layers=1, pretrain_tasks=("mlm", "sap"))
    ref = build_ref_pretrain(cfg)
    gen_keys(ref, "tiny_rxr")
    B, seed = 2, 61
    arrs = {"seed": np.int64(seed), "B": np.int64(B), "txt_len": np.int64(160)}
    for task in ("mlm", "sap"):
        b = synthetic.make_batch(cfg, task, B, seed=seed, txt_len=160, ragged=True)
        with torch.no_grad():
            arrs[f"{task}_loss"] = npy(ref(dict(b), task, True))
            outs = ref(dict(b), task, False)
        if task == "mlm":
            arrs["mlm_scores_sub"] = sub(outs, 4099)
            arrs["mlm_scores_rowmax"] = npy(outs.max(1).values)
            arrs["mlm_scores_argmax"] = npy(outs.argmax(1))
        else:
            arrs.update(sap_global=npy(outs[0]), sap_local=npy(outs[1]), sap_fused=npy(outs[2]))
    save("tasks_tiny_rxr", **arrs)

    def outputs():
        out = {}
        with torch.no_grad():
            b = synthetic.make_batch(cfg, "mlm", B, seed=seed, txt_len=160, ragged=True)
            out["rxr mlm_loss"] = ref(dict(b), "mlm", True)
            sc = ref(dict(b), "mlm", False)
            out["rxr mlm_scores"] = torch.from_numpy(sub(sc, 4099).copy())
            out["rxr mlm_rowmax"] = sc.max(1).values
            b = synthetic.make_batch(cfg, "sap", B, seed=seed, txt_len=160, ragged=True)
            out["rxr sap_loss"] = ref(dict(b), "sap", True)
            o = ref(dict(b), "sap", False)
            out.update({"rxr sap_global": o[0], "rxr sap_local": o[1], "rxr sap_fused": o[2]})
        return out
    merge_autocast_errors(ref, "tiny_rxr", outputs)


def gen_graph():
    """Fine-tune bookkeeping: the reference's GraphMap / FloydGraph (map_nav_src/models/graph_utils.py) driven through
    agent.rollout's per-step updates (map_nav_src/r2r/agent.py:447-449,471-494,556-559) on synthetic observation
    streams; the agent methods themselves (_nav_gmap_variable, _map_cand_to_bev) are restated in oracle/graph_ref.py."""
    import json
    print("fine-tune graph bookkeeping (GraphMap / FloydGraph)")
    sys.path.insert(0, os.path.join(REF, "map_nav_src"))
    from models import graph_utils
    from models.bev_utils import transfrom3D
    from oracle import graph_ref
    B, T, H, seed = 3, 7, 16, 51
    obs_all, ended_all = synthetic.make_nav_episodes(B, T, seed)
    g = torch.Generator().manual_seed(seed)
    gmaps = [graph_utils.GraphMap(ob["viewpoint"]) for ob in obs_all[0]]
    for i, ob in enumerate(obs_all[0]):
        gmaps[i].update_graph(ob)
    steps = []
    for t in range(T):
        obs, ended = obs_all[t], ended_all[t]
        if t > 0:                                             # agent.py:556-559: new observations extend the graphs
            for i, ob in enumerate(obs):
                if not ended_all[t - 1][i]:
                    gmaps[i].update_graph(ob)
        for i, gmap in enumerate(gmaps):
            if not ended[i]:
                gmap.node_step_ids[obs[i]["viewpoint"]] = t + 1
        C = max(len(ob["candidate"]) for ob in obs)
        avg = torch.randn(B, H, generator=g)
        pano = torch.randn(B, C + 2, H, generator=g)
        for i, gmap in enumerate(gmaps):                      # agent.py:485-494
            if not ended[i]:
                vp = obs[i]["viewpoint"]
                gmap.update_node_embed(vp, avg[i].clone(), rewrite=True)
                gmap.update_node_pc(vp, torch.tensor([[float(len(gmap.node_pc))]]) if vp not in gmap.node_pc
                                    else gmap.node_pc[vp][0], torch.zeros(1, dtype=torch.bool), torch.zeros(1, 1))
                for j, cc in enumerate(obs[i]["candidate"]):
                    if not gmap.graph.visited(cc["viewpointId"]):
                        gmap.update_node_embed(cc["viewpointId"], pano[i, j].clone())
        nv = graph_ref.nav_gmap_variable(obs, gmaps)
        rec = {"avg": avg.tolist(), "pano": pano.tolist(), "gmap_vpids": nv["gmap_vpids"],
               "gmap_step_ids": nv["gmap_step_ids"], "gmap_visited_masks": nv["gmap_visited_masks"],
               "no_vp_left": nv["no_vp_left"],
               "gmap_pos_fts": [p.tolist() for p in nv["gmap_pos_fts"]],
               "gmap_pair_dists": [p.tolist() for p in nv["gmap_pair_dists"]],
               "gmap_img_embeds": [e.tolist() for e in nv["gmap_img_embeds"]],
               "gather_order1": [], "cand_cells": [], "start_pos_fts": [], "paths": []}
        for i, (ob, gmap) in enumerate(zip(obs, gmaps)):
            pc, _, _ = gmap.gather_node_pc(ob["viewpoint"], 1)            # the stored "pc" is the node's visit index
            names = list(gmap.node_pc.keys())
            rec["gather_order1"].append([names[int(v)] for v in pc.reshape(-1).tolist()])
            cells = graph_ref.map_cand_to_bev(ob, 21, 0.5, transfrom3D)
            rec["cand_cells"].append((cells[:, 1] * 21 + cells[:, 0]).tolist())
            rec["start_pos_fts"].append(gmap.get_pos_fts(ob["viewpoint"], [gmap.start_vp], ob["heading"],
                                                         ob["elevation"]).tolist())
            rec["paths"].append({vp: gmap.graph.path(ob["viewpoint"], vp) for vp in gmap.node_positions})
        steps.append(rec)
    save("graph_nav", seed=np.int64(seed), B=np.int64(B), T=np.int64(T), H=np.int64(H),
         steps_json=np.array(json.dumps(steps)))


build_ref_pretrain_cached = {}


def main():
    assert os.path.isdir(REF), "the reference is only mounted in the build container"
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _install_shims()
    if "--ce" in sys.argv:
        gen_ce()
        return
    if "--graph" in sys.argv:
        gen_graph()
        return
    if "--rxr" in sys.argv:
        gen_rxr()
        return
    if "--configs" in sys.argv:
        gen_configs()
        return
    if "--curve" in sys.argv:
        gen_curve()
        return
    if "--fullsize" in sys.argv:
        gen_fullsize()
        return
    if "--modules" in sys.argv or "--autocast" in sys.argv:
        tiny = BevBertConfig.tiny()
        ref = build_ref_pretrain(tiny)
        if "--modules" in sys.argv:
            gen_modules(ref, tiny)
        if "--autocast" in sys.argv:
            gen_autocast(ref, tiny)
            arrs = {}
            gen_autocast_errors(ref, tiny, "tiny_b3_ragged", 3, 7, True, arrs)
            gen_autocast_errors(ref, tiny, "tiny_b2_fixed", 2, 8, False, arrs)
            rvr = BevBertConfig.tiny(image_feat_size=768, obj_feat_size=768, obj_prob_size=50,
                                     pretrain_tasks=("mlm", "mrc", "sap", "og"))
            gen_autocast_errors(build_ref_pretrain(rvr), rvr, "tiny_rvr", 4, 31, True, arrs, ("mlm", "mrc", "sap", "og"))
            objlin = BevBertConfig.tiny(image_feat_size=512, obj_feat_size=640, obj_prob_size=50, num_l_layers=1,
                                        num_x_layers=1, pretrain_tasks=("mrc", "og"))
            gen_autocast_errors(build_ref_pretrain(objlin), objlin, "tiny_objlin", 4, 32, True, arrs, ("mrc", "og"))
            full = BevBertConfig()
            gen_autocast_errors(build_ref_pretrain(full), full, "r2r_b2", 2, 1000, False, arrs)
            save("ref_autocast_errors", **arrs)
        return

    tiny = BevBertConfig.tiny()
    ref = build_ref_pretrain(tiny)
    build_ref_pretrain_cached["m"] = ref
    gen_keys(ref, "tiny")
    gen_splat(ref, tiny)
    gen_tasks(ref, tiny, "tiny_b3_ragged", B=3, seed=7, ragged=True, with_grads=True)
    gen_tasks(ref, tiny, "tiny_b2_fixed", B=2, seed=8, ragged=False, with_grads=False)
    gen_modules(ref, tiny)
    gen_autocast(ref, tiny)
    gen_configs()
    gen_nav(tiny)
    gen_adamw()
    gen_graph()
    gen_rxr()
    gen_curve()

    rvr = BevBertConfig.tiny(image_feat_size=768, obj_feat_size=768, obj_prob_size=50,
                             pretrain_tasks=("mlm", "mrc", "sap", "og"))
    gen_objects(rvr, "tiny_rvr", ("mlm", "mrc", "sap", "og"), seed=31)
    objlin = BevBertConfig.tiny(image_feat_size=512, obj_feat_size=640, obj_prob_size=50, num_l_layers=1,
                                num_x_layers=1, pretrain_tasks=("mrc", "og"))
    gen_objects(objlin, "tiny_objlin", ("mrc", "og"), seed=32)

    full = BevBertConfig()
    ref = build_ref_pretrain(full)
    gen_keys(ref, "r2r")
    gen_tasks(ref, full, "r2r_b2", B=2, seed=1000, ragged=False, with_grads=False)


if __name__ == "__main__":
    main()
