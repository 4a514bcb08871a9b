"""Pin the CPU oracle (oracle/bevbert_ref.py) against golden vectors captured from the
reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import bevbert_ref as R
from tests.helpers import load_golden, max_abs, rule_state_dict, sub
from vln_bevbert_amd import synthetic
from vln_bevbert_amd.config import BevBertConfig

FP32_TOL = 2e-4   # reference-vs-restatement, fp32 CPU, different op order (observed ~1e-5)


def test_pose_matrix_and_bevpos():
    g = load_golden("pose_matrix")
    assert np.array_equal(synthetic.pose_matrix(g["xyzhe"]), g["T"])
    g = load_golden("bevpos_polar")
    for d in (11, 14, 21):
        assert max_abs(R.bevpos_polar(d).numpy(), g[f"d{d}"]) < 1e-6


def test_splat_edge_cases():
    g = load_golden("splat_edge")
    pts, nod = torch.from_numpy(g["pts"])[None], torch.from_numpy(g["no_depth"])
    feat = torch.from_numpy(g["feat"])
    sem = torch.nn.functional.one_hot(torch.from_numpy(g["sem_ids"]), 40).double()
    bev, bsem, bsm = R.project_bev(pts, nod, feat, sem)
    assert np.array_equal(bev.numpy(), g["bev"])           # same sums in the same order -> bit exact
    assert np.array_equal(bsem.numpy().astype(np.uint8), g["bev_sems"])
    assert np.array_equal(bsm.numpy(), g["bev_sem_masks"])
    idx = R.cell_index(pts[0], nod[0])
    # spot checks of the documented branches (half-to-even, borders, thresholds)
    assert idx[0] == -1                                     # no depth
    assert idx[1] == 21 * 10 + 10 and idx[2] == 21 * 10 + 10
    assert idx[3] == 21 * 12 + 12 and idx[4] == 21 * 8 + 8
    assert idx[7] == 21 * 10 + 20 and idx[8] == -1
    assert idx[9] == 21 * 10 + 0 and idx[10] == -1          # round(-0.5) = -0 is inside
    assert idx[11] >= 0 and idx[12] == -1 and idx[13] == -1


def test_lift_splat_b2():
    cfg = BevBertConfig.tiny()
    g = load_golden("splat_b2")
    b = synthetic.make_batch(cfg, "sap", 2, seed=int(g["seed"]), ragged=True)
    pc, nod = R.lift_points(b["depths"], b["T_c2w"], b["T_w2c"], b["S_w2c"])
    assert np.array_equal(nod.numpy(), g["no_depth"])
    assert max_abs(pc.numpy(), g["pc_ego"]) < 1e-5
    # cell assignment must agree with the reference's own points on every point
    ref_idx = R.cell_index(torch.from_numpy(g["pc_ego"]), nod)
    my_idx = R.cell_index(pc, nod)
    assert (ref_idx == my_idx).all(), f"{int((ref_idx != my_idx).sum())} points changed cell"
    out = R.lift_splat(cfg, b)
    assert max_abs(sub(out["bev_fts"], 7), g["bev_fts_sub"]) < 1e-5
    assert max_abs(out["bev_fts"].abs().sum(-1).numpy(), g["bev_fts_cell_l1"]) < 1e-2
    assert np.array_equal(out["bev_pos_fts"].numpy(), g["bev_pos_fts"])
    assert np.array_equal(out["bev_masks"].numpy(), g["bev_masks"])
    assert np.array_equal(out["bev_sems"].numpy().astype(np.uint8), g["bev_sems"])
    assert np.array_equal(out["bev_sem_masks"].numpy(), g["bev_sem_masks"])
    # compact class-id form gives the same semantics
    b2 = synthetic.make_batch(cfg, "sap", 2, seed=int(g["seed"]), ragged=True, sems_as="ids")
    out2 = R.lift_splat(cfg, b2)
    assert np.array_equal(out2["bev_sems"].numpy(), out["bev_sems"].numpy())


def _check_tasks(cfg, tag, keys_file, check_grads):
    g = load_golden(f"tasks_{tag}")
    sd = rule_state_dict(keys_file)
    B, seed, ragged = int(g["B"]), int(g["seed"]), bool(g["ragged"])
    with torch.no_grad():
        b = synthetic.make_batch(cfg, "mlm", B, seed=seed, ragged=ragged)
        assert max_abs(R.pretrain_forward(sd, cfg, b, "mlm").numpy(), g["mlm_loss"]) < FP32_TOL
        sc = R.pretrain_forward(sd, cfg, b, "mlm", compute_loss=False)
        assert max_abs(sub(sc, 13), g["mlm_scores_sub"]) < FP32_TOL
        b = synthetic.make_batch(cfg, "sap", B, seed=seed, ragged=ragged)
        assert max_abs(R.pretrain_forward(sd, cfg, b, "sap").numpy(), g["sap_loss"]) < FP32_TOL
        gl, ll, fl, _, _ = R.pretrain_forward(sd, cfg, b, "sap", compute_loss=False)
        assert max_abs(gl.numpy(), g["sap_global"]) < FP32_TOL
        assert max_abs(ll.numpy(), g["sap_local"]) < FP32_TOL
        assert max_abs(fl.numpy(), g["sap_fused"]) < FP32_TOL
        bb = dict(b)
        bb.update(R.lift_splat(cfg, bb))
        gm, bev = R.cmt_forward(sd, cfg, bb)
        assert max_abs(gm.numpy(), g["gmap_embeds"]) < FP32_TOL
        assert max_abs(sub(bev, 11), g["bev_embeds_sub"]) < FP32_TOL
        b = synthetic.make_batch(cfg, "masksem", B, seed=seed, ragged=ragged)
        assert max_abs(R.pretrain_forward(sd, cfg, b, "masksem").numpy(), g["masksem_loss"]) < FP32_TOL
        lg, lb = R.pretrain_forward(sd, cfg, b, "masksem", compute_loss=False)
        assert max_abs(lg.numpy(), g["masksem_logits"]) < FP32_TOL
        assert np.array_equal(lb.numpy().astype(np.uint8), g["masksem_labels"])
        b = synthetic.make_batch(cfg, "sem", B, seed=seed, ragged=ragged)
        for tok in ("sattn", "embed", "cattn"):
            cfg.sem_pred_token = tok
            lg, _ = R.pretrain_forward(sd, cfg, b, "sem", compute_loss=False)
            assert max_abs(sub(lg, 3), g[f"sem_{tok}_logits_sub"]) < FP32_TOL
        cfg.sem_pred_token = "cattn"
    if check_grads:
        for task in ("mlm", "sap", "masksem"):
            leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
            leaf["mlm_head.predictions.decoder.weight"] = leaf["bert.embeddings.word_embeddings.weight"]
            b = synthetic.make_batch(cfg, task, B, seed=seed, ragged=ragged)
            R.pretrain_forward(leaf, cfg, b, task).mean().backward()
            sq, n = 0.0, 0
            seen = set()
            for k, v in leaf.items():
                if v.grad is None or id(v) in seen:
                    continue
                seen.add(id(v))
                n += 1
                sq += float((v.grad.double() ** 2).sum())
                gk = f"{task}_grad::{k}"
                if gk in g.files:
                    ref = g[gk]
                    step = 97 if v.numel() > 4096 else 1
                    scale = max(1e-6, float(np.abs(ref).max()))
                    assert max_abs(sub(v.grad, step), ref) < 2e-3 * scale + 1e-7, gk
            assert n == int(g[f"{task}_n_params_with_grad"]), (task, n)
            assert abs(sq - float(g[f"{task}_grad_sqnorm"])) < 1e-3 * float(g[f"{task}_grad_sqnorm"])


def test_tasks_tiny_ragged_with_grads():
    _check_tasks(BevBertConfig.tiny(), "tiny_b3_ragged", "pretrain_state_dict_keys_tiny.txt", True)


def test_tasks_tiny_fixed():
    _check_tasks(BevBertConfig.tiny(), "tiny_b2_fixed", "pretrain_state_dict_keys_tiny.txt", False)


def test_tasks_full_r2r_config():
    _check_tasks(BevBertConfig(), "r2r_b2", "pretrain_state_dict_keys_r2r.txt", False)


@pytest.mark.parametrize("tag,kw,tasks", [
    ("tiny_rvr", dict(image_feat_size=768, obj_feat_size=768, obj_prob_size=50,
                      pretrain_tasks=("mlm", "mrc", "sap", "og")), ("mlm", "mrc", "sap", "og")),
    ("tiny_objlin", dict(image_feat_size=512, obj_feat_size=640, obj_prob_size=50, num_l_layers=1, num_x_layers=1,
                         pretrain_tasks=("mrc", "og")), ("mrc", "og")),
    ("tiny_ce", dict(bev_dim=11, bev_res=1.0, depth_feat_size=128, loc_feat_size=4, nav_type_vocab=2, sem_classes=0,
                     pretrain_tasks=("mlm", "sap")), ("mlm", "sap")),     # continuous-environment fork (bevbert_ce)
])
def test_object_token_tasks(tag, kw, tasks):
    """REVERIE-style object tokens: panorama + BEV object branch, MRC and OG heads (pretrain_cmt.py:272-297,367-389);
    and the continuous-environment fork's model (depth-feature branch, 11x11 BEV at 1 m, no semantics), whose golden
    vectors come from bevbert_ce/pretrain/pretrain_src."""
    cfg = BevBertConfig.tiny(**kw)
    g = load_golden(f"tasks_{tag}")
    sd = rule_state_dict(f"pretrain_state_dict_keys_{tag}.txt")
    B, seed = int(g["B"]), int(g["seed"])
    for task in tasks:
        b = synthetic.make_batch(cfg, task, B, seed=seed, ragged=True)
        leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        if "mlm_head.predictions.decoder.weight" in leaf:
            leaf["mlm_head.predictions.decoder.weight"] = leaf["bert.embeddings.word_embeddings.weight"]
        loss = R.pretrain_forward(leaf, cfg, b, task)
        assert max_abs(loss.detach().numpy(), g[f"{task}_loss"]) < FP32_TOL, task
        with torch.no_grad():
            outs = R.pretrain_forward(sd, cfg, b, task, compute_loss=False)
        if task == "mrc":
            assert outs[0].shape[0] == int(g["mrc_n"]) and max_abs(sub(outs[0], 7), g["mrc_pred_sub"]) < FP32_TOL
        elif task == "og":
            assert max_abs(outs.numpy(), g["og_logits"]) < FP32_TOL
        elif task == "sap":
            assert max_abs(outs[2].numpy(), g["sap_fused"]) < FP32_TOL
        loss.mean().backward()
        seen, sq, n = set(), 0.0, 0
        for k, v in leaf.items():
            if v.grad is None or id(v) in seen:
                continue
            seen.add(id(v))
            n += 1
            sq += float((v.grad.double() ** 2).sum())
            gk = f"{task}_grad::{k}"
            if gk in g.files:
                ref = g[gk]
                scale = max(1e-6, float(np.abs(ref).max()))
                assert max_abs(sub(v.grad, 97 if v.numel() > 4096 else 1), ref) < 2e-3 * scale + 1e-7, gk
        assert n == int(g[f"{task}_n_params_with_grad"]), (task, n)
        assert abs(sq - float(g[f"{task}_grad_sqnorm"])) < 1e-3 * float(g[f"{task}_grad_sqnorm"])


def test_nav_api():
    cfg = BevBertConfig.tiny()
    g = load_golden("nav_tiny")
    sd = rule_state_dict("nav_state_dict_keys.txt")
    B = 3
    pb = synthetic.make_batch(cfg, "sap", B, seed=int(g["seed"]), ragged=True)
    with torch.no_grad():
        txt_masks = torch.arange(pb["txt_ids"].shape[1])[None] < pb["txt_lens"][:, None]
        txt = R.nav_forward(sd, cfg, "language", {"txt_ids": pb["txt_ids"], "txt_masks": txt_masks})
        assert max_abs(sub(txt, 7), g["txt_embeds_sub"]) < FP32_TOL
        ends = np.cumsum(pb["traj_step_lens"]) - 1
        pano, pm = R.nav_forward(sd, cfg, "panorama", {
            "view_img_fts": pb["traj_view_img_fts"][ends], "loc_fts": pb["traj_loc_fts"][ends],
            "nav_types": pb["traj_nav_types"][ends], "view_lens": pb["traj_vp_view_lens"][ends]})
        assert np.array_equal(pm.numpy(), g["pano_masks"])
        assert max_abs(sub(pano, 5), g["pano_embeds_sub"]) < FP32_TOL
        G = int(pb["gmap_lens"].max())
        gen = torch.Generator().manual_seed(99)
        gimg = torch.randn(B, G, 768, generator=gen)
        gimg[:, 0] = 0
        lifted = R.lift_splat(cfg, pb)
        out = R.nav_forward(sd, cfg, "navigation", {
            "txt_embeds": txt, "txt_masks": txt_masks, "gmap_img_embeds": gimg,
            "gmap_step_ids": pb["gmap_step_ids"], "gmap_pos_fts": pb["gmap_pos_fts"],
            "gmap_masks": torch.arange(G)[None] < pb["gmap_lens"][:, None],
            "gmap_pair_dists": pb["gmap_pair_dists"], "gmap_visited_masks": pb["gmap_visited_masks"],
            "gmap_vpids": pb["gmap_vpids"], "bev_fts": lifted["bev_fts"], "bev_pos_fts": lifted["bev_pos_fts"],
            "bev_masks": lifted["bev_masks"], "bev_nav_masks": pb["bev_nav_masks"],
            "bev_cand_idxs": pb["bev_cand_idxs"],
            "bev_cand_vpids": [[None] + c[-1] for c in pb["traj_cand_vpids"]]})
        assert max_abs(out["gmap_embeds"].numpy(), g["nav_gmap_embeds"]) < FP32_TOL
        for k in ("global", "local", "fused"):
            assert max_abs(out[f"{k}_logits"].numpy(), g[f"nav_{k}"]) < FP32_TOL


def test_adamw_and_schedule():
    g = load_golden("adamw")
    for wd in (0.01, 0.0):
        p = torch.from_numpy(g["p0"]).clone()
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for k in range(4):
            R.adamw_step(p, torch.from_numpy(g["grads"][k]), m, v, k + 1, 5e-5 * (k + 1) / 4, wd)
            assert np.array_equal(p.numpy(), g[f"wd{wd}"][k])
    for s, lr in zip(g["lr_steps"], g["lrs"]):
        assert R.warmup_linear_lr(int(s), 5e-5, 10000, 100000) == pytest.approx(float(lr), rel=1e-12)
    assert R.no_decay_key("bert.embeddings.LayerNorm.weight") and R.no_decay_key("a.dense.bias")
    assert not R.no_decay_key("bert.img_embeddings.img_layer_norm.weight")   # decayed: substring rule


def test_rxr_vocabulary_tasks():
    """BASELINE.json configs[3]: xlm-roberta vocabulary (250 002 tokens, 514 positions), 160-token instructions."""
    cfg = BevBertConfig.rxr(num_l_layers=1, num_x_layers=1, num_pano_layers=1, pretrain_tasks=("mlm", "sap"))
    g = load_golden("tasks_tiny_rxr")
    sd = rule_state_dict("pretrain_state_dict_keys_tiny_rxr.txt")
    assert sd["bert.embeddings.word_embeddings.weight"].shape == (250002, 768)
    assert sd["bert.embeddings.position_embeddings.weight"].shape == (514, 768)
    B, seed, L = int(g["B"]), int(g["seed"]), int(g["txt_len"])
    with torch.no_grad():
        b = synthetic.make_batch(cfg, "mlm", B, seed=seed, txt_len=L, ragged=True)
        assert 80 < b["txt_ids"].shape[1] <= L and int(b["txt_ids"].max()) > 30522      # ragged: padded to the batch max
        assert max_abs(R.pretrain_forward(sd, cfg, b, "mlm").numpy(), g["mlm_loss"]) < FP32_TOL
        scores = R.pretrain_forward(sd, cfg, b, "mlm", compute_loss=False)
        assert scores.shape[1] == 250002 and max_abs(sub(scores, 4099), g["mlm_scores_sub"]) < FP32_TOL
        assert np.array_equal(scores.argmax(1).numpy(), g["mlm_scores_argmax"])
        b = synthetic.make_batch(cfg, "sap", B, seed=seed, txt_len=L, ragged=True)
        assert max_abs(R.pretrain_forward(sd, cfg, b, "sap").numpy(), g["sap_loss"]) < FP32_TOL
        outs = R.pretrain_forward(sd, cfg, b, "sap", compute_loss=False)
        assert max_abs(outs[2].numpy(), g["sap_fused"]) < FP32_TOL


def _module_inputs():
    """The inputs of tests/golden/modules_tiny.npz, re-derived from its generator seed (make_golden.gen_modules)."""
    g = torch.Generator().manual_seed(4242)
    d = {"x": torch.randn(2, 11, 768, generator=g)}
    d["m"] = torch.arange(11)[None] < torch.tensor([11, 6])[:, None]
    d["lang"] = torch.randn(2, 9, 768, generator=g)
    d["visn"] = torch.randn(2, 7, 768, generator=g)
    d["spr"] = torch.randn(2, 7, 7, generator=g)
    d["lm"] = torch.arange(9)[None] < torch.tensor([9, 4])[:, None]
    d["vm"] = torch.arange(7)[None] < torch.tensor([5, 7])[:, None]
    d["pano"] = torch.randn(3, 36, 768, generator=g)
    d["pm"] = torch.arange(36)[None] < torch.tensor([36, 20, 5])[:, None]
    d["vf"] = torch.randn(3, 36, 512, generator=g)
    d["lf"] = torch.randn(3, 36, 7, generator=g)
    d["nt"] = torch.randint(0, 3, (3, 36), generator=g)
    d["bf"] = torch.randn(2, 441, 768, generator=g)
    d["bp"] = torch.randn(2, 441, 10, generator=g)
    d["bn"] = torch.rand(2, 441, generator=g) < 0.1
    return d


def test_per_module_vectors():
    """One BertLayer, one GraphLXRTXLayer per forward (with / without graph_sprels), the panorama encoder alone, the
    two input-embedding sums: the oracle's building blocks against the reference's own modules."""
    from tests.helpers import rule_state_dict
    g = load_golden("modules_tiny")
    sd = rule_state_dict("pretrain_state_dict_keys_tiny.txt")
    cfg = BevBertConfig.tiny()
    d = _module_inputs()
    nh = cfg.num_attention_heads
    tol = 2e-5
    assert max_abs(R.bert_layer(sd, "bert.lang_encoder.layer.0", d["x"], R.neg_mask(d["m"]), nh), g["bert_layer"]) < tol
    p = "bert.global_encoder.encoder.x_layers.0"
    el, ev = R.neg_mask(d["lm"]), R.neg_mask(d["vm"])
    assert max_abs(R.x_layer_visn(sd, p, nh, d["lang"], el, d["visn"], ev, d["spr"][:, None]), g["x_visn_sprels"]) < tol
    assert max_abs(R.x_layer_visn(sd, p, nh, d["lang"], el, d["visn"], ev), g["x_visn"]) < tol
    assert max_abs(R.x_layer_lang2visn(sd, p, nh, d["lang"], el, d["visn"], ev), g["x_lang2visn"]) < tol
    assert max_abs(R.x_layer_visn2visn(sd, p, nh, d["visn"], ev), g["x_visn2visn"]) < tol
    e = d["pano"]
    for i in range(cfg.num_pano_layers):
        e = R.pano_layer(sd, f"bert.img_embeddings.pano_encoder.layers.{i}", e, ~d["pm"], nh)
    e = R.layer_norm(sd, "bert.img_embeddings.pano_encoder.norm", e)
    assert max_abs(e[d["pm"]], g["pano_encoder"][g["pano_valid"]]) < tol          # padded query rows carry garbage
    ip = "bert.img_embeddings"
    emb = R.layer_norm(sd, ip + ".img_layer_norm", R.linear(sd, ip + ".img_linear", d["vf"])) \
        + R.layer_norm(sd, ip + ".loc_layer_norm", R.linear(sd, ip + ".loc_linear", d["lf"])) \
        + sd[ip + ".nav_type_embedding.weight"][d["nt"]] + sd["bert.embeddings.token_type_embeddings.weight"][1]
    assert max_abs(R.layer_norm(sd, ip + ".layer_norm", emb), g["img_embed_sum_ln"]) < tol
    be = R.bev_input_embedding(sd, "bert.local_encoder", d["bf"], d["bp"], d["bn"])
    assert max_abs(be.reshape(-1)[::5], g["bev_input_embedding_sub"]) < tol


def test_oracle_matches_the_reference_at_full_size():
    """The oracle against the reference itself at BASELINE.json configs[1]'s real size (tests/golden/tasks_r2r_fullsize.npz,
    make_golden.py --fullsize): batch 64, 80 tokens, full R2R model -- per-sample SAP and MLM losses; batch 16 -- gradient
    norm and the sub-sampled named gradients.  bench.py's cpu_baseline times this oracle at these sizes; the GPU suite
    compares the HIP path with the same file (test_full_size_parity_vs_reference).  ~20 s of CPU."""
    import os
    g = load_golden("tasks_r2r_fullsize")
    cfg = BevBertConfig()
    keys = "pretrain_state_dict_keys_r2r.txt"
    sd = rule_state_dict(keys)
    old = torch.get_num_threads()
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    try:
        Bf, Bb, L = int(g["fwd_batch"]), int(g["bwd_batch"]), int(g["txt_len"])
        for task in ("sap", "mlm"):
            b = synthetic.make_batch(cfg, task, Bf, seed=int(g["fwd_seed"]), txt_len=L)
            with torch.no_grad():
                got = R.pretrain_forward(sd, cfg, b, task).numpy()
            want = g[f"{task}_loss"]
            assert got.shape == want.shape
            assert max_abs(got, want) < FP32_TOL * max(1.0, float(np.abs(want).max())), (task, max_abs(got, want))
        names = sorted({k.split("::", 1)[1] for k in g.files if "_grad::" in k and not k.startswith("ref_autocast")})
        for task in ("sap", "mlm"):
            b = synthetic.make_batch(cfg, task, Bb, seed=int(g["bwd_seed"]), txt_len=L)
            osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
            osd["mlm_head.predictions.decoder.weight"] = osd["bert.embeddings.word_embeddings.weight"]
            loss = R.pretrain_forward(osd, cfg, b, task).mean()
            allp = list({id(v): v for v in osd.values()}.values())
            grads = torch.autograd.grad(loss, allp, allow_unused=True)
            sq = float(sum((x.double() ** 2).sum() for x in grads if x is not None))
            ref_sq = float(g[f"{task}_grad_sqnorm"])
            assert abs(sq - ref_sq) < 1e-3 * ref_sq, (task, sq, ref_sq)
            by_id = {id(p_): x for p_, x in zip(allp, grads)}
            for n in names:
                gk = f"{task}_grad::{n}"
                x = by_id[id(osd[n])]
                if gk not in g.files:
                    assert x is None, gk                      # the reference left .grad None for it as well
                    continue
                got = sub(x, 97 if x.numel() > 4096 else 1)
                scale = max(1e-6, float(np.abs(g[gk]).max()))
                assert max_abs(got, g[gk]) < 1e-3 * scale + 1e-7, (gk, max_abs(got, g[gk]), scale)
            del grads, loss, osd
    finally:
        torch.set_num_threads(old)
