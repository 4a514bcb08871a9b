"""End-to-end parity of the HIP path on the MI355X: the product model (vln_bevbert_amd) against the golden vectors
captured from the reference (tests/golden/*.npz) and against the CPU oracle, forward and backward, fp32 and bf16."""
import numpy as np
import pytest
import torch

from oracle import bevbert_ref as R
from tests.helpers import load_golden, max_abs, rule_state_dict, sub
from vln_bevbert_amd import synthetic
from vln_bevbert_amd.config import BevBertConfig

pytestmark = pytest.mark.gpu
DEV = "cuda"
FP32_TOL = 1e-3          # north_star: "within 1e-3 fp32"


def bf16_close(got, want, what):
    """north_star "1e-2 bf16", read as SURVEY section 7 fixes it: mean-abs error relative to the output's abs-max
    <= 1e-2 (the reference's own autocast-bf16 forward sits at ~1e-3 by this measure), and max-abs <= 6e-2 * absmax."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    fin = np.isfinite(want)
    assert (np.isfinite(got) == fin).all(), what
    scale = max(1e-6, float(np.abs(want[fin]).max()))
    err = np.abs(got[fin] - want[fin])
    assert err.mean() / scale < 1e-2 and err.max() / scale < 6e-2, (what, err.mean() / scale, err.max() / scale)


def bf16_grad_close(got, ref, what):
    """bf16 gradients of single tensors: relative L2 error of the sampled entries.  Small gradients that are sums of
    cancelling bf16-rounded paths (word embeddings behind the whole text encoder, 3-row type tables) sit at 0.1-0.25;
    analytically-zero gradients (softmax shift invariance: sprel bias, the 1-wide head's bias) are checked absolutely.
    The fp32 mode pins the same tensors to 2e-3, and the global gradient norm is checked separately."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if float(np.abs(ref).max()) < 1e-6:
        assert float(np.abs(got).max()) < 5e-2, (what, got)
        return
    l2 = float(np.linalg.norm(got - ref) / max(1e-12, np.linalg.norm(ref)))
    assert l2 < 0.3, (what, l2)


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vln_bevbert_amd import lib
    lib.load()
    return True


def build(cfg, keys_file, dtype, nav=False):
    from vln_bevbert_amd.nav_model import GlocalTextPathNavCMT
    from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    m = (GlocalTextPathNavCMT if nav else GlocalTextPathCMTPreTraining)(cfg)
    m.load_state_dict(rule_state_dict(keys_file))
    if not nav:
        m.tie_weights()
    arena = m.finalize(DEV, dtype)
    return m.eval(), arena


def _check_tasks(env, cfg, tag, keys_file, dtype, check_grads=False):
    g = load_golden(f"tasks_{tag}")
    B, seed, ragged = int(g["B"]), int(g["seed"]), bool(g["ragged"])
    model, arena = build(cfg, keys_file, dtype)
    fp32 = dtype == torch.float32

    def cmp(got, key, step=None):
        got = got.detach().float().cpu()
        ref = g[key]
        got = sub(got, step) if step else got.numpy()
        if fp32:
            assert max_abs(got, ref) < FP32_TOL, (key, max_abs(got, ref))
        else:
            bf16_close(got, ref, key)

    mk = lambda task: synthetic.batch_to(synthetic.make_batch(cfg, task, B, seed=seed, ragged=ragged), DEV)
    with torch.no_grad():
        cmp(model(mk("mlm"), "mlm"), "mlm_loss")
        cmp(model(mk("mlm"), "mlm", compute_loss=False), "mlm_scores_sub", 13)
        cmp(model(mk("sap"), "sap"), "sap_loss")
        gl, ll, fl, _, _ = model(mk("sap"), "sap", compute_loss=False)
        cmp(gl, "sap_global"); cmp(ll, "sap_local"); cmp(fl, "sap_fused")
        b = mk("sap")
        model.lift_splat(b)
        gm, bev, _, _ = model.bert(*model._cmt_args(b))
        cmp(gm, "gmap_embeds"); cmp(bev, "bev_embeds_sub", 11)
        cmp(model(mk("masksem"), "masksem"), "masksem_loss")
        lg, lb = model(mk("masksem"), "masksem", compute_loss=False)
        cmp(lg, "masksem_logits")
        assert np.array_equal(lb.cpu().numpy().astype(np.uint8), g["masksem_labels"])
        for tok in ("sattn", "embed", "cattn"):
            model.sem_pred_token = tok
            lg, _ = model(mk("sem"), "sem", compute_loss=False)
            cmp(lg, f"sem_{tok}_logits_sub", 3)
        model.sem_pred_token = "cattn"
    if check_grads:
        for task in ("mlm", "sap", "masksem"):
            arena.zero_grad()
            model(mk(task), task).mean().backward()
            arena.sync()
            sq = float((arena.grads.double() ** 2).sum())
            ref_sq = float(g[f"{task}_grad_sqnorm"])
            assert abs(sq - ref_sq) < (2e-3 if fp32 else 5e-2) * ref_sq, (task, sq, ref_sq)
            for name, p in model.named_parameters():
                gk = f"{task}_grad::{name}"
                if gk in g.files:
                    ref = g[gk]
                    got = sub(p.main_grad.cpu(), 97 if p.numel() > 4096 else 1)
                    scale = max(1e-6, float(np.abs(ref).max()))
                    if fp32:
                        assert max_abs(got, ref) < 2e-3 * scale + 1e-7, (gk, max_abs(got, ref), scale)
                    else:
                        bf16_grad_close(got, ref, gk)
            # parameters the task does not use keep an exactly-zero gradient (find_unused_parameters semantics)
            used = {k[len(task) + 7:] for k in g.files if k.startswith(f"{task}_grad::")}
            assert len(used) > 0


def test_tiny_ragged_fp32_forward_and_grads(env):
    _check_tasks(env, BevBertConfig.tiny(), "tiny_b3_ragged", "pretrain_state_dict_keys_tiny.txt", torch.float32, True)


def test_tiny_fixed_fp32(env):
    _check_tasks(env, BevBertConfig.tiny(), "tiny_b2_fixed", "pretrain_state_dict_keys_tiny.txt", torch.float32)


def test_tiny_ragged_bf16_forward_and_grads(env):
    _check_tasks(env, BevBertConfig.tiny(), "tiny_b3_ragged", "pretrain_state_dict_keys_tiny.txt", torch.bfloat16, True)


def test_full_r2r_config_fp32(env):
    _check_tasks(env, BevBertConfig(), "r2r_b2", "pretrain_state_dict_keys_r2r.txt", torch.float32)


def test_full_r2r_config_bf16(env):
    _check_tasks(env, BevBertConfig(), "r2r_b2", "pretrain_state_dict_keys_r2r.txt", torch.bfloat16)


OBJ_CASES = [
    ("tiny_rvr", dict(image_feat_size=768, obj_feat_size=768, obj_prob_size=50,
                      pretrain_tasks=("mlm", "mrc", "sap", "og")), ("mlm", "mrc", "sap", "og")),
    ("tiny_objlin", dict(image_feat_size=512, obj_feat_size=640, obj_prob_size=50, num_l_layers=1, num_x_layers=1,
                         pretrain_tasks=("mrc", "og")), ("mrc", "og")),
    ("tiny_ce", dict(bev_dim=11, bev_res=1.0, depth_feat_size=128, loc_feat_size=4, nav_type_vocab=2, sem_classes=0,
                     pretrain_tasks=("mlm", "sap")), ("mlm", "sap")),     # continuous-environment fork (bevbert_ce)
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag,kw,tasks", OBJ_CASES)
def test_object_token_tasks(env, tag, kw, tasks, dtype):
    """REVERIE-style object tokens on the HIP path vs the reference's golden vectors (forward + gradients)."""
    cfg = BevBertConfig.tiny(**kw)
    g = load_golden(f"tasks_{tag}")
    model, arena = build(cfg, f"pretrain_state_dict_keys_{tag}.txt", dtype)
    fp32 = dtype == torch.float32
    B, seed = int(g["B"]), int(g["seed"])

    def cmp(got, ref, what):
        got = got.detach().float().cpu().numpy()
        if fp32:
            assert max_abs(got, ref) < FP32_TOL, (what, max_abs(got, ref))
        else:
            bf16_close(got, ref, what)

    for task in tasks:
        b = synthetic.batch_to(synthetic.make_batch(cfg, task, B, seed=seed, ragged=True), DEV)
        arena.zero_grad()
        loss = model(b, task)
        cmp(loss, g[f"{task}_loss"], f"{task}_loss")
        loss.mean().backward()
        arena.sync()
        with torch.no_grad():
            outs = model(b, task, compute_loss=False)
        if task == "og":
            cmp(outs, g["og_logits"], "og_logits")
        elif task == "mrc":
            assert outs[0].shape[0] == int(g["mrc_n"])
            cmp(torch.from_numpy(sub(outs[0].float().cpu(), 7).copy()), g["mrc_pred_sub"], "mrc_pred")
        elif task == "sap":
            cmp(outs[2], g["sap_fused"], "sap_fused")
        sq = float((arena.grads.double() ** 2).sum())
        ref_sq = float(g[f"{task}_grad_sqnorm"])
        assert abs(sq - ref_sq) < (2e-3 if fp32 else 6e-2) * ref_sq, (task, sq, ref_sq)
        for name, p in model.named_parameters():
            gk = f"{task}_grad::{name}"
            if gk in g.files:
                ref = g[gk]
                got = sub(p.main_grad.cpu(), 97 if p.numel() > 4096 else 1)
                scale = max(1e-6, float(np.abs(ref).max()))
                if fp32:
                    assert max_abs(got, ref) < 2e-3 * scale + 1e-7, (gk, max_abs(got, ref), scale)
                else:
                    bf16_grad_close(got, ref, gk)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_nav_api(env, dtype):
    cfg = BevBertConfig.tiny()
    g = load_golden("nav_tiny")
    model, _ = build(cfg, "nav_state_dict_keys.txt", dtype, nav=True)
    fp32 = dtype == torch.float32
    B = 3
    pb = synthetic.make_batch(cfg, "sap", B, seed=int(g["seed"]), ragged=True)
    lifted = R.lift_splat(cfg, pb)
    d = synthetic.batch_to(pb, DEV)

    def cmp(got, ref, step=None):
        got = got.detach().float().cpu()
        got = sub(got, step) if step else got.numpy()
        if fp32:
            assert max_abs(got, ref) < FP32_TOL
        else:
            bf16_close(got, ref, "nav")

    with torch.no_grad():
        txt_masks = torch.arange(d["txt_ids"].shape[1], device=DEV)[None] < d["txt_lens"][:, None]
        txt = model("language", {"txt_ids": d["txt_ids"], "txt_masks": txt_masks})
        cmp(txt, g["txt_embeds_sub"], 7)
        ends = torch.from_numpy(np.cumsum(pb["traj_step_lens"]) - 1).to(DEV)
        pano, pm = model("panorama", {"view_img_fts": d["traj_view_img_fts"][ends], "obj_img_fts": None,
                                      "loc_fts": d["traj_loc_fts"][ends], "nav_types": d["traj_nav_types"][ends],
                                      "view_lens": d["traj_vp_view_lens"][ends], "obj_lens": None})
        assert np.array_equal(pm.cpu().numpy(), g["pano_masks"])
        cmp(pano, g["pano_embeds_sub"], 5)
        G = int(pb["gmap_lens"].max())
        gen = torch.Generator().manual_seed(99)
        gimg = torch.randn(B, G, 768, generator=gen)
        gimg[:, 0] = 0
        out = model("navigation", {
            "txt_embeds": txt, "txt_masks": txt_masks, "gmap_img_embeds": gimg.to(DEV),
            "gmap_step_ids": d["gmap_step_ids"], "gmap_pos_fts": d["gmap_pos_fts"],
            "gmap_masks": torch.arange(G, device=DEV)[None] < d["gmap_lens"][:, None],
            "gmap_pair_dists": d["gmap_pair_dists"], "gmap_visited_masks": d["gmap_visited_masks"],
            "gmap_vpids": pb["gmap_vpids"], "bev_fts": lifted["bev_fts"].to(DEV).to(txt.dtype),
            "bev_pos_fts": lifted["bev_pos_fts"].to(DEV), "bev_masks": lifted["bev_masks"].to(DEV),
            "bev_nav_masks": d["bev_nav_masks"], "bev_cand_idxs": d["bev_cand_idxs"],
            "bev_cand_vpids": [[None] + c[-1] for c in pb["traj_cand_vpids"]],
            "obj_embeds": None, "obj_masks": None})
        cmp(out["gmap_embeds"], g["nav_gmap_embeds"])
        for k in ("global", "local", "fused"):
            cmp(out[f"{k}_logits"], g[f"nav_{k}"])


@pytest.mark.parametrize("dim,res", [(11, 1.0), (14, 0.5)])
def test_bev_geometry_variants_against_oracle(env, dim, res):
    """The path is parametric in the BEV geometry: 11x11 @ 1 m is the continuous-environment fork
    (bevbert_ce/pretrain/pretrain_src/model/pretrain_cmt.py:16-17), 14x14 the north star's literal shape."""
    from vln_bevbert_amd import weights
    from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    cfg = BevBertConfig.tiny(num_l_layers=1, num_x_layers=1, vocab_size=400, bev_dim=dim, bev_res=res)
    model = GlocalTextPathCMTPreTraining(cfg)
    sd = weights.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict(sd)
    model.tie_weights()
    model.finalize(DEV, torch.float32)
    model.eval()
    for task in ("sap", "masksem", "mlm"):
        b = synthetic.make_batch(cfg, task, 3, seed=60 + dim, ragged=True)
        with torch.no_grad():
            got = model(synthetic.batch_to(b, DEV), task).cpu()
            want = R.pretrain_forward(sd, cfg, b, task)
        assert got.shape == want.shape and max_abs(got.numpy(), want.numpy()) < FP32_TOL, (task, dim)


def _oracle_train(cfg, sd0, tasks, batches, lr_fn, wd=0.01, max_norm=5.0):
    """CPU oracle of the reference's hot loop with dropout disabled (train_r2r.py:247-313)."""
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    sd["mlm_head.predictions.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    names = [k for k in sd if k != "mlm_head.predictions.decoder.weight"]
    state = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k]), [0]) for k in names}
    losses = []
    for step, (task, b) in enumerate(zip(tasks, batches), 1):
        loss = R.pretrain_forward(sd, cfg, b, task).mean()
        grads = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
        gd = {k: g for k, g in zip(names, grads)}
        for k in names:                               # zero_grad() keeps zeros for params that ever had a grad
            if gd[k] is None and state[k][2][0] > 0:
                gd[k] = torch.zeros_like(sd[k])
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in gd.values() if g is not None)).float()
        coef = min(1.0, max_norm / (float(total) + 1e-6))
        with torch.no_grad():
            for k in names:
                if gd[k] is None:
                    continue
                m, v, n = state[k]
                n[0] += 1
                R.adamw_step(sd[k], gd[k] * coef, m, v, n[0], lr_fn(step), 0.0 if R.no_decay_key(k) else wd)
        losses.append(float(loss.detach()))
    return losses


def test_training_curve_matches_oracle_fp32(env):
    """Loss curves overlap (north_star) -- run with dropout disabled on both sides, fp32, 12 steps."""
    from vln_bevbert_amd.train import PretrainTrainer, TaskSampler
    cfg = BevBertConfig.tiny(num_l_layers=1, num_x_layers=1, vocab_size=600)
    from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from vln_bevbert_amd import weights
    model = GlocalTextPathCMTPreTraining(cfg)
    sd0 = weights.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict(sd0)
    model.tie_weights()
    arena = model.finalize(DEV, torch.float32)
    model.train()
    model.set_dropout(0.0)
    n = 12
    sampler = TaskSampler("mlm.5.sap.5.masksem.1", seed=1)
    tasks = [sampler.next() for _ in range(n)]
    batches = [synthetic.make_batch(cfg, t, 2, seed=50 + i, ragged=True) for i, t in enumerate(tasks)]
    trainer = PretrainTrainer(model, arena, learning_rate=1e-4, warmup_steps=4, num_train_steps=40)
    got = [float(trainer.step(t, synthetic.batch_to(b, DEV))) for t, b in zip(tasks, batches)]
    from vln_bevbert_amd.train import warmup_linear_lr
    want = _oracle_train(cfg, sd0, tasks, batches, lambda s: warmup_linear_lr(s, 1e-4, 4, 40))
    err = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(got, want))
    assert err < 2e-2, (got, want)
    assert abs(got[0] - want[0]) < 1e-3 * max(1.0, abs(want[0]))


def test_training_step_bf16_full_size_runs_and_learns(env):
    """BASELINE configs[1] shapes at a reduced batch: bf16, dropout on; loss is finite and falls on a fixed batch."""
    from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from vln_bevbert_amd.train import PretrainTrainer
    cfg = BevBertConfig()
    torch.manual_seed(0)
    model = GlocalTextPathCMTPreTraining(cfg)
    arena = model.finalize(DEV, torch.bfloat16)
    model.train()
    model.set_dropout(0.1)
    trainer = PretrainTrainer(model, arena, learning_rate=1e-4, warmup_steps=2, num_train_steps=50)
    b = synthetic.batch_to(synthetic.make_batch(cfg, "sap", 8, seed=9, sems_as="ids"), DEV)
    losses = [float(trainer.step("sap", b)) for _ in range(8)]
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
    for t in ("mlm", "masksem"):
        bb = synthetic.batch_to(synthetic.make_batch(cfg, t, 8, seed=10, sems_as="ids"), DEV)
        assert np.isfinite(float(trainer.step(t, bb)))


def test_rccl_reducer_path_single_rank(env):
    """The data-parallel exchange (side stream, text-embedding hook, in-place all-reduce of arena slices) on real RCCL
    with a one-rank group: the collectives are identities, so the run must match one without them.  Every dropout
    mask comes from the library's counter-based stream (seed, step), so two runs differ only by the summation order
    of the fp32 atomics in the embedding / graph-bias gradients (measured: <= 1e-8 on 1-3 of 52 M parameters)."""
    import os
    import socket
    import torch.distributed as dist
    from vln_bevbert_amd import weights
    from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from vln_bevbert_amd.train import PretrainTrainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg = BevBertConfig.tiny(num_l_layers=1, num_x_layers=1, vocab_size=400)
        results = []
        # the first pass only settles the hipBLASLt plans (the first launch of a problem times its candidates and leaves
        # the product of whichever ran last): the two compared passes must use the final algorithms throughout
        for force in (None, False, True):
            model = GlocalTextPathCMTPreTraining(cfg)
            model.load_state_dict(weights.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
            model.tie_weights()
            arena = model.finalize(DEV, torch.bfloat16)
            model.train()
            model.set_dropout(0.1)
            tr = PretrainTrainer(model, arena, warmup_steps=2, num_train_steps=20, force_collectives=bool(force))
            assert tr.reducer.active == bool(force) and tr.overlap == bool(force)
            for i, task in enumerate(("sap", "mlm", "masksem", "sap")):
                tr.step(task, synthetic.batch_to(synthetic.make_batch(cfg, task, 2, seed=80 + i, ragged=True), DEV))
            torch.cuda.synchronize()
            if force is not None:
                results.append(arena.params.clone())
        assert float((results[0] - results[1]).abs().max()) < 1e-6
    finally:
        dist.destroy_process_group()


def test_batches_from_resident_grid_feature_store(env):
    """f1: a batch that names rows of a device-resident GridFeatureStore (fp16 features, uint8 class ids, read in place
    by the splat kernel) gives exactly the losses of the same batch shipped as tensors, and matches the oracle run on
    the fp16-rounded features."""
    from vln_bevbert_amd import weights
    from vln_bevbert_amd.feature_store import GridFeatureStore
    from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    cfg = BevBertConfig.tiny(num_l_layers=1, num_x_layers=1, vocab_size=400)
    model = GlocalTextPathCMTPreTraining(cfg)
    sd = weights.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict(sd)
    model.tie_weights()
    model.finalize(DEV, torch.float32)
    model.eval()
    # a "dataset" of 6 viewpoints; each batch draws 3 of them
    pool = synthetic.make_batch(cfg, "sap", 6, seed=21, ragged=True, sems_as="ids")
    keys = [f"scan_{i}" for i in range(6)]
    store = GridFeatureStore(keys, pool["rgbs"], pool["depths"], pool["sems"], DEV)
    assert store.nbytes() == 6 * (2352 * 768 * 2 + 2352 * 4 + 2352)
    pick = [4, 1, 5]
    for task in ("sap", "masksem", "mlm"):
        b = synthetic.make_batch(cfg, task, 3, seed=33, ragged=True, sems_as="ids")
        b["rgbs"] = pool["rgbs"][pick].half().float()            # what the store holds, widened
        b["depths"], b["sems"] = pool["depths"][pick], pool["sems"][pick]
        with torch.no_grad():
            model(synthetic.batch_to(b, DEV), task)      # settles the hipBLASLt plans of this task's shapes
            as_tensors = model(synthetic.batch_to(b, DEV), task).cpu()
            via_store = model(store.attach(synthetic.batch_to(b, DEV), [keys[i] for i in pick]), task).cpu()
            ob = dict(b)
            ob["sems"] = torch.from_numpy(np.eye(cfg.sem_classes)[b["sems"].reshape(3, -1).numpy()])
            want = R.pretrain_forward(sd, cfg, ob, task)
        assert torch.equal(via_store, as_tensors), task
        assert max_abs(via_store.numpy(), want.numpy()) < FP32_TOL, task


def test_finetune_bev_from_store_rows_of_visited_neighbours(env):
    """f3: the fine-tune BEV of a step = lift + splat over the current viewpoint AND its visited 1-hop neighbours
    (GraphMap.gather_node_pc + agent.splat, map_nav_src/models/graph_utils.py:129-144, r2r/agent.py:143-192).  The
    product reads the neighbours' features straight from the resident store (multi-row sample_rows); the oracle
    concatenates materialised point clouds like the reference.  Cell ids and fp32 BEV features must agree exactly."""
    from vln_bevbert_amd import ops
    from vln_bevbert_amd.feature_store import GridFeatureStore
    from vln_bevbert_amd.graph_map import GraphMapBatch
    cfg = BevBertConfig()
    B, T, n_nodes = 3, 6, 14
    obs_all, ended_all = synthetic.make_nav_episodes(B, T, seed=77, n_nodes=n_nodes)
    g = torch.Generator().manual_seed(77)
    keys = [f"scan{i}_e{i}_v{n}" for i in range(B) for n in range(n_nodes)]
    N = len(keys)
    rgbs = torch.randn(N, 12, 14, 14, 64, generator=g).half()              # 64 channels keep the store small
    depths = torch.rand(N, 12, 14, 14, generator=g) * 0.5
    depths[torch.rand(depths.shape, generator=g) < 0.05] = 0.0
    sems = torch.randint(0, 40, (N, 12, 14, 14), generator=g).to(torch.uint8)
    store = GridFeatureStore(keys, rgbs, depths, sems, DEV)
    gm = GraphMapBatch([ob["viewpoint"] for ob in obs_all[0]], 8, DEV)
    gm.update_graph(obs_all[0])
    pix = ops.pixel_scale(cfg.grid_hw, torch.device(DEV))
    K = cfg.bev_dim ** 2
    multi = 0
    for t in range(T):
        obs, ended = obs_all[t], ended_all[t]
        if t > 0:
            gm.update_graph(obs, ended_all[t - 1])
        gm.remember_views(obs, [f"{ob['scan']}_{ob['viewpoint']}" for ob in obs], store, ended)
        bi = gm.bev_inputs(obs, store, pc_order=1, bev_dim=cfg.bev_dim, bev_res=cfg.bev_res)
        multi += int(bi["grid_rows"].shape[1] > 1)
        cell, order, start = ops.bev_lift_bin(bi["depths"], bi["T_c2w"], bi["T_w2c"], bi["S_w2c"], pix, cfg.bev_dim,
                                              cfg.bev_res)
        bev, _, _ = ops.bev_splat_mean(store.rgbs, order, start, K, out_dtype=torch.float32, rows=bi["grid_rows"])
        # oracle: materialise the concatenated inputs the way the reference stores and gathers them
        rows = bi["grid_rows"].cpu().long()
        feat = store.rgbs.cpu().float()[rows].reshape(B, -1, 64)
        pc, nod = R_lift(bi, cfg)
        want_cell = R.cell_index(pc, nod, cfg.bev_dim, cfg.bev_res)
        assert torch.equal(cell.cpu().long(), want_cell), t
        want, _, _ = R.project_bev(pc, nod, feat, None, cfg.bev_dim, cfg.bev_res)
        assert torch.equal(bev.cpu(), want), t
        # candidate cells feed bev_nav_masks; the centre cell is the [stop] token
        assert bool(bi["bev_nav_masks"][:, (K - 1) // 2].all())
    assert multi >= 2          # the walks did revisit neighbourhoods: several steps splat more than one panorama


def R_lift(bi, cfg):
    return R.lift_points(bi["depths"].cpu(), bi["T_c2w"].cpu(), bi["T_w2c"].cpu(), bi["S_w2c"].cpu(), cfg.grid_hw)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_phase_a_gradients_are_final_when_the_text_hook_fires(env, dtype):
    """The overlapped all-reduce (train.GradReducer.phase_a) reduces the arena region [split, end) -- map encoders and
    heads -- as soon as d loss / d text-embeddings is complete.  That is only correct if every kernel that writes that
    region has been issued by then (a one-rank RCCL group cannot show a violation: its all-reduce is the identity).
    Snapshot the region at the moment the hook fires and compare with the region after the whole backward."""
    from vln_bevbert_amd import ops, weights
    from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from vln_bevbert_amd.train import PretrainTrainer
    cfg = BevBertConfig.tiny(num_l_layers=2, num_x_layers=2, vocab_size=400)
    model = GlocalTextPathCMTPreTraining(cfg)
    model.load_state_dict(weights.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
    model.tie_weights()
    arena = model.finalize(DEV, dtype)
    model.train()
    model.set_dropout(0.1)
    split = PretrainTrainer(model, arena, overlap=False).reducer.split
    assert 0 < split < arena.numel
    snap = {}

    def at_hook(g):
        ops.WgradStream.flush_all()                  # what GradReducer._launch does before it issues the collective
        torch.cuda.synchronize()
        snap["region"] = arena.grads[split:].clone()
        return g

    def fwd_hook(mod, inputs, output):
        if output.requires_grad and torch.is_grad_enabled():
            output.register_hook(at_hook)

    handle = model.bert.lang_encoder.register_forward_hook(fwd_hook)
    try:
        for step, task in enumerate(("sap", "mlm", "masksem")):
            snap.clear()
            ops.RT.new_step(500 + step)
            arena.zero_grad()
            b = synthetic.batch_to(synthetic.make_batch(cfg, task, 3, seed=90 + step, ragged=True), DEV)
            model(b, task).mean().backward()
            arena.sync()
            torch.cuda.synchronize()
            assert "region" in snap, task                # the hook fired: every task here reads the text encoder
            final = arena.grads[split:]
            late = (snap["region"] != final).nonzero()
            if late.numel():
                off = int(late[0]) + split
                name = [n for n, (o, k) in arena.slices.items() if o <= off < o + k]
                raise AssertionError(f"{task}: {late.shape[0]} gradient elements of [split, end) changed after the "
                                     f"text hook, first in {name}")
            assert float(final.abs().sum()) > 0
    finally:
        handle.remove()
