"""world_size-2 gloo test (CPU) of the data-parallel gradient exchange: the in-place two-phase arena all-reduce used by
PretrainTrainer gives, together with the 1/world pre-scale, the gradient of the global batch; ranks draw disjoint
batches and agree on the task sequence without a collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bevbert_ref as R
from vln_bevbert_amd import synthetic, weights
from vln_bevbert_amd.config import BevBertConfig


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from vln_bevbert_amd.train import GradReducer, TaskSampler
    cfg = BevBertConfig.tiny(num_l_layers=1, num_x_layers=1, vocab_size=300)
    model = GlocalTextPathCMTPreTraining(cfg)
    sd = weights.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict(sd)
    model.tie_weights()
    arena = model.finalize("cpu", torch.float32)
    split = min(arena.slices[n][0] for n in arena.slices
                if n.startswith("bert.local_encoder") or n.startswith("bert.global_encoder") or not n.startswith("bert."))
    reducer = GradReducer(arena.grads, split)
    assert reducer.world == world and 0 < split < arena.numel

    # each rank: oracle gradient of ITS batch (the oracle is the checker; the reducer is the code under test),
    # written into the arena exactly where the HIP backward kernels would accumulate it
    task = TaskSampler(seed=0).next()
    b = synthetic.make_batch(cfg, "sap", 2, seed=1000 + rank)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    leaf["mlm_head.predictions.decoder.weight"] = leaf["bert.embeddings.word_embeddings.weight"]
    R.pretrain_forward(leaf, cfg, b, "sap").mean().backward()
    for n, p in model.named_parameters():
        if leaf[n].grad is not None:
            p.main_grad.copy_(leaf[n].grad)
    local = arena.grads.clone()
    # round 5: regions that follow backward through the map encoders go out FIRST (the heads, then an x-layer's run), in
    # any order; phase A then reduces only what they left of [split, end) -- every element exactly once
    n_all = arena.numel
    heads_lo = min(o for n, (o, k) in arena.slices.items() if not n.startswith("bert."))
    reducer.launch_region(heads_lo, n_all)
    mid_lo, mid_hi = split + (heads_lo - split) // 3, split + (heads_lo - split) // 2
    reducer.launch_region(mid_lo, mid_hi)
    reducer.phase_a()                       # the rest of map encoders + heads (overlaps with the text encoder's backward)
    assert reducer._remaining() == [(0, split)] and sorted(reducer._done) == [(split, mid_lo), (mid_lo, mid_hi), (mid_hi, heads_lo), (heads_lo, n_all)]
    reducer.launch_region(split // 3, split // 2)       # an out-of-order middle region (opt-in per-layer text phases)
    assert reducer._remaining() == [(0, split // 3), (split // 2, split)]
    reducer.finish()                        # ... and whatever is left, exactly once
    assert reducer._done == [] and reducer._works == []
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    assert torch.allclose(arena.grads, sum(gathered), rtol=0, atol=1e-6)
    # 1/world pre-scale == gradient of the mean loss over the global batch
    if rank == 0:
        bb = [synthetic.make_batch(cfg, "sap", 2, seed=1000 + r) for r in range(world)]
        leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        leaf["mlm_head.predictions.decoder.weight"] = leaf["bert.embeddings.word_embeddings.weight"]
        tot = sum(R.pretrain_forward(leaf, cfg, x, "sap").mean() for x in bb) / world
        tot.backward()
        n = "bert.local_encoder.encoder.x_layers.0.visn_inter.dense.weight"
        o, k = arena.slices[n]
        err = (arena.grads[o:o + k].view_as(leaf[n].grad) / world - leaf[n].grad).abs().max()
        q.put(("ok", float(err), task, float((local - gathered[1]).abs().max())))
    else:
        q.put(("ok", 0.0, task, 1.0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gradient_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[0] == "ok" for r in res)
    assert max(r[1] for r in res) < 1e-5                 # averaged arena gradient == global-batch gradient
    assert len({r[2] for r in res}) == 1                 # same task on every rank, no broadcast needed
    assert max(r[3] for r in res) > 0                    # ranks really drew different batches


def _worker_broadcast(rank, world, port, q):
    """Ranks start from DIFFERENT weights (the reference seeds each rank with seed + rank, train_r2r.py:85-88); the
    trainer's construction must leave every replica with rank 0's parameters (DDP's wrap-time broadcast)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from vln_bevbert_amd.train import PretrainTrainer
    cfg = BevBertConfig.tiny(num_l_layers=1, num_x_layers=1, vocab_size=300)
    torch.manual_seed(100 + rank)
    model = GlocalTextPathCMTPreTraining(cfg)           # init_weights draws from the rank-specific generator
    arena = model.finalize("cpu", torch.float32)
    mine = arena.params.clone()
    everyone = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    differed = float((everyone[0] - everyone[1]).abs().max())
    arena.exp_avg = torch.full_like(arena.params, float(rank))        # a resumed run carries optimiser state as well
    arena.exp_avg_sq = torch.full_like(arena.params, float(rank))
    PretrainTrainer(model, arena, rank=rank, world_size=world)
    same = torch.equal(arena.params, everyone[0]) and float(arena.exp_avg.abs().max()) == 0.0
    # the nn.Parameter objects are views of the arena: the model itself now holds rank 0's weights
    w = model.global_sap_head.net[0].weight
    o, k = arena.slices["global_sap_head.net.0.weight"]
    same = same and torch.equal(w.detach().reshape(-1), everyone[0][o:o + k])
    q.put(("ok", differed, same))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_trainer_broadcasts_rank0_state_to_all_replicas():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_broadcast, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[0] == "ok" and r[2] for r in res)
    assert min(r[1] for r in res) > 1e-3                 # the replicas really started apart


# ---------------------------------------------------------------------------------------------------------------
# torch-API bridge of the arena (arena.py) + train.ArenaDataParallel on two gloo ranks: a toy module whose backward
# writes its weight gradient straight into ``main_grad`` (as every Linear of the product does), trained with the
# reference's loop body (loss.backward(); clip_grad_norm_; torch optimiser; zero_grad) on rank-specific halves of a
# batch, must follow the single-process run on the whole batch.
class _ArenaLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x)
        ctx.w = w
        return x @ w.detach().t()

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        w = ctx.w
        w.arena.touch(w)
        w.main_grad.add_(dy.t() @ x)                 # gradient written behind autograd's back
        return dy @ w.detach(), None


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.a = torch.nn.Parameter(torch.randn(6, 4, generator=g))
        self.b = torch.nn.Parameter(torch.randn(3, 6, generator=g))
        self.unused = torch.nn.Parameter(torch.ones(2))

    def finalize(self, device, dtype=torch.float32):
        from vln_bevbert_amd.arena import ParamArena
        self.arena = ParamArena(self, device, dtype)
        return self.arena

    def forward(self, x):
        from vln_bevbert_amd.vilmodel import ensure_arena
        ensure_arena(self)
        return _ArenaLinearFn.apply(torch.tanh(_ArenaLinearFn.apply(x, self.a)), self.b)


def _toy_run(rank, world, port, out):
    import torch.distributed as dist
    from vln_bevbert_amd.train import ArenaDataParallel
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                     # replicas start different: the wrap-time broadcast aligns them
    model = _Toy()
    if world > 1 and rank > 0:
        with torch.no_grad():
            model.a.add_(1.0)
    model.finalize("cpu")
    net = ArenaDataParallel(model) if world > 1 else model
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    g = torch.Generator().manual_seed(9)
    xs, ys = torch.randn(5, 8, 4, generator=g), torch.randn(5, 8, 3, generator=g)
    for step in range(5):
        x, y = xs[step], ys[step]
        if world > 1:                                  # each rank its half of the batch
            x, y = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
        loss = ((net(x) - y) ** 2).mean()
        loss.backward()
        assert model.a.grad is not None and model.unused.grad is None
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        opt.zero_grad()                                # set_to_none=True: the next forward zeroes the arena
    if rank == 0:
        torch.save({"a": model.a.detach().clone(), "b": model.b.detach().clone()}, out)
    if world > 1:
        dist.destroy_process_group()


def test_arena_data_parallel_follows_the_single_process_run_through_a_torch_optimizer(tmp_path):
    import torch.multiprocessing as mp
    single, double = str(tmp_path / "single.pt"), str(tmp_path / "double.pt")
    _toy_run(0, 1, 0, single)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_toy_run, args=(2, port, double), nprocs=2, join=True)
    a, b = torch.load(single), torch.load(double)
    for k in ("a", "b"):
        assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-6), (k, (a[k] - b[k]).abs().max())


class _ToyBranch(_Toy):
    """``extra`` is used by rank 0 only (a task head another rank's batch does not reach)."""

    def __init__(self):
        super().__init__()
        self.extra = torch.nn.Parameter(torch.full((3, 3), 0.5))
        self.use_extra = False

    def forward(self, x):
        y = super().forward(x)
        return _ArenaLinearFn.apply(y, self.extra) if self.use_extra else y


def _union_run(rank, world, port, out):
    import torch.distributed as dist
    from vln_bevbert_amd.train import ArenaDataParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _ToyBranch()
    model.use_extra = rank == 0
    model.finalize("cpu")
    net = ArenaDataParallel(model)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(9)
    xs, ys = torch.randn(3, 8, 4, generator=g), torch.randn(3, 8, 3, generator=g)
    for step in range(3):
        x, y = xs[step][rank * 4:(rank + 1) * 4], ys[step][rank * 4:(rank + 1) * 4]
        ((net(x) - y) ** 2).mean().backward()
        # used on rank 0 only: BOTH ranks receive the averaged gradient (DDP find_unused_parameters=True semantics)
        assert model.extra.grad is not None, rank
        assert model.unused.grad is None
        opt.step()
        opt.zero_grad()
    torch.save({"extra": model.extra.detach().clone(), "a": model.a.detach().clone()}, out + f".{rank}")
    dist.destroy_process_group()


def test_a_parameter_used_on_one_rank_only_is_updated_on_every_rank(tmp_path):
    """arena._publish exchanges the set of touched parameters, so a parameter only rank 0's batch reaches gets
    ``.grad`` (the averaged gradient) on rank 1 as well and the replicas stay identical."""
    out = str(tmp_path / "u.pt")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_union_run, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["extra"], r1["extra"]) and torch.equal(r0["a"], r1["a"])
    assert not torch.equal(r0["extra"], torch.full((3, 3), 0.5))


def _bf16_exchange_run(rank, world, port, out, exchange="bf16_a2a"):
    from vln_bevbert_amd.train import GradReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(50 + rank)
    n = 10_007                                             # not a multiple of the world size: exercises the padding
    local = torch.randn(n, generator=g) * torch.logspace(-4, 2, n)          # seven decades of magnitudes
    want = local.clone()
    dist.all_reduce(want)                                  # fp32 reference (the default exchange)
    got = local.clone()
    red = GradReducer(got, split=4000, exchange=exchange)
    assert red.active and red.exchange == exchange
    red.phase_a()                                          # [4000, n) first, as the backward hook would
    red.launch_region(1000, 2500)
    red.finish()                                           # the rest
    if rank == 0:
        torch.save({"got": got, "want": want}, out)
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["bf16", "bf16_a2a"])
def test_bf16_gradient_exchange_sums_in_fp32_and_stays_within_two_bf16_roundings(tmp_path, exchange):
    """BEVBERT_GRAD_EXCHANGE=bf16 (reduce-scatter + all-gather: capturable) / bf16_a2a (all-to-all + fp32 sum +
    all-gather): half the bytes per link; against the fp32 all-reduce every element is within the bf16 rounding of each
    rank's contribution plus that of the result (each 2^-9 relative), in any region order."""
    out = str(tmp_path / "x.pt")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_bf16_exchange_run, args=(2, port, out, exchange), nprocs=2, join=True)
    r = torch.load(out)
    got, want = r["got"], r["want"]
    assert bool(torch.isfinite(got).all())
    # |error| <= 2^-9 (|a| + |b|) + 2^-9 |a + b|  <=  3 * 2^-9 * (|a| + |b|); bound it by the result's own scale per decade
    rel = float((got - want).norm() / want.norm())
    assert rel < 4e-3, rel
    assert float((got - want).abs().max()) <= 3 * 2 ** -8 * float(want.abs().max())
