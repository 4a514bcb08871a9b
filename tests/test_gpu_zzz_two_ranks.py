"""Two data-parallel ranks of the PRODUCT (HIP kernels, PretrainTrainer, GradReducer with its hooks and side stream) on ONE
MI355X: two processes share cuda:0 and exchange gradients over gloo (RCCL refuses two ranks on one device; gloo stages
device tensors through the host).  What a two-GPU run would show about correctness, minus RCCL itself: the wrap-time
broadcast, the per-region exchange issued from backward hooks, 1/world folded into the clip, identical replicas after every
step -- against the same steps done by ONE process that runs both ranks' batches and sums the gradients itself."""
import os
import socket

import pytest
import torch

from vln_bevbert_amd import synthetic
from vln_bevbert_amd.config import BevBertConfig

pytestmark = pytest.mark.gpu
TASKS = ("sap", "mlm", "masksem", "sap")
HP = dict(learning_rate=1e-3, warmup_steps=2, num_train_steps=20, betas=(0.9, 0.98), weight_decay=0.01, grad_norm=1.0)


def _cfg():
    return BevBertConfig.tiny(num_l_layers=2, num_x_layers=2, vocab_size=400)


def _model(seed_offset=0):
    from vln_bevbert_amd import weights
    from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    model = GlocalTextPathCMTPreTraining(_cfg())
    sd = weights.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    if seed_offset:      # replicas start DIFFERENT (the reference seeds ranks differently): the broadcast has to fix it
        g = torch.Generator().manual_seed(seed_offset)
        sd = {k: (v + 0.01 * torch.randn(v.shape, generator=g)).to(v.dtype) if v.is_floating_point() else v for k, v in sd.items()}
    model.load_state_dict(sd)
    model.tie_weights()
    arena = model.finalize("cuda", torch.float32)
    model.train()
    model.set_dropout(0.0)
    return model, arena


def _batch(i, task, rank):
    return synthetic.batch_to(synthetic.make_batch(_cfg(), task, 2, seed=500 + 10 * i + rank, ragged=True), "cuda")


def _worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from vln_bevbert_amd.train import PretrainTrainer
        model, arena = _model(seed_offset=17 * rank)
        tr = PretrainTrainer(model, arena, rank=rank, world_size=world, **HP)
        assert tr.reducer.active and tr.reducer.world == world and tr.overlap
        start = arena.params.detach().cpu().numpy().copy()      # numpy: pickled by value (tensors travel as fd handles,
                                                                # which die with this process)
        losses, sums = [], []
        for i, task in enumerate(TASKS):
            loss = tr.forward_backward(task, _batch(i, task, rank))
            torch.cuda.synchronize()
            losses.append(float(loss))
            sums.append(arena.grads.detach().cpu().numpy().copy())
            tr.optimizer_step()
        torch.cuda.synchronize()
        end = arena.params.detach().cpu().numpy().copy()
        # a StaticBatch step on two ranks: the host-staged exchange cannot be captured, the trainer issues it eagerly and
        # the replicas stay identical
        from vln_bevbert_amd.static_step import StaticBatch
        assert tr.capture_ok is False
        sb = StaticBatch(_cfg(), "sap", synthetic.make_batch(_cfg(), "sap", 2, seed=900 + rank, ragged=True, sems_as="ids"), "cuda")
        for _ in range(3):
            tr.step("sap", sb)
        assert sb.graph is None
        torch.cuda.synchronize()
        q.put((rank, "ok", start, losses, sums, end, tr.reducer.queue_report, arena.params.detach().cpu().numpy().copy()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, "error", f"{type(e).__name__}: {e}\n{traceback.format_exc()[-1500:]}"))


def test_two_ranks_on_one_gpu_train_like_one_process_summing_both_batches():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    from vln_bevbert_amd import lib, ops
    from vln_bevbert_amd.train import PretrainTrainer, warmup_linear_lr
    lib.load()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(2):
            item = q.get(timeout=420)
            got[item[0]] = item
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for r in range(2):
        assert got[r][1] == "ok", got[r][2]
    t = torch.from_numpy
    (_, _, start0, loss0, sums0, end0, rep0, late0), (_, _, start1, loss1, sums1, end1, _, late1) = got[0], got[1]
    start0, start1, end0, end1 = t(start0), t(start1), t(end0), t(end1)
    assert torch.equal(t(late0), t(late1)) and not torch.equal(t(late0), end0)      # ... and after the StaticBatch steps
    sums0, sums1 = [t(x) for x in sums0], [t(x) for x in sums1]
    # wrap-time broadcast: rank 1 was built with other weights and starts from rank 0's
    assert torch.equal(start0, start1)
    assert rep0 == {"backend": "gloo", "checked": False}
    # the summed gradients and the replicas agree bit for bit on both ranks after every step
    for a, b in zip(sums0, sums1):
        assert torch.equal(a, b)
    assert torch.equal(end0, end1) and not torch.equal(end0, start0)
    # one process, both batches: sum of the two backward passes, 1/2 folded into the clip, same AdamW
    model, arena = _model()
    assert torch.equal(arena.params.detach().cpu(), start0)
    tr = PretrainTrainer(model, arena, **HP)
    for i, task in enumerate(TASKS):
        local, ls = [], []
        for r in range(2):
            ops.RT.new_step(7 + i, plan_key=None)
            ls.append(float(tr._forward_backward(task, _batch(i, task, r))))
            torch.cuda.synchronize()
            local.append(arena.grads.detach().clone())
        # another process may have timed its way to another library GEMM algorithm for a shape: last bits, not more
        for mine, theirs in zip(ls, (loss0[i], loss1[i])):
            assert abs(mine - theirs) <= 1e-5 * max(1.0, abs(theirs)), (task, ls, loss0[i], loss1[i])
        total = local[0] + local[1]
        ref = sums0[i].cuda()
        assert float((total - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), (task, float((total - ref).abs().max()))
        arena.grads.copy_(ref)
        arena.clip_and_step(warmup_linear_lr(i + 1, HP["learning_rate"], HP["warmup_steps"], HP["num_train_steps"]),
                            HP["betas"], 1e-6, HP["weight_decay"], HP["grad_norm"], grad_pre_scale=0.5)
    torch.cuda.synchronize()
    err = float((arena.params.detach().cpu() - end0).abs().max())
    assert err <= 1e-6, err
