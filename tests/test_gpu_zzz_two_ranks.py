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


def test_bench_multi_rank_path_rehearsed_with_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` end to end -- self-launched ranks, trainer with world_size 2, barriers, maximum over
    ranks, the `rccl` block with its region timeline, ONE line from rank 0 -- with both ranks on GPU 0 and gloo underneath
    (BEVBERT_BENCH_SHARE_GPU=1).  Not a measurement (the line says so); the first run on a multi-GPU node must not be the
    first run of this code."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BEVBERT_BENCH_SHARE_GPU="1", BEVBERT_BENCH_WATCHDOG_S="600")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    detail = os.path.join(root, "gpurun_out", "bench_detail_rehearsal.json")
    os.makedirs(os.path.dirname(detail), exist_ok=True)
    pr = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "11", "--warmup", "2",
                         "--launch", "eager", "--no-side", "--no-stream", "--no-cpu-baseline", "--no-fwd", "--no-kernel-pass",
                         "--detail", detail], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert pr.returncode == 0, pr.stderr[-2000:]
    line = pr.stdout.strip().splitlines()[-1]
    d = json.loads(line)
    assert len(line) < 6144
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    assert d["steps"] == 11 and d["value"] > 0 and abs(d["value"] - 11 * 128 / (11 * d["ms_per_step"] / 1e3)) < 0.01 * d["value"]
    assert "REHEARSAL" in d["data"] and d["scaling"] == "weak"
    assert d["rccl"]["ranks"] == 2 and d["rccl"]["backend"] == "gloo" and d["rccl"]["allreduce_bytes_per_step"] > 9e8
    full = json.load(open(detail))
    assert len(full["rccl"]["region_timeline"]) >= 1
