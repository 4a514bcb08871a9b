"""Synthetic-batch pre-training harness: the build's counterpart of the reference's hot loop
(pretrain_src/train_r2r.py:247-313) -- task mix, loss.mean(), clip 5.0, warm-up-linear LR, AdamW -- plus the
data-parallel gradient exchange that replaces DistributedDataParallel (pretrain_src/utils/misc.py:64-77).

Data parallelism: one process per GPU, batches sharded by rank (seed + rank), weights replicated.
The only data-path collective is the gradient SUM all-reduce; it runs IN PLACE on slices of the flat gradient
arena (no bucket copies), on a side stream, in two phases so that it overlaps with the rest of backward:

    phase A  [first map-encoder parameter, end of arena)   launched when d(loss)/d(text embeddings) is complete,
             i.e. every kernel of the two map encoders and the heads has been enqueued (~57 % of the bytes);
             it overlaps with the backward of the 9-layer text encoder and the panorama encoder;
    phase B  [0, first map-encoder parameter)               launched when backward returns.

The 1/world averaging is folded into the clip kernel (grad_pre_scale), and parameters unused by the step's task
simply contribute zeros (the semantics of find_unused_parameters=True without the graph traversal).  The task of
each step is drawn from a generator seeded identically on every rank, so the reference's per-step task-id broadcast
(pretrain_src/data/loader.py:56-59) needs no collective at all.
"""
import os
import random

import torch
import torch.distributed as dist

from . import ops
from .static_step import GraphedStep, StaticBatch


def warmup_linear_lr(step, base_lr, warmup_steps, total_steps):
    """optim/sched.py:17-30."""
    f = step / warmup_steps if step < warmup_steps else max(0, (total_steps - step) / (total_steps - warmup_steps))
    lr = base_lr * f
    return lr if lr > 0 else 1e-8


def load_gemm_tuning():
    """Load scripts/tune_gemms.py's hipBLASLt solution table (read-only) if one was shipped; returns #entries or 0."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_results.csv")
    if os.environ.get("BEVBERT_GEMM_TUNING", "1") != "1" or not os.path.exists(path):
        return 0
    try:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(False)
        tunable.record_untuned_enable(False) if hasattr(tunable, "record_untuned_enable") else None
        tunable.set_filename(path + ".unused")      # never overwrite the shipped table
        ok = tunable.read_file(path)
        return len(tunable.get_results()) if ok else 0
    except Exception:
        return 0


class TaskSampler:
    """MetaLoader's ratio sampling (data/loader.py:18-62) with a generator shared by all ranks."""

    def __init__(self, task_ratio="mlm.5.sap.5.masksem.1", seed=0):
        parts = task_ratio.split(".")
        self.tasks = parts[::2]
        ratios = [int(r) for r in parts[1::2]]
        self.pool = [t for t, r in zip(self.tasks, ratios) for _ in range(r)]
        self.rng = random.Random(seed)

    def next(self):
        return self.rng.choice(self.pool)


class GradReducer:
    """In-place SUM all-reduce of a flat gradient buffer in [split, end) then [0, split) (see module docstring)."""

    def __init__(self, flat_grads, split, group=None, force=False, exchange=None):
        """force=True issues the collectives even for a single-rank group (used to exercise the RCCL path on one GPU).
        exchange (BEVBERT_GRAD_EXCHANGE):
          "fp32"     in-place SUM all-reduce of the fp32 arena: the reference's DDP semantics, 956 MB per step for the
                     R2R model (default);
          "bf16"     gradients travel as bf16: cast -> reduce-scatter (RCCL sums each hop in fp32 and forwards bf16:
                     at most W - 1 roundings of 2^-9 relative) -> all-gather of the bf16 result: half the bytes on every
                     xGMI link, and -- unlike all-to-all -- both collectives survive hipGraph stream capture, so a step
                     with this exchange is captured like one with the fp32 all-reduce;
          "bf16_a2a" bf16 on the wire with fp32 ACCUMULATION: cast -> all-to-all of the W shards -> fp32 sum of the W
                     received pieces -> all-gather: two roundings whatever W is, but RCCL 2.26's all-to-all takes the
                     process down under stream capture, so steps with it are issued eagerly."""
        import os
        self.exchange = exchange or os.environ.get("BEVBERT_GRAD_EXCHANGE", "fp32")
        assert self.exchange in ("fp32", "bf16", "bf16_a2a"), self.exchange
        self.flat = flat_grads
        self.split = int(split)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_available() and dist.is_initialized())
        self.cuda = flat_grads.is_cuda
        self.stream = None
        if self.cuda and self.active:
            from .hwqueues import side_stream
            self.stream = side_stream(flat_grads.device)
        self.queue_report = None     # where the collectives run relative to the compute stream (_settle_collective_queue)
        self._works = []
        self._phase_a_done = False
        self._done = []          # [lo, hi) regions already issued in this step
        self.occupied = None     # sorted (start, end) of the tensors in the buffer (set_occupied): padding gaps are skipped
        self.timeline = None     # set to [] to collect (lo, hi, start event, end event) per region (bench.py's rccl block);
                                 # both events sit on the reducer's stream, the end one right behind its own collective

    _settled = {}        # (device index, id of the group asked for) -> (group used, report): once per process

    def settle_collective_queue(self):
        """Once, before the first step (every rank, same point of the program): if the group's collectives wait behind the
        compute stream's work (collectives_wait_behind_compute), and the group is the default one, move the gradient
        exchange to a group of its own whose communicator takes a stream off the compute stream's hardware queue
        (hwqueues.steer_stream_pool, then new_group).  BEVBERT_COLLECTIVE_QUEUE_CHECK=0 skips the whole check."""
        if self.queue_report is not None or not (self.cuda and self.active) or \
                os.environ.get("BEVBERT_COLLECTIVE_QUEUE_CHECK", "1") != "1":
            return self.queue_report
        if dist.get_backend(self.group) != "nccl":          # gloo moves device tensors through the host: no stream to place
            self.queue_report = {"backend": dist.get_backend(self.group), "checked": False}
            return self.queue_report
        settled = GradReducer._settled.get((self.flat.device.index, id(self.group)))
        if settled is not None:                               # an earlier reducer of this process did the work
            self.group, self.queue_report = settled
            return self.queue_report
        key = (self.flat.device.index, id(self.group))
        from .hwqueues import steer_stream_pool
        # the communicator of a group whose first collective is still to come takes the NEXT pool stream: make that one a
        # stream on the reducer's own queue (the reducer's stream carries only the event edges of the same collectives;
        # the other side streams -- deferred weight gradients, keep bits -- are dealt onto the remaining queues)
        st = steer_stream_pool(self.flat.device, like=self.stream)
        rep = {"steered": st and {"pool": st["pool"], "next": st["next"], "wanted": sum(st["wanted"])},
               "waits_behind_compute": self.collectives_wait_behind_compute()}
        if rep["waits_behind_compute"] and self.group is None:
            # the default group had its communicator (and its stream) before we came: exchange on a group of our own
            st = steer_stream_pool(self.flat.device, like=self.stream)
            self.group = dist.new_group(backend=dist.get_backend())
            rep["own_group"] = {"steered": st and {"pool": st["pool"], "next": st["next"]},
                                "waits_behind_compute": self.collectives_wait_behind_compute()}
        self.queue_report = rep
        GradReducer._settled[key] = (self.group, rep)
        return rep

    def collectives_wait_behind_compute(self, busy_ms=3.0):
        """Diagnosis (all ranks call it together): does a collective issued from the reducer's stream run only after work
        enqueued EARLIER on the compute stream -- i.e. does the collective library's stream share the compute stream's
        in-order hardware queue (hwqueues.py)?  Then eager steps get no overlap of the exchange with backward, whatever the
        hooks do.  ~busy_ms of fills on the compute stream, then a 1 KB all-reduce from the side stream: which finishes
        first.  Returns True / False for the whole group (MAX over ranks: one rank whose collective waits holds everyone
        up), None when collectives are inactive or on CPU."""
        if not (self.cuda and self.active and self.stream is not None):
            return None
        dev = self.flat.device
        main = torch.cuda.current_stream(dev)
        busy = torch.empty(64 << 20, dtype=torch.float32, device=dev)
        t = torch.zeros(256, device=dev)
        dist.all_reduce(t, group=self.group)                  # communicator and its stream exist from here on
        dist.barrier(group=self.group)
        torch.cuda.synchronize(dev)
        from .hwqueues import pick_copy_stream
        issue = pick_copy_stream(dev)                         # a stream known to be off the compute stream's queue: the
        m1, c1 = torch.cuda.Event(), torch.cuda.Event()       # collective's stream waits for an event on the issuing one
        for _ in range(max(4, int(busy_ms / 0.06))):
            busy.fill_(1.0)
        m1.record(main)
        with torch.cuda.stream(issue):                        # no wait on `main`: the probe asks about queues, not events
            dist.all_reduce(t, group=self.group, async_op=True).wait()
            c1.record(issue)
        while not c1.query():
            pass
        waited = torch.tensor([1.0 if m1.query() else 0.0], device=dev)
        torch.cuda.synchronize(dev)
        dist.all_reduce(waited, op=dist.ReduceOp.MAX, group=self.group)
        return bool(waited.item() > 0)

    def drop_pending(self):
        """Forget the collectives of an aborted step (failed graph capture): the step is issued again eagerly."""
        self._works, self._done, self._phase_a_done = [], [], False

    def launch_region(self, lo, hi):
        """Reduce [lo, hi) now: every gradient in it has been enqueued (caller's guarantee).  Regions may be issued in
        any order; finish() covers whatever is left."""
        lo, hi = max(0, int(lo)), min(int(hi), self.flat.numel())
        if self.active and hi > lo:
            self._launch(lo, hi)
            self._done.append((lo, hi))

    def _remaining(self):
        gaps, at = [], 0
        for lo, hi in sorted(self._done):
            if lo > at:
                gaps.append((at, lo))
            at = max(at, hi)
        if at < self.flat.numel():
            gaps.append((at, self.flat.numel()))
        if self.occupied is not None:
            # tensors start on 1024-element boundaries: between two regions that end / start on tensor borders lies padding
            # that no kernel writes -- not worth a collective of its own (r05: eight zero-size launches per step)
            gaps = [(lo, hi) for lo, hi in gaps if self._holds_a_tensor(lo, hi)]
        return gaps

    def _holds_a_tensor(self, lo, hi):
        if self.occupied is None:
            return True
        import bisect
        i = bisect.bisect_right(self._occ_starts, lo) - 1
        return (i >= 0 and self.occupied[i][1] > lo) or (i + 1 < len(self._occ_starts) and self._occ_starts[i + 1] < hi)

    def set_occupied(self, slices):
        """slices: iterable of (offset, numel) of the tensors living in the flat buffer (ParamArena.slices.values())."""
        self.occupied = sorted((int(o), int(o) + int(k)) for o, k in slices)
        self._occ_starts = [o for o, _ in self.occupied]

    def _exchange_bf16(self, view):
        """SUM over ranks of ``view`` (fp32, in place) with bf16 on the wire (see __init__ for the two variants)."""
        W, n = self.world, view.numel()
        per = -(-n // W)
        send = torch.zeros(W * per, dtype=torch.bfloat16, device=view.device)
        send[:n].copy_(view)                                   # fp32 -> bf16 (round to nearest even)
        if self.exchange == "bf16":
            mine = torch.empty(per, dtype=torch.bfloat16, device=view.device)
            dist.reduce_scatter_tensor(mine, send, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(send, mine, group=self.group)
        else:
            recv = torch.empty_like(send)
            # RCCL moves bf16 natively; gloo (the CPU tests) does not know the type in all-to-all: raw bytes there
            as_wire = (lambda t: t) if view.is_cuda else (lambda t: t.view(torch.uint8))
            dist.all_to_all_single(as_wire(recv), as_wire(send), group=self.group)
            mine = recv.view(W, per).float().sum(0).to(torch.bfloat16)      # fp32 sum of the W pieces of MY shard
            dist.all_gather_into_tensor(as_wire(send), as_wire(mine), group=self.group)
        view.copy_(send[:n])                                   # bf16 -> fp32

    def _launch(self, lo, hi):
        if hi <= lo:
            return
        view = self.flat[lo:hi]
        if self.cuda:
            ops.WgradStream.flush_all()                              # deferred weight-gradient launches go out first
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())     # everything enqueued so far has produced `view`
            for side in ops.Branches.side_streams():                 # ... including side streams that carry work of
                if ops.Branches.enabled or side not in ops.WgradStream.streams or ops.WgradStream.dirty:   # this step
                    self.stream.wait_stream(side)
            with torch.cuda.stream(self.stream):
                if self.timeline is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                    self.timeline.append((lo, hi, e0, e1))
                if self.exchange != "fp32":
                    self._exchange_bf16(view)                  # stream-ordered on the reducer's stream
                else:
                    self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                if self.timeline is not None:
                    # the collective runs on the process group's own stream: join it into the reducer's stream now (a
                    # device-side wait, the host goes on) so that the END event is this region's completion and not, as
                    # until round 5, "after every collective of the step had been issued"
                    if self._works and self.exchange == "fp32":
                        self._works[-1].wait()
                    e1.record(self.stream)
        elif self.exchange != "fp32":
            self._exchange_bf16(view)
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def launch_uncovered(self, lo, hi):
        """Reduce what has not been issued yet of [lo, hi) (regions inside it may have gone out from earlier hooks)."""
        for a, b in self._remaining():
            a, b = max(a, int(lo)), min(b, int(hi))
            if b > a and self._holds_a_tensor(a, b):      # the clipped piece of a kept gap may be padding only
                self.launch_region(a, b)

    def phase_a(self, *_):
        """Call when every gradient in [split, end) has been enqueued (tensor hook on the text embeddings): reduces what
        the per-layer hooks of the map encoders have not sent already."""
        if self.active and not self._phase_a_done:
            self._phase_a_done = True
            self.launch_uncovered(self.split, self.flat.numel())

    def finish(self):
        """Call after backward: reduces what phase A did not cover and joins the side stream."""
        if self.active:
            for lo, hi in self._remaining():
                self._launch(lo, hi)
            for w in self._works:
                w.wait()
            if self.stream is not None:
                torch.cuda.current_stream().wait_stream(self.stream)
        self._works = []
        self._phase_a_done = False
        self._done = []


class PretrainTrainer:
    def __init__(self, model, arena, learning_rate=5e-5, warmup_steps=10000, num_train_steps=100000,
                 betas=(0.9, 0.98), weight_decay=0.01, grad_norm=5.0, seed=0, rank=0, world_size=1, overlap=True,
                 force_collectives=False):
        self.model, self.arena = model, arena
        arena.publish_grads = False      # this class drives the arena itself (flat clip + AdamW, in-place all-reduce, graphs)
        self.lr, self.warmup, self.total = learning_rate, warmup_steps, num_train_steps
        self.betas, self.wd, self.grad_norm = betas, weight_decay, grad_norm
        self.seed, self.rank, self.world = seed, rank, world_size
        self.global_step = 0
        import os
        self.use_graphs = os.environ.get("BEVBERT_GRAPHS", "1") == "1" and arena.device.type == "cuda"
        # captured steps fork the independent model branches (text || panorama encoder, global-map || BEV encoder) onto
        # a third stream: in a graph the extra fork / join edges cost no host time (issued eagerly they cost ~3 ms of
        # events per step and make the step host-bound), and the small text / map kernels overlap the 28 224-row ones.
        # Measured in one call on one MI355X at batch 64: eager 18.83, graph with the weight-gradient stream only 19.32,
        # + branch stream 18.58 ms/step.  BEVBERT_GRAPH_BRANCHES=0 turns it off.
        self.graph_branches = os.environ.get("BEVBERT_GRAPH_BRANCHES", "1") == "1"
        # the gradient-arena fill runs on a side stream beside the forward pass (arena.zero_grad); BEVBERT_OVERLAP_ZERO=0: in line
        self.overlap_zero = os.environ.get("BEVBERT_OVERLAP_ZERO", "1") == "1" and arena.device.type == "cuda"
        self.graph_error = None            # set (and use_graphs cleared) if a capture ever fails
        self._graph_pool = None
        first_map = min(arena.slices[n][0] for n in arena.slices
                        if n.startswith("bert.local_encoder") or n.startswith("bert.global_encoder")
                        or not n.startswith("bert."))
        # the arena keeps registration order: embeddings, lang_encoder, img_embeddings come before the map encoders
        self.reducer = GradReducer(arena.grads, first_map, force=force_collectives)
        self.reducer.set_occupied(arena.slices.values())
        self.reducer.settle_collective_queue()
        self.overlap = overlap and self.reducer.active
        # RCCL 2.26's all-to-all under stream capture takes the process down (segmentation fault on the MI355X box,
        # one-rank group, gpurun_out r03w) while all-reduce, reduce-scatter and all-gather capture fine: only steps with
        # the all-to-all form of the bf16 exchange are issued eagerly
        self.capture_ok = not (self.reducer.active and self.reducer.exchange == "bf16_a2a")
        if self.reducer.active and dist.get_backend(self.reducer.group) != "nccl":
            self.capture_ok = False      # a host-staged exchange (gloo on device tensors) is not stream work: eager steps
        if self.reducer.active:
            self.broadcast_state()             # replicas start from rank 0's weights, as under DistributedDataParallel
        # Without collectives the same moment of backward -- d loss / d text-embeddings complete: every kernel of the two
        # map encoders and the heads has been enqueued, 57 % of the parameter bytes -- is used to issue the batched
        # reductions queued so far (split-K partial sums -> arena, LayerNorm / bias column sums): they then run on the
        # weight-gradient stream beside the text encoder's backward instead of all at the END of backward, in front of
        # clip + AdamW on the critical path (multi_accum alone is ~0.6 ms per step at batch 64).  MEASURED SLOWER (round 4,
        # same box, two runs each: 18.40 vs 17.89 ms per step): the flush makes the weight-gradient streams join, which
        # serialises work that otherwise overlaps the text encoder's backward.  Off unless BEVBERT_EARLY_FLUSH=1.
        self.early_flush = os.environ.get("BEVBERT_EARLY_FLUSH", "0") == "1" and arena.device.type == "cuda"
        if self.early_flush and not self.overlap:
            model.bert.lang_encoder.register_forward_hook(self._hook_flush)
        if self.overlap:
            self.install_map_layer_hooks()
            model.bert.lang_encoder.register_forward_hook(self._hook_text)
            # finer pipeline for phase B (embeddings + text + panorama encoders, 43 % of the gradient bytes): when the
            # gradient w.r.t. the INPUT of text layer k is complete, every kernel of the text layers >= k has been
            # enqueued (or deferred: GradReducer._launch flushes the weight-gradient queue first), so their arena region
            # goes out while the earlier layers are still in backward.  BEVBERT_REDUCE_TEXT_LAYERS: comma-separated layer
            # indices, "auto" (default) = the layers at one and two thirds of the stack, "" = off.  The hook-time
            # finality of every region is tested on the GPU (test_text_layer_regions_are_final_when_their_hooks_fire).
            for k, lo, hi in self.text_layer_regions(model, arena):
                model.bert.lang_encoder.layer[k].register_forward_pre_hook(self._make_layer_hook(lo, hi))

    @staticmethod
    def map_layer_regions(model, arena):
        """Arena regions that follow backward THROUGH the map encoders (round 5; before, the first collective -- 57 % of the
        bytes -- waited for both encoders, ~70 % of a SAP step's backward): {"heads": (lo, hi), ("local" | "global", k): (lo, hi)
        for the x-layers k >= 1}.  Layer k's region is the contiguous run of its own parameters WITHOUT the K | V
        projections of its cross-attention: those are packed with every layer's into one operand (ops.hoisted_kv, laid out
        inside layer 0's run) whose gradient is final only when the whole encoder is done.  Layer 0, the packed K | V, the
        encoder's input embeddings and what follows the layers (sprel_linear) stay with the catch-all of phase A."""
        out = {}
        heads = [(o, o + k) for n, (o, k) in arena.slices.items() if not n.startswith("bert.")]
        if heads:
            lo, hi = min(lo for lo, _ in heads), max(hi for _, hi in heads)
            # same rule as for the x-layer runs below: the span goes out from the FIRST x-layer hook, so an
            # encoder tensor registered between two heads would be reduced before its gradient is final -- then the heads
            # stay with the catch-all of phase A instead
            if not any(n.startswith("bert.") and lo <= o < hi for n, (o, c) in arena.slices.items()):
                out["heads"] = (lo, hi)
        for enc in ("local", "global"):
            mod = getattr(model.bert, f"{enc}_encoder").encoder
            kv = {n for group in mod.arena_groups(f"bert.{enc}_encoder.encoder.") for n in group}
            for k in range(1, len(mod.x_layers)):
                pre = f"bert.{enc}_encoder.encoder.x_layers.{k}."
                own = sorted((o, o + c) for n, (o, c) in arena.slices.items() if n.startswith(pre) and n not in kv)
                if not own:
                    continue
                lo, hi = own[0][0], own[-1][1]
                # nothing foreign inside the run (tensors start on 1024-element boundaries: gaps are padding)
                inside = [n for n, (o, c) in arena.slices.items() if lo <= o < hi and not (n.startswith(pre) and n not in kv)]
                if not inside:
                    out[(enc, k)] = (lo, hi)
        return out

    def install_map_layer_hooks(self):
        regions = self.map_layer_regions(self.model, self.arena)
        self._map_regions = regions
        self._map_uses, self._map_fired, self._heads_sent = {}, {}, False

        def make(enc):
            def hook(k, fired):
                key = (enc, k)
                if not fired:                                      # forward: one more use of layer k
                    self._map_uses[key] = self._map_uses.get(key, 0) + 1
                    return
                self._map_fired[key] = self._map_fired.get(key, 0) + 1
                if not self._heads_sent and "heads" in regions:     # every head's backward precedes any x-layer's
                    self._heads_sent = True
                    self.reducer.launch_region(*regions["heads"])
                if key in regions and self._map_fired[key] == self._map_uses.get(key, 0):
                    self.reducer.launch_region(*regions[key])
            return hook
        for enc in ("local", "global"):
            getattr(self.model.bert, f"{enc}_encoder").encoder.region_hook = make(enc)

    def _reset_map_hooks(self):
        if getattr(self, "_map_regions", None) is not None:
            self._map_uses, self._map_fired, self._heads_sent = {}, {}, False

    @staticmethod
    def text_layer_regions(model, arena, spec=None):
        """[(layer k, lo, hi)]: arena region of the text layers k .. (next hooked layer - 1), last hooked layer first."""
        import os
        if spec is None:
            spec = os.environ.get("BEVBERT_REDUCE_TEXT_LAYERS", "auto")
        n_layers = len(model.bert.lang_encoder.layer)
        if spec.strip() == "auto":
            layers = {round(n_layers / 3), round(2 * n_layers / 3)}
        else:
            layers = {int(x) for x in spec.split(",") if x.strip()}
        hi = max(o + k for n, (o, k) in arena.slices.items() if n.startswith("bert.lang_encoder."))
        out = []
        for k in sorted({x for x in layers if 0 < x < n_layers}, reverse=True):
            lo = min(o for n, (o, _) in arena.slices.items() if n.startswith(f"bert.lang_encoder.layer.{k}."))
            out.append((k, lo, hi))
            hi = lo
        return out

    def _make_layer_hook(self, lo, hi):
        def pre_hook(module, inputs):
            x = inputs[0]
            if torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled():
                x.register_hook(lambda g: (self.reducer.launch_region(lo, hi), g)[1])
        return pre_hook

    def _hook_flush(self, module, inputs, output):
        if output.requires_grad and torch.is_grad_enabled():
            output.register_hook(lambda g: (ops.WgradStream.flush_all(), g)[1])
        return output

    def _hook_text(self, module, inputs, output):
        if output.requires_grad and torch.is_grad_enabled():
            output.register_hook(lambda g: (self.reducer.phase_a(), g)[1])
        return output

    def broadcast_state(self, src=0):
        """DistributedDataParallel's wrap-time synchronisation (pretrain_src/utils/misc.py:64-72: DDP broadcasts
        rank 0's parameters and buffers when it wraps the model).  The reference seeds every rank differently
        (train_r2r.py:85-88) and loads its LXMERT / XLM-R checkpoints with strict=False, so every parameter the
        checkpoint lacks (image / BEV / map embeddings, all heads) starts rank-specific; only the broadcast makes the
        replicas identical.  Here: ONE collective over the flat parameter arena (plus the optimiser state when a run
        is resumed), then the bf16 compute copy is rebuilt from the received masters."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        a = self.arena
        dist.broadcast(a.params, src=src, group=self.reducer.group)
        if a.exp_avg is not None:
            dist.broadcast(a.exp_avg, src=src, group=self.reducer.group)
            dist.broadcast(a.exp_avg_sq, src=src, group=self.reducer.group)
            dist.broadcast(a.chunk_steps, src=src, group=self.reducer.group)
        for b in self.model.buffers():
            dist.broadcast(b, src=src, group=self.reducer.group)
        a.sync_shadow()

    def _step_seed(self):
        return (self.seed + self.rank) * 1000003 + self.global_step                  # per-rank dropout stream

    def forward_backward(self, task, batch):
        """Forward + backward + gradient exchange of one batch; leaves the (summed) gradients in ``arena.grads``."""
        self.global_step += 1
        ops.RT.new_step(self._step_seed(), plan_key=self._plan_key(task, batch))
        return self._forward_backward(task, batch)

    @staticmethod
    def _plan_key(task, batch):
        """Identity of a step's kernel sequence for ops.ATTN_BITS: static batches carry their shape signature; batches
        of the reference API (data-dependent shapes) take the inline path."""
        return (task, batch.signature) if isinstance(batch, StaticBatch) else None

    def _forward_backward(self, task, batch):
        self._reset_map_hooks()
        self.arena.zero_grad(overlap=self.overlap_zero)
        try:
            if isinstance(batch, StaticBatch):
                loss = self.model.loss_mean(batch.tensors, task)
            else:
                loss = self.model(batch, task, compute_loss=True).mean()             # train_r2r.py:263
        finally:
            self.arena.wait_zero()             # (also on an exception: a capture can only end with every stream joined)
        if self.reducer is not None and self.reducer.timeline is not None:           # bench.py's region timeline
            self.backward_start = torch.cuda.Event(enable_timing=True)
            self.backward_start.record()
        if loss.is_cuda and loss.dim() == 0:
            # the seed gradient of backward(): one persistent device scalar instead of a ones_like fill per step
            one = self.__dict__.get("_one")
            if one is None or one.dtype != loss.dtype or one.device != loss.device:
                one = self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
            loss.backward(gradient=one)
        else:
            loss.backward()
        self.arena.sync()                      # side-stream work has written its gradients
        self.reducer.finish()
        return loss.detach()

    def optimizer_step(self, lr=None):
        """clip_grad_norm_ + AdamW + schedule (train_r2r.py:278-313) on the flat arena.  ``lr=None``: the schedule value
        of the current global step is computed here and written to the device-resident slot."""
        if lr is None:
            lr = warmup_linear_lr(self.global_step, self.lr, self.warmup, self.total)
        self.arena.clip_and_step(lr, self.betas, 1e-6, self.wd, self.grad_norm, grad_pre_scale=1.0 / self.world)

    def step(self, task, batch):
        """One optimisation step on one batch (gradient_accumulation_steps == 1, as every shipped config).

        A ``static_step.StaticBatch`` runs eagerly ``GRAPH_WARMUP`` times, is then captured into a hipGraph (forward,
        backward with the weight-gradient stream, clip, AdamW -- and, on several GPUs, the in-place all-reduce on its
        side stream) and replayed from then on: one launch per step instead of ~1 000."""
        if isinstance(batch, StaticBatch) and self.use_graphs and self.capture_ok and batch.capturable and ops.RT.trace is None:
            return self._static_step(task, batch)
        loss = self.forward_backward(task, batch)
        self.optimizer_step()
        return loss

    # ---- captured steps ----------------------------------------------------------------------------------------
    GRAPH_WARMUP = 2

    def _static_step(self, task, sb):
        assert sb.task == task
        self.global_step += 1
        lr = warmup_linear_lr(self.global_step, self.lr, self.warmup, self.total)
        a = self.arena
        if a.exp_avg is None:                       # optimiser state exists before anything is captured
            a.exp_avg = torch.zeros_like(a.params)
            a.exp_avg_sq = torch.zeros_like(a.params)
        ops.RT.new_step(self._step_seed(), plan_key=None if (sb.graph is not None and sb.graph.owner is self)
                        else self._plan_key(task, sb))          # salt -> device word (a 4-byte fill on the stream)
        a.set_lr(lr)                                # learning rate -> device word
        gs = sb.graph
        if gs is not None and gs.owner is self:
            a.upload_flags()
            gs.graph.replay()
            return gs.loss.clone()
        branches = ops.Branches.enabled
        if sb.eager_runs < self.GRAPH_WARMUP:
            # the warm-up runs use the stream layout of the capture: the order in which the deferred weight-gradient work
            # is issued (and with it the scratch addresses in the cached reduction task tables) depends on it
            sb.eager_runs += 1
            ops.Branches.enabled = branches or self.graph_branches
            try:
                loss = self._forward_backward(task, sb)
            finally:
                ops.Branches.enabled = branches
            self.optimizer_step(lr=None)
            return loss
        # capture: flags / plans / side streams are warm, nothing below allocates outside the graph's pool
        a.upload_flags()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        if self._graph_pool is None:
            # all step graphs of a trainer share one private memory pool: they are replayed one at a time on one stream
            # and exchange nothing through pool memory, so the pool is as large as the largest step, not the sum
            self._graph_pool = torch.cuda.graph_pool_handle()
        ops.Branches.enabled = branches or self.graph_branches
        # thread-local capture mode: other threads of the process keep making HIP calls while this one captures -- the
        # loader's producer thread allocates pinned staging buffers and device buffer sets (loader.BucketManager), and
        # ProcessGroupNCCL's watchdog polls the completion events of earlier eager collectives; in "global" mode any such
        # call invalidates the capture (hipErrorStreamCaptureInvalidated; seen in round 4 with the first pin_memory() of a
        # buffer set).  Calls made by THIS thread and by autograd's backward thread on its behalf are still checked.
        mode = "thread_local"
        if self.reducer.active:
            # the device is idle (synchronize above): give the watchdog one polling period to retire the eager works;
            # collectives issued DURING capture are not handed to it
            import time
            time.sleep(0.3)
        err = None
        try:
            with torch.cuda.graph(graph, pool=self._graph_pool, capture_error_mode=mode):
                try:
                    # offsets restart; the salt word is read by the kernels at replay; the keep-bit generation of every
                    # attention site is captured as a side branch of the graph (ops.ATTN_BITS)
                    ops.RT.new_step(0, write_salt=False, plan_key=self._plan_key(task, sb))
                    loss = self._forward_backward(task, sb)
                    a.clip_and_step(None, self.betas, 1e-6, self.wd, self.grad_norm, grad_pre_scale=1.0 / self.world)
                except Exception as e:      # noqa: BLE001 -- still inside the capture: join the forked side streams so that
                    err = e                 # the capture can END (an unjoined capture stays open on this ROCm, see ops)
                    ops.join_captured_side_streams(extra=[self.reducer.stream])
        except Exception as e:      # noqa: BLE001 -- ending the capture failed as well
            err = err or e
        if err is not None:
            # a step that cannot be captured (e.g. a collective library that refuses stream capture) is not fatal: nothing
            # has executed, the trainer says so loudly and goes on eagerly
            e = err
            import warnings
            self.graph_error = f"{type(e).__name__}: {e}"[:400]
            self.use_graphs = False
            warnings.warn(f"hipGraph capture of the {task} step failed, continuing with eager steps: {self.graph_error}")
            ops.Branches.enabled = branches
            from . import lib
            lib.load().bevbert_hip_error_reset()     # the capture's error code must not surface in the next launch check
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the failed hipGraph capture could not be ended: the stream is still capturing and the "
                                   f"process cannot issue work any more ({self.graph_error})") from e
            torch.cuda.synchronize()
            # host state the aborted capture left behind points into its dead memory pool: deferred weight-gradient
            # closures, queued reduction records, collective work handles -- forget them, the step is redone eagerly
            ops.WgradStream.drop_pending()
            ops.ReduceQueue.drop_pending()
            self.reducer.drop_pending()
            ops.RT.new_step(self._step_seed(), plan_key=self._plan_key(task, sb))
            loss = self._forward_backward(task, sb)
            self.optimizer_step(lr=None)
            return loss
        ops.Branches.enabled = branches
        gs = GraphedStep(graph, loss)
        gs.owner = self
        sb.graph = gs
        graph.replay()                              # the capture recorded the step; this replay executes it
        return loss.clone()


# ---------------------------------------------------------------------------------------------------------------
# Drop-in for the reference's own loops (pretrain_src/utils/misc.py:64-77 wrap_model, train_r2r.py:247-313,
# map_nav_src/r2r/agent_base.py:122-123,174-217): a script that replaces
#     from utils.misc import wrap_model        ->   from vln_bevbert_amd.train import wrap_model
# keeps everything else -- build_optimizer / torch.optim over model.parameters(), loss.backward(),
# clip_grad_norm_(model.parameters(), ...), GradScaler, optimizer.step(), optimizer.zero_grad(), model.state_dict()
# (keys 'module.*' when wrapped, which the reference's saver strips: pretrain_src/utils/save.py:33-35).
class ArenaDataParallel(torch.nn.Module):
    """What DistributedDataParallel is to the reference (utils/misc.py:70), for a model whose gradients live in a
    ParamArena: wrap-time broadcast of rank 0's parameters and buffers, and one in-place all-reduce of the flat
    gradient arena (averaged) when a backward pass ends -- torch's DDP cannot do it, its reducer copies ``p.grad`` out
    of AccumulateGrad hooks that never fire here.  ``find_unused_parameters=True`` semantics come for free (unused
    parameters contribute zeros and keep ``.grad is None``)."""

    def __init__(self, module, process_group=None, compute_dtype=None):
        super().__init__()
        self.module = module
        arena = getattr(module, "arena", None)
        if arena is None:
            dev = next(module.parameters()).device
            arena = module.finalize(dev, compute_dtype or torch.float32)
        self.arena = arena
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
            arena.allreduce_group = process_group if process_group is not None else dist.group.WORLD
            dist.broadcast(arena.params, src=0, group=process_group)
            for b in module.buffers():
                dist.broadcast(b, src=0, group=process_group)
            arena.sync_shadow()

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def no_sync(self):
        """Gradient accumulation without communication (DDP.no_sync): the arena keeps summing locally."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old = self.arena.defer_allreduce
            self.arena.defer_allreduce = True
            try:
                yield
            finally:
                self.arena.defer_allreduce = old
        return ctx()


def wrap_model(model, device, local_rank, compute_dtype=None):
    """pretrain_src/utils/misc.py:64-77 with the same signature: move to the device, place the parameters in the arena
    (fp32 unless ``compute_dtype`` says bf16; left to the first forward -- which looks at autocast -- when None and
    single-process), wrap for data parallelism when ``local_rank != -1``."""
    model.to(device)
    if local_rank != -1:
        return ArenaDataParallel(model, compute_dtype=compute_dtype)
    if compute_dtype is not None and getattr(model, "arena", None) is None:
        model.finalize(device, compute_dtype)
    return model
