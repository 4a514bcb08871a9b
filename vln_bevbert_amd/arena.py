"""Flat parameter / gradient / optimiser-state arenas (memory laid out for one 288 GB HBM3E stack).

The reference keeps 567 separate parameter tensors, lets autograd allocate a gradient per tensor, copies them
into DDP buckets for NCCL, and steps a Python-loop AdamW over every tensor (pretrain_src/optim/adamw.py:53-112,
pretrain_src/utils/misc.py:64-77).  Here every parameter is a view into ONE fp32 buffer; gradients, Adam moments
and (in mixed precision) the bf16 compute copy are parallel buffers with the same offsets:

    params  fp32 [N]   masters            (N = 238.8 M + padding for the R2R model: 0.96 GB)
    grads   fp32 [N]   accumulation target of every backward kernel; the all-reduce runs on slices of it in place
    exp_avg, exp_avg_sq fp32 [N]
    shadow  bf16 [N]   refreshed by the AdamW kernel in the same pass that updates the masters

Tensors start on 1024-element boundaries so the optimiser kernel can use one flag byte per 1024-element chunk
(bit0 weight decay, bit1 "has had a gradient": the reference skips parameters whose .grad is None).  Parameters
listed in a *group* (packed QKV / KV projections) are laid out back to back so the group is one GEMM operand.
"""
import math

import torch

from . import lib
from .lib import ptr, stream


def call(name, *args):
    from . import ops          # traced C-ABI call (bench.py's kernel-timing pass)
    return ops.call(name, *args)

CHUNK = 1024
_ZERO_KERNEL = __import__("os").environ.get("BEVBERT_ZERO_KERNEL", "0") == "1"      # A/B: fill kernel instead of a memset command
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")   # optim/misc.py:14 (substring match)


def _round_up(n, m):
    return (n + m - 1) // m * m


class ParamArena:
    def __init__(self, module, device, compute_dtype=torch.float32, groups=()):
        """groups: iterable of lists of parameter names to lay out contiguously (same decay class)."""
        self.device = torch.device(device)
        self.compute_dtype = compute_dtype
        named = [(n, p) for n, p in module.named_parameters()]
        by_name = dict(named)
        in_group = {}
        for gi, g in enumerate(groups):
            for n in g:
                in_group[n] = gi
        layout, seen_groups = [], set()
        for n, p in named:
            if n in in_group:
                gi = in_group[n]
                if gi in seen_groups:
                    continue
                seen_groups.add(gi)
                layout.append(list(groups[gi]))
            else:
                layout.append([n])
        self.slices = {}
        off = 0
        seg_meta = []   # (start, end_padded, decay)
        for seg in layout:
            start = off
            decays = set()
            for n in seg:
                k = by_name[n].numel()
                self.slices[n] = (off, k)
                off += k
                decays.add(not any(nd in n for nd in NO_DECAY))
            assert len(decays) == 1, f"group {seg} mixes weight-decay classes"
            off = _round_up(off, CHUNK)
            seg_meta.append((start, off, decays.pop(), seg))
        self.numel = off
        self.n_params = sum(p.numel() for _, p in named)
        dev = self.device
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = None
        self.exp_avg_sq = None
        self.shadow = torch.zeros(off, dtype=compute_dtype, device=dev) if compute_dtype != torch.float32 else None
        self._flags_host = torch.zeros(off // CHUNK, dtype=torch.uint8)
        self._seg_of = {}
        for start, end, decay, seg in seg_meta:
            if decay:
                self._flags_host[start // CHUNK:end // CHUNK] |= 1
            for n in seg:
                self._seg_of[n] = (start // CHUNK, end // CHUNK)
        self.flags = self._flags_host.to(dev)
        self.chunk_steps = torch.zeros(off // CHUNK, dtype=torch.int32, device=dev)   # per-parameter state["step"]
        self._touched = set()
        self._flags_dirty = False
        # torch-API bridge (see publish_grads below)
        self.publish_grads = True
        self.allreduce_group = None        # set by train.ArenaDataParallel: gradients are averaged over it when published
        self.defer_allreduce = False       # ArenaDataParallel.no_sync()
        self._cb_queued = False
        self._fresh = []                   # parameters written during the running backward pass
        self._published = []               # parameters whose .grad currently aliases the arena
        self._param_versions = None
        self.step_count = 0
        self._scalars = torch.zeros(4, dtype=torch.float32, device=dev)       # [grad norm, clip multiplier, lr, -]
        self._partials = torch.zeros(1024, dtype=torch.float32, device=dev)
        # re-point every parameter at its arena view
        self.named = named
        for n, p in named:
            o, k = self.slices[n]
            view = self.params[o:o + k].view(p.shape)
            view.copy_(p.data.to(dev, torch.float32))
            p.data = view
            p.main_grad = self.grads[o:o + k].view(p.shape)
            p.compute = self.shadow[o:o + k].view(p.shape) if self.shadow is not None else p.data
            p.arena = self
            p.arena_name = n
        self._index = {n: i for i, (n, _) in enumerate(named)}
        self.sync_shadow()
        # the compute copy matches the masters as of now: a later write through the parameters (load_state_dict after
        # wrap_model, as map_nav_src/r2r/agent_base.py does; a torch optimiser) moves a version counter and the next
        # forward refreshes the copy (maybe_refresh_shadow) -- also when that write happens before the FIRST forward
        self._param_versions = sum(p._version for _, p in named)

    # ---- views over contiguous groups --------------------------------------------------------
    def packed(self, names, shape):
        """(compute view, fp32 grad view) over consecutive parameters ``names`` reshaped to ``shape``."""
        o0, _ = self.slices[names[0]]
        total = 0
        for n in names:
            o, k = self.slices[n]
            assert o == o0 + total, f"{names} are not contiguous in the arena"
            total += k
        assert total == math.prod(shape)
        src = self.shadow if self.shadow is not None else self.params
        return src[o0:o0 + total].view(shape), self.grads[o0:o0 + total].view(shape)

    # ---- bookkeeping -------------------------------------------------------------------------
    def touch(self, param):
        n = param.arena_name
        if n not in self._touched:
            self._touched.add(n)
            a, b = self._seg_of[n]
            self._flags_host[a:b] |= 2
            self._flags_dirty = True
        if self.publish_grads:
            self._note_backward_write(param)

    # ---- torch-API bridge: the reference's loops run unchanged ---------------------------------
    # train_r2r.py:263-313 and map_nav_src/r2r/agent_base.py:174-217 do
    #     loss.backward(); [scaler.unscale_(opt)]; clip_grad_norm_(model.parameters(), 5.0); optimizer.step(); optimizer.zero_grad()
    # on ``p.grad`` with a torch optimiser.  Backward kernels here write into ``grads`` (``p.main_grad``) behind autograd's
    # back, so with ``publish_grads`` (default; PretrainTrainer turns it off and drives the arena itself):
    #   * the first gradient write of a backward pass queues an autograd-engine callback; at the end of that pass it issues
    #     the deferred weight-gradient work, joins the side streams (``sync``), averages over ``allreduce_group`` if one is
    #     set, and points ``p.grad`` of every parameter written so far at its arena view -- parameters that never received
    #     a gradient keep ``.grad is None`` exactly as under the reference (its AdamW skips them, optim/adamw.py:66);
    #   * ``optimizer.zero_grad()`` (set_to_none, torch's default) leaves stale sums in the arena: the next forward sees
    #     ``.grad is None`` on a published parameter and zeroes the arena first (``maybe_lazy_zero``, called from the model's
    #     forward entry); with set_to_none=False torch zeroes the views in place and nothing is needed;
    #   * a torch optimiser updates the fp32 masters through ``p``: the bf16 compute copy is refreshed at the next
    #     forward when any parameter's version counter moved (``maybe_refresh_shadow``).
    def _note_backward_write(self, param):
        if not self._cb_queued:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._publish)
            except RuntimeError:           # not inside a backward pass (a test touching parameters by hand)
                return
            self._cb_queued = True
        self._fresh.append(param)

    def _publish(self):
        self._cb_queued = False
        self.sync()
        # end of a backward pass in a caller-owned loop (nobody calls arena.zero_grad there): every queued reduction has
        # been issued and the current stream waits for the side streams, so the scratch ring may start over
        if self.device.type == "cuda":
            from . import ops
            ops.RT.scratch.reset()
        fresh = self._fresh
        if self.allreduce_group is not None and not self.defer_allreduce:
            import torch.distributed as dist
            dist.all_reduce(self.grads, group=self.allreduce_group)
            self.grads.div_(dist.get_world_size(self.allreduce_group))
            # DistributedDataParallel(find_unused_parameters=True) hands the reduced gradient of a parameter to EVERY
            # rank when any rank used it; a rank that did not use it locally must publish it too, or its optimiser
            # skips the update the other ranks make and the replicas drift apart: exchange the touched set (MAX)
            used = torch.zeros(len(self.named), dtype=torch.int32, device=self.grads.device)
            if fresh:
                used[torch.tensor(sorted({self._index[p.arena_name] for p in fresh}), device=used.device)] = 1
            dist.all_reduce(used, op=dist.ReduceOp.MAX, group=self.allreduce_group)
            mine = {id(p) for p in fresh}
            fresh = list(fresh) + [self.named[i][1] for i in used.nonzero().flatten().tolist()
                                   if id(self.named[i][1]) not in mine]
        for p in fresh:
            if p.grad is None:
                p.grad = p.main_grad
                self._published.append(p)
        self._fresh.clear()

    def maybe_lazy_zero(self):
        """Zero the gradient arena if the caller dropped published gradients (zero_grad(set_to_none=True)) -- of ANY
        published parameter: an optimiser that owns only some of them clears only those."""
        if self._published and any(p.grad is None for p in self._published):
            for p in self._published:
                p.grad = None
            self._published.clear()
            self.zero_grad()

    def maybe_refresh_shadow(self):
        """Rebuild the bf16 compute copy if a torch-API optimiser (or load_state_dict, or a DDP-style broadcast) wrote
        the fp32 masters through the parameters since the last look."""
        if self.shadow is None or not self.publish_grads:
            return
        v = 0
        for _, p in self.named:
            v += p._version
        if v != self._param_versions:
            if self._param_versions is not None:
                self.sync_shadow()
            self._param_versions = v

    def sync_shadow(self):
        if self.shadow is not None:
            call("bevbert_cast_f32", ptr(self.params), ptr(self.shadow), self.numel, lib.dtype_code(self.compute_dtype),
                 stream())

    def sync(self):
        """Make the current stream wait for the side streams that carry work of this step (weight-gradient stream,
        model branches).  Backward kernels write parameter gradients straight into ``grads`` as a side effect autograd
        does not see, so whoever reads or overwrites the arena next (optimiser, all-reduce, zero_grad, a test) must
        order itself after them.  Streams without pending work are left alone: inside a graph capture a wait on a
        stream that is not part of the capture would be an error."""
        if self.device.type == "cuda":
            from . import ops
            ops.WgradStream.flush_all()
            cur = torch.cuda.current_stream(self.device)
            for side in ops.Branches.side_streams():
                if ops.Branches.enabled or side not in ops.WgradStream.streams or ops.WgradStream.dirty:
                    cur.wait_stream(side)
            ops.WgradStream.dirty = False
            ops.WgradStream.release()

    def backward_done(self):
        """Call after ``loss.backward()`` when an external loop / optimiser reads the gradients: issues the deferred
        weight-gradient and reduction launches and makes the current stream wait for them (== ``sync``)."""
        self.sync()

    def zero_grad(self, overlap=False):
        """Zero the gradient arena.  ``overlap=True`` (the trainer's step): the 0.96 GB fill goes to a side stream, ordered
        behind everything enqueued so far (the previous step's AdamW reads the gradients), and runs beside the forward
        pass -- which is bound by the matrix cores and leaves HBM idle; ``wait_zero()`` joins it before backward."""
        self.sync()
        if self.device.type == "cuda":
            from . import ops
            ops.RT.scratch.reset()        # partial-sum addresses repeat from step to step (ops.ReduceQueue caches on them)
            if overlap:
                if getattr(self, "_zero_stream", None) is None:
                    from .hwqueues import side_stream
                    self._zero_stream = side_stream(self.device)
                self._zero_stream.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self._zero_stream):
                    if _ZERO_KERNEL:
                        self.grads.zero_()
                    else:
                        call("bevbert_zero", ptr(self.grads), self.grads.numel() * 4, stream())
                self._zero_pending = True
                return
            call("bevbert_zero", ptr(self.grads), self.grads.numel() * 4, stream())
            return
        self.grads.zero_()

    def wait_zero(self):
        """Order the current stream behind a ``zero_grad(overlap=True)`` (no-op otherwise)."""
        if getattr(self, "_zero_pending", False):
            torch.cuda.current_stream(self.device).wait_stream(self._zero_stream)
            self._zero_pending = False

    def load_state_dict_into(self, module, sd, strict=True):
        out = module.load_state_dict(sd, strict=strict)     # copies in place into the arena views
        self.sync_shadow()
        return out

    # ---- optimiser ---------------------------------------------------------------------------
    def set_lr(self, lr):
        """Learning rate of the next optimiser step, kept in device memory (one 4-byte fill on the stream): the AdamW
        kernel reads it there, so a step captured in a hipGraph follows the schedule without being re-captured."""
        self._scalars[2:3].fill_(float(lr))
        self._lr_set = True

    def upload_flags(self):
        """Host -> device copy of the per-chunk flags when a parameter was touched for the first time (never inside a
        graph capture: the copy reads pageable host memory)."""
        if self._flags_dirty:
            self.flags.copy_(self._flags_host, non_blocking=True)
            self._flags_dirty = False

    def clip_and_step(self, lr=None, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, max_norm=5.0, grad_pre_scale=1.0):
        """clip_grad_norm_(max_norm) + AdamW.step (train_r2r.py:295-313) in three launches, no host sync.
        ``lr=None``: use the device-resident learning rate as set by ``set_lr`` (graph-captured steps)."""
        self.sync()
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.params)
            self.exp_avg_sq = torch.zeros_like(self.params)
        if lr is not None:
            self.set_lr(lr)
        elif not getattr(self, "_lr_set", False):
            raise RuntimeError("clip_and_step(lr=None) reads the device-resident learning rate: call set_lr() first "
                               "(the slot starts at 0, which would silently turn AdamW and its weight decay into a no-op)")
        if not torch.cuda.is_current_stream_capturing():
            self.upload_flags()
        self.step_count += 1
        call("bevbert_grad_norm_clip", ptr(self.grads), self.numel, float(grad_pre_scale),
             float(max_norm if max_norm is not None else -1.0), ptr(self._partials), ptr(self._scalars), stream())
        call("bevbert_adamw_step", ptr(self.params), ptr(self.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq),
             ptr(self.shadow), ptr(self.flags), ptr(self.chunk_steps), self.numel, self._scalars[1:].data_ptr(),
             self._scalars[2:].data_ptr(), 0.0, float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
             stream())

    def grad_norm(self):
        """L2 norm computed by the last clip_and_step (device scalar; reading it syncs)."""
        return self._scalars[0]
