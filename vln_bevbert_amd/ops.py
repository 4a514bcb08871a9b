"""Torch-facing wrappers over the C ABI (include/bevbert_hip.h) and their autograd Functions.

PyTorch here is plumbing: device memory, streams and the autograd tape.  The Linear layers stay on the vendor BLAS
(north star) but are issued to hipBLASLt directly through the C ABI (bevbert_gemm, cached per-shape plans).  Everything else on the hot path --
attention, bias/dropout/residual/LayerNorm, bias+GELU, the BEV splat, gmap aggregation, the optimiser -- is a
hand-written HIP kernel reached through ``lib.call``; none of them has a CPU or eager fallback.  The library GEMMs are
the one exception: a problem hipBLASLt's direct path cannot take (no algorithm for the layout, the plan budget is
spent, BEVBERT_LT_GEMM=0) is handed to torch's GEMM -- the same library underneath, 4x the host cost per call -- and
every such problem is reported once through ``warnings`` (``ops.GEMM_FALLBACKS`` counts them), never silently.

Gradient sinks: a parameter that lives in a ParamArena (arena.py) carries ``main_grad`` (fp32 view of the flat
gradient arena).  Backward kernels accumulate straight into that view and the Function returns ``None`` for the
parameter, so there are no per-parameter AccumulateGrad kernels, no bucket copies for the all-reduce, and the
optimiser sees one flat buffer.  Plain tensors (unit tests) get ordinary returned gradients.
"""
import math
import os as _os

import torch
import torch.nn.functional as F

from . import lib
from .lib import call as _raw_call
from .lib import dtype_code, ptr, stream

HEAD_DIM = 64


# ----------------------------------------------------------------------------- runtime state
def _hash32(x):
    """common.h bb_hash32 ("lowbias32") on the host."""
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


class _Runtime:
    """Dropout stream + scratch buffers.

    A dropout site's mask is a pure function of (seed, offset, step salt, element index).  ``seed`` is fixed for the
    life of the process and ``offset`` advances by the element count of every dropout site, so each site draws from a
    disjoint counter range -- both are launch ARGUMENTS and freeze into a captured hipGraph.  What changes from step to
    step is the salt: one 32-bit word in device memory (registered with the library through bevbert_set_step_salt)
    that ``new_step`` rewrites with a 4-byte fill on the stream; a replayed graph therefore draws fresh masks."""

    SEED = 0x5EED

    def __init__(self):
        self.seed = self.SEED
        self.offset = 0
        self.attn_impl = 0      # 0 auto, 1 exact kernels, 2 MFMA kernels
        # bf16 mode with an fp32 RESIDUAL STREAM (finalize(..., residual=torch.float32)): the post-norm blocks keep their
        # LayerNorm outputs and residual sums in fp32 next to the bf16 copy the GEMMs read -- torch.autocast's arithmetic
        self.res32 = False
        self._ws = {}
        self._ws_ptr = {}
        self._salt = None

    def next_offset(self, n):
        off = self.offset
        self.offset += int(n)
        return off

    def salt_word(self, step_seed):
        step_seed = int(step_seed) & 0xFFFFFFFFFFFFFFFF
        v = _hash32(_hash32(step_seed & 0xFFFFFFFF) ^ (step_seed >> 32))
        return v - (1 << 32) if v >= (1 << 31) else v          # as int32 bit pattern

    def new_step(self, step_seed, write_salt=True, plan_key=None):
        """Start the dropout stream of a step: offsets restart at 0 and the device salt becomes hash(step_seed).
        ``write_salt=False`` only restarts the offsets (graph replay: the caller has already written the salt).
        ``plan_key``: identity of the step's shape (task + batch signature) for ``ATTN_BITS`` -- a step whose sequence of
        attention-dropout sites is known from an earlier step with the same key generates all its keep-bit workspaces
        up front on a side stream."""
        self.offset = 0
        if write_salt and torch.cuda.is_available():
            if self._salt is None:
                self._salt = torch.zeros(1, dtype=torch.int32, device="cuda")
                lib.load().bevbert_set_step_salt(self._salt.data_ptr())
            self._salt.fill_(self.salt_word(step_seed))
        ATTN_BITS.begin(plan_key)

    def workspace(self, device, nfloats):
        # one scratch buffer per (device, stream): branches of the model run concurrently on separate streams
        key = stream()
        buf = self._ws.get(key)
        if buf is None or buf.numel() < nfloats:
            buf = torch.empty(max(int(nfloats), 512 * 3 * 3072), dtype=torch.float32, device=device)
            self._ws[key] = buf
        return buf


    def gemm_workspace(self, device, stream_handle):
        """device pointer of the hipBLASLt workspace of a stream"""
        key = ("lt", stream_handle)
        p = self._ws_ptr.get(key)
        if p is None:
            buf = torch.empty(_LT_WS_BYTES, dtype=torch.uint8, device=device)
            self._ws[key] = buf
            p = self._ws_ptr[key] = buf.data_ptr()
        return p


# hipBLASLt workspace per launching stream: solutions that need more (split-K / stream-K partial tiles of the wide problems)
# are not candidates.  BEVBERT_LT_WS_MB raises it (a choice table made with a larger workspace needs it at run time too).
_LT_WS_BYTES = int(_os.environ.get("BEVBERT_LT_WS_MB", "32")) << 20


class _AttnBitsPlanner:
    """Keep-bit workspaces of a step's attention-dropout sites, generated ahead of the forward on a side stream.

    The mask of a site is a pure function of (seed, offset, step salt, element index) and the (shape, offset) sequence
    of a step repeats from step to step for the same task and batch shapes.  The first step with a given ``plan_key``
    records the sequence (its sites generate their bits inline, in front of their forward kernel); every later step
    with that key launches ALL its bevbert_attn_drop_bits calls when the step starts, on one side stream, into buffers
    that belong to the plan -- the hashing (one 32-bit mix per element pair: ~80 us of pure VALU work per 441 x 441
    site at batch 64) then runs beside the library GEMMs of the text and panorama encoders instead of in front of
    every attention kernel, and each attention forward only waits for its site's event.  Works eagerly and inside a
    captured step (the side stream is forked from and joined to the capturing stream).  BEVBERT_ATTN_BITS_AHEAD=0
    turns it off (A/B measurements)."""

    def __init__(self):
        self.enabled = _os.environ.get("BEVBERT_ATTN_BITS_AHEAD", "1") == "1"
        self.plans = {}          # key -> {"sites": [sig], "bufs": [tensor]}
        self.key = None
        self.seen = []
        self.ready = None        # [(sig, bits, event)] of the running step
        self.idx = 0
        self.stream = None
        self.hits = self.misses = 0

    def begin(self, key):
        if self.key is not None and self.seen and self.key not in self.plans:
            self.plans[self.key] = {"sites": list(self.seen), "bufs": [None] * len(self.seen)}
        self.key, self.seen, self.idx, self.ready = key, [], 0, None
        if not self.enabled or key is None or not torch.cuda.is_available():
            return
        plan = self.plans.get(key)
        if plan is None:
            return
        dev = torch.cuda.current_device()
        if self.stream is None:
            self.stream = torch.cuda.Stream(dev)
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)             # after the salt fill (and, in a capture, part of the captured graph)
        ready = []
        with torch.cuda.stream(self.stream):
            for i, sig in enumerate(plan["sites"]):
                B, nh, Lq, Lk, p, off = sig
                if plan["bufs"][i] is None:
                    plan["bufs"][i] = torch.empty(_drop_bits_words(B, nh, Lq, Lk), dtype=torch.int64, device="cuda")
                bits = plan["bufs"][i]
                call("bevbert_attn_drop_bits", ptr(bits), B, nh, Lq, Lk, float(p), self._seed(), int(off), stream())
                ev = torch.cuda.Event()
                ev.record(self.stream)
                ready.append((sig, bits, ev))
        self.ready = ready

    @staticmethod
    def _seed():
        return RT.seed

    def get(self, B, nh, Lq, Lk, p, off, device):
        """(workspace, bits_ready) for the next attention-dropout site of the running step."""
        sig = (B, nh, Lq, Lk, float(p), int(off))
        i = self.idx
        self.idx += 1
        self.seen.append(sig)
        if self.ready is not None and i < len(self.ready) and self.ready[i][0] == sig:
            _, bits, ev = self.ready[i]
            torch.cuda.current_stream().wait_event(ev)
            self.hits += 1
            return bits, 1
        self.misses += 1
        return torch.empty(_drop_bits_words(B, nh, Lq, Lk), dtype=torch.int64, device=device), 0


RT = _Runtime()
ATTN_BITS = _AttnBitsPlanner()


class Branches:
    """Two-stream execution of independent model branches (MI355X: kernels of the small text / panorama / global-map
    branches do not fill 256 CUs; overlapping them with each other and with the BEV branch does).

    ``fork()`` makes the side stream wait for everything enqueued so far on the current stream; code inside
    ``with br.side():`` is enqueued on the side stream; ``join(*tensors)`` makes the current stream wait for the side
    stream and tells the caching allocator that the given side-allocated tensors are now used on the current stream.
    Autograd replays every backward op on the stream its forward ran on and inserts the cross-stream waits itself."""

    # Issued eagerly the extra fork / join events cost ~3 ms of host time per step at batch 64 and the step becomes
    # host-bound (round 2, one call: 19.27 with vs 19.35 ms without), so the side stream is OFF for eager steps
    # (BEVBERT_STREAMS=1 turns it on); CAPTURED steps turn it on themselves (train.PretrainTrainer.graph_branches),
    # where the edges cost nothing on the host: 18.58 vs 19.32 ms/step.
    enabled = _os.environ.get("BEVBERT_STREAMS", "0") == "1"
    _streams = {}

    def __init__(self, device):
        self.device = device
        self.on = Branches.enabled and device.type == "cuda"
        if self.on:
            key = device.index
            if key not in Branches._streams:
                Branches._streams[key] = torch.cuda.Stream(device)
            self.stream = Branches._streams[key]
            self.main = torch.cuda.current_stream(device)

    @classmethod
    def side_streams(cls):
        out = []
        for st in cls._streams.values():
            if all(st is not o for o in out):
                out.append(st)
        return out

    def fork(self, *tensors):
        if self.on:
            self.stream.wait_stream(self.main)
            for t in tensors:
                if t is not None:
                    t.record_stream(self.stream)

    def side(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.on else contextlib.nullcontext()

    def join(self, *tensors):
        if self.on:
            self.main.wait_stream(self.stream)
            for t in tensors:
                if t is not None:
                    t.record_stream(self.main)
def join_captured_side_streams(extra=()):
    """Recovery step of a FAILED stream capture: make the capturing (current) stream wait for every side stream that was
    forked into the capture (keep-bit stream, model-branch stream, weight-gradient streams, the reducer's stream).
    hipStreamEndCapture refuses to end a capture with unjoined forks (hipErrorStreamCaptureUnjoined) and -- on ROCm 7.2 --
    then leaves the origin stream IN capture mode, so that every later launch of the process fails; with the forks
    joined the capture ends normally and its graph is simply dropped."""
    if not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
        return 0
    cur = torch.cuda.current_stream()
    seen, n = {cur.cuda_stream}, 0
    cands = [ATTN_BITS.stream] + list(Branches._streams.values()) + list(WgradStream.streams) + list(extra)
    for st in cands:
        if st is None or st.cuda_stream in seen:
            continue
        seen.add(st.cuda_stream)
        with torch.cuda.stream(st):
            capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            cur.wait_stream(st)
            n += 1
    return n


TRACE = None     # dict name -> [(start_event, end_event)] while bench.py's kernel-timing pass is active


class WgradStream:
    """Weight-gradient work (split-K GEMM, partial-sum accumulate, bias column sums) on its own HIP stream.

    Backward's critical path is the dgrad chain; the dW / db kernels only feed the optimiser.  At the 5 120-token shapes
    of the text branch neither kind fills 256 CUs (<= 240 workgroups per GEMM), so running them concurrently shortens
    the step.  Events are expensive on the host (~20 us for record + wait), so the work is DEFERRED: backward nodes
    ``submit`` closures, and every ``BATCH`` submissions from one producing stream are flushed behind a single event
    (events come from a small reusable pool).  Operands are kept alive until ``release()`` (ParamArena.sync, which
    also joins the stream) instead of being tracked by the caching allocator -- with 288 GB of HBM the extra lifetime
    of one backward's activation gradients is free."""

    # Round-1 in-run A/B at batch 64: 24.9 ms/step with neither side stream, 23.3 with this stream alone (22.5 after the
    # LayerNorm / GELU reduction tails moved here too), 22.8 with the branch stream alone, 23.6-24.5 with both on one
    # shared stream.
    enabled = _os.environ.get("BEVBERT_WGRAD_STREAM", "1") == "1"
    BATCH = int(_os.environ.get("BEVBERT_WGRAD_BATCH", "6"))
    DEFER_FINALIZE = _os.environ.get("BEVBERT_DEFER_FINALIZE", "1") == "1"      # A/B knob for the split reductions
    # own stream even next to ops.Branches (three streams).  Round 1 shared one side stream because three streams had
    # stalled at batch 64: stream-K library GEMMs spinning on each other across streams (the
    # package sets TENSILE_STREAMK_DATA_PARALLEL=1); three to four streams measure fastest
    OWN_STREAM = _os.environ.get("BEVBERT_WGRAD_OWN_STREAM", "1") == "1"
    stream = None
    streams = []
    # BEVBERT_WGRAD_STREAMS=2: batches of deferred work alternate between two streams (independent weight-gradient GEMMs
    # of different layers next to each other).  Only sensible inside captured steps, where the extra events are free.
    NSTREAMS = int(_os.environ.get("BEVBERT_WGRAD_STREAMS", "2"))
    _rr = 0
    _target = None       # stream the deferred closures are being issued on right now
    dirty = False        # work has been enqueued on the stream since the last join (ParamArena.sync)
    _keep = []
    _pending = {}        # producing stream handle -> (torch stream, [closures])
    _events = []
    _next_event = 0

    @classmethod
    def active(cls, device):
        return cls.enabled and TRACE is None and device.type == "cuda"

    @classmethod
    def submit(cls, device, fn, *keep):
        """Run ``fn`` (C-ABI launches only) on the weight-gradient stream once its operands -- everything enqueued so
        far on the current stream -- are ready.  ``keep``: tensors ``fn`` reads or writes, plus the ORIGINAL gradient
        tensor autograd handed to the node: holding that object keeps its use count above one, which stops the
        engine from accumulating another gradient into its storage in place while the deferred read is pending."""
        if not cls.active(device):
            fn()
            return
        h = lib.stream()
        slot = cls._pending.get(h)
        if slot is None:
            slot = cls._pending[h] = (torch.cuda.current_stream(device), [])
        slot[1].append(fn)
        cls._keep.extend(keep)
        if len(slot[1]) >= cls.BATCH:
            cls._flush(slot)

    @classmethod
    def _flush(cls, slot, final=False):
        producer, fns = slot
        if not fns:
            return
        if cls.stream is None:
            shared = Branches._streams.get(producer.device.index) if Branches.enabled and not cls.OWN_STREAM else None
            cls.stream = shared if shared is not None else torch.cuda.Stream(producer.device)
            Branches._streams["wgrad"] = cls.stream       # joined by ParamArena.sync / GradReducer like the branches
            cls._events = [torch.cuda.Event() for _ in range(64)]
            cls.streams = [cls.stream]
            for i in range(1, cls.NSTREAMS):              # further streams: batches of deferred work go round robin
                st = torch.cuda.Stream(producer.device)
                Branches._streams[f"wgrad{i}"] = st
                cls.streams.append(st)
        if final or len(cls.streams) == 1:
            target = cls.stream
        else:
            cls._rr += 1
            target = cls.streams[cls._rr % len(cls.streams)]
        if producer.cuda_stream != target.cuda_stream:              # same stream: already in order
            ev = cls._events[cls._next_event % len(cls._events)]
            cls._next_event += 1
            ev.record(producer)
            target.wait_event(ev)
        if final:        # the batched reductions read what every weight-gradient stream produced, and the first
            for st in cls.streams[1:]:      # stages of the queued column reductions, wherever those were launched
                target.wait_stream(st)
            ReduceQueue.wait_producers(target)
        lib.set_stream_override(target.cuda_stream)
        cls._target = target
        cls.dirty = True
        try:
            for fn in fns:
                fn()
        finally:
            lib.set_stream_override(None)
            cls._target = None
            fns.clear()

    @classmethod
    def flush_all(cls):
        """Issue everything deferred so far, then -- in one launch -- the pending second stages of the column
        reductions (ReduceQueue): their first stages were enqueued on the producing streams before this call."""
        if ReduceQueue.jobs or ReduceQueue.accum_jobs:
            dev = torch.device("cuda", torch.cuda.current_device())
            if cls.active(dev):
                h = lib.stream()
                slot = cls._pending.get(h)
                if slot is None:
                    slot = cls._pending[h] = (torch.cuda.current_stream(dev), [])
                slot[1].append(lambda: ReduceQueue.flush(dev))
                mine = slot
                for other in cls._pending.values():
                    if other is not mine:
                        cls._flush(other)
                cls._flush(mine, final=True)
                return
            ReduceQueue.wait_producers(torch.cuda.current_stream(dev))
            ReduceQueue.flush(dev)
        for slot in cls._pending.values():
            cls._flush(slot)

    @classmethod
    def release(cls):
        cls._keep.clear()

    @classmethod
    def drop_pending(cls):
        """Forget deferred closures of an aborted step (failed graph capture) instead of running them later."""
        for _, fns in cls._pending.values():
            fns.clear()
        cls._keep.clear()


class ScratchRing:
    """Bump allocator over device buffers for the short-lived fp32 partial sums of the two-stage column reductions.
    ``reset()`` at the start of every step (and at the end of every backward pass: arena._publish) makes the addresses
    REPEAT from step to step (same task -> same sequence of allocations), which is what lets ReduceQueue keep its task
    tables -- they hold raw pointers -- in device memory instead of rebuilding and re-uploading them every step.

    The first buffer grows to what a training step needs, up to ``nbytes`` (BEVBERT_SCRATCH_MB).  A backward pass that
    queues more than that before its reductions are issued -- a fine-tune rollout differentiates through all its
    navigation steps at once (map_nav_src/r2r/agent.py:339-420) -- continues in further buffers of the same size, kept and
    reused in the same order by the following passes (up to BEVBERT_SCRATCH_MAX_MB in total)."""

    INITIAL = 256 << 20

    def __init__(self, nbytes=1 << 30, max_total=64 << 30):
        self.nbytes = nbytes            # size of one buffer (BEVBERT_SCRATCH_MB)
        self.max_total = max(max_total, nbytes)
        self.size = 0                   # bytes of the current buffer: the first one GROWS to what a step needs
        self.buf = None
        self.base = 0
        self.off = 0
        self.ci = 0                     # index of the current buffer
        self._chunks = []               # [buf, base, size] per buffer; [0] is the growing one
        self._old = []                  # outgrown buffers are NEVER freed: queued records of the running step and steps
        #                                 captured before the growth (another task's hipGraph) keep pointing into them

    def reset(self):
        self.off = self.ci = 0
        if self._chunks:
            self.buf, self.base, self.size = self._chunks[0]

    def total_bytes(self):
        return sum(c[2] for c in self._chunks)

    def alloc(self, nbytes, device):
        n = (int(nbytes) + 255) & ~255
        if n > self.nbytes:
            raise lib.BevBertHipError(f"scratch ring: {n} bytes requested, buffer size {self.nbytes} (BEVBERT_SCRATCH_MB)")
        if self.off + n > self.size:
            if self.ci == 0 and self.size < self.nbytes:
                # grow (warm-up steps): a new, larger buffer; from the next reset on every allocation of the step lives
                # in it, so the addresses repeat again -- which the cached task tables and captured steps rely on
                if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
                    raise lib.BevBertHipError("scratch ring would have to grow during graph capture: run one more "
                                              "eager step first, or start larger (BEVBERT_SCRATCH_INITIAL_MB)")
                new = min(self.nbytes, max(2 * self.size, self.off + n, self.INITIAL))
                if self.buf is not None:
                    self._old.append(self.buf)
                self.buf = torch.empty(new, dtype=torch.uint8, device=device)
                self.base, self.size, self.off = self.buf.data_ptr(), new, 0
                self._chunks[:1] = [[self.buf, self.base, self.size]]
            elif self.ci + 1 < len(self._chunks):
                self._enter(self.ci + 1)                # a buffer an earlier pass of this length left behind
            elif not (ReduceQueue.jobs or ReduceQueue.accum_jobs):
                self._enter(0)                          # nothing queued points into the buffers: start over
            else:
                # reductions of this pass are still queued: their partial sums must stay where they are
                if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
                    raise lib.BevBertHipError("scratch ring would need another buffer during graph capture: run one more "
                                              "eager step first, or raise BEVBERT_SCRATCH_MB")
                if self.total_bytes() + self.nbytes > self.max_total:
                    raise lib.BevBertHipError(
                        f"one backward pass queued more than {self.total_bytes() >> 20} MB of partial sums for its column "
                        "reductions: raise BEVBERT_SCRATCH_MAX_MB if that is intended")
                buf = torch.empty(self.nbytes, dtype=torch.uint8, device=device)
                self._chunks.append([buf, buf.data_ptr(), self.nbytes])
                self._enter(len(self._chunks) - 1)
        p = self.base + self.off
        self.off += n
        return p

    def _enter(self, ci):
        self.ci, self.off = ci, 0
        self.buf, self.base, self.size = self._chunks[ci]

    def tensor(self, shape, dtype, device):
        """A tensor view of freshly bumped ring memory (for operands that go through tensor-typed call paths)."""
        nbytes = math.prod(shape) * torch.empty((), dtype=dtype).element_size()
        p = self.alloc(nbytes, device)
        o = p - self.base
        return self.buf[o:o + nbytes].view(dtype).view(shape)


SCRATCH = ScratchRing(int(_os.environ.get("BEVBERT_SCRATCH_MB", "6144")) << 20,       # ~2.5 GB / step at batch 64
                      int(_os.environ.get("BEVBERT_SCRATCH_MAX_MB", "65536")) << 20)
ScratchRing.INITIAL = int(_os.environ.get("BEVBERT_SCRATCH_INITIAL_MB", "256")) << 20


class ReduceQueue:
    """Pending second stages of the column reductions of a backward pass (LayerNorm gamma / beta / bias, GELU bias,
    projection biases).  Issued one by one they are ~110 launches of 6-8 us per training step -- a tenth of the step's
    launches and ~1 ms of GPU time spent on kernels of a few dozen workgroups.  Here the first stages leave their
    per-block partial sums in the scratch ring, the queue collects (partials, outputs) records, and ``flush`` runs them
    all in ONE launch (bevbert_multi_finalize) on the weight-gradient stream.  The task table of a given record list
    is built once and kept on the device (the records hold raw pointers; ScratchRing makes them repeat)."""

    jobs = []
    accum_jobs = []
    table_bytes = {}     # device address of a task table -> algorithmic bytes of one launch over it (bench.py's rooflines)
    producers = {}       # raw stream handle -> torch stream on which first stages of pending records were launched
    _tables = {}
    _accum_tables = {}
    _dtype = None
    _adtype = None

    @classmethod
    def _note_producer(cls):
        """The first stage of the record being added was launched on the stream C-ABI launches go to right now (the
        autograd stream, a branch stream, or the weight-gradient stream a deferred closure runs on).  The second
        stage must wait for every such stream, whatever else happens to order them (a side-stream producer
        whose deferred-work slot is empty would otherwise leave no dependency edge)."""
        st = WgradStream._target
        if st is None:
            if not torch.cuda.is_available():
                return
            st = torch.cuda.current_stream()
        cls.producers[st.cuda_stream] = st

    @classmethod
    def wait_producers(cls, consumer):
        """Make ``consumer`` (a torch stream) wait for everything enqueued so far on the producing streams."""
        for h, st in cls.producers.items():
            if h != consumer.cuda_stream:
                consumer.wait_stream(st)
        cls.producers = {}

    @classmethod
    def drop_pending(cls):
        """Forget the records of an aborted step (failed graph capture): they reference memory of a dead capture."""
        cls.jobs, cls.accum_jobs, cls.producers = [], [], {}

    @classmethod
    def add_accum(cls, partials_ptr, sink_ptr, S, n, dtype):
        """sink[0:n] += sum of the S partial slices at partials_ptr (split-K weight-gradient products)."""
        cls.accum_jobs.append((partials_ptr, sink_ptr, S, n, dtype))
        cls._note_producer()

    @classmethod
    def _build_accum(cls, jobs, device):
        import numpy as np
        if cls._adtype is None:
            cls._adtype = np.dtype([("partials", "<u8"), ("sink", "<u8"), ("n4_total", "<u8"), ("off4", "<u4"),
                                    ("n4", "<u4"), ("S", "<i4"), ("dtype", "<i4")])
        parts = []
        for partials_ptr, sink_ptr, S, n, dt in jobs:
            n4 = n // 4
            off = np.arange(0, n4, 4096, dtype=np.int64)
            t = np.zeros(len(off), dtype=cls._adtype)
            t["partials"] = partials_ptr
            t["sink"] = sink_ptr + off * 16
            t["n4_total"] = n4
            t["off4"] = off
            t["n4"] = np.minimum(4096, n4 - off)
            t["S"] = S
            t["dtype"] = dt
            parts.append(t)
        table = np.concatenate(parts) if parts else np.zeros(0, dtype=cls._adtype)
        dev = torch.from_numpy(table.view(np.uint8).copy()).to(device)
        # every partial slice read once (its own dtype), the sink read and written once (fp32)
        cls.table_bytes[dev.data_ptr()] = int(sum(S * n * (4 if dt == lib.F32 else 2) + 8 * n for _, _, S, n, dt in jobs))
        return dev, len(table)

    @classmethod
    def add(cls, partials_ptr, nblocks, nwhich, C, outs, accumulate=1):
        cls.jobs.append((partials_ptr, nblocks, nwhich, C, outs[0] or 0, outs[1] or 0, outs[2] or 0, accumulate))
        cls._note_producer()

    @classmethod
    def _build(cls, jobs, device):
        import numpy as np
        if cls._dtype is None:
            cls._dtype = np.dtype([("partials", "<u8"), ("out", "<u8"), ("nblocks", "<i4"), ("row_stride", "<i4"),
                                   ("col0", "<i4"), ("ncols", "<i4"), ("accumulate", "<i4"), ("pad", "<i4")])
        parts = []
        for partials_ptr, nblocks, nwhich, C, o0, o1, o2, acc in jobs:
            ntile = (C + 63) // 64
            tiles = np.arange(ntile, dtype=np.int64)
            for which, out in enumerate((o0, o1, o2)[:nwhich]):
                if not out:
                    continue
                t = np.zeros(ntile, dtype=cls._dtype)
                t["partials"] = partials_ptr
                t["out"] = out + tiles * 256
                t["nblocks"] = nblocks
                t["row_stride"] = nwhich * C
                t["col0"] = which * C + tiles * 64
                t["ncols"] = np.minimum(64, C - tiles * 64)
                t["accumulate"] = acc
                parts.append(t)
        table = np.concatenate(parts) if parts else np.zeros(0, dtype=cls._dtype)
        dev = torch.from_numpy(table.view(np.uint8).copy()).to(device)
        # fp32 partial sums read once, outputs written (and read when accumulating)
        cls.table_bytes[dev.data_ptr()] = int(4 * (table["nblocks"].astype(np.int64) * table["ncols"]).sum()
                                              + 4 * (table["ncols"] * (1 + (table["accumulate"] != 0))).sum()) if len(table) else 0
        return dev, len(table)

    @classmethod
    def flush(cls, device):
        """Launch the pending second stages (on the stream C-ABI launches currently go to).  Records that accumulate
        into the SAME output vector (a parameter used twice in one backward: REVERIE's object tokens share
        img_linear / img_layer_norm with the views) must not run concurrently: they go into successive launches."""
        if cls.accum_jobs:
            akey = tuple(cls.accum_jobs)
            cls.accum_jobs = []
            aent = cls._accum_tables.get(akey)
            if aent is None:
                if torch.cuda.is_current_stream_capturing():
                    raise lib.BevBertHipError("accumulate task table missing during graph capture (warm-up steps build it)")
                if len(cls._accum_tables) > 256:
                    cls._accum_tables.clear()
                rounds, seen = [[]], [set()]
                for job in akey:                      # a weight used twice in one backward: successive launches
                    r = 0
                    while job[1] in seen[r]:
                        r += 1
                        if r == len(rounds):
                            rounds.append([])
                            seen.append(set())
                    rounds[r].append(job)
                    seen[r].add(job[1])
                aent = cls._accum_tables[akey] = [cls._build_accum(tuple(r), device) for r in rounds]
            for table, n in aent:
                call("bevbert_multi_accum", table.data_ptr(), n, stream())
        if not cls.jobs:
            return
        key = tuple(cls.jobs)
        cls.jobs = []
        ent = cls._tables.get(key)
        if ent is None:
            if torch.cuda.is_current_stream_capturing():
                raise lib.BevBertHipError("reduction task table missing during graph capture (warm-up steps build it)")
            if len(cls._tables) > 256:
                cls._tables.clear()
            rounds, seen = [[]], [set()]
            for job in key:
                outs = {o for o in job[4:7] if o}
                r = 0
                while outs & seen[r]:
                    r += 1
                    if r == len(rounds):
                        rounds.append([])
                        seen.append(set())
                rounds[r].append(job)
                seen[r] |= outs
            ent = cls._tables[key] = [cls._build(tuple(r), device) for r in rounds]
        for table, n in ent:
            call("bevbert_multi_finalize", table.data_ptr(), n, stream())


def call(name, *args):
    """C-ABI call; when TRACE is armed, bracket the launch with HIP events on the launching stream."""
    if TRACE is None:
        return _raw_call(name, *args)
    key = name
    if name == "bevbert_attn_fwd":
        key = f"{name}[Lq={args[10]},Lk={args[11]}]"
    elif name == "bevbert_attn_bwd":
        key = f"{name}[Lq={args[16]},Lk={args[17]}]"
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    _raw_call(name, *args)
    e.record()
    TRACE.setdefault(key, []).append((s, e, args))


def _gemm(kind, fn, m, n, k):
    """Library GEMM (hipBLASLt via torch); when TRACE is armed, time it with HIP events keyed by its shape."""
    if TRACE is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = fn()
    e.record()
    TRACE.setdefault(f"gemm:{kind}[M={m},N={n},K={k}]", []).append((s, e, (m, n, k)))
    return out


# Library GEMMs go straight to hipBLASLt through the C ABI (bevbert_gemm): ~7 us of host time per call instead of the
# ~28 us of torch.mm / F.linear dispatch -- the training step is host-bound at batch 64 (bench.py reports both clocks).
# BEVBERT_LT_GEMM=0 routes them through torch instead (same library underneath); A/B knob.
_LT_ENABLED = _os.environ.get("BEVBERT_LT_GEMM", "1") == "1"
_LT_AUTOTUNE = int(_os.environ.get("BEVBERT_LT_AUTOTUNE", "32"))
_LT_UNSUPPORTED = set()
# Every new problem costs one timing pass (32 candidates x 10 launches + a sync) the first time it is seen.  The R2R step
# has ~100 problems; real batches add data-dependent row counts (masked tokens, selected cells, trajectory lengths).
# Past this many plans new problems stay on torch's own GEMM path (the library's single heuristic pick, no timing pass)
# so that an unbounded variety of shapes cannot turn into an unbounded number of stalls.
_LT_PLAN_BUDGET = int(_os.environ.get("BEVBERT_LT_PLAN_BUDGET", "8192"))


_LT_PLANS = {}
GEMM_FALLBACKS = {}      # (kind, M, N, K) -> calls that went through torch's GEMM instead of the direct hipBLASLt path


def _warn_fallback(kind, M, N, K, why="no direct hipBLASLt plan"):
    key = (kind, int(M), int(N), int(K))
    n = GEMM_FALLBACKS.get(key, 0)
    GEMM_FALLBACKS[key] = n + 1
    if n == 0:
        import warnings
        warnings.warn(f"vln_bevbert_amd: {kind} GEMM M={M} N={N} K={K} runs through torch ({why}); same library, "
                      f"~4x the host cost per call", RuntimeWarning, stacklevel=3)
GEMM_TUNING_FILE = _os.environ.get("BEVBERT_GEMM_TABLE",
                                   _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "gemm_tuning.txt"))
_tuning_loaded = False


def load_gemm_tuning_table(path=None):
    """Import the shipped hipBLASLt choice table (bevbert_gemm_tuning_import); returns the number of rows (0 when the
    file is missing or was made with another library version -- the plans then time their candidates on first use)."""
    global _tuning_loaded
    _tuning_loaded = True
    path = path or GEMM_TUNING_FILE
    if not _os.path.exists(path):
        return 0
    with open(path, "rb") as f:
        n = lib.load().bevbert_gemm_tuning_import(f.read())
    return max(n, 0)


def save_gemm_tuning_table(path):
    """Write the choices of every plan tuned so far (plus the imported rows) for later runs."""
    l = lib.load()
    need = l.bevbert_gemm_tuning_export(None, 0)
    import ctypes
    buf = ctypes.create_string_buffer(need)
    l.bevbert_gemm_tuning_export(buf, need)
    with open(path, "wb") as f:
        f.write(buf.value)
    return buf.value.count(b"\n") - 1


def _lt_gemm(a, b, out, bias, M, N, K, opA, opB, lda, ldb, ldc, batch=1, sa=0, sb=0, sc=0, accumulate=0, c_in=None):
    """out (+)= op(a) . op(b) (+ bias) on hipBLASLt via the C ABI; False if the library has no kernel for the shape.
    ``c_in`` (with accumulate=1): out = product + c_in, the addend being a separate buffer of out's layout."""
    key = (M, N, K, opA, opB, lda, ldb, ldc, batch, a.dtype, out.dtype, None if bias is None else bias.dtype,
           accumulate)
    plan = _LT_PLANS.get(key)
    if plan is None:
        if len(_LT_PLANS) >= _LT_PLAN_BUDGET:
            return False
        if not _tuning_loaded:
            load_gemm_tuning_table()
        plan = lib.load().bevbert_gemm_plan(M, N, K, opA, opB, lda, ldb, ldc, batch, sa, sb, sc, dtype_code(a),
                                            dtype_code(out), -1 if bias is None else dtype_code(bias), accumulate,
                                            _LT_WS_BYTES, _LT_AUTOTUNE)
        if plan < 0:
            raise lib.BevBertHipError(f"bevbert_gemm_plan failed ({plan}): {lib.load().bevbert_last_error().decode()}")
        _LT_PLANS[key] = plan
    if plan in _LT_UNSUPPORTED:
        return False
    st = stream()
    if c_in is not None:
        rc = lib.load().bevbert_gemm_run_add(plan, a.data_ptr(), b.data_ptr(), c_in.data_ptr(), out.data_ptr(),
                                             None if bias is None else bias.data_ptr(),
                                             RT.gemm_workspace(a.device, st), _LT_WS_BYTES, st)
    else:
        rc = _LT_RUN(plan, a.data_ptr(), b.data_ptr(), out.data_ptr(), None if bias is None else bias.data_ptr(),
                     RT.gemm_workspace(a.device, st), _LT_WS_BYTES, st)
    if rc == -3:
        _LT_UNSUPPORTED.add(plan)
        return False
    if rc != 0:
        raise lib.BevBertHipError(f"bevbert_gemm_run failed ({rc}): {lib.load().bevbert_last_error().decode()}")
    return True


def _LT_RUN(*args):
    global _LT_RUN
    _LT_RUN = lib.load().bevbert_gemm_run        # bind once; later calls go straight to the ctypes function
    return _LT_RUN(*args)


def _rows(t):
    """2-D row-major view (rows, C) of a tensor with unit inner stride and its row stride."""
    t2 = t.reshape(-1, t.shape[-1])
    if t2.stride(1) != 1 or (t2.shape[0] > 1 and t2.stride(0) < t2.shape[1]):
        t2 = t2.contiguous()
    return t2, (t2.stride(0) if t2.shape[0] > 1 else t2.shape[1])


def _lt_ok(*ts):
    return _LT_ENABLED and all(t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) for t in ts)


def _linear_fwd(x, w_c, b_c):
    """y = x w_c^T (+ b_c)."""
    N, K = w_c.shape
    if _lt_ok(x, w_c) and x.dtype == w_c.dtype and w_c.stride(1) == 1 and x.numel() > 0:
        x2, lda = _rows(x)
        M = x2.shape[0]
        y = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
        if _lt_gemm(x2, w_c, y, b_c, M, N, K, 0, 1, lda, w_c.stride(0), N):
            return y
    if x.is_cuda:
        _warn_fallback("fwd", x.numel() // max(1, K), N, K)
    return F.linear(x, w_c, b_c)


def _linear_dgrad(dy2, w_c, add=None):
    """dx (M x K) = dy2 (M x N) w_c (N x K) (+ add, an (M x K) tensor folded in as the GEMM's beta = 1 addend)."""
    N, K = w_c.shape
    if add is not None:
        add = add.reshape(-1, K)
        if not add.is_contiguous() or add.dtype != dy2.dtype:
            add = add.to(dy2.dtype).contiguous()
    if _lt_ok(dy2, w_c) and dy2.dtype == w_c.dtype and w_c.stride(1) == 1 and dy2.numel() > 0:
        d2, lda = _rows(dy2)
        M = d2.shape[0]
        dx = torch.empty(M, K, dtype=dy2.dtype, device=dy2.device)
        if add is None:
            if _lt_gemm(d2, w_c, dx, None, M, K, N, 0, 0, lda, w_c.stride(0), K):
                return dx
        elif _lt_gemm(d2, w_c, dx, None, M, K, N, 0, 0, lda, w_c.stride(0), K, accumulate=1, c_in=add):
            return dx
    if dy2.is_cuda:
        _warn_fallback("dgrad", dy2.shape[0], K, N)
    return dy2.mm(w_c) if add is None else torch.addmm(add, dy2, w_c)


def _linear_wgrad(dy2, x2, S=1, scratch=False):
    """(S x) N x K partial products dy2^T x2 over S equal chunks of the token axis (compute dtype); ``scratch``: the
    product lives in the scratch ring (it is consumed by the batched accumulate at the end of the backward pass)."""
    M, N = dy2.shape
    K = x2.shape[1]
    if _lt_ok(dy2, x2) and dy2.dtype == x2.dtype and M > 0:
        d2, lda = _rows(dy2)
        xx, ldb = _rows(x2)
        if S == 1 or (lda == N and ldb == K):
            shape = (S, N, K) if S > 1 else (N, K)
            part = SCRATCH.tensor(shape, dy2.dtype, dy2.device) if scratch else \
                torch.empty(shape, dtype=dy2.dtype, device=dy2.device)
            Ms = M // S
            if _lt_gemm(d2, xx, part, None, N, K, Ms, 1, 0, lda, ldb, K, S, Ms * lda, Ms * ldb, N * K):
                return part
    if dy2.is_cuda:
        _warn_fallback("wgrad", N, K, M)
    if S > 1:
        return _on_launch_stream(lambda: torch.bmm(dy2.view(S, M // S, N).transpose(1, 2), x2.view(S, M // S, K)))
    return _on_launch_stream(lambda: dy2.t().mm(x2))


def _on_launch_stream(fn):
    """Run a torch op on the stream the C-ABI launches currently go to (fallback paths inside a WgradStream section)."""
    if lib._override is None:
        return fn()
    with torch.cuda.stream(WgradStream._target or WgradStream.stream):
        return fn()


_SPLITK_ENABLED = _os.environ.get("BEVBERT_SPLITK", "1") == "1"     # A/B knob
_SPLITK_MAX = int(_os.environ.get("BEVBERT_SPLITK_MAX", "16"))


def _split_k(M, N, K):
    """Number of token-axis chunks for a weight-gradient GEMM dW(N x K) = dy^T(N x M) x(M x K).

    The output is small (9..36 tiles of 256x256) and the reduction axis M is long (5 120 .. 28 224 tokens), so a plain
    GEMM leaves most of the 256 CUs idle (measured 140-250 TFLOP/s); a batched GEMM over S chunks of M fills them
    (600-880 TFLOP/s, scripts/bench_wgrad.py).  Aim at 144-256 workgroups, keep >= 640 tokens per chunk."""
    if not _SPLITK_ENABLED:
        return 1
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    s = 1
    while s * 2 <= min(_SPLITK_MAX, M // 640) and s * 2 * tiles <= 256 and M % (s * 2) == 0:
        s *= 2
    return s


def _wgrad_into(sink, dy2, x2):
    """sink (fp32 arena view, N x K) += dy2^T @ x2 with host-side split-K and a fused partial-sum + accumulate."""
    M, N = dy2.shape
    K = x2.shape[1]
    if dy2.dtype == torch.float32 and not (_LT_ENABLED and dy2.is_cuda):
        if dy2.is_cuda:
            _warn_fallback("wgrad", N, K, M, "BEVBERT_LT_GEMM=0")
        _gemm("wgrad", lambda: _on_launch_stream(lambda: sink.addmm_(dy2.t(), x2)), N, K, M)
        return None
    S = _split_k(M, N, K) if dy2.dtype != torch.float32 else 1
    if not (S > 1 and dy2.is_contiguous() and x2.is_contiguous()):
        S = 1
    batched = WgradStream.DEFER_FINALIZE and (N * K) % 4 == 0 and dy2.is_cuda
    part = _gemm("wgrad", lambda: _linear_wgrad(dy2, x2, S, scratch=batched), N, K, M)
    if batched:           # folded into the arena by ONE launch per backward pass, together with every other weight's
        ReduceQueue.add_accum(part.data_ptr(), sink.data_ptr(), S, N * K, dtype_code(part))
        WgradStream._keep.append(part)     # (a product that came from torch's fallback GEMM must outlive the flush)
        return None
    if (N * K) % 4 == 0:
        call("bevbert_accum_partials", ptr(part), ptr(sink), S, N * K, dtype_code(part), stream())
    else:
        _on_launch_stream(lambda: sink.add_(part if S == 1 else part.sum(0)))
    return part


def embedding_grad_small(ids, d, sink, table_rows):
    """sink (fp32 arena view, table_rows x H) += scatter-sum of d's rows by ids, for tables of a few rows: sliced partial
    sums in the scratch ring (bevbert_embedding_grad_sliced), folded in by the step's batched column reduction
    (ReduceQueue / bevbert_multi_finalize: 16 row lanes per 64 columns, so hundreds of slices are fine)."""
    rows, H = d.shape
    per = 64
    while (rows + per - 1) // per * table_rows > 4096:        # keep the launch at a few thousand workgroups
        per *= 2
    slices = (rows + per - 1) // per
    part = SCRATCH.alloc(slices * table_rows * H * 4, d.device)
    call("bevbert_embedding_grad_sliced", ptr(ids), ptr(d), part, rows, H, table_rows, per, dtype_code(d), stream())
    ReduceQueue.add(part, slices, 1, table_rows * H, (sink.data_ptr(), None, None))


def _param_grads(w_sink, b_sink, dyc, xc):
    """dW += dy^T x and db += colsum(dy) into the gradient arena (the deferred body of a Linear's backward)."""
    if w_sink is not None:
        part = _wgrad_into(w_sink, dyc, xc)
        if part is not None and lib._override is not None:
            WgradStream._keep.append(part)
    if b_sink is not None:
        C = dyc.shape[1]
        if WgradStream.DEFER_FINALIZE and dyc.shape[0] > 0:
            nb = _partial_rows(dyc.shape[0])
            part = SCRATCH.alloc(nb * C * 4, dyc.device)
            call("bevbert_colsum_partials", ptr(dyc), part, dyc.shape[0], C, dtype_code(dyc), stream())
            ReduceQueue.add(part, nb, 1, C, (ptr(b_sink), None, None))
        else:
            ws = RT.workspace(dyc.device, 512 * C)
            call("bevbert_colsum", ptr(dyc), ptr(b_sink), ptr(ws), dyc.shape[0], C, dtype_code(dyc), 1, stream())


_PARTIAL_ROWS = {}


def _partial_rows(rows):
    """number of per-block partial rows the two-stage column reductions produce for `rows` input rows"""
    nb = _PARTIAL_ROWS.get(rows)
    if nb is None:
        nb = _PARTIAL_ROWS[rows] = lib.load().bevbert_colsum_partial_rows(rows)
    return nb


def _sink(param):
    """fp32 accumulation target of a parameter, or None for plain tensors."""
    return getattr(param, "main_grad", None)


def _mark_touched(param):
    param = getattr(param, "table", param)          # RowOfTable: the parameter is the table
    arena = getattr(param, "arena", None)
    if arena is not None:
        arena.touch(param)


class RowOfTable:
    """Row ``r`` of an arena-resident embedding table used as the broadcast ``bias`` of a fused LayerNorm: the token-type
    row that the reference adds to every panorama token (vilmodel.py:518-521 ``+ type_embed_layer(ones)``).  The
    gradient of a broadcast term is the column sum of the LayerNorm's input gradient, i.e. exactly the kernel's dbias
    output: it goes through the deterministic two-stage column reduction straight into the table's gradient row.  (As a
    torch broadcast add its gradient was a torch ``sum`` over 11 520 rows -- whose result depended on what else the GPU
    was running: the last source of run-to-run noise found in round 4.)"""

    def __init__(self, table, r):
        assert getattr(table, "main_grad", None) is not None or not table.requires_grad, \
            "RowOfTable: the table must live in a ParamArena (its gradient row is written by the LayerNorm backward)"
        self.table, self.r = table, int(r)
        self.dtype, self.requires_grad = table.dtype, table.requires_grad

    def detach(self):
        return self.table.detach()[self.r]

    @property
    def main_grad(self):
        mg = getattr(self.table, "main_grad", None)
        return None if mg is None or not self.table.requires_grad else mg[self.r]


def _compute(param):
    """compute-dtype view of a parameter (bf16 shadow in mixed precision, the master itself in fp32)."""
    return getattr(param, "compute", param)


def _f32(param):
    return param.detach() if param.dtype == torch.float32 else param.detach().float()


class _UseParam(torch.autograd.Function):
    """Bridge for the few tiny parameters consumed by plain torch ops (e.g. sprel_linear): hands out the fp32
    master and routes the incoming gradient into the arena instead of ``.grad``."""

    @staticmethod
    def forward(ctx, p):
        ctx.p = p
        return p.detach().view_as(p)

    @staticmethod
    def backward(ctx, g):
        p = ctx.p
        sink = _sink(p)
        if sink is None:
            return g
        _mark_touched(p)
        sink.add_(g.to(sink.dtype))
        return None


def use_param(p):
    return _UseParam.apply(p)


# ----------------------------------------------------------------------------- K3 LayerNorm family
class _BiasDropResLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, residual, gamma, beta, eps, drop_p, inplace_z, post1=None, post2=None, return_z=False):
        assert x.is_contiguous() and x.dim() >= 2
        H = x.shape[-1]
        rows = x.numel() // H
        y = torch.empty_like(x)
        need_grad = any(ctx.needs_input_grad)
        plain = bias is None and residual is None and drop_p == 0
        z = (x if (inplace_z or plain) else torch.empty_like(x)) if need_grad else None
        mean = torch.empty(rows, dtype=torch.float32, device=x.device) if need_grad else None
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if need_grad else None
        off = RT.next_offset(x.numel()) if drop_p > 0 else 0
        if residual is not None:
            assert residual.is_contiguous() and residual.shape == x.shape and residual.dtype == x.dtype
        if post1 is not None or post2 is not None:
            # y = LN(x + bias) + post1 + post2: the sums that follow the LayerNorm ride on its store (rowops.hip)
            assert residual is None and drop_p == 0, "post terms: plain bias + LayerNorm only"
            for t in (post1, post2):
                assert t is None or (t.is_contiguous() and t.shape == x.shape and t.dtype == x.dtype)
            call("bevbert_layernorm_post_fwd", ptr(x), ptr(_f32(bias)) if bias is not None else None, ptr(_f32(gamma)),
                 ptr(_f32(beta)), ptr(post1), ptr(post2), ptr(y), None if plain else ptr(z), ptr(mean), ptr(rstd), rows,
                 H, float(eps), dtype_code(x), stream())
        else:
            call("bevbert_bias_dropout_residual_layernorm_fwd", ptr(x), ptr(_f32(bias)) if bias is not None else None,
                 ptr(residual), ptr(_f32(gamma)), ptr(_f32(beta)), ptr(y), None if plain else ptr(z), ptr(mean),
                 ptr(rstd), rows, H, float(eps), dtype_code(x), float(drop_p), RT.seed, off, stream())
        ctx.save_for_backward(z, mean, rstd)
        ctx.params = (bias, gamma, beta)
        ctx.cfg = (rows, H, float(drop_p), RT.seed, off, residual is not None)
        ctx.posts = (post1 is not None, post2 is not None)
        ctx.return_z = return_z
        if return_z:
            # pre-norm blocks (transformer.py:170-182): z = residual + dropout(x + bias) is the NEW residual stream and
            # y = LayerNorm(z) feeds the next sub-layer; the gradient arriving at z is added to LayerNorm's input gradient
            # inside the backward kernel (bevbert_layernorm_bwd_add)
            assert need_grad and z is not None and z is not x
            return y, z.view_as(z)
        return y

    @staticmethod
    def backward(ctx, dy, dz_in=None):
        z, mean, rstd = ctx.saved_tensors
        bias, gamma, beta = ctx.params
        rows, H, drop_p, seed, off, has_res = ctx.cfg
        if dy is None:              # only z was used downstream: LayerNorm itself contributes nothing
            dy = torch.zeros_like(z)
        dy = dy.contiguous()
        add = None
        if ctx.return_z and dz_in is not None:
            add = dz_in.contiguous()
            assert add.dtype == dy.dtype and add.shape == dy.shape
        dz = torch.empty_like(dy)
        dx = torch.empty_like(dy) if (drop_p > 0 and has_res) else None
        dev = dy.device
        ws = RT.workspace(dev, lib.load().bevbert_colsum_workspace_floats(3 * H))
        outs = []
        for p in (gamma, beta, bias):
            if p is None:
                outs.append((None, None, 0))
            elif p is bias and not getattr(p, "requires_grad", True):
                outs.append((None, None, None))          # a frozen bias (e.g. fix_lang_embedding): no gradient wanted
            elif _sink(p) is not None:
                outs.append((_sink(p), None, 1))
                _mark_touched(p)
            else:
                t = torch.empty(H, dtype=torch.float32, device=dev)
                outs.append((t, t, 0))
        (dg, rg, ag), (db, rb, ab), (dbi, rbi, abi) = outs
        assert ag == ab and (bias is None or abi is None or abi == ag), "mixed arena / plain parameters in one LayerNorm"
        # without a residual branch only dx is needed (it is the single input gradient)
        if not has_res and drop_p > 0:
            dx, dz_ptr = dz, None
        else:
            dz_ptr = dz
        if ag == 1 and WgradStream.DEFER_FINALIZE and dev.type == "cuda":
            # arena parameters: the kernel leaves its per-block partial sums in the scratch ring; the second stage of
            # the reduction joins the step's other pending reductions (ReduceQueue: one launch, off the critical path)
            nb = _partial_rows(rows)
            part = SCRATCH.alloc(nb * 3 * H * 4, dev)
            call("bevbert_layernorm_bwd_add", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(_f32(gamma)), ptr(dz_ptr), ptr(dx),
                 ptr(add), None, None, None, part, rows, H, dtype_code(dy), drop_p, seed, off, 1, stream())
            ReduceQueue.add(part, nb, 3, H, (ptr(dg), ptr(db), ptr(dbi)))
        else:
            call("bevbert_layernorm_bwd_add", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(_f32(gamma)), ptr(dz_ptr), ptr(dx),
                 ptr(add), ptr(dg), ptr(db), ptr(dbi), ptr(ws), rows, H, dtype_code(dy), drop_p, seed, off, ag, stream())
        gx = dx if dx is not None else dz
        gres = dz if has_res else None
        cast = lambda r, p: None if r is None else r.to(p.dtype)
        g1, g2 = (dy if has else None for has in ctx.posts)      # the post terms were added after the affine
        return (gx, cast(rbi, bias) if bias is not None else None, gres, cast(rg, gamma), cast(rb, beta), None, None, None,
                g1, g2, None)


class _BiasDropResLN32(torch.autograd.Function):
    """LayerNorm(dropout(x + bias) + residual) with the fp32 residual stream of ``RT.res32``: x bf16 (a GEMM output),
    residual fp32 (the previous block's ``y32``) or bf16 (where a stream starts); returns (y16, y32).  The backward sums the
    two output gradients in the kernel (bf16 from the GEMMs that read y16, fp32 from the residual add that read y32) and
    returns dz in fp32 to an fp32 residual."""

    @staticmethod
    def forward(ctx, x, bias, residual, gamma, beta, eps, drop_p):
        assert x.is_contiguous() and x.dtype == torch.bfloat16 and x.dim() >= 2
        H = x.shape[-1]
        rows = x.numel() // H
        need_grad = any(ctx.needs_input_grad)
        y16 = torch.empty_like(x)
        y32 = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        z32 = torch.empty(x.shape, dtype=torch.float32, device=x.device) if need_grad else None
        mean = torch.empty(rows, dtype=torch.float32, device=x.device) if need_grad else None
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if need_grad else None
        off = RT.next_offset(x.numel()) if drop_p > 0 else 0
        assert residual.is_contiguous() and residual.shape == x.shape and residual.dtype in (torch.float32, torch.bfloat16)
        call("bevbert_layernorm_res32_fwd", ptr(x), ptr(_f32(bias)) if bias is not None else None, ptr(residual),
             dtype_code(residual), ptr(_f32(gamma)), ptr(_f32(beta)), ptr(y16), ptr(y32), ptr(z32), ptr(mean), ptr(rstd),
             rows, H, float(eps), float(drop_p), RT.seed, off, stream())
        ctx.save_for_backward(z32, mean, rstd)
        ctx.params = (bias, gamma, beta)
        ctx.cfg = (rows, H, float(drop_p), RT.seed, off, residual.dtype)
        return y16, y32

    @staticmethod
    def backward(ctx, dy16, dy32):
        z32, mean, rstd = ctx.saved_tensors
        bias, gamma, beta = ctx.params
        rows, H, drop_p, seed, off, res_dtype = ctx.cfg
        dev = z32.device
        if dy16 is None and dy32 is None:
            dy32 = torch.zeros_like(z32)
        dy16 = dy16.contiguous() if dy16 is not None else None
        dy32 = dy32.contiguous() if dy32 is not None else None
        dz32 = torch.empty_like(z32)
        dx16 = torch.empty(z32.shape, dtype=torch.bfloat16, device=dev)
        outs = []
        for p in (gamma, beta, bias):
            if p is None:
                outs.append((None, None, 0))
            elif p is bias and not getattr(p, "requires_grad", True):
                outs.append((None, None, None))
            elif _sink(p) is not None:
                outs.append((_sink(p), None, 1))
                _mark_touched(p)
            else:
                t = torch.empty(H, dtype=torch.float32, device=dev)
                outs.append((t, t, 0))
        (dg, rg, ag), (db, rb, ab), (dbi, rbi, abi) = outs
        assert ag == ab and (bias is None or abi is None or abi == ag), "mixed arena / plain parameters in one LayerNorm"
        if ag == 1 and WgradStream.DEFER_FINALIZE and dev.type == "cuda":
            nb = _partial_rows(rows)
            part = SCRATCH.alloc(nb * 3 * H * 4, dev)
            call("bevbert_layernorm_res32_bwd", ptr(dy16), ptr(dy32), ptr(z32), ptr(mean), ptr(rstd), ptr(_f32(gamma)),
                 ptr(dz32), ptr(dx16), None, None, None, part, rows, H, drop_p, seed, off, 1, stream())
            ReduceQueue.add(part, nb, 3, H, (ptr(dg), ptr(db), ptr(dbi)))
        else:
            ws = RT.workspace(dev, lib.load().bevbert_colsum_workspace_floats(3 * H))
            call("bevbert_layernorm_res32_bwd", ptr(dy16), ptr(dy32), ptr(z32), ptr(mean), ptr(rstd), ptr(_f32(gamma)),
                 ptr(dz32), ptr(dx16), ptr(dg), ptr(db), ptr(dbi), ptr(ws), rows, H, drop_p, seed, off, ag, stream())
        cast = lambda r, p: None if r is None else r.to(p.dtype)
        gres = dz32 if res_dtype == torch.float32 else dz32.to(torch.bfloat16)      # (a bf16 residual: where a stream starts)
        return dx16, cast(rbi, bias) if bias is not None else None, gres, cast(rg, gamma), cast(rb, beta), None, None


def bias_dropout_residual_layernorm(x, bias, residual, gamma, beta, eps, drop_p=0.0, training=False,
                                    inplace_z=True):
    """LayerNorm(dropout(x + bias) + residual)  -- vilmodel.py:150-154,189-193."""
    p = float(drop_p) if training else 0.0
    if RT.res32 and residual is not None and x.dtype == torch.bfloat16 and x.is_cuda:
        # fp32 residual stream: the previous block left its fp32 output on the bf16 tensor the model passes around
        r32 = getattr(residual, "_res32", None)
        y16, y32 = _BiasDropResLN32.apply(x, bias, r32 if r32 is not None else residual, gamma, beta, eps, p)
        y16._res32 = y32
        return y16
    return _BiasDropResLN.apply(x, bias, residual, gamma, beta, eps, p, inplace_z, None, None)


def bias_dropout_residual_prenorm(x, bias, residual, gamma, beta, eps, drop_p=0.0, training=False):
    """(LayerNorm(z), z) with z = residual + dropout(x + bias): one launch for the residual add of a pre-norm block AND the
    LayerNorm that opens the next sub-layer (transformer.py:170-182); backward likewise (the gradient reaching z from the
    rest of the stream is folded into the LayerNorm backward kernel).  Inference / no-grad callers get the two tensors
    from the same launch too."""
    p = float(drop_p) if training else 0.0
    if not (torch.is_grad_enabled() and (x.requires_grad or residual.requires_grad)):
        assert x.is_contiguous() and residual.is_contiguous() and residual.shape == x.shape and residual.dtype == x.dtype
        H = x.shape[-1]
        rows = x.numel() // H
        y, z = torch.empty_like(x), torch.empty_like(x)
        off = RT.next_offset(x.numel()) if p > 0 else 0
        call("bevbert_bias_dropout_residual_layernorm_fwd", ptr(x), ptr(_f32(bias)) if bias is not None else None,
             ptr(residual), ptr(_f32(gamma)), ptr(_f32(beta)), ptr(y), ptr(z), None, None, rows, H, float(eps),
             dtype_code(x), p, RT.seed, off, stream())
        return y, z
    return _BiasDropResLN.apply(x, bias, residual, gamma, beta, eps, p, False, None, None, True)


def bias_layernorm_plus(x, bias, gamma, beta, eps, post1, post2=None):
    """(LayerNorm(x + bias) + post1) + post2 in one launch -- the sums of the embedding compositions
    (vilmodel.py:494-532, 589-593); fp32 results equal the separate adds bit for bit (same order of additions)."""
    return _BiasDropResLN.apply(x, bias, None, gamma, beta, eps, 0.0, True, post1.contiguous(),
                                None if post2 is None else post2.contiguous())


def layernorm(x, gamma, beta, eps):
    return _BiasDropResLN.apply(x.contiguous(), None, None, gamma, beta, eps, 0.0, False, None, None)


# ----------------------------------------------------------------------------- dropout (+ residual, + cast)
class _DropoutAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, p, out_dtype):
        assert x.is_contiguous() and x.numel() % 4 == 0
        out_dtype = out_dtype or x.dtype
        y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        if residual is not None:
            assert residual.is_contiguous() and residual.shape == x.shape and residual.dtype == out_dtype
        off = RT.next_offset(x.numel())
        call("bevbert_dropout_add", ptr(x), ptr(residual), ptr(y), x.numel(), dtype_code(x), dtype_code(y), p, RT.seed,
             off, stream())
        ctx.cfg = (p, RT.seed, off, x.dtype, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, off, in_dtype, has_res = ctx.cfg
        dx = None
        if ctx.needs_input_grad[0]:
            dy = dy.contiguous()
            dx = torch.empty_like(dy)
            call("bevbert_dropout_add", ptr(dy), None, ptr(dx), dy.numel(), dtype_code(dy), dtype_code(dx), p, seed, off,
                 stream())
            if dx.dtype != in_dtype:
                dx = dx.to(in_dtype)
        return dx, (dy if has_res else None), None, None


def dropout(x, p, training, residual=None, out_dtype=None):
    """residual + nn.Dropout(p)(x) on the library's counter-based mask stream (reproducible from (seed, step) alone);
    ``out_dtype`` fuses the cast of fp32 loader features to the compute dtype."""
    if not training or p <= 0.0:
        y = x if out_dtype is None or out_dtype == x.dtype else x.to(out_dtype)
        return y if residual is None else residual + y
    return _DropoutAdd.apply(x.contiguous(), residual, float(p), out_dtype)


# ----------------------------------------------------------------------------- SAP loss tail
class _SapLoss(torch.autograd.Function):
    """loss (B,) of forward_sap behind the three heads (pretrain_cmt.py:225-275) in one launch; see bevbert_sap_loss_fwd."""

    @staticmethod
    def forward(ctx, graw, lraw, fraw, visited, gmap_lens, nav_masks, cand_idxs, src, vis_c, glabels, llabels):
        B, G = graw.shape
        K = lraw.shape[1]
        dev = graw.device
        assert lraw.dtype == graw.dtype and (fraw is None or fraw.dtype == graw.dtype)
        graw, lraw = graw.contiguous(), lraw.contiguous()
        fraw = None if fraw is None else fraw.contiguous()
        buf = torch.empty(B * (G + K + 2), dtype=torch.float32, device=dev)
        loss, dG, dL, dF = buf[:B], buf[B:B + B * G], buf[B + B * G:B + B * (G + K)], buf[B + B * (G + K):]
        as_u8 = lambda t: t.contiguous().view(torch.uint8)
        call("bevbert_sap_loss_fwd", ptr(graw), ptr(lraw), ptr(fraw), ptr(as_u8(visited)), ptr(gmap_lens.contiguous()),
             ptr(as_u8(nav_masks)), ptr(cand_idxs.contiguous()), ptr(src.contiguous()), ptr(as_u8(vis_c)),
             ptr(glabels.contiguous()), ptr(llabels.contiguous()), ptr(loss), ptr(dG), ptr(dL), ptr(dF), B, G, K,
             nav_masks.shape[1], dtype_code(graw), stream())
        ctx.save_for_backward(buf)
        ctx.dims = (B, G, K, graw.dtype, fraw is not None)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (buf,) = ctx.saved_tensors
        B, G, K, dt, has_f = ctx.dims
        dG, dL, dF = buf[B:B + B * G], buf[B + B * G:B + B * (G + K)], buf[B + B * (G + K):]
        out = torch.empty(B * (G + K + 1), dtype=dt, device=buf.device)
        dgr, dlr, dfr = out[:B * G].view(B, G), out[B * G:B * (G + K)].view(B, K), out[B * (G + K):].view(B, 1)
        call("bevbert_sap_loss_bwd", ptr(dG), ptr(dL), ptr(dF), ptr(dloss.contiguous().float()), ptr(dgr), ptr(dlr),
             ptr(dfr) if has_f else None, B, G, K, dtype_code(out), stream())
        return (dgr, dlr, dfr if has_f else None) + (None,) * 8


def sap_loss_supported(graw, lraw):
    return graw.is_cuda and graw.shape[1] <= 64 and lraw.shape[1] <= 62 and graw.dtype in (torch.float32, torch.bfloat16)


def sap_loss(graw, lraw, fraw, visited, gmap_lens, nav_masks, cand_idxs, src, vis_c, glabels, llabels):
    """(B,) loss of the SAP task from the raw head outputs: graw (B,G), lraw (B,K), fraw (B,1) or None."""
    return _SapLoss.apply(graw, lraw, fraw, visited, gmap_lens, nav_masks, cand_idxs, src, vis_c, glabels, llabels)


class _CrossEntropy(torch.autograd.Function):
    """F.cross_entropy(logits.float(), target, reduction="none") without the fp32 copy of the logits."""

    @staticmethod
    def forward(ctx, logits, target):
        rows, C = logits.shape
        logits = logits.contiguous()
        out = torch.empty(2, rows, dtype=torch.float32, device=logits.device)
        call("bevbert_cross_entropy_fwd", ptr(logits), ptr(target.contiguous()), ptr(out[0]), ptr(out[1]), rows, C,
             dtype_code(logits), stream())
        ctx.save_for_backward(logits, target, out)
        return out[0]

    @staticmethod
    def backward(ctx, dloss):
        logits, target, out = ctx.saved_tensors
        rows, C = logits.shape
        d = torch.empty_like(logits)
        call("bevbert_cross_entropy_bwd", ptr(logits), ptr(target.contiguous()), ptr(out[1]),
             ptr(dloss.contiguous().float()), ptr(d), rows, C, dtype_code(logits), stream())
        return d, None


def cross_entropy_rows(logits, target):
    """(rows,) fp32 losses of (rows, C) logits in the compute dtype (the MLM head's vocabulary rows)."""
    if logits.is_cuda and logits.dtype in (torch.float32, torch.bfloat16):
        return _CrossEntropy.apply(logits, target)
    return torch.nn.functional.cross_entropy(logits.float(), target, reduction="none")


# ----------------------------------------------------------------------------- K4 bias + GELU
class _BiasGelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias):
        assert x.is_contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        y = torch.empty_like(x)
        call("bevbert_bias_gelu_fwd", ptr(x), ptr(_f32(bias)), ptr(y), rows, C, dtype_code(x), stream())
        ctx.save_for_backward(x)
        ctx.bias = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        bias = ctx.bias
        C = x.shape[-1]
        rows = x.numel() // C
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        ws = RT.workspace(dy.device, 512 * C)
        sink = _sink(bias)
        if sink is not None:
            _mark_touched(bias)
            if WgradStream.DEFER_FINALIZE:      # second reduction stage batched with the step's others (see _BiasDropResLN)
                nb = _partial_rows(rows)
                part = SCRATCH.alloc(nb * C * 4, dy.device)
                call("bevbert_bias_gelu_bwd", ptr(dy), ptr(x), ptr(_f32(bias)), ptr(dx), None, part, rows, C,
                     dtype_code(dy), 1, stream())
                ReduceQueue.add(part, nb, 1, C, (ptr(sink), None, None))
                return dx, None
            call("bevbert_bias_gelu_bwd", ptr(dy), ptr(x), ptr(_f32(bias)), ptr(dx), ptr(sink), ptr(ws), rows, C,
                 dtype_code(dy), 1, stream())
            return dx, None
        db = torch.empty(C, dtype=torch.float32, device=dy.device)
        call("bevbert_bias_gelu_bwd", ptr(dy), ptr(x), ptr(_f32(bias)), ptr(dx), ptr(db), ptr(ws), rows, C,
             dtype_code(dy), 0, stream())
        return dx, db.to(bias.dtype)


def bias_gelu(x, bias):
    """gelu_erf(x + bias) -- vilmodel.py:31-37,177-180."""
    return _BiasGelu.apply(x, bias)


# ----------------------------------------------------------------------------- library GEMM with arena wgrad
class _Linear(torch.autograd.Function):
    """y = x W^T (+ b) on hipBLASLt; backward writes dW / db straight into the gradient arena."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_c, b_c, tap=False):
        y = _gemm("fwd", lambda: _linear_fwd(x, w_c, b_c), x.numel() // x.shape[-1], w_c.shape[0], w_c.shape[1])
        ctx.save_for_backward(x, w_c)
        ctx.params = (weight, bias)
        ctx.tap = tap
        # tap: the input ALSO feeds a residual connection.  It is handed back as a second output, so that the residual's
        # gradient arrives HERE and is folded into the input-gradient GEMM (dx = dy W + d_res, beta = 1) -- autograd would
        # otherwise add the two gradients of x with a separate elementwise kernel (~50 of them per training step)
        return (y, x.view_as(x)) if tap else y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, w_c = ctx.saved_tensors
        weight, bias = ctx.params
        if dy is None:                      # only the residual tap carried a gradient
            return dres, None, None, None, None, None
        dy2 = dy.reshape(-1, dy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        M, N, K = dy2.shape[0], dy2.shape[1], x2.shape[1]
        dx = _gemm("dgrad", lambda: _linear_dgrad(dy2, w_c, dres), M, K, N).view(x.shape) if ctx.needs_input_grad[0] else None
        gw = gb = None
        w_sink = _sink(weight) if weight.requires_grad else None
        b_sink = _sink(bias) if (bias is not None and bias.requires_grad) else None
        C = dy2.shape[1]
        if weight.requires_grad and w_sink is None:
            gw = _linear_wgrad(dy2, x2).to(weight.dtype)
        if bias is not None and bias.requires_grad and (b_sink is None or C % 4 != 0):
            if C % 4 != 0:                                  # e.g. the 1-wide heads: a library reduction is fine
                s = dy2.float().sum(0)
                if b_sink is not None:
                    _mark_touched(bias)
                    b_sink.add_(s)
                    b_sink = None
                else:
                    gb = s.to(bias.dtype)
            else:
                ws = RT.workspace(dy.device, 512 * C)
                dyc = dy2 if dy2.is_contiguous() else dy2.contiguous()
                t = torch.empty(C, dtype=torch.float32, device=dy.device)
                call("bevbert_colsum", ptr(dyc), ptr(t), ptr(ws), dyc.shape[0], C, dtype_code(dyc), 0, stream())
                gb = t.to(bias.dtype)
        if w_sink is not None or b_sink is not None:        # arena parameters: accumulate on the weight-gradient stream
            dyc = dy2 if dy2.is_contiguous() else dy2.contiguous()
            xc = x2 if x2.is_contiguous() else x2.contiguous()
            if w_sink is not None:
                _mark_touched(weight)
            if b_sink is not None:
                _mark_touched(bias)
            WgradStream.submit(dy.device, lambda: _param_grads(w_sink, b_sink, dyc, xc), dyc, xc, dy)
        return dx, gw, gb, None, None, None


def linear(x, weight, bias=None, w_c=None, b_c=None):
    """F.linear with compute-dtype weights; ``weight``/``bias`` are the master parameters (gradient owners)."""
    if w_c is None:
        w_c = _compute(weight)
    if bias is not None and b_c is None:
        b_c = _compute(bias)
    return _Linear.apply(x, weight, bias, w_c, b_c)


def linear_res(x, weight, bias=None):
    """(linear(x), x) for an input that also feeds a residual connection: use the SECOND output as the residual and
    the gradient of the residual branch is folded into this layer's input-gradient GEMM (see _Linear.forward)."""
    if not (x.requires_grad and torch.is_grad_enabled()) or getattr(x, "_res32", None) is not None:
        # (fp32 residual stream: the residual is x's fp32 twin, its gradient joins x's inside the LayerNorm backward kernel)
        return linear(x, weight, bias), x
    return _Linear.apply(x, weight, bias, _compute(weight), None if bias is None else _compute(bias), True)


class _PackedParam:
    """A contiguous run of arena parameters used as one GEMM operand (packed QKV / KV projections)."""

    def __init__(self, params, compute, main_grad):
        self.params, self.compute, self.main_grad = params, compute, main_grad
        self.requires_grad = any(p.requires_grad for p in params)
        self.dtype = params[0].dtype
        self.arena = getattr(params[0], "arena", None)

    def touch(self):
        for p in self.params:
            _mark_touched(p)


class _LinearPacked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pw, pb, tap=False):
        ctx.save_for_backward(x)
        ctx.packed = (pw, pb)
        y = _gemm("fwd", lambda: _linear_fwd(x, pw.compute, pb.compute), x.numel() // x.shape[-1],
                  pw.compute.shape[0], pw.compute.shape[1])
        return (y, x.view_as(x)) if tap else y          # residual tap: see _Linear.forward

    @staticmethod
    def backward(ctx, dy, dres=None):
        (x,) = ctx.saved_tensors
        pw, pb = ctx.packed
        if dy is None:
            return dres, None, None, None
        dy2 = dy.reshape(-1, dy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        M, N, K = dy2.shape[0], dy2.shape[1], x2.shape[1]
        dx = _gemm("dgrad", lambda: _linear_dgrad(dy2, pw.compute, dres), M, K, N).view(x.shape) if ctx.needs_input_grad[0] else None
        if pw.requires_grad:
            pw.touch()
            pb.touch()
            C = dy2.shape[1]
            dyc = dy2 if dy2.is_contiguous() else dy2.contiguous()
            xc = x2 if x2.is_contiguous() else x2.contiguous()
            WgradStream.submit(dy.device, lambda: _param_grads(pw.main_grad, pb.main_grad, dyc, xc), dyc, xc, dy)
        return dx, None, None, None


def linear_packed(x, pw, pb):
    return _LinearPacked.apply(x, pw, pb)


HOIST_KV = _os.environ.get("BEVBERT_HOIST_KV", "1") == "1"      # A/B knob


class _KVGradHolder:
    """The (B, Lk, layers * 2H) gradient buffer of a hoisted K/V projection, allocated when the first attention backward
    asks for its slice."""

    def __init__(self, n_layers, width):
        self.n, self.width, self.buf = n_layers, width, None

    def grad_slice(self, layer, like):
        if self.buf is None:
            self.buf = torch.empty(like.shape[:-1] + (self.n * self.width,), dtype=like.dtype, device=like.device)
        return self.buf[..., layer * self.width:(layer + 1) * self.width]


class _HoistedKV(torch.autograd.Function):
    """The key / value projections of ALL cross-attention layers of an encoder in one GEMM.

    The context of the cross-attention is the same tensor in every layer (the text states in the map encoders --
    ``lang_feats`` is never updated, vilmodel.py:383-398,446-463 -- or the BEV / map tokens in the MLM direction), so
    layers x (x W_kv^T) is one (rows, layers * 2H, C) problem: at the 5 120 text rows of the step that is 96 output
    tiles of 256 x 256 instead of four launches of 24.  Backward: each layer's attention writes dK / dV into its column
    slice of one buffer (``_KVGradHolder``); when the last one has run, ONE K-concatenated input-gradient GEMM
    (rows x C, K = layers * 2H) and ONE weight-gradient GEMM (layers * 2H x C) into the arena follow."""

    @staticmethod
    def forward(ctx, x, pw, pb, n_layers):
        ctx.save_for_backward(x)
        ctx.packed = (pw, pb)
        y = _gemm("fwd", lambda: _linear_fwd(x, pw.compute, pb.compute), x.numel() // x.shape[-1],
                  pw.compute.shape[0], pw.compute.shape[1])
        width = y.shape[-1] // n_layers
        ctx.holder = _KVGradHolder(n_layers, width)
        return tuple(y[..., i * width:(i + 1) * width] for i in range(n_layers))

    @staticmethod
    def backward(ctx, *grads):
        (x,) = ctx.saved_tensors
        pw, pb = ctx.packed
        h = ctx.holder
        some = next((g for g in grads if g is not None), None)
        if some is None:
            return None, None, None, None
        for i, g in enumerate(grads):
            dst = h.grad_slice(i, some)
            if g is None:
                dst.zero_()                        # a layer whose output reached no loss
            elif g.data_ptr() != dst.data_ptr() or g.stride() != dst.stride():
                dst.copy_(g)                       # a gradient that did not come from the attention backward (tests)
        dy2 = h.buf.reshape(-1, h.buf.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        M, N, K = dy2.shape[0], dy2.shape[1], x2.shape[1]
        dx = _gemm("dgrad", lambda: _linear_dgrad(dy2, pw.compute), M, K, N).view(x.shape) if ctx.needs_input_grad[0] else None
        if pw.requires_grad:
            pw.touch()
            pb.touch()
            xc = x2 if x2.is_contiguous() else x2.contiguous()
            WgradStream.submit(dy2.device, lambda: _param_grads(pw.main_grad, pb.main_grad, dy2, xc), dy2, xc, h.buf)
        return dx, None, None, None


def hoisted_kv(context, pw, pb, n_layers):
    """[(B, Lk, 2H) K|V view of layer i] for the cross-attention layers whose packed parameters ``pw`` (layers * 2H, C) /
    ``pb`` (layers * 2H) describe; pass the views as ``kv=`` to BertOutAttention.forward."""
    outs = _HoistedKV.apply(context, pw, pb, n_layers)
    if torch.is_grad_enabled() and any(o.requires_grad for o in outs):
        holder = outs[0].grad_fn.holder if hasattr(outs[0].grad_fn, "holder") else None
        if holder is not None:
            for i, o in enumerate(outs):
                o._kv_grad_slot = (holder, i)
    return outs


def linear_packed_res(x, pw, pb):
    """(packed projection of x, x as residual tap) -- see linear_res."""
    if not (x.requires_grad and torch.is_grad_enabled()) or getattr(x, "_res32", None) is not None:
        return _LinearPacked.apply(x, pw, pb), x
    return _LinearPacked.apply(x, pw, pb, True)


# ----------------------------------------------------------------------------- K2 attention
def _strides(q, k, v, o):
    for t in (q, k, v, o):
        assert t.dim() == 3 and t.stride(2) == 1, "attention operands are (B, L, nh*64) with unit inner stride"
    import ctypes
    arr = (ctypes.c_int64 * 8)(q.stride(1), k.stride(1), v.stride(1), o.stride(1),
                               q.stride(0), k.stride(0), v.stride(0), o.stride(0))
    return arr


_DROP_BITS_WORDS = {}


def _drop_bits_words(B, nh, Lq, Lk):
    key = (B, nh, Lq, Lk)
    n = _DROP_BITS_WORDS.get(key)
    if n is None:
        n = _DROP_BITS_WORDS[key] = lib.load().bevbert_attn_drop_bits_words(B, nh, Lq, Lk)
    return n


class _Attention(torch.autograd.Function):
    """mode 'self': qkv packed (B,L,3H);  mode 'cross': q (B,Lq,H) + kv packed (B,Lk,2H);  mode 'sep': q,k,v."""

    @staticmethod
    def forward(ctx, mode, a, b_, c_, key_mask, bias, nh, drop_p, impl):
        if mode == "self":
            H = a.shape[-1] // 3
            q, k, v = a[..., :H], a[..., H:2 * H], a[..., 2 * H:]
        elif mode == "cross":
            H = a.shape[-1]
            q, k, v = a, b_[..., :H], b_[..., H:]
        else:
            H = a.shape[-1]
            q, k, v = a, b_, c_
        assert H == nh * HEAD_DIM
        B, Lq, Lk = q.shape[0], q.shape[1], k.shape[1]
        o = torch.empty(B, Lq, H, dtype=q.dtype, device=q.device)
        need_grad = any(ctx.needs_input_grad)
        lse = torch.empty(B, nh, Lq, dtype=torch.float32, device=q.device) if need_grad else None
        off = RT.next_offset(B * nh * Lq * Lk) if drop_p > 0 else 0
        scale = 1.0 / math.sqrt(HEAD_DIM)
        if key_mask is not None:
            assert key_mask.dtype == torch.float32 and key_mask.shape == (B, Lk) and key_mask.is_contiguous()
        if bias is not None:
            assert bias.dtype == torch.float32 and bias.shape == (B, Lq, Lk) and bias.is_contiguous()
        bits, bits_ready = None, 0
        if drop_p > 0 and q.dtype == torch.bfloat16 and impl != 1:
            # keep-bit workspace of the dropout mask (1 bit / element in the forward's and in the backward's lane
            # layout: 2 x 19 MB at 64x12x441x441), filled by the library ahead of the forward kernel; both directions
            # read bits through the scalar cache instead of hashing per element
            # attn_small.hip (opt-in, BEVBERT_ATTN_SMALL=1) hashes inline whatever the query count
            short_keys = Lk <= 96 and bias is None and _os.environ.get("BEVBERT_ATTN_SMALL") == "1"
            if (Lq * Lk >= 32768 or Lk > 256) and not short_keys:
                bits, bits_ready = ATTN_BITS.get(B, nh, Lq, Lk, drop_p, off, q.device)
            else:       # small score matrices: the forward hashes inline and leaves the bits for the backward (capi.hip)
                bits = torch.empty(_drop_bits_words(B, nh, Lq, Lk), dtype=torch.int64, device=q.device)
        call("bevbert_attn_fwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(key_mask), ptr(bias),
             _strides(q, k, v, o), B, nh, Lq, Lk, HEAD_DIM, scale, dtype_code(q), impl, float(drop_p), RT.seed, off,
             ptr(bits), bits_ready, stream())
        ctx.save_for_backward(a, b_, c_, key_mask, bias, o, lse, bits)
        ctx.cfg = (mode, nh, float(drop_p), RT.seed, off, impl, scale)
        ctx.kv_slot = getattr(b_, "_kv_grad_slot", None) if mode == "cross" else None     # see hoisted_kv
        return o

    @staticmethod
    def backward(ctx, do):
        a, b_, c_, key_mask, bias, o, lse, bits = ctx.saved_tensors
        mode, nh, drop_p, seed, off, impl, scale = ctx.cfg
        do = do.contiguous()
        if mode == "self":
            H = a.shape[-1] // 3
            q, k, v = a[..., :H], a[..., H:2 * H], a[..., 2 * H:]
            da = torch.empty_like(a)
            dq, dk, dv = da[..., :H], da[..., H:2 * H], da[..., 2 * H:]
            grads = (da, None, None)
        elif mode == "cross":
            H = a.shape[-1]
            q, k, v = a, b_[..., :H], b_[..., H:]
            dq = torch.empty_like(a)
            # K/V projected for all layers of an encoder at once (hoisted_kv): the gradient goes straight into this
            # layer's column slice of the shared (B, Lk, layers * 2H) buffer, which feeds ONE input-gradient GEMM
            dkv = ctx.kv_slot[0].grad_slice(ctx.kv_slot[1], b_) if ctx.kv_slot is not None else torch.empty_like(b_)
            dk, dv = dkv[..., :H], dkv[..., H:]
            grads = (dq, dkv, None)
        else:
            q, k, v = a, b_, c_
            dq, dk, dv = torch.empty_like(a), torch.empty_like(b_), torch.empty_like(c_)
            grads = (dq, dk, dv)
        B, Lq, Lk = q.shape[0], q.shape[1], k.shape[1]
        delta = torch.empty(B, nh, Lq, dtype=torch.float32, device=q.device)
        # the bias is shared by the heads: the kernels store per-head gradients (no atomics), summed here in a fixed order
        dbias_h = torch.zeros(B, nh, Lq, Lk, dtype=torch.float32, device=q.device) \
            if (bias is not None and ctx.needs_input_grad[5]) else None
        assert do.shape == o.shape
        call("bevbert_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq), ptr(dk),
             ptr(dv), ptr(dbias_h), ptr(key_mask), ptr(bias), _strides(q, k, v, o), B, nh, Lq, Lk, HEAD_DIM, scale,
             dtype_code(q), impl, drop_p, seed, off, ptr(bits), stream())
        dbias = None if dbias_h is None else dbias_h.sum(1)
        return (None,) + grads + (None, dbias, None, None, None)


def attention_self(qkv, key_mask, bias, nh, drop_p=0.0, training=False):
    return _Attention.apply("self", qkv, None, None, key_mask, bias, nh, drop_p if training else 0.0, RT.attn_impl)


def attention_cross(q, kv, key_mask, nh, drop_p=0.0, training=False):
    return _Attention.apply("cross", q, kv, None, key_mask, None, nh, drop_p if training else 0.0, RT.attn_impl)


def attention(q, k, v, key_mask=None, bias=None, nh=12, drop_p=0.0, training=False, impl=None):
    return _Attention.apply("sep", q, k, v, key_mask, bias, nh, drop_p if training else 0.0,
                            RT.attn_impl if impl is None else impl)


# ----------------------------------------------------------------------------- K5 embeddings
class _EmbedLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, word, pos, typ, gamma, beta, eps, type_index, word_c, pos_c, typ_c, pad_idx=-1):
        B, L = ids.shape
        H = word_c.shape[1]
        rows = B * L
        y = torch.empty(B, L, H, dtype=word_c.dtype, device=ids.device)
        need_grad = any(ctx.needs_input_grad)
        z = torch.empty_like(y) if need_grad else None
        mean = torch.empty(rows, dtype=torch.float32, device=ids.device) if need_grad else None
        rstd = torch.empty(rows, dtype=torch.float32, device=ids.device) if need_grad else None
        ids = ids.contiguous()
        call("bevbert_embed_sum_layernorm_fwd", ptr(ids), ptr(word_c), ptr(pos_c), ptr(typ_c[type_index]),
             ptr(_f32(gamma)), ptr(_f32(beta)), ptr(y), ptr(z), ptr(mean), ptr(rstd), rows, L, H, float(eps),
             dtype_code(y), 0.0, 0, 0, stream())
        ctx.save_for_backward(ids, z, mean, rstd)
        ctx.params = (word, pos, typ, gamma, beta, type_index, int(pad_idx))
        return y

    @staticmethod
    def backward(ctx, dy):
        ids, z, mean, rstd = ctx.saved_tensors
        word, pos, typ, gamma, beta, type_index, pad_idx = ctx.params
        B, L = ids.shape
        H = z.shape[-1]
        rows = B * L
        dy = dy.contiguous()
        dz = torch.empty_like(dy)
        ws = RT.workspace(dy.device, 512 * 3 * H)
        sg, sb = _sink(gamma), _sink(beta)
        assert (sg is None) == (sb is None)
        # the broadcast token-type row: its gradient is the column sum of dz = the kernel's third (dbias) output, through
        # the deterministic two-stage reduction (a torch sum over the 5 120 rows would depend on the GPU's load)
        styp = _sink(typ)[type_index] if (sg is not None and typ.requires_grad and _sink(typ) is not None) else None
        if sg is not None:
            _mark_touched(gamma); _mark_touched(beta)
            if styp is not None:
                _mark_touched(typ)
            if WgradStream.DEFER_FINALIZE:
                nb = _partial_rows(rows)
                part = SCRATCH.alloc(nb * 3 * H * 4, dy.device)
                call("bevbert_layernorm_bwd", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(_f32(gamma)), ptr(dz), None,
                     None, None, None, part, rows, H, dtype_code(dy), 0.0, 0, 0, 1, stream())
                ReduceQueue.add(part, nb, 3, H, (ptr(sg), ptr(sb), ptr(styp)))
            else:
                call("bevbert_layernorm_bwd", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(_f32(gamma)), ptr(dz), None,
                     ptr(sg), ptr(sb), ptr(styp), ptr(ws), rows, H, dtype_code(dy), 0.0, 0, 0, 1, stream())
            rg = rb = None
        else:
            rg = torch.empty(H, dtype=torch.float32, device=dy.device)
            rb = torch.empty(H, dtype=torch.float32, device=dy.device)
            call("bevbert_layernorm_bwd", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(_f32(gamma)), ptr(dz), None,
                 ptr(rg), ptr(rb), None, ptr(ws), rows, H, dtype_code(dy), 0.0, 0, 0, 0, stream())
        dz2 = dz.reshape(rows, H)
        dzf = dz2.float()

        def word_grad(t):
            call("bevbert_embedding_grad", ptr(ids), ptr(dz2), ptr(t), rows, H, pad_idx, dtype_code(dz2), stream())

        makers = ((word, word_grad),
                  (pos, lambda t: _on_launch_stream(lambda: t[:L].add_(dzf.view(B, L, H).sum(0)))),
                  (typ, lambda t: _on_launch_stream(lambda: t[type_index].add_(dzf.sum(0)))))
        outs, deferred = [], []
        for p, make in makers:
            if not p.requires_grad:
                outs.append(None)
            elif p is typ and styp is not None:
                outs.append(None)              # written by the LayerNorm backward's column reduction above
            elif _sink(p) is not None:
                _mark_touched(p)
                deferred.append((make, _sink(p)))
                outs.append(None)
            else:
                t = torch.zeros(p.shape, dtype=torch.float32, device=dy.device)
                make(t)
                outs.append(t.to(p.dtype))
        if deferred:
            # every write into a parameter's gradient sink goes through the weight-gradient stream: the word table also
            # receives the tied MLM decoder's dW there (a deferred, non-atomic read-modify-write), the type table the
            # panorama branch's row-1 gradient -- one stream keeps the writers of a sink in program order
            WgradStream.submit(dy.device, lambda: [m(t) for m, t in deferred], dz2, dzf, ids, dy)
        return (None, outs[0], outs[1], outs[2], rg, rb, None, None, None, None, None, None)


def embed_sum_layernorm(ids, word, pos, typ, gamma, beta, eps, type_index=0, padding_idx=None):
    """BertEmbeddings (vilmodel.py:62-77): LN(word[ids] + pos[0..L) + type[type_index]).  ``padding_idx``: rows of the
    word table that receive no lookup gradient (nn.Embedding(padding_idx=0), vilmodel.py:50)."""
    return _EmbedLN.apply(ids, word, pos, typ, gamma, beta, eps, type_index, _compute(word), _compute(pos),
                          _compute(typ), -1 if padding_idx is None else int(padding_idx))


# ----------------------------------------------------------------------------- K6 segment gather
class SegmentCSR:
    """Host-built CSR (and its transpose) describing out[r] = sum_e w[e] * src[idx[e]].

    ``capacity`` (entries) fixes the size of the device arrays, so that a later batch of the same shape bucket can be
    written into the SAME buffers (``update``) -- the kernels only read the ranges the row pointers describe."""

    def __init__(self, rowptr, idx, w, n_src, device, capacity=None):
        self.n_out, self.n_src = len(rowptr) - 1, int(n_src)
        self.capacity = int(capacity) if capacity is not None else len(idx)
        pack, packw = self._pack(rowptr, idx, w)
        di = torch.from_numpy(pack).to(device, non_blocking=True)
        dw = torch.from_numpy(packw).to(device, non_blocking=True)
        self._di, self._dw = di, dw
        n0, n1, n2 = self.n_out + 1, self.capacity, self.n_src + 1
        self.rowptr, self.idx = di[:n0], di[n0:n0 + n1]
        self.t_rowptr, self.t_idx = di[n0 + n1:n0 + n1 + n2], di[n0 + n1 + n2:]
        self.w, self.t_w = dw[:n1], dw[n1:]

    def _pack(self, rowptr, idx, w):
        import numpy as np
        rowptr = np.asarray(rowptr, dtype=np.int32)
        idx = np.asarray(idx, dtype=np.int32)
        w = np.asarray(w, dtype=np.float32)
        assert len(rowptr) == self.n_out + 1 and len(idx) <= self.capacity, "segment CSR does not fit its buffers"
        # transpose: for each src row, the (out row, weight) pairs that read it
        out_of_e = np.repeat(np.arange(self.n_out, dtype=np.int32), np.diff(rowptr))
        order = np.argsort(idx, kind="stable")
        t_rowptr = np.zeros(self.n_src + 1, dtype=np.int32)
        np.add.at(t_rowptr, idx + 1, 1)
        t_rowptr = np.cumsum(t_rowptr).astype(np.int32)
        pad = np.zeros(self.capacity - len(idx), dtype=np.int32)
        padw = pad.astype(np.float32)
        pack = np.concatenate([rowptr, idx, pad, t_rowptr, out_of_e[order], pad]).astype(np.int32)
        packw = np.concatenate([w, padw, w[order], padw]).astype(np.float32)
        return pack, packw

    def update(self, rowptr, idx, w):
        """Write another aggregation of the same shape (rows, sources, <= capacity entries) into the device arrays."""
        pack, packw = self._pack(rowptr, idx, w)
        self._di.copy_(torch.from_numpy(pack), non_blocking=True)
        self._dw.copy_(torch.from_numpy(packw), non_blocking=True)


class _SegmentWsum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, csr):
        assert src.is_contiguous() and src.dim() == 2 and src.shape[0] == csr.n_src
        out = torch.empty(csr.n_out, src.shape[1], dtype=src.dtype, device=src.device)
        call("bevbert_segment_wsum", ptr(src), ptr(csr.rowptr), ptr(csr.idx), ptr(csr.w), ptr(out), csr.n_out,
             src.shape[1], dtype_code(src), stream())
        ctx.csr = csr
        return out

    @staticmethod
    def backward(ctx, dout):
        csr = ctx.csr
        dout = dout.contiguous()
        dsrc = torch.empty(csr.n_src, dout.shape[1], dtype=dout.dtype, device=dout.device)
        call("bevbert_segment_wsum", ptr(dout), ptr(csr.t_rowptr), ptr(csr.t_idx), ptr(csr.t_w), ptr(dsrc), csr.n_src,
             dout.shape[1], dtype_code(dout), stream())
        return dsrc, None


def segment_wsum(src, csr):
    return _SegmentWsum.apply(src, csr)


# ----------------------------------------------------------------------------- K1 BEV splat (no gradient)
def pixel_scale(hw, device, vfov=math.radians(90)):
    """((u + .5 - c) / f) in fp32 exactly as bev_utils.py:91-137 builds it (f = hw / (2 tan(vfov/2)), c = hw/2)."""
    f = torch.tensor(hw / (2.0 * math.tan(vfov / 2.0)), dtype=torch.float32)
    c = torch.tensor(hw / 2.0, dtype=torch.float32)
    return ((torch.arange(hw, dtype=torch.float32) + 0.5 - c) / f).to(device)


@torch.no_grad()
def bev_lift_bin(depths, T_c2w, T_w2c, S_w2c, pix, dim, res, depth_scale=10.0, y_clip=0.5):
    B, V = depths.shape[0], depths.shape[1]
    hw = depths.shape[-1]
    P = V * hw * hw
    dev = depths.device
    cell = torch.empty(B, P, dtype=torch.int32, device=dev)
    order = torch.zeros(B, P, dtype=torch.int32, device=dev)
    cell_start = torch.empty(B, dim * dim + 1, dtype=torch.int32, device=dev)
    f = lambda t: t.contiguous().float()
    d, a, b_, c_ = f(depths), f(T_c2w), f(T_w2c), f(S_w2c)
    call("bevbert_bev_lift_bin", ptr(d), ptr(a), ptr(b_), ptr(c_), ptr(pix), B, V, hw, float(depth_scale), dim,
         float(res), float(y_clip), ptr(cell), ptr(order), ptr(cell_start), stream())
    return cell, order, cell_start


@torch.no_grad()
def bev_bin_points(points, drop_mask, dim, res, y_clip=0.5):
    B, P = points.shape[0], points.shape[1]
    dev = points.device
    cell = torch.empty(B, P, dtype=torch.int32, device=dev)
    order = torch.zeros(B, P, dtype=torch.int32, device=dev)
    cell_start = torch.empty(B, dim * dim + 1, dtype=torch.int32, device=dev)
    pts = points.contiguous().float()
    dm = drop_mask.contiguous().to(torch.uint8)
    call("bevbert_bev_bin_points", ptr(pts), ptr(dm), B, P, dim, float(res), float(y_clip), ptr(cell), ptr(order),
         ptr(cell_start), stream())
    return cell, order, cell_start


@torch.no_grad()
def bev_splat_mean(feat, order, cell_start, K, out_dtype=None, sems=None, n_classes=40, rows=None):
    """feat (B,P,C) f32/bf16/f16 -> (B,K,C); sems: (B,P) uint8 ids or (B,P,S) float64 one-hot or None.
    rows (B,) or (B,R) int32: feat / sems are (N,P0,...) stores (feature_store.GridFeatureStore) and sample b's points
    are the R store rows rows[b] back to back (P = R * P0 = order.shape[1])."""
    C = feat.shape[-1]
    B, P = order.shape
    R = 1 if rows is None or rows.dim() == 1 else rows.shape[1]
    assert P == R * feat.shape[1] if rows is not None else feat.shape[:2] == (B, P)
    assert feat.is_contiguous() if rows is not None else True
    feat = feat.contiguous()
    out_dtype = out_dtype or (feat.dtype if feat.dtype != torch.float16 else torch.float32)
    out = torch.empty(B, K, C, dtype=out_dtype, device=feat.device)
    sem_ids = sem_dense = out_sem = out_mask = None
    S = n_classes
    if sems is not None:
        if sems.dim() == 2:
            sem_ids = sems.contiguous().to(torch.uint8)
            assert sem_ids.shape[1] * R == P and (rows is not None or sem_ids.shape[0] == B)
        else:
            sem_dense = sems.contiguous().to(torch.float64)
            S = sems.shape[-1]
        out_sem = torch.empty(B, K, S, dtype=torch.uint8, device=feat.device)
        out_mask = torch.empty(B, K, dtype=torch.uint8, device=feat.device)
    call("bevbert_bev_splat_mean", ptr(feat), dtype_code(feat), ptr(order), ptr(cell_start), ptr(out),
         dtype_code(out_dtype), B, P, K, C, ptr(sem_ids), ptr(sem_dense), S, ptr(out_sem), ptr(out_mask),
         ptr(rows.contiguous() if rows is not None else None), R, stream())
    return out, out_sem, out_mask


def dropout_keep_mask(n, drop_p, seed, offset, device):
    out = torch.empty(n, dtype=torch.uint8, device=device)
    call("bevbert_dropout_keep_mask", ptr(out), n, float(drop_p), int(seed), int(offset), stream())
    return out.bool()


def attn_drop_bits(B, nh, Lq, Lk, drop_p, seed, offset, device):
    """Keep-bit workspace of one attention call ([forward layout | backward layout], int64 words); see
    include/bevbert_hip.h bevbert_attn_drop_bits."""
    bits = torch.empty(_drop_bits_words(B, nh, Lq, Lk), dtype=torch.int64, device=device)
    call("bevbert_attn_drop_bits", ptr(bits), B, nh, Lq, Lk, float(drop_p), int(seed), int(offset), stream())
    return bits


def gemm_plan_count():
    """Number of hipBLASLt plans the library holds (grows when a new GEMM problem shows up: a new shape bucket)."""
    return int(lib.load().bevbert_gemm_plan_count())
