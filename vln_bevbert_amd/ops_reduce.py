"""Deferred work of a backward pass (ops.py re-exports everything here): the weight-gradient stream, the scratch ring for
partial sums and the batched second stages of all two-stage reductions (``ReduceQueue``: one bevbert_multi_finalize and one
bevbert_multi_accum launch per backward)."""
import math
import os as _os

import torch

from . import lib
from .lib import stream
from .ops_core import ATTN_BITS, Branches, RT, call


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
        return cls.enabled and RT.trace is None and device.type == "cuda"

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
            from .hwqueues import side_stream
            cls.stream = shared if shared is not None else side_stream(producer.device)
            Branches._streams["wgrad"] = cls.stream       # joined by ParamArena.sync / GradReducer like the branches
            cls._events = [torch.cuda.Event() for _ in range(64)]
            cls.streams = [cls.stream]
            for i in range(1, cls.NSTREAMS):              # further streams: batches of deferred work go round robin
                st = side_stream(producer.device)
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
RT.scratch = SCRATCH          # every allocation goes through RT.scratch (a test swaps in a small ring)


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


def _on_launch_stream(fn):
    """Run a torch op on the stream the C-ABI launches currently go to (fallback paths inside a WgradStream section)."""
    if lib._override is None:
        return fn()
    with torch.cuda.stream(WgradStream._target or WgradStream.stream):
        return fn()


_PARTIAL_ROWS = {}


def _partial_rows(rows):
    """number of per-block partial rows the two-stage column reductions produce for `rows` input rows"""
    nb = _PARTIAL_ROWS.get(rows)
    if nb is None:
        nb = _PARTIAL_ROWS[rows] = lib.load().bevbert_colsum_partial_rows(rows)
    return nb
