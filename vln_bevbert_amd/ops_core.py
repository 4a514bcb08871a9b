"""Runtime state of the torch-facing wrappers (ops.py re-exports everything here): the per-step dropout state and launch
streams (``RT``), the keep-bit planner of the attention sites, the model-branch streams, the traced C-ABI call, and the
parameter-arena helpers (gradient sinks, compute copies) every autograd Function of the package uses."""
import os as _os

import torch

from . import lib
from .lib import call as _raw_call
from .lib import ptr, stream


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
    trace = None
    paths = {}            # "bevbert_attn_fwd[Lq=..,Lk=..] -> kernel": count, filled while ``trace`` is armed
    scratch = None

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
            from .hwqueues import side_stream
            self.stream = side_stream(torch.device("cuda", dev))
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
                from .hwqueues import side_stream
                Branches._streams[key] = side_stream(device)
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


# RT.trace: dict name -> [(start_event, end_event, args)] while bench.py's kernel-timing pass is active (None otherwise);
# RT.scratch: the scratch ring of the partial sums (ops_reduce.py installs it).  Both live on the shared runtime object so
# that a caller (a test, bench.py) that swaps them reaches every module of the package.


_ROWS_ARG = {"bevbert_layernorm_res32_fwd": 11, "bevbert_layernorm_res32_bwd": 12, "bevbert_bias_dropout_residual_layernorm_fwd": 9,
             "bevbert_layernorm_bwd": 11, "bevbert_layernorm_bwd_add": 12, "bevbert_bias_gelu_fwd": 3, "bevbert_bias_gelu_bwd": 6}


def call(name, *args):
    """C-ABI call; when RT.trace is armed, bracket the launch with HIP events on the launching stream."""
    if RT.trace is None:
        return _raw_call(name, *args)
    key = name
    if name == "bevbert_attn_fwd":
        key = f"{name}[Lq={args[10]},Lk={args[11]}]"
    elif name == "bevbert_attn_bwd":
        key = f"{name}[Lq={args[16]},Lk={args[17]}]"
    elif name in _ROWS_ARG:                     # the row kernels by problem size too (5 120 text rows vs 28 224 BEV rows)
        key = f"{name}[rows={args[_ROWS_ARG[name]]}]"
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    _raw_call(name, *args)
    e.record()
    RT.trace.setdefault(key, []).append((s, e, args))
    if name in ("bevbert_attn_fwd", "bevbert_attn_bwd"):       # which kernel the library picked for this shape
        path = lib.load().bevbert_attn_last_path(int(name.endswith("bwd"))).decode()
        RT.paths[f"{key} -> {path}"] = RT.paths.get(f"{key} -> {path}", 0) + 1


def _gemm(kind, fn, m, n, k):
    """Library GEMM (hipBLASLt via torch); when RT.trace is armed, time it with HIP events keyed by its shape."""
    if RT.trace is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = fn()
    e.record()
    RT.trace.setdefault(f"gemm:{kind}[M={m},N={n},K={k}]", []).append((s, e, (m, n, k)))
    return out


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


_DROP_BITS_WORDS = {}


def _drop_bits_words(B, nh, Lq, Lk):
    key = (B, nh, Lq, Lk)
    n = _DROP_BITS_WORDS.get(key)
    if n is None:
        n = _DROP_BITS_WORDS[key] = lib.load().bevbert_attn_drop_bits_words(B, nh, Lq, Lk)
    return n
