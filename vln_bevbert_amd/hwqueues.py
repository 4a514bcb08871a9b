"""Which of the process's hardware queues a HIP stream is served by -- measured, because the runtime does not say.

A process on ROCm has GPU_MAX_HW_QUEUES (4) in-order hardware queues; its HIP streams are dealt onto them in creation order.
Two streams on one queue do not overlap: work enqueued on the second waits behind everything enqueued earlier on the first.
For the training step that is harmless between its own side streams (the captured step is replayed through the graph
executor's streams), but it decides whether work issued from OUTSIDE the step overlaps it:

  * the loader's refill (loader.BucketManager.copy_stream): on the compute stream's queue the host->device copies of batch
    k + 1, enqueued during step k, run after step k's AdamW, and step k + 1 starts one copy time late (round 6: 0.54 - 0.59 ms
    idle per step, live-loader rate 0.95 x resident; off that queue 0.998 x);
  * the collective library's stream (ProcessGroupNCCL takes the next stream of torch's pool when it creates a
    communicator): on the compute stream's queue an all-reduce issued from a backward hook runs behind the backward
    kernels enqueued so far AND holds up the ones enqueued after it -- no overlap at all in eager steps.

More queues are not the answer: GPU_MAX_HW_QUEUES = 5 / 6 / 8 (or a stream of its own priority level, which gets a queue of
its own) slow the whole step from 17.6 ms to 23.1 / 24.2 / 27.5 ms (`profiles/r06x_*`); 2 and 3 crash the runtime.
"""
import os

import torch


_PROBE_BUFFERS = {}


def shares_hw_queue(stream, main, device, busy_ms=3.0):
    """True when work enqueued on ``stream`` waits behind work enqueued earlier on ``main``: the two HIP streams are served
    by the same in-order hardware queue (a process has GPU_MAX_HW_QUEUES = 4 of them, streams are dealt onto them in
    creation order).  Probe: ~busy_ms of fills on ``main``, then a 1 KB host->device copy on ``stream``; which finishes
    first, seen from the host."""
    device = torch.device(device)
    bufs = _PROBE_BUFFERS.get(device.index)
    if bufs is None:                                   # one pinned source for all probes (pinning synchronises the device)
        bufs = _PROBE_BUFFERS[device.index] = (torch.empty(1024, dtype=torch.uint8).pin_memory(),
                                               torch.empty(1024, dtype=torch.uint8, device=device))
    src, dst = bufs
    busy = torch.empty(64 << 20, dtype=torch.float32, device=device)      # 256 MB from the caching allocator, returned below
    with torch.cuda.stream(stream):
        dst.copy_(src, non_blocking=True)              # first use of the stream (queue acquisition) outside the probe
    torch.cuda.synchronize(device)
    reps = max(4, int(busy_ms / 0.06))                 # a 256 MB fill takes ~0.06 ms at 4-5 TB/s
    m1, c1 = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(main):
        for _ in range(reps):
            busy.fill_(1.0)
        m1.record(main)
    with torch.cuda.stream(stream):
        dst.copy_(src, non_blocking=True)
        c1.record(stream)
    while not c1.query():
        pass
    shared = m1.query()                                # the copy came out only after everything on `main` had run
    torch.cuda.synchronize(device)
    return shared


def pick_copy_stream(device, owner=None, candidates=6):
    """The loader's copy stream: the first of a few fresh streams that does NOT sit on the compute stream's hardware queue.
    A refill enqueued during step k on a stream that shares the compute stream's queue executes behind ALL of step k's
    packets in that queue, i.e. after AdamW, and step k + 1 -- which waits for it -- starts one copy time late (round 6,
    `BEVBERT_STEP_EVENTS=1 python bench.py`: 0.54 - 0.59 ms idle in front of every step, 3 % of the step).  On any other
    queue the copy lands behind a side branch of the captured step (weight gradients, row sums), which is finished before
    the clip + AdamW tail starts, and the copy hides under that tail.  A stream of its own priority level would get a
    hardware queue of its own -- and a fifth active queue slows the whole step by 4 ms (same measurement, also
    GPU_MAX_HW_QUEUES = 5, 6, 8: 23.1, 24.2, 27.5 ms against 17.6).  BEVBERT_COPY_STREAM_PROBE=0: first stream, unprobed."""
    main = torch.cuda.current_stream(device)
    if os.environ.get("BEVBERT_COPY_STREAM_PROBE", "1") != "1" or torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(device)
    tried = []
    keep = []                                           # rejected candidates stay alive while probing: the next stream
    for _ in range(candidates):                         # must not be handed the queue slot of a released one
        st = torch.cuda.Stream(device)
        keep.append(st)
        shared = shares_hw_queue(st, main, device)
        tried.append(bool(shared))
        if not shared:
            break
    if owner is not None:
        owner.copy_stream_probe = {"shares_compute_queue": tried, "picked": len(tried) - 1 if not tried[-1] else None}
    return keep[-1] if not tried[-1] else keep[0]


_SIDE = {}        # device index -> side streams handed out so far (kept distinct from each other while queues last)


def side_stream(device, candidates=8):
    """A stream for work that is meant to overlap the compute stream (deferred weight gradients, keep-bit generation,
    the reducer, the arena fill): a fresh stream that does not share the compute stream's hardware queue and, while there
    are queues left (4 per process), none of the side streams handed out before.  Eager steps depend on it -- the same
    eager step measured 20.0 and 22.0 ms with two positions of torch's stream pool (`profiles/r06ab_*`); captured steps are
    replayed through the graph executor's own streams and do not.  Inside a stream capture, or with
    BEVBERT_COPY_STREAM_PROBE=0, the next pool stream is returned unprobed."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if os.environ.get("BEVBERT_COPY_STREAM_PROBE", "1") != "1" or torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(device)
    main = torch.cuda.current_stream(device)
    mine = _SIDE.setdefault(device.index, [])
    keep, fallback, picked = [], None, None
    for _ in range(candidates):
        st = torch.cuda.Stream(device)
        keep.append(st)
        if shares_hw_queue(st, main, device):
            continue
        if any(shares_hw_queue(st, other, device) for other in mine[:3]):
            fallback = fallback or st                   # off the compute queue, but beside another side stream
            continue
        picked = st
        break
    picked = picked or fallback or keep[0]
    mine.append(picked)
    return picked


def steer_stream_pool(device, like=None, limit=64):
    """Call right before the FIRST collective of a process group on ``device`` (ProcessGroupNCCL takes the next stream of
    torch's round-robin stream pool for the communicator it creates then).  Walks the pool once (it is cyclic: the walk ends
    when the first stream comes around again), probes every stream, and leaves the pool positioned so that the NEXT stream
    handed out is a wanted one: with ``like`` a stream that shares ``like``'s hardware queue (the reducer's side stream,
    which carries nothing but the event edges of the same collectives), without it any stream off the compute stream's
    queue.  Returns {"pool": n, "wanted": [...], "next": j} (None off-GPU or with BEVBERT_STEER_POOL=0).
    Verify with train.GradReducer.collectives_wait_behind_compute()."""
    device = torch.device(device)
    if device.type != "cuda" or os.environ.get("BEVBERT_STEER_POOL", "1") != "1" or torch.cuda.is_current_stream_capturing():
        return None
    main = torch.cuda.current_stream(device)

    def wanted(st):
        if like is not None:
            return st.cuda_stream == like.cuda_stream or bool(shares_hw_queue(st, like, device))
        return not shares_hw_queue(st, main, device)

    first = torch.cuda.Stream(device)
    handles, ok, keep = [first.cuda_stream], [wanted(first)], [first]
    for _ in range(limit):
        st = torch.cuda.Stream(device)
        if st.cuda_stream == handles[0]:
            break                                        # wrapped around: the next draw is pool position 1
        keep.append(st)
        handles.append(st.cuda_stream)
        ok.append(wanted(st))
    else:
        return {"pool": None, "wanted": ok, "next": None}
    n = len(handles)
    j = next((k for k in range(1, n + 1) if ok[k % n]), None)
    if j is None:
        return {"pool": n, "wanted": ok, "next": None}
    for _ in range(j - 1):
        torch.cuda.Stream(device)
    return {"pool": n, "wanted": ok, "next": j % n}
