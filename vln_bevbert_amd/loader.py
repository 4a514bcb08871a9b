"""Feeding the captured step: shape buckets, double-buffered static batches, a copy stream.

The reference overlaps the host->device copy of batch i+1 with the step on batch i through ``PrefetchLoader``
(pretrain_src/data/loader.py:78-120: ``move_to_cuda(non_blocking=True)`` one batch ahead) and draws the task of every
step in ``MetaLoader`` (loader.py:18-62).  Everything it then does on the HOST inside the forward -- positions of the
masked tokens, the SAP fusion table, the global-map aggregation lists (vilmodel.py:632-666, pretrain_cmt.py:339-356) --
is loader work here (``StaticBatch._host_side``), done by the producer thread of ``StreamingLoader`` next to the copies:

    source iterator (task, host batch)            main thread
          |  producer thread                           |
          v                                            v
    BucketManager.acquire: shape signature -> buffer set of that bucket (round robin over ``depth`` sets; waits for the
        step that last read the set), ``StaticBatch.load`` on the COPY stream, ``ready`` event
          |------------------ queue (depth - 1) ------->|  current stream waits ``ready``; trainer.step(task, sb)
          |<----------------- release(sb): ``done`` event on the compute stream

A bucket is what a hipGraph is captured on: the signature is (task, B, text shape, panorama shape, map width, padded row
counts); buffer sets of a bucket are separate ``StaticBatch`` objects, each with its own captured step.  Buckets are
kept LRU up to ``max_buckets``; what the manager did (buckets created / evicted, refills, bytes copied) is counted so
that ``bench.py --stream`` can report it.
"""
import collections
import queue
import os
import threading
import time

import torch

from .hwqueues import pick_copy_stream, shares_hw_queue  # noqa: F401 (re-exported)
from .static_step import StaticBatch


class BucketManager:
    """Buffer sets per shape bucket.  Ownership protocol of a buffer set (a refill must never touch buffers a
    step may still read, and with ``depth`` sets and ``prefetch`` queued batches up to ``prefetch + 2`` batches are in
    flight): ``acquire`` hands a set out with ``in_use = True``; only ``release`` (called by the consumer AFTER it has
    enqueued the step that reads the set) clears the flag and records the ``done`` event; a later ``acquire`` that
    lands on the same set blocks on a condition variable until the flag is clear, then makes the copy stream wait for
    ``done`` -- so neither the device buffers nor the host-side fields (CPU copies, row counts, CSR) of a set change
    while the consumer holds it or the GPU still reads it."""

    WAIT_TIMEOUT_S = 120.0       # a consumer that never releases is a bug: say so instead of hanging forever

    def __init__(self, cfg, device, depth=2, max_buckets=16, grid_store=None):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.cfg, self.device, self.depth, self.max_buckets = cfg, device, depth, max_buckets
        self.grid_store = grid_store
        self.buckets = collections.OrderedDict()        # signature -> {"sets": [StaticBatch], "next": int}
        self.copy_stream_probe = None
        self.copy_stream = pick_copy_stream(self.device, self) if self.device.type == "cuda" else None
        self.stats = collections.Counter()
        self.trace = None               # set to [] to collect (t_enter, wait for the buffer set [s], t_exit, signature) per acquire
        self._cv = threading.Condition()
        self.stopped = False            # set by stop(): a producer waiting for a buffer set gives up promptly

    def _on_copy_stream(self):
        return torch.cuda.stream(self.copy_stream) if self.copy_stream is not None else _NullCtx()

    def _wait_free(self, sb, stop=None):
        """Block until the consumer has released ``sb``; then order the copy stream behind the step that read it.
        ``stop``: the calling loader's own stop token (a threading.Event)."""
        t0 = time.perf_counter()
        with self._cv:
            while getattr(sb, "in_use", False):
                if self.stopped or (stop is not None and stop.is_set()):
                    raise LoaderStopped("the consumer of this loader is gone" if not self.stopped else
                                        "BucketManager.stop(): the manager was shut down")
                if not self._cv.wait(timeout=1.0) and time.perf_counter() - t0 > self.WAIT_TIMEOUT_S:
                    raise RuntimeError("BucketManager: a buffer set was never released (call release(sb) after the step "
                                       "on it has been enqueued); depth must be >= 2 for the producer to run ahead")
        done = getattr(sb, "done", None)
        if done is not None:
            if self.copy_stream is not None:
                self.copy_stream.wait_event(done)       # device order: refill after the step that last read the buffers
            done.synchronize()                          # host order: the host-side fields are rewritten below too
        self.stats["wait_s"] += time.perf_counter() - t0      # time spent waiting (consumer + GPU), not loader work

    def acquire(self, task, batch, grid_keys=None, stop=None):
        """Device-resident StaticBatch holding ``batch`` (host tensors, collate schema).  Blocks while the buffer set it
        is about to overwrite is still held by the consumer or read by an earlier step; ``stop`` (a threading.Event owned
        by the calling loader) ends such a wait with LoaderStopped."""
        t0 = time.perf_counter()
        host = StaticBatch.plan(self.cfg, task, batch)
        sig = host["signature"]
        b = self.buckets.get(sig)
        if b is None:
            b = {"sets": [], "next": 0}
            self.buckets[sig] = b
            self.stats["buckets_created"] += 1
            while len(self.buckets) > self.max_buckets:
                old_sig = next(iter(self.buckets))
                old = self.buckets[old_sig]
                for sb in old["sets"]:
                    self._wait_free(sb, stop)
                del self.buckets[old_sig]
                self.stats["buckets_evicted"] += 1
        self.buckets.move_to_end(sig)
        if len(b["sets"]) < self.depth:                  # first uses of a bucket: allocate another buffer set
            with self._on_copy_stream():
                sb = StaticBatch(self.cfg, task, batch, self.device, grid_store=self.grid_store, grid_keys=grid_keys)
                sb.ready = self._record()
            sb.done = None
            b["sets"].append(sb)
            self.stats["buffer_sets_allocated"] += 1
        else:
            sb = b["sets"][b["next"] % self.depth]
            b["next"] += 1
            self._wait_free(sb, stop)
            with self._on_copy_stream():
                sb.load(batch, grid_keys=grid_keys, host=host)
                sb.ready = self._record()
            self.stats["refills"] += 1
        with self._cv:
            sb.in_use = True
        self.stats["bytes_h2d"] += sum(v.numel() * v.element_size() for k, v in batch.items()
                                       if torch.is_tensor(v) and not (self.grid_store is not None and k in ("rgbs", "depths", "sems")))
        self.stats["loader_s"] += time.perf_counter() - t0
        if self.trace is not None:
            self.trace.append((t0, self.stats["wait_s"], time.perf_counter(), hash(sig) & 0xffff))
        return sb

    def _record(self):
        if self.copy_stream is None:
            return None
        ev = torch.cuda.Event()
        ev.record(self.copy_stream)
        return ev

    def stop(self):
        """Shut the MANAGER down for good: every producer waiting for a buffer set gives up.  A loader that ends does not
        call this (the flag used to be set by StreamingLoader.close and never cleared, so the second loader on
        the same manager -- the next epoch, same buckets, same captured graphs -- lost batches silently at its first
        back-pressure wait); it sets its own token and calls wake()."""
        with self._cv:
            self.stopped = True
            self._cv.notify_all()

    def wake(self):
        """Make waiting producers re-check their stop tokens."""
        with self._cv:
            self._cv.notify_all()

    def release(self, sb):
        """Call right after the step on ``sb`` has been enqueued: its buffers may be refilled once that step is done."""
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            sb.done = ev
        with self._cv:
            sb.in_use = False
            self._cv.notify_all()

    def captured_graphs(self):
        return sum(sb.graph is not None for b in self.buckets.values() for sb in b["sets"])


def quiet_gc():
    """Call once after warm-up (model built, buffer sets allocated, steps captured).  Python's cyclic collector stops every
    thread of the process; a generation-2 pass over the objects a training process has by then (modules, parameters, captured
    graphs, task tables, plans) takes 80 - 90 ms (measured, round 6: `gc.callbacks` around the worst idle gaps of a
    live-loader run), i.e. four to five steps during which the producer thread builds nothing -- more than the two buffer
    sets per bucket can hide.  ``gc.freeze()`` moves everything alive now out of the collector's reach; what later steps
    allocate is still collected, in passes that take microseconds."""
    import gc
    gc.collect()
    gc.freeze()


class LoaderStopped(RuntimeError):
    """Raised inside the producer thread when BucketManager.stop() ends a wait for a buffer set."""


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class StreamingLoader:
    """Iterate ``(task, StaticBatch)``: a producer thread stays ``prefetch`` batches ahead of the consumer."""

    def __init__(self, source, manager, prefetch=1):
        self.source, self.manager = source, manager
        self.q = queue.Queue(maxsize=max(1, prefetch))
        self._stop = False
        self._stop_token = threading.Event()        # this loader's own: the manager outlives it (next epoch, same buckets)
        self.error = None
        self.thread = threading.Thread(target=self._produce, daemon=True)

    def _produce(self):
        try:
            if self.manager.device.type == "cuda":
                torch.cuda.set_device(self.manager.device)
            st = self.manager.stats
            src = iter(self.source)
            while not self._stop:
                t0 = time.perf_counter()
                try:
                    item = next(src)
                except StopIteration:
                    break
                st["source_s"] += time.perf_counter() - t0          # waiting for the collate workers / the dataset
                task, batch = item[0], item[1]
                keys = item[2] if len(item) > 2 else None
                sb = self.manager.acquire(task, batch, grid_keys=keys, stop=self._stop_token)
                t0 = time.perf_counter()
                self.q.put((task, sb))
                st["queue_s"] += time.perf_counter() - t0           # waiting for the consumer to take the previous batch
        except LoaderStopped as e:
            if not self._stop:          # nobody closed THIS loader: the stream must not end as if the source were exhausted
                self.error = e
            # else: close() while this thread waited for a buffer set -- not an error
        except Exception as e:          # noqa: BLE001 -- surfaced in the consumer thread
            self.error = e
        self.q.put(None)

    def __iter__(self):
        self.thread.start()
        while True:
            item = self.q.get()
            if item is None:
                if self.error is not None:
                    raise self.error
                return
            task, sb = item
            if sb.ready is not None:
                torch.cuda.current_stream(self.manager.device).wait_event(sb.ready)
            yield task, sb

    def release(self, sb):
        self.manager.release(sb)

    def close(self):
        """Stop the producer.  Items still queued are released (nobody will run a step on them), and a producer that is
        waiting for a buffer set the consumer never released -- it stopped early, or raised before release(sb) -- is woken
        up instead of sitting out the two-minute timeout of that wait."""
        self._stop = True
        self._stop_token.set()
        self.manager.wake()
        while self.thread.is_alive():
            try:
                item = self.q.get(timeout=0.1)
            except queue.Empty:
                continue
            if isinstance(item, tuple) and len(item) == 2 and hasattr(item[1], "in_use"):
                self.manager.release(item[1])
