"""Sharded on-disk cache of the per-viewpoint grid features.

The reference reads three HDF5 files inside the training loop, one gzip-compressed dataset per viewpoint
(precompute_features/grid_mp3d_clip.py:168-183 writes `vit_b16_224_clip_patch.hdf5`: key '<scan>_<viewpoint>' ->
(12, 196, 768) float16; grid_depth.py:122-131 `depth_14x14.hdf5`: (12, 14, 14) float32; `semantic_14x14.hdf5`: uint8;
read back by pretrain_src/data/dataset.py:110-118 and map_nav_src/utils/data.py:9-29).  h5py + gzip in the loop is what
this replaces: the features are converted ONCE into a few large safetensors shards

    shard_00000.safetensors   rgbs (n, 2352, 768) f16 | depths (n, 12, 14, 14) f32 | sems (n, 2352) u8
    index.json                {"keys": [...], "shard_of": [...], "row_of": [...], "shape": {...}}

that are read sequentially (one big read per shard, no decompression) straight into the device-resident
``feature_store.GridFeatureStore``.  ``convert_hdf5`` reads the HDF5 files with h5py or, where that is missing, through
libhdf5 itself (``hdf5_reader``); ``write_shards`` / ``load_store`` need neither.
"""
import json
import os

import numpy as np
import torch
from safetensors.numpy import load_file, save_file

from .feature_store import GridFeatureStore

INDEX = "index.json"


def write_shards(items, out_dir, shard_size=512):
    """items: iterable of (key, rgbs (V, hw*hw | hw, hw, C) fp16-castable, depths (V, hw, hw), sems (V, hw, hw) ids).
    Writes shards of up to `shard_size` viewpoints and the index; returns the number of viewpoints."""
    os.makedirs(out_dir, exist_ok=True)
    keys, shard_of, row_of, seen = [], [], [], set()
    buf = {"rgbs": [], "depths": [], "sems": []}
    shape = {}

    def flush():
        if not buf["rgbs"]:
            return
        sid = (shard_of[-1] if shard_of else 0)
        save_file({"rgbs": np.stack(buf["rgbs"]), "depths": np.stack(buf["depths"]), "sems": np.stack(buf["sems"])},
                  os.path.join(out_dir, f"shard_{sid:05d}.safetensors"))
        for v in buf.values():
            v.clear()

    for key, rgbs, depths, sems in items:
        rgbs, depths, sems = np.asarray(rgbs), np.asarray(depths), np.asarray(sems)
        V, hw = depths.shape[0], depths.shape[-1]
        C = rgbs.shape[-1]
        cur = {"V": int(V), "hw": int(hw), "C": int(C)}
        if not shape:
            shape.update(cur)
        elif shape != cur:
            raise ValueError(f"viewpoint {key}: shape {cur} differs from the cache's {shape}")
        if key in seen:
            raise ValueError(f"duplicate key {key}")
        seen.add(key)
        sid = len(keys) // shard_size
        if keys and sid != shard_of[-1]:
            flush()
        keys.append(key)
        shard_of.append(sid)
        row_of.append(len(buf["rgbs"]))
        buf["rgbs"].append(rgbs.reshape(V * hw * hw, C).astype(np.float16))
        buf["depths"].append(depths.reshape(V, hw, hw).astype(np.float32))
        buf["sems"].append(sems.reshape(V * hw * hw).astype(np.uint8))
    flush()
    with open(os.path.join(out_dir, INDEX), "w") as f:
        json.dump({"keys": keys, "shard_of": shard_of, "row_of": row_of, "shape": shape, "shard_size": shard_size}, f)
    return len(keys)


def read_index(cache_dir):
    with open(os.path.join(cache_dir, INDEX)) as f:
        return json.load(f)


def load_store(cache_dir, device, keys=None, stats=None):
    """Build a GridFeatureStore from a cache directory (optionally only the viewpoints in `keys`, e.g. one split).

    Per shard: one sequential read (safetensors, no decompression), one copy into a PINNED staging buffer, one
    asynchronous host->device copy out of it.  Two staging buffers alternate, so the file read of shard s+1 overlaps
    the PCIe transfer of shard s.  ``stats`` (a dict) receives bytes, seconds spent reading / staging / waiting and the
    end-to-end GB/s -- the figure DESIGN.md quotes for filling the 38 GB R2R store."""
    import time
    idx = read_index(cache_dir)
    want = None if keys is None else set(keys)
    missing = [] if want is None else sorted(want - set(idx["keys"]))
    if missing:
        raise KeyError(f"{len(missing)} viewpoints are not in the cache, e.g. {missing[:3]}")
    by_shard = {}
    for k, s, r in zip(idx["keys"], idx["shard_of"], idx["row_of"]):
        if want is None or k in want:
            by_shard.setdefault(s, []).append((k, r))
    sh = idx["shape"]
    n_total = sum(len(v) for v in by_shard.values())
    P = sh["V"] * sh["hw"] * sh["hw"]
    device = torch.device(device)
    cuda = device.type == "cuda"
    # the store is allocated once at its final size (38 GB for R2R) and filled shard by shard: no second copy in HBM
    rgbs = torch.empty(n_total, P, sh["C"], dtype=torch.float16, device=device)
    depths = torch.empty(n_total, sh["V"], sh["hw"], sh["hw"], dtype=torch.float32, device=device)
    sems = torch.empty(n_total, P, dtype=torch.uint8, device=device)
    dst = {"rgbs": rgbs, "depths": depths, "sems": sems}
    n_max = max(len(v) for v in by_shard.values())
    stage = [None, None]
    events = [None, None]
    if cuda:
        stage = [{n: torch.empty((n_max,) + tuple(t.shape[1:]), dtype=t.dtype, pin_memory=True) for n, t in dst.items()}
                 for _ in range(2)]
        copy_stream = torch.cuda.Stream(device)
    t_read = t_stage = t_wait = 0.0
    nbytes = 0
    t_all = time.perf_counter()
    out_keys, at = [], 0
    for j, s in enumerate(sorted(by_shard)):
        t0 = time.perf_counter()
        t = load_file(os.path.join(cache_dir, f"shard_{s:05d}.safetensors"))
        t_read += time.perf_counter() - t0
        rows = np.asarray([r for _, r in by_shard[s]], dtype=np.int64)
        out_keys += [k for k, _ in by_shard[s]]
        full = len(rows) == t["rgbs"].shape[0] and np.array_equal(rows, np.arange(len(rows)))
        n = len(rows)
        if cuda:
            t0 = time.perf_counter()
            if events[j & 1] is not None:
                events[j & 1].synchronize()               # the transfer that last used this staging buffer is done
            t_wait += time.perf_counter() - t0
        t0 = time.perf_counter()
        for name in ("rgbs", "depths", "sems"):
            a = t[name] if full else t[name][rows]
            src = torch.from_numpy(np.ascontiguousarray(a))
            nbytes += src.numel() * src.element_size()
            if cuda:
                stage[j & 1][name][:n].copy_(src)
            else:
                dst[name][at:at + n].copy_(src)
        t_stage += time.perf_counter() - t0
        if cuda:
            with torch.cuda.stream(copy_stream):
                for name in ("rgbs", "depths", "sems"):
                    dst[name][at:at + n].copy_(stage[j & 1][name][:n], non_blocking=True)
                events[j & 1] = torch.cuda.Event()
                events[j & 1].record(copy_stream)
        at += n
    if cuda:
        t0 = time.perf_counter()
        copy_stream.synchronize()
        torch.cuda.current_stream(device).wait_stream(copy_stream)
        t_wait += time.perf_counter() - t0
    if stats is not None:
        dt = time.perf_counter() - t_all
        stats.update(bytes=nbytes, shards=len(by_shard), viewpoints=n_total, seconds=round(dt, 3),
                     read_s=round(t_read, 3), stage_s=round(t_stage, 3), wait_s=round(t_wait, 3),
                     GBps=round(nbytes / dt / 1e9, 2))
    return GridFeatureStore(out_keys, rgbs, depths, sems, device)


def convert_hdf5(rgb_file, depth_file, sem_file, out_dir, shard_size=512, reader=None):
    """One-off conversion of the reference's three HDF5 stores (dataset.py:110-118 reads them per sample) into the sharded
    cache.  The files are opened with h5py when it is importable, otherwise through libhdf5's C API
    (``hdf5_reader``: same keys, same element values -- tests/test_host_logic.py checks it on files written by h5py with
    the reference's own ``create_dataset`` calls).  ``reader``: a callable path -> file object, to force one of the two."""
    from . import hdf5_reader
    opener = reader or hdf5_reader.open_file

    def items():
        with opener(rgb_file) as fr, opener(depth_file) as fd, opener(sem_file) as fs:
            for key in fr.keys():
                for name, f in (("depth", fd), ("semantic", fs)):
                    if key not in f:
                        raise KeyError(f"viewpoint {key} of {rgb_file} is missing from the {name} file")
                yield key, fr[key][...], fd[key][...], fs[key][...]
    return write_shards(items(), out_dir, shard_size)
