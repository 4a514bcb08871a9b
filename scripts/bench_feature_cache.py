#!/usr/bin/env python3
"""Throughput of filling the device-resident grid-feature store from the sharded cache (row f2).
Writes a synthetic cache of N viewpoints (3.6 MB each: 12 x 196 x 768 fp16 + depths + class ids, the R2R layout) under
--dir, drops nothing from the page cache (a second pass shows the warm figure), loads it twice with
feature_cache.load_store and prints one JSON line per pass.  R2R's 10 567 viewpoints are 38 GB."""
import argparse
import json
import os
import shutil
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vln_bevbert_amd import feature_cache  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1200)
    ap.add_argument("--dir", default="/tmp/bevbert_cache_bench")
    ap.add_argument("--shard", type=int, default=128)
    a = ap.parse_args()
    shutil.rmtree(a.dir, ignore_errors=True)
    rng = np.random.default_rng(0)
    base = rng.standard_normal((12, 196, 768)).astype(np.float16)

    def items():
        for i in range(a.n):
            yield (f"s{i // 90}_v{i}", np.roll(base, i, axis=1), (rng.random((12, 14, 14)) * 0.6).astype(np.float32),
                   rng.integers(0, 40, (12, 14, 14)).astype(np.uint8))
    t0 = time.perf_counter()
    feature_cache.write_shards(items(), a.dir, shard_size=a.shard)
    t_write = time.perf_counter() - t0
    size = sum(os.path.getsize(os.path.join(a.dir, f)) for f in os.listdir(a.dir))
    print(json.dumps({"wrote_GB": round(size / 1e9, 2), "viewpoints": a.n, "seconds": round(t_write, 1)}), flush=True)
    for label in ("first pass (files just written: page cache warm)", "second pass"):
        stats = {}
        store = feature_cache.load_store(a.dir, "cuda", stats=stats)
        torch.cuda.synchronize()
        stats["pass"] = label
        stats["store_GiB"] = round(store.nbytes() / 2 ** 30, 2)
        stats["r2r_38GB_would_take_s"] = round(38e9 / (stats["GBps"] * 1e9), 1)
        print(json.dumps(stats), flush=True)
        del store
    shutil.rmtree(a.dir, ignore_errors=True)


if __name__ == "__main__":
    main()
