#!/usr/bin/env python3
"""Real HDF5 feature files for the converter tests, written the way the reference's precompute scripts write theirs:

    rgb    precompute_features/grid_mp3d_clip.py:168-180   create_dataset(key, data=fts, dtype='float16', compression='gzip')
    depth  precompute_features/grid_depth.py:122-131       create_dataset(key, data=depth_item)          (native dtype)
    sem    precompute_features/grid_sem.py:146-155         create_dataset(key, data=ids, dtype='uint8', compression='gzip')

and read back the way the reference reads them (map_nav_src/utils/data.py:22-27: ``h5py.File(path, 'r')[key][...]``): the
read-back arrays are the expected values (readback.npz).  Needs h5py; this image has one under /opt/conda:

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_fixtures.py [out_dir = tests/golden/hdf5]

Shapes: 12 views x 14 x 14 cells like the reference's files; 6 feature channels instead of 768 to keep the fixture small."""
import os
import sys

import h5py
import numpy as np

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "hdf5")
os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(20260927)
keys = [f"scan{i}_{vp}" for i, vp in enumerate(["b7a1", "0f3c", "9d22", "41ee", "c005"])]     # written out of name order
with h5py.File(os.path.join(out, "rgb.hdf5"), "w") as f:
    for k in keys:
        fts = (rng.standard_normal((12, 196, 6)) * 3).astype(np.float32)          # the extractor's fp32 output
        f.create_dataset(k, data=fts, dtype='float16', compression='gzip')
with h5py.File(os.path.join(out, "depth.hdf5"), "w") as f:
    for i, k in enumerate(keys):
        d = rng.random((12, 14, 14)) * 10.0                                       # float64 like numpy arithmetic leaves it
        f.create_dataset(k, data=d if i % 2 else d.astype(np.float32))
with h5py.File(os.path.join(out, "sem.hdf5"), "w") as f:
    for k in keys:
        f.create_dataset(k, data=rng.integers(0, 41, (12, 14, 14)), dtype='uint8', compression='gzip')
back = {}
for name in ("rgb", "depth", "sem"):
    with h5py.File(os.path.join(out, name + ".hdf5"), "r") as f:
        back[name + "/keys"] = np.array(list(f.keys()))
        for k in f.keys():
            back[f"{name}/{k}"] = f[k][...]
np.savez_compressed(os.path.join(out, "readback.npz"), **back)
print("h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version, "->", out, sorted(os.listdir(out)))
