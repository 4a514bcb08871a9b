"""Shared test helpers (CPU side)."""
import os

import numpy as np
import torch

from vln_bevbert_amd import weights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def read_shapes(fname):
    shapes = {}
    with open(os.path.join(GOLDEN, fname)) as f:
        for line in f:
            k, shp = line.split(" ", 1)
            shapes[k] = tuple(int(v) for v in shp.strip().strip("()").split(",") if v.strip())
    return shapes


_SD_CACHE = {}


def rule_state_dict(fname):
    """state_dict filled by the key-name weight rule for the key/shape list in tests/golden/<fname>."""
    if fname not in _SD_CACHE:
        _SD_CACHE[fname] = weights.fill_state_dict(read_shapes(fname))
    return _SD_CACHE[fname]


def sub(t, step):
    if torch.is_tensor(t):
        t = t.detach().cpu().numpy()
    return t.reshape(-1)[::step]


def max_abs(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    fin = np.isfinite(a) & np.isfinite(b)
    assert (np.isfinite(a) == np.isfinite(b)).all(), "inf/nan pattern differs"
    assert (a[~fin] == b[~fin]).all() if (~fin).any() else True
    return float(np.abs(a[fin] - b[fin]).max()) if fin.any() else 0.0
