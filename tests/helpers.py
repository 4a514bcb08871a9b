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


def oracle_train(cfg, sd0, tasks, batches, lr_fn, wd=0.01, max_norm=5.0):
    """CPU oracle of the reference's hot loop with dropout disabled (train_r2r.py:247-313)."""
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    sd["mlm_head.predictions.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    names = [k for k in sd if k != "mlm_head.predictions.decoder.weight"]
    state = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k]), [0]) for k in names}
    from oracle import bevbert_ref as R
    losses = []
    for step, (task, b) in enumerate(zip(tasks, batches), 1):
        loss = R.pretrain_forward(sd, cfg, b, task).mean()
        grads = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
        gd = {k: g for k, g in zip(names, grads)}
        for k in names:                               # zero_grad() keeps zeros for params that ever had a grad
            if gd[k] is None and state[k][2][0] > 0:
                gd[k] = torch.zeros_like(sd[k])
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in gd.values() if g is not None)).float()
        coef = min(1.0, max_norm / (float(total) + 1e-6))
        with torch.no_grad():
            for k in names:
                if gd[k] is None:
                    continue
                m, v, n = state[k]
                n[0] += 1
                R.adamw_step(sd[k], gd[k] * coef, m, v, n[0], lr_fn(step), 0.0 if R.no_decay_key(k) else wd)
        losses.append(float(loss.detach()))
    return losses
