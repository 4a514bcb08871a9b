"""Deterministic, implementation-independent weight rule.

No checkpoints can be shipped (no network, 955 MB fp32), so every side that needs
"the same model" -- the golden-vector generator that imports the reference
(tests/golden/make_golden.py), the CPU oracle, and this package -- fills a
``state_dict`` from nothing but the key names and shapes:

    seed(key) = crc32(key) ^ base_seed        (per-tensor torch.Generator)
    *.LayerNorm.weight / norm*.weight / "*layer_norm.weight" / "net.2.weight" /
    "*_embeddings.1.weight"                   -> 1 + 0.1*N(0,1)
    any other *.weight                        -> 0.02*N(0,1)   (BERT init range,
                                                 configs/r2r_model.json "initializer_range")
    *.bias / predictions.bias                 -> 0.02*N(0,1)   (non-zero on purpose:
                                                 a dropped bias must fail parity)

Tied tensor: mlm_head.predictions.decoder.weight aliases
bert.embeddings.word_embeddings.weight (pretrain_src/model/pretrain_cmt.py:109-112).
"""
import re
import zlib

import torch

_LN_PAT = re.compile(
    r"(LayerNorm|layer_norm|\.norm\d?|_embeddings\.1|\.net\.2)\.weight$"
)
TIED = {"mlm_head.predictions.decoder.weight": "bert.embeddings.word_embeddings.weight"}


def is_layernorm_weight(key: str) -> bool:
    return _LN_PAT.search(key) is not None


def fill_tensor(key: str, shape, base_seed: int = 0) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ base_seed) & 0x7FFFFFFF)
    x = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if key.endswith(".weight") and is_layernorm_weight(key):
        return 1.0 + 0.1 * x
    return 0.02 * x


def fill_state_dict(shapes: dict, base_seed: int = 0) -> dict:
    """shapes: {key: shape}. Returns {key: fp32 CPU tensor}, with tied keys aliased."""
    out = {}
    for k, shp in shapes.items():
        if k in TIED:
            continue
        out[k] = fill_tensor(k, shp, base_seed)
    for k, src in TIED.items():
        if k in shapes:
            out[k] = out[src]
    return out
