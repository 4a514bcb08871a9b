"""vln_bevbert_amd: MI355X-native (gfx950) implementation of BEVBert's cross-modal transformer hot path.

Public surface (mirrors the reference's module boundary):
    GlocalTextPathCMT, GlocalTextPathCMTPreTraining      (pretrain_src/model/{vilmodel,pretrain_cmt}.py)
    GlocalTextPathNavCMT, VLNBert                        (map_nav_src/models/{vilmodel,model}.py)
    BevBertConfig, ParamArena, PretrainTrainer, synthetic batches
Compute goes through libbevbert_hip.so (include/bevbert_hip.h); there is no CPU / eager fallback.
"""
__version__ = "0.1.0"

import os as _os

# hipBLASLt's gfx950 kernels are stream-K capable: with few output tiles their workgroups split the K loop and SPIN on
# flags written by peer workgroups, assuming the whole grid is resident.  This package runs library GEMMs on three or
# four streams at once (main, model branch, weight-gradient streams); two such kernels can then each hold part of the
# chip while waiting for peers that cannot be scheduled -- the "three-stream stall" of round 1, reproduced in round 2
# (BEVBERT_SPLITK_MAX=1 with the captured stream layout never returns; with this variable set it runs).  Data-parallel
# grids have no inter-workgroup waits; the measured cost on the default configuration is within run-to-run noise
# (18.77 vs 18.66 ms/step).  Set before the library makes its first launch; export TENSILE_STREAMK_DATA_PARALLEL=0 to undo.
_os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")

from .config import BevBertConfig  # noqa: F401


def __getattr__(name):
    # heavy modules are imported lazily so that `import vln_bevbert_amd` stays cheap on hosts without a GPU
    import importlib
    table = {
        "GlocalTextPathCMT": "vilmodel", "GlocalTextPathCMTPreTraining": "pretrain_cmt",
        "GlocalTextPathNavCMT": "nav_model", "VLNBert": "nav_model", "ParamArena": "arena",
        "PretrainTrainer": "train", "GradReducer": "train",
    }
    if name in table:
        return getattr(importlib.import_module(f"{__name__}.{table[name]}"), name)
    raise AttributeError(name)
