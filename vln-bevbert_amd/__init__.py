"""vln-bevbert_amd: MI355X-native (gfx950) implementation of BEVBert's cross-modal transformer hot path.

Public surface (mirrors the reference's module boundary):
    GlocalTextPathCMT, GlocalTextPathCMTPreTraining      (pretrain_src/model/{vilmodel,pretrain_cmt}.py)
    GlocalTextPathNavCMT, VLNBert                        (map_nav_src/models/{vilmodel,model}.py)
    BevBertConfig, ParamArena, PretrainTrainer, synthetic batches
Compute goes through libbevbert_hip.so (include/bevbert_hip.h); there is no CPU / eager fallback.
"""
__version__ = "0.1.0"

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
