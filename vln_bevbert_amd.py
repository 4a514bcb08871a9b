"""Import shim: the package directory is named ``vln-bevbert_amd`` (not a valid Python identifier), so
``import vln_bevbert_amd`` lands here and this file swaps itself for the real package loaded from that directory."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "vln-bevbert_amd")
_spec = _ilu.spec_from_file_location("vln_bevbert_amd", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["vln_bevbert_amd"] = _mod
_spec.loader.exec_module(_mod)
