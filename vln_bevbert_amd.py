"""Import shim: the package directory is named ``vln-bevbert_amd`` (not a valid
Python identifier), so ``import vln_bevbert_amd`` resolves here and this module
turns itself into a package whose ``__path__`` is that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "vln-bevbert_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
