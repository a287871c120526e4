"""Import alias: ``import zklc_amd`` == the package directory
``zk-light-client-implementation_amd`` (whose name is not a Python identifier).
``zklc_amd.x`` and ``from zklc_amd import x`` give the SAME module object (one set of classes: an exception raised
by a module imported one way is caught by its name imported the other way)."""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys

_REAL = "zk-light-client-implementation_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`zklc_amd.sub[.sub]` -> the module object of `zk-light-client-implementation_amd.sub[.sub]`"""

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(__name__ + "."):
            return None
        real = importlib.import_module(_REAL + fullname[len(__name__):])
        spec = importlib.machinery.ModuleSpec(fullname, self, is_package=hasattr(real, "__path__"))
        spec.loader_state = (real, real.__spec__, getattr(real, "__loader__", None))
        return spec

    def create_module(self, spec):
        return spec.loader_state[0]

    def exec_module(self, module):
        # importlib has just stamped the ALIAS spec / loader on the real module: put the real ones back, so that
        # __package__ == __spec__.parent holds (relative imports, importlib.reload, inspect keep working)
        alias = module.__spec__
        if alias is not None and getattr(alias, "loader_state", None) and alias.loader_state[0] is module:
            module.__spec__, module.__loader__ = alias.loader_state[1], alias.loader_state[2]


sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
