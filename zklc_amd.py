"""Import alias: ``import zklc_amd`` == the package directory
``zk-light-client-implementation_amd`` (whose name is not a Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("zk-light-client-implementation_amd")
sys.modules[__name__] = _pkg
