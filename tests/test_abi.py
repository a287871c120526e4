"""The C-ABI library loads and exports every symbol include/zklc.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    hdr = open(os.path.join(ROOT, "include", "zklc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(zklc_[a-z0-9_]+)\s*\(", hdr)))


@pytest.fixture(scope="module")
def lib_path():
    import importlib
    b = importlib.import_module("zk-light-client-implementation_amd.build")
    return b.build(verbose=False)


def test_header_symbols_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert "zklc_ed25519_verify_batch" in names and "zklc_init" in names
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n


def test_python_binding_covers_header(lib_path):
    import zklc_amd
    zklc_amd.load()
    assert sorted(zklc_amd.declared_symbols()) == _declared()


def test_no_gpu_is_an_error_not_a_fallback(lib_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import zklc_amd
    with pytest.raises(zklc_amd.ZklcError) as e:
        zklc_amd.Context(0)
    assert e.value.code == -4  # ZKLC_ERR_NO_DEVICE


def test_strerror_and_version(lib_path):
    import zklc_amd
    lib = zklc_amd.load()
    assert lib.zklc_abi_version() >= 1
    assert lib.zklc_strerror(0) == b"ok"
    assert b"invalid" in lib.zklc_strerror(-1)


def test_import_alias_keeps_the_real_module_specs():
    """`zklc_amd.x` is the module object of `zk-light-client-implementation_amd.x` WITH its own spec -- relative imports
    inside lazily importing functions raise no ImportWarning (__package__ == __spec__.parent), importlib.reload and inspect work"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import importlib, inspect; import zklc_amd; import zklc_amd.pipeline as pl; "
            "import zklc_amd.plonky2.builder as b; "
            "assert pl.__spec__.name == 'zk-light-client-implementation_amd.pipeline' and pl.__package__ == pl.__spec__.parent; "
            "assert sys.modules['zklc_amd.pipeline'] is sys.modules['zk-light-client-implementation_amd.pipeline']; "
            "from zklc_amd.plonky2.recursion import RecursionProver; importlib.reload(pl); "
            "assert inspect.getsourcefile(b).endswith('builder.py'); print('ok')" % root)
    r = subprocess.run([sys.executable, "-W", "error::ImportWarning", "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-1500:]
