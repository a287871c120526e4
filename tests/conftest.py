import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def near_sets():
    return ["ed25519_near_c1_small.json", "ed25519_near_c1_small_skip.json", "ed25519_near_c2_100.json",
            "ed25519_near_epoch_block01.json", "ed25519_near_epoch_random01.json"]


def near_set_arrays(j):
    """-> (msg, approvals list[bytes], validators list[bytes])"""
    msg = bytes.fromhex(j["msg"])
    approvals = [bytes.fromhex(e["approval"]) for e in j["entries"]]
    # a borsh ValidatorStake::V1 = account_id (u32 len + bytes) || key_type || pk || stake; the reference only slices the tail
    validators = [len(e["account_id"]).to_bytes(4, "little") + e["account_id"].encode() + bytes.fromhex(e["validator_tail"])
                  for e in j["entries"]]
    return msg, approvals, validators


@pytest.fixture(scope="session")
def hostsim():
    """g++ build of the kernel arithmetic headers (tests/hostsim) -- CPU tests only."""
    import ctypes
    d = os.path.join(ROOT, "tests", "hostsim")
    so = os.path.join(d, "libhostsim.so")
    src = os.path.join(d, "hostsim.cpp")
    hdrs = [os.path.join(ROOT, "zk-light-client-implementation_amd", "csrc", f)
            for f in os.listdir(os.path.join(ROOT, "zk-light-client-implementation_amd", "csrc")) if f.endswith(".cuh")]
    newest = max(os.path.getmtime(p) for p in [src] + hdrs)
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["g++", "-O2", "-DZKLC_FE_BOUND_CHECKS", "-shared", "-fPIC", "-o", so, src])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def zctx():
    import zklc_amd
    ctx = zklc_amd.Context(0)
    yield ctx
    ctx.close()
