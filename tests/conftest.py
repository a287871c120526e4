import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
STASH = {}       # results the sequential GPU tests leave for the pipeline parity tests of the same session (HPi5, epoch_CRTZ)
# circuits are built once per machine (zklc_amd/plonky2/circuit_cache.py): the pipeline tests load the Ed25519 circuit the session's
# ApprovalProver built; ZKLC_CIRCUIT_CACHE= (empty) disables
os.environ.setdefault("ZKLC_CIRCUIT_CACHE", os.path.join(ROOT, ".circuit_cache"))


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


_GPU_STATE = {}


def _gpu_unavailable_reason():
    """None if a gfx950 context can be created (library built, device present); else why not.  Checked once."""
    if "reason" not in _GPU_STATE:
        if os.path.exists("/dev/kfd"):      # a GPU box: never skip -- a missing library or context must FAIL there
            _GPU_STATE["reason"] = None
            return None
        try:
            import zklc_amd
            zklc_amd.Context(0).close()
            _GPU_STATE["reason"] = None
        except Exception as e:  # ZklcError (no device), OSError / ImportError (library not built)
            _GPU_STATE["reason"] = "%s: %s" % (type(e).__name__, e)
    return _GPU_STATE["reason"]


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a gfx950: the gpu-marked tests are skipped (with the reason), not errored.  With
    `-m gpu` on such a box every test is reported as skipped, which the driver reads as "nothing ran"."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items:
        return
    why = _gpu_unavailable_reason()
    if why is not None:
        skip = pytest.mark.skip(reason="no MI355X context: " + why)
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def zctx():
    import zklc_amd
    ctx = zklc_amd.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def approval_prover(zctx):
    """ApprovalProver with the reference's per-signature circuit (2^18 rows x 234 wires) built ONCE for the session: its
    construction and the compilation of its generator program are ~40 s of host Python, and four test modules prove with it"""
    from zklc_amd.signatures import ApprovalProver
    ap = ApprovalProver(zctx)
    ap.ed25519_circuit(41)
    yield ap
    ap.close()


@pytest.fixture(scope="session")
def block_prover(zctx, approval_prover):
    """BlockProver over the session's ApprovalProver: the SHA-256 and recursion circuits of the header chains and of the joins are
    built by the first block proof of the session and reused by the others (5-block window, 6-block epoch branch)"""
    from zklc_amd.prove_bft import BlockProver
    bp = BlockProver(zctx, approval_prover)
    yield bp
    bp.hashes.sha.close()       # the recursion prover and the Ed25519 circuit belong to `approval_prover`, closed by its fixture
    bp.prims.close()
