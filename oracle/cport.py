"""ctypes loader for the oracle's C restatement (oracle/c/*.c -> oracle/_build/libzklc_oracle.so).
TEST INFRASTRUCTURE / CPU BASELINE ONLY -- see oracle/__init__.py."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libzklc_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "c")])
    return _SO


def load():
    global _lib
    if _lib is None:
        srcs = [os.path.join(_HERE, "c", f) for f in os.listdir(os.path.join(_HERE, "c"))]
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
            build()
        lib = ctypes.CDLL(_SO)
        lib.zklc_oracle_ed25519_verify.restype = ctypes.c_int
        lib.zklc_oracle_ed25519_verify.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64]
        lib.zklc_oracle_ed25519_verify_batch.restype = ctypes.c_int
        lib.zklc_oracle_ed25519_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                                         ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int]
        lib.zklc_oracle_sha512.restype = None
        lib.zklc_oracle_sha512.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p]
        _lib = lib
    return _lib


def ed25519_verify(pk, sig, msg):
    return bool(load().zklc_oracle_ed25519_verify(pk, sig, msg, len(msg)))


def ed25519_verify_batch(pks, sigs, msgs, msg_len, msg_stride, n, nthreads=1):
    """numpy uint8 arrays in, (ok uint8[n], threads_used) out."""
    import numpy as np
    ok = np.zeros(n, dtype=np.uint8)
    used = load().zklc_oracle_ed25519_verify_batch(pks.ctypes.data, sigs.ctypes.data, msgs.ctypes.data, msg_len, msg_stride, n,
                                                   ok.ctypes.data, nthreads)
    return ok, used


def sha512(msg):
    out = ctypes.create_string_buffer(64)
    load().zklc_oracle_sha512(msg, len(msg), out)
    return out.raw
