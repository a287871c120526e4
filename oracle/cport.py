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
        u64p = ctypes.c_void_p
        lib.zklc_oracle_gl_ntt.restype = ctypes.c_int
        lib.zklc_oracle_gl_ntt.argtypes = [u64p, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int]
        lib.zklc_oracle_gl_lde.restype = ctypes.c_int
        lib.zklc_oracle_gl_lde.argtypes = [u64p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, u64p, ctypes.c_int]
        lib.zklc_oracle_poseidon_gl_permute.restype = None
        lib.zklc_oracle_poseidon_gl_permute.argtypes = [u64p]
        lib.zklc_oracle_gl_merkle_commit.restype = ctypes.c_int
        lib.zklc_oracle_gl_merkle_commit.argtypes = [u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, u64p,
                                                     ctypes.c_int]
        lib.zklc_oracle_bn254_gen_points.restype = None
        lib.zklc_oracle_bn254_gen_points.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, u64p]
        lib.zklc_oracle_bn254_msm_naive.restype = ctypes.c_int
        lib.zklc_oracle_bn254_msm_naive.argtypes = [u64p, u64p, ctypes.c_uint64, u64p]
        lib.zklc_oracle_bn254_msm.restype = ctypes.c_int
        lib.zklc_oracle_bn254_msm.argtypes = [u64p, u64p, ctypes.c_uint64, u64p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
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


# ---- Goldilocks (oracle/c/goldilocks_oracle.c) ----
def gl_ntt(data, inverse=False, nthreads=1):
    """uint64 [batch, n] -> transformed copy (natural order in and out)"""
    import numpy as np
    a = np.ascontiguousarray(data, dtype=np.uint64).copy()
    if a.ndim == 1:
        a = a[None, :]
    batch, n = a.shape
    load().zklc_oracle_gl_ntt(a.ctypes.data, n.bit_length() - 1, batch, int(inverse), nthreads)
    return a.reshape(np.shape(data))


def gl_lde(coeffs, rate_bits, shift=7, nthreads=1):
    import numpy as np
    a = np.ascontiguousarray(coeffs, dtype=np.uint64)
    if a.ndim == 1:
        a = a[None, :]
    batch, n = a.shape
    out = np.zeros((batch, n << rate_bits), dtype=np.uint64)
    load().zklc_oracle_gl_lde(a.ctypes.data, n.bit_length() - 1, rate_bits, batch, shift, out.ctypes.data, nthreads)
    return out


def poseidon_gl_permute(state):
    import numpy as np
    a = np.array(state, dtype=np.uint64)
    load().zklc_oracle_poseidon_gl_permute(a.ctypes.data)
    return [int(x) for x in a]


def gl_merkle_commit(mat, cap_height, nthreads=1):
    """mat uint64 [width, n] (poly-major) -> list of levels ([m,4] arrays), leaves first, cap last"""
    import numpy as np
    a = np.ascontiguousarray(mat, dtype=np.uint64)
    width, n = a.shape
    log_leaves = n.bit_length() - 1
    words = sum(4 << (log_leaves - l) for l in range(log_leaves - cap_height + 1))
    tree = np.zeros(words, dtype=np.uint64)
    load().zklc_oracle_gl_merkle_commit(a.ctypes.data, n, log_leaves, width, cap_height, tree.ctypes.data, nthreads)
    levels, off = [], 0
    for l in range(log_leaves - cap_height + 1):
        m = n >> l
        levels.append(tree[off:off + 4 * m].reshape(m, 4))
        off += 4 * m
    return levels


# ---- BN254 (oracle/c/bn254_oracle.c) ----
def bn254_gen_points(n, a=7, b=11):
    """n affine points (a + i*b) * G in gnark Montgomery layout, uint64 [n, 8]"""
    import numpy as np
    out = np.zeros((n, 8), dtype=np.uint64)
    load().zklc_oracle_bn254_gen_points(a, b, n, out.ctypes.data)
    return out


def bn254_msm(points, scalars, nthreads=1, naive=False):
    """-> (uint64[8] affine gnark layout, is_infinity, threads_used)"""
    import numpy as np
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(8, dtype=np.uint64)
    used = ctypes.c_int(1)
    if naive:
        inf = load().zklc_oracle_bn254_msm_naive(pts.ctypes.data, sc.ctypes.data, pts.shape[0], out.ctypes.data)
    else:
        inf = load().zklc_oracle_bn254_msm(pts.ctypes.data, sc.ctypes.data, pts.shape[0], out.ctypes.data, nthreads, ctypes.byref(used))
    return out, bool(inf), used.value
