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


# ---- the complete plonky2 prover (oracle/c/plonky2_prover_oracle.c) ----
class _OGate(ctypes.Structure):
    _fields_ = [("type", ctypes.c_uint32), ("p", ctypes.c_uint32 * 4), ("selector_index", ctypes.c_uint32),
                ("group_start", ctypes.c_uint32), ("group_end", ctypes.c_uint32), ("extra_off", ctypes.c_uint32)]


class _OParams(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in (
        "degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_selectors", "num_challenges", "rate_bits",
        "cap_height", "proof_of_work_bits", "num_query_rounds", "quotient_degree_factor", "num_partial_products",
        "num_gate_constraints", "num_public_inputs", "hasher", "num_gates", "num_arities")] + [("arity_bits", ctypes.c_uint32 * 8)]


PLONKY2_PROVE_STAGES = ("preprocess", "wires_commit", "partial_products", "quotient", "openings", "fri", "proof", "threads")


def plonky2_prove(data, wires, public_inputs, nthreads=0, verifier_data=False):
    """One complete CPU proof with the C + OpenMP restatement.  `data`: a built circuit in the host builder's representation
    (read-only attribute access: degree_bits, config, gates with .code / .params, groups, selector_indices, k_is, constants,
    sigmas, ..: the description the GPU library is given as well); wires uint64 [num_wires, n]; Poseidon-Goldilocks config.
    -> (proof bytes = ProofWithPublicInputs::to_bytes, {stage: seconds}) and, with verifier_data=True, the verifier-only data
    {"constants_sigmas_cap": [..], "circuit_digest": ..} in the JSON schema of the reference as a third item.  Raises ValueError for a circuit with gates outside
    the recursion set, AssertionError when the witness does not satisfy the copy constraints."""
    import numpy as np
    cfg = data.config
    fri = cfg["fri_config"]
    p = _OParams()
    p.degree_bits, p.num_wires, p.num_routed_wires = data.degree_bits, cfg["num_wires"], cfg["num_routed_wires"]
    p.num_constants, p.num_selectors, p.num_challenges = data.num_constants, len(data.groups), cfg["num_challenges"]
    p.rate_bits, p.cap_height, p.proof_of_work_bits = fri["rate_bits"], fri["cap_height"], fri["proof_of_work_bits"]
    p.num_query_rounds = fri["num_query_rounds"]
    p.quotient_degree_factor, p.num_partial_products = data.quotient_degree_factor, data.num_partial_products
    p.num_gate_constraints, p.num_public_inputs = data.num_gate_constraints, data.num_public_inputs
    p.hasher, p.num_gates, p.num_arities = 0, len(data.gates), len(data.fri_arity_bits)
    for i, a in enumerate(data.fri_arity_bits):
        p.arity_bits[i] = a
    gates = (_OGate * len(data.gates))()
    extra = []
    PRIME = 2**64 - 2**32 + 1
    for i, g in enumerate(data.gates):
        gates[i].type = g.code
        for k in range(4):
            gates[i].p[k] = g.params[k]
        s, e = data.groups[data.selector_indices[i]]
        gates[i].selector_index, gates[i].group_start, gates[i].group_end = data.selector_indices[i], s, e
        gates[i].extra_off = len(extra)
        if g.code == 13:       # CosetInterpolationGate: barycentric weights, then the subgroup points
            w = pow(1753635133440165772, 1 << (32 - g.subgroup_bits), PRIME)
            extra += list(g.weights) + [pow(w, j, PRIME) for j in range(1 << g.subgroup_bits)]
    ex = np.array(extra + [0], dtype=np.uint64)
    kis = np.array(data.k_is, dtype=np.uint64)
    consts = np.ascontiguousarray(data.constants, dtype=np.uint64)
    sig = np.ascontiguousarray(data.sigmas, dtype=np.uint64)
    w = np.ascontiguousarray(wires, dtype=np.uint64)
    n = 1 << data.degree_bits
    assert w.shape == (cfg["num_wires"], n) and consts.shape == (data.num_constants, n) and sig.shape == (cfg["num_routed_wires"], n)
    pis = np.array([int(x) for x in public_inputs] + [0], dtype=np.uint64)
    assert len(pis) - 1 == data.num_public_inputs
    out = np.zeros(1 << 20, dtype=np.uint8)
    ln = ctypes.c_uint64()
    secs = (ctypes.c_double * 8)()
    cap_n = 1 << min(fri["cap_height"], data.degree_bits + fri["rate_bits"])
    vd = np.zeros(4 * cap_n + 4, dtype=np.uint64)
    lib = load()
    lib.zklc_oracle_plonky2_prove.restype = ctypes.c_int
    rc = lib.zklc_oracle_plonky2_prove(ctypes.byref(p), gates, ctypes.c_void_p(ex.ctypes.data), ctypes.c_void_p(kis.ctypes.data),
                                       ctypes.c_void_p(consts.ctypes.data), ctypes.c_void_p(sig.ctypes.data),
                                       ctypes.c_void_p(w.ctypes.data), ctypes.c_void_p(pis.ctypes.data),
                                       ctypes.c_void_p(out.ctypes.data), ctypes.c_uint64(out.size), ctypes.byref(ln), secs,
                                       ctypes.c_int(nthreads), ctypes.c_void_p(vd.ctypes.data))
    if rc == -2:
        raise AssertionError("copy constraints are not satisfied by the witness (Z does not close)")
    if rc != 0:
        raise ValueError("zklc_oracle_plonky2_prove: rc %d (unsupported gate / invalid argument / buffer)" % rc)
    res = (bytes(out[:ln.value]), dict(zip(PLONKY2_PROVE_STAGES, [float(x) for x in secs])))
    if verifier_data:
        h = lambda k: {"elements": [int(x) for x in vd[4 * k:4 * k + 4]]}
        res += ({"constants_sigmas_cap": [h(k) for k in range(cap_n)], "circuit_digest": h(cap_n)},)
    return res
