"""GPU prover binding: `CircuitData.prove` of the reference (near_bft_finality/src/prove_crypto/ed25519.rs:60,100,
recursion.rs:95) over the C ABI (include/zklc.h: zklc_plonky2_circuit_create / zklc_plonky2_prove).

There is no CPU path: proving needs a `Context` (a GPU); only `poseidon_gate_rows` (witness generation of
PoseidonGate rows, a host function of the library) works without one.
"""
import ctypes

import numpy as np

from .. import _lib
from . import serialization as S

HASH_GL, HASH_BN128 = S.HASH_GL, S.HASH_BN128


class GateC(ctypes.Structure):
    _fields_ = [("type", ctypes.c_uint32), ("p", ctypes.c_uint32 * 4), ("selector_index", ctypes.c_uint32),
                ("group_start", ctypes.c_uint32), ("group_end", ctypes.c_uint32), ("extra_off", ctypes.c_uint32)]


class ParamsC(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in (
        "degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_selectors", "num_challenges", "rate_bits",
        "cap_height", "proof_of_work_bits", "num_query_rounds", "quotient_degree_factor", "num_partial_products",
        "num_gate_constraints", "num_public_inputs", "hasher", "num_gates", "num_arities")] + [("arity_bits", ctypes.c_uint32 * 8)]


def poseidon_gate_rows(inputs, swap=None):
    """inputs uint64 [n, 12] (+ swap uint64 [n]) -> PoseidonGate wire rows uint64 [n, 135]"""
    a = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, 12)
    n = a.shape[0]
    sw = None if swap is None else np.ascontiguousarray(swap, dtype=np.uint64)
    rows = np.zeros((n, 135), dtype=np.uint64)
    rc = _lib.load().zklc_poseidon_gl_gate_rows(a.ctypes.data, None if sw is None else sw.ctypes.data, n, rows.ctypes.data)
    if rc != 0:
        raise ValueError("zklc_poseidon_gl_gate_rows: invalid argument")
    return rows


class Prover:
    """One circuit resident on one GPU (zklc_plonky2_circuit): preprocessed polynomials + the prover's working set."""

    def __init__(self, ctx, data, hasher=HASH_GL):
        self.ctx, self.data, self.hasher = ctx, data, hasher
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        if data._container is not None:
            # a circuit loaded from a container file: the library takes parameters, gates and matrices from the file's sections
            rc = self._lib.zklc_plonky2_circuit_create_from_container(ctx._h, data._container._h, int(hasher), ctypes.byref(h))
            data._container.release_pages()      # the matrices are in HBM now: the mapped pages go back to the page cache (RSS)
        else:
            from .container import native_arguments
            p, gates, ex, kis = native_arguments(data, hasher)
            consts = np.ascontiguousarray(data.constants, dtype=np.uint64)
            sig = np.ascontiguousarray(data.sigmas, dtype=np.uint64)
            rc = self._lib.zklc_plonky2_circuit_create(ctx._h, ctypes.byref(p), gates, ex.ctypes.data if ex.size else None, ex.size,
                                                       kis.ctypes.data, consts.ctypes.data, sig.ctypes.data, ctypes.byref(h))
        ctx._check(rc)
        self._h = h
        self.common = data.common_data()
        self.proof_bytes = int(self._lib.zklc_plonky2_proof_bytes(h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.zklc_plonky2_circuit_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def verifier_data(self):
        """verifier_only_circuit_data.json of the reference (constants_sigmas_cap, circuit_digest)"""
        cap_n = 1 << min(self.common["fri_params"]["config"]["cap_height"],
                         self.data.degree_bits + self.common["fri_params"]["config"]["rate_bits"])
        cap = np.zeros(32 * cap_n, dtype=np.uint8)
        dig = np.zeros(32, dtype=np.uint8)
        self.ctx._check(self._lib.zklc_plonky2_verifier_data(self._h, cap.ctypes.data, dig.ctypes.data))
        rd = S._Reader(bytes(cap) + bytes(dig), self.hasher)
        return {"constants_sigmas_cap": [rd.hash() for _ in range(cap_n)], "circuit_digest": rd.hash()}

    def prove_bytes(self, wires, public_inputs):
        """wires uint64 [num_wires, n] (host) -> proof bytes (ProofWithPublicInputs::to_bytes)"""
        w = np.ascontiguousarray(wires, dtype=np.uint64)
        assert w.shape == (self.data.config["num_wires"], self.data.n)
        pis = np.array([int(x) for x in public_inputs], dtype=np.uint64)
        assert len(pis) == self.data.num_public_inputs
        out = np.zeros(self.proof_bytes, dtype=np.uint8)
        ln = ctypes.c_uint64()
        rc = self._lib.zklc_plonky2_prove(self.ctx._h, self._h, w.ctypes.data, pis.ctypes.data if len(pis) else None,
                                          out.ctypes.data, out.size, ctypes.byref(ln))
        self.ctx._check(rc)
        return bytes(out[:ln.value])

    def prove_host_ptr(self, wires_ptr, public_inputs):
        """wires_ptr: address of a host (ideally pinned) uint64 [num_wires, n] matrix -> proof bytes"""
        pis = np.array([int(x) for x in public_inputs], dtype=np.uint64)
        out = np.zeros(self.proof_bytes, dtype=np.uint8)
        ln = ctypes.c_uint64()
        rc = self._lib.zklc_plonky2_prove(self.ctx._h, self._h, wires_ptr, pis.ctypes.data if len(pis) else None,
                                          out.ctypes.data, out.size, ctypes.byref(ln))
        self.ctx._check(rc)
        return bytes(out[:ln.value])

    def prove(self, wires, public_inputs):
        """-> proof in the reference's proof.json schema"""
        return S.proof_from_bytes(self.prove_bytes(wires, public_inputs), self.common, self.hasher)

    def prove_dev(self, d_wires_ptr, public_inputs, stream=None):
        pis = np.array([int(x) for x in public_inputs], dtype=np.uint64)
        out = np.zeros(self.proof_bytes, dtype=np.uint8)
        ln = ctypes.c_uint64()
        rc = self._lib.zklc_plonky2_prove_dev(self.ctx._h, stream, self._h, d_wires_ptr, pis.ctypes.data if len(pis) else None,
                                              out.ctypes.data, out.size, ctypes.byref(ln))
        self.ctx._check(rc)
        return bytes(out[:ln.value])

    def last_challenges(self):
        buf = np.zeros(64, dtype=np.uint64)
        k = self._lib.zklc_plonky2_last_challenges(self._h, buf.ctypes.data, 64)
        return [int(x) for x in buf[:k]]

    def last_timings(self):
        buf = np.zeros(8, dtype=np.float64)
        k = self._lib.zklc_plonky2_last_timings(self._h, buf.ctypes.data, 8)
        names = ["wires_commit", "partial_products", "quotient", "openings", "fri_commit", "pow", "queries", "total"]
        return dict(zip(names[:k], [float(x) for x in buf[:k]]))
