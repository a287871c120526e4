"""CPU check that a wire matrix satisfies every gate constraint of a circuit (the device evaluators compiled for the host,
tests/hostsim): used by the tests of circuits too large for the Python oracle prover."""
import ctypes

import numpy as np

from zklc_amd.plonky2 import gates as G
from zklc_amd.plonky2.builder import P, root_of_unity
from oracle import poseidon_gl as pgl


def gate_constraint_failures(hostsim, data, wires, public_inputs, alphas=(0x123456789ABCDEF1, 0xFEDCBA9876543211)):
    """rows whose alpha-weighted constraint sum is non-zero (empty list = every gate constraint holds)"""
    f = hostsim.hostsim_p2_eval_gate
    f.restype = None
    f.argtypes = [ctypes.c_uint32] + [ctypes.c_void_p] * 3 + [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 2 + \
                 [ctypes.c_uint32, ctypes.c_void_p]
    pih = np.array(pgl.hash_no_pad([int(x) for x in public_inputs]), dtype=np.uint64)
    aa = np.array(alphas, dtype=np.uint64)
    nsel = len(data.groups)
    consts = np.ascontiguousarray(data.constants[nsel:].T)
    consts = np.concatenate([consts, np.zeros((data.n, 1), dtype=np.uint64)], axis=1)
    wt = np.ascontiguousarray(wires.T)
    sel = data.constants[:nsel]
    extras, params = [], []
    for g in data.gates:
        e = np.zeros(1, dtype=np.uint64)
        if g.code == G.COSET_INTERPOLATION:
            w = root_of_unity(g.subgroup_bits)
            e = np.array(list(g.weights) + [pow(w, j, P) for j in range(1 << g.subgroup_bits)], dtype=np.uint64)
        extras.append(e)
        params.append(np.array(g.params, dtype=np.uint32))
    out = np.zeros(2, dtype=np.uint64)
    bad = []
    for r in range(data.n):
        gi = [int(sel[s_, r]) for s_ in range(nsel) if int(sel[s_, r]) != (1 << 32) - 1]
        assert len(gi) == 1
        gi = gi[0]
        g = data.gates[gi]
        f(g.code, params[gi].ctypes.data, extras[gi].ctypes.data, wt[r].ctypes.data, wt.shape[1], consts[r].ctypes.data,
          consts.shape[1] - 1, pih.ctypes.data, aa.ctypes.data, 2, out.ctypes.data)
        if out[0] or out[1]:
            bad.append((r, g.id()[:30]))
    return bad
