"""The device gate evaluators (csrc/plonky2_gates.cuh, compiled by g++) against the oracle's evaluators
(oracle/plonky2_gates.py, pinned by the reference's golden proofs) on random wires/constants: for each of the 19 gate
types the alpha-weighted sum of the constraints must agree for both challenges."""
import ctypes
import random

import numpy as np
import pytest

import zklc_amd
from zklc_amd.plonky2 import gates as G, synthetic as SY
from oracle import goldilocks as gl
from oracle import plonky2_gates as OG

P = gl.P
CFG = {"num_wires": 234, "num_routed_wires": 80, "num_constants": 2}
GATES = [
    G.NoopGate(), G.ConstantGate(2), G.PublicInputGate(), G.ArithmeticGate(20), G.ArithmeticExtensionGate(10), G.MulExtensionGate(13),
    G.BaseSumGate(63, 2), G.BaseSumGate(10, 4), G.PoseidonGate(), G.PoseidonMdsGate(), G.RandomAccessGate(4, 4, 2),
    G.RandomAccessGate(1, 20, 0), G.ReducingGate(43), G.ReducingExtensionGate(32), G.ExponentiationGate(66),
    G.CosetInterpolationGate(4, 6, SY.barycentric_weights(4)), G.CosetInterpolationGate(3, 3, SY.barycentric_weights(3)),
    G.U32ArithmeticGate(6), G.U32AddManyGate(3, 9), G.U32AddManyGate(11, 5), G.U32SubtractionGate(11), G.U32RangeCheckGate(8),
    G.ComparisonGate(32, 16), G.ComparisonGate(10, 5), G.U32InterleaveGate(3), G.UninterleaveToU32Gate(2), G.UninterleaveToB32Gate(2),
]


def _call(hostsim, g, wires, consts, pih, alphas):
    f = hostsim.hostsim_p2_eval_gate
    f.restype = None
    f.argtypes = [ctypes.c_uint32] + [ctypes.c_void_p] * 3 + [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 2 + \
                 [ctypes.c_uint32, ctypes.c_void_p]
    params = np.array(g.params, dtype=np.uint32)
    extra = np.zeros(1, dtype=np.uint64)
    if g.code == G.COSET_INTERPOLATION:
        w = SY.root_of_unity(g.subgroup_bits)
        extra = np.array(list(g.weights) + [pow(w, j, P) for j in range(1 << g.subgroup_bits)], dtype=np.uint64)
    wa, ca = np.array(wires, dtype=np.uint64), np.array(consts + [0], dtype=np.uint64)
    pa, aa = np.array(pih, dtype=np.uint64), np.array(alphas, dtype=np.uint64)
    out = np.zeros(2, dtype=np.uint64)
    f(g.code, params.ctypes.data, extra.ctypes.data, wa.ctypes.data, len(wires), ca.ctypes.data, len(consts), pa.ctypes.data,
      aa.ctypes.data, 2, out.ctypes.data)
    return [int(x) for x in out]


@pytest.mark.parametrize("g", GATES, ids=lambda g: g.id()[:40])
def test_gate_evaluator_matches_oracle(hostsim, g):
    rng = random.Random(hash(g.id()) & 0xFFFF)
    og = OG.gate_from_id(g.id())
    assert og.num_constraints == g.num_constraints and og.degree == g.degree and og.num_constants == g.num_constants
    for trial in range(3):
        small = trial == 2      # small values exercise the range-check products around their roots
        wires = [rng.randrange(4) if small else rng.randrange(P) for _ in range(max(g.num_wires, 1))]
        consts = [rng.randrange(P) for _ in range(g.num_constants)]
        pih = [rng.randrange(P) for _ in range(4)]
        alphas = [rng.randrange(P), rng.randrange(P)]
        cs = og.eval(OG.BaseK, consts, wires, pih)
        assert len(cs) == g.num_constraints
        want = [OG.reduce_with_powers(OG.BaseK, cs, a) for a in alphas]
        assert _call(hostsim, g, wires, consts, pih, alphas) == want


def test_poseidon_gate_lazy_form_equals_the_round_by_round_form(hostsim):
    """p2_eval_poseidon_lazy (opt-in: full rounds as whole-round pieces, the 22 partial rounds as linear forms of the S-box wires over
    the PGL_LAZY_* tables, loose values) against p2_eval_poseidon (the default: canonical, round by round) and the oracle, on random wires, on
    wires at the edges of the field, and on a SATISFYING assignment (a real permutation: every constraint zero)."""
    g = G.PoseidonGate()
    og = OG.gate_from_id(g.id())
    lazy = hostsim.hostsim_p2_eval_poseidon_lazy
    lazy.restype = None
    lazy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    rng = random.Random(77)
    edge = [0, 1, P - 1, P - 2, (1 << 32) - 1, 1 << 32, P - (1 << 32), (1 << 63)]
    for trial in range(40):
        if trial < 30:
            wires = [rng.randrange(P) for _ in range(g.num_wires)]
        else:
            wires = [rng.choice(edge) for _ in range(g.num_wires)]
        wires[24] = rng.randrange(2) if trial % 2 else wires[24]
        alphas = [rng.randrange(P), rng.randrange(P)]
        got = _call(hostsim, g, wires, [], [0] * 4, alphas)
        out = np.zeros(2, dtype=np.uint64)
        wa, aa = np.array(wires, dtype=np.uint64), np.array(alphas, dtype=np.uint64)      # kept alive across the call
        for mode in (0, 1, 2):    # lazy with unrolled / rolled partial rounds, loose round by round
            lazy(wa.ctypes.data, aa.ctypes.data, 2, out.ctypes.data, mode)
            assert got == [int(x) for x in out], (trial, mode)
        cs = og.eval(OG.BaseK, [], wires, [0] * 4)
        assert got == [OG.reduce_with_powers(OG.BaseK, cs, a) for a in alphas], trial
    # satisfying rows from the witness generator of the gate (host function of the product library): every constraint is zero
    from zklc_amd.plonky2.prover import poseidon_gate_rows
    from oracle import poseidon_gl as OP
    ins = np.array([[rng.randrange(P) for _ in range(12)] for _ in range(4)], dtype=np.uint64)
    rows = poseidon_gate_rows(ins, np.array([0, 1, 0, 1], dtype=np.uint64))
    for inp, sw, row in zip(ins, (0, 1, 0, 1), rows):
        x = [int(t) for t in inp]
        if sw:
            x = x[4:8] + x[:4] + x[8:]
        assert [int(t) for t in row[12:24]] == OP.permute(x)
        assert _call(hostsim, g, [int(t) for t in row], [], [0] * 4, [rng.randrange(P), rng.randrange(P)]) == [0, 0]
        broken = [int(t) for t in row]
        broken[65 + 13] = (broken[65 + 13] + 1) % P        # one S-box wire of the second partial block
        assert _call(hostsim, g, broken, [], [0] * 4, [rng.randrange(P), rng.randrange(P)]) != [0, 0]
        for wires_, zero in ((row, True), (np.array(broken, dtype=np.uint64), False)):
            wa = np.ascontiguousarray(wires_, dtype=np.uint64)
            aa = np.array([rng.randrange(P), rng.randrange(P)], dtype=np.uint64)
            out = np.zeros(2, dtype=np.uint64)
            for mode in (0, 1, 2):
                lazy(wa.ctypes.data, aa.ctypes.data, 2, out.ctypes.data, mode)
                assert ([int(t) for t in out] == [0, 0]) == zero


def test_filter_matches_oracle(hostsim):
    f = hostsim.hostsim_p2_filter
    f.restype = ctypes.c_uint64
    f.argtypes = [ctypes.c_uint32] * 3 + [ctypes.c_uint64, ctypes.c_uint32]
    rng = random.Random(5)
    for _ in range(50):
        start = rng.randrange(5)
        end = start + 1 + rng.randrange(7)
        row = rng.randrange(start, end)
        s = rng.choice([rng.randrange(P), row, 0xFFFFFFFF, rng.randrange(start, end)])
        many = rng.randrange(2)
        assert f(row, start, end, s, many) == OG.compute_filter(OG.BaseK, row, (start, end), s, bool(many))


def test_extension_field_ops(hostsim):
    f = hostsim.hostsim_gl2_op
    f.restype = None
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    rng = random.Random(9)

    def call(op, a, b, e=0):
        aa, bb, out = np.array(a, dtype=np.uint64), np.array(b, dtype=np.uint64), np.zeros(2, dtype=np.uint64)
        f(op, aa.ctypes.data, bb.ctypes.data, e, out.ctypes.data)
        return (int(out[0]), int(out[1]))
    edge = [(0, 0), (1, 0), (0, 1), (P - 1, P - 1), (P - 1, 0)]
    vals = edge + [(rng.randrange(P), rng.randrange(P)) for _ in range(40)]
    for a in vals:
        for b in rng.sample(vals, 5):
            assert call(0, a, b) == gl.ext_add(a, b) and call(1, a, b) == gl.ext_sub(a, b) and call(2, a, b) == gl.ext_mul(a, b)
        assert call(3, a, a) == gl.ext_mul(a, a)
        if a != (0, 0):
            assert gl.ext_mul(call(4, a, a), a) == (1, 0)
        e = rng.getrandbits(40)
        from oracle.plonky2_verifier import ext_pow
        assert call(5, a, a, e) == ext_pow(a, e)


def test_addmany_tile_walk_matches_the_per_gate_evaluator(hostsim):
    """the U32AddMany LDS-tile evaluator (csrc/plonky2_prover.hip: p2_quotient_addmany_tile_kernel; lane-level pieces p2_amt_routed /
    p2_amt_consume in plonky2_gates.cuh) walked for one LDE point -- four waves, phases of 48 limb columns, the real Ed25519 circuit's
    eight variants on the host plan's slots -- gives every variant the sum its own evaluator (p2_eval_u32_add_many) gives"""
    variants = [(11, 5), (13, 5), (15, 4), (16, 4), (3, 9), (5, 9), (7, 8), (9, 6)]
    slots = [4, 2, 5, 3, 6, 1, 7, 0]          # the plan printed by the GPU run: heaviest first onto the least loaded wave
    rng = random.Random(7)
    f = hostsim.hostsim_p2_addmany_tile
    f.restype = None
    f.argtypes = [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 3 + [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    for trial, (vs, sl) in enumerate([(variants, slots), (variants[:3], [0, 0xFFFFFFFF, 1, 0xFFFFFFFF, 2] + [0xFFFFFFFF] * 3)]):
        wires = [rng.randrange(P) for _ in range(234)]
        alphas = [rng.randrange(P), rng.randrange(P)]
        va = np.array([x for v in vs for x in v], dtype=np.uint32)
        sa = np.array(sl, dtype=np.uint32)
        wa, aa = np.array(wires, dtype=np.uint64), np.array(alphas, dtype=np.uint64)
        out = np.zeros(2 * len(vs), dtype=np.uint64)
        k0 = 22 * trial          # the kernels start the gate constraints at alpha^k0 (after the Z / partial-product terms)
        f(va.ctypes.data, len(vs), sa.ctypes.data, wa.ctypes.data, aa.ctypes.data, 2, out.ctypes.data, k0)
        for k, (na, ops) in enumerate(vs):
            want = _call(hostsim, G.U32AddManyGate(na, ops), wires, [], [0] * 4, alphas)
            want = [w * pow(a, k0, P) % P for w, a in zip(want, alphas)]
            assert [int(x) % P for x in out[2 * k:2 * k + 2]] == want, (trial, k, na, ops)
