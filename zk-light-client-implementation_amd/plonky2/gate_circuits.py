"""Gate constraints restated IN-CIRCUIT (plonky2 `Gate::eval_unfiltered_circuit`), for the recursive verifier (recursion.py).

The constraint of every gate type that can occur in an inner circuit is written once over an adapter `K` whose values are
extension-field targets of the outer circuit; `CircuitK` turns K.add / K.sub / K.mul into ArithmeticExtensionGate /
MulExtensionGate operations (fusing a product with the following addition into one operation, like plonky2's
`mul_add_extension`).  The constraint formulas are those of
  gnark-plonky2-verifier/plonk/gates/*.go (standard gates; file list in csrc/plonky2_gates.cuh) and
  crypto/plonky2_u32/src/gates/{arithmetic_u32,add_many_u32,subtraction_u32,range_check_u32,comparison}.rs (in-tree),
i.e. the same ones the GPU quotient kernels evaluate (csrc/plonky2_gates.cuh) -- here they become rows of the outer circuit.
PoseidonGate follows poseidon_gate.go:84-181 except that the partial rounds take the naive form (full constant layer, dense MDS
layer = one PoseidonMdsGate row each), as plonky2's in-circuit evaluator does when the MDS gate fits the routed wires: the S-box
inputs (the only values the gate's wires pin) are the same.
"""
import functools

import numpy as np

from . import gates as G
from .builder import P, root_of_unity

W = 7
_SMALL = 1 << 33


@functools.lru_cache(maxsize=1)
def poseidon_constants():
    from .. import _lib
    rc, first, prc = np.zeros(360, dtype=np.uint64), np.zeros(12, dtype=np.uint64), np.zeros(22, dtype=np.uint64)
    circ, diag = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
    _lib.load().zklc_poseidon_gl_constants(rc.ctypes.data, first.ctypes.data, prc.ctypes.data, circ.ctypes.data, diag.ctypes.data)
    return {"rc": [int(x) for x in rc], "fp_first": [int(x) for x in first], "fp_rc": [int(x) for x in prc],
            "circ": [int(x) for x in circ], "diag": [int(x) for x in diag]}


def _is_small(c):
    return c < _SMALL or c > P - _SMALL


class CircuitK:
    """Field adapter over extension targets of a RecursiveCircuitBuilder.  Values are lazy:
    ("c", k) a constant, ("t", e) a target, ("s", k, e) = k*e, ("p", k, a, b) = k*a*b."""

    def __init__(self, b):
        self.b = b
        self.zero, self.one = ("c", 0), ("c", 1)

    @staticmethod
    def const(c):
        return ("c", c % P)

    @staticmethod
    def lift(e):
        return ("t", e)

    def mat(self, v):
        b = self.b
        k = v[0]
        if k == "t":
            return v[1]
        if k == "c":
            return b.constant_ext(v[1])
        if k == "s":
            return b.arithmetic_ext(v[1], 0, v[2], b.one_ext(), b.zero_ext())
        return b.arithmetic_ext(v[1], 0, v[2], v[3], b.zero_ext())

    def scale(self, c, v):
        c %= P
        if c == 1:
            return v
        if c == 0:
            return self.zero
        k = v[0]
        if k == "c":
            return ("c", c * v[1] % P)
        if not _is_small(c):       # a one-off constant: keep the gate constants shared, bring it in as a target
            return ("p", 1, self.b.constant_ext(c), self.mat(v))
        if k == "t":
            return ("s", c, v[1])
        if k == "s":
            return ("s", c * v[1] % P, v[2]) if _is_small(c * v[1] % P) else ("s", c, self.mat(v))
        return ("p", c * v[1] % P, v[2], v[3]) if _is_small(c * v[1] % P) else ("s", c, self.mat(v))

    def _lin(self, v):
        """v as (coefficient, target)"""
        k = v[0]
        if k == "t":
            return 1, v[1]
        if k == "s":
            return v[1], v[2]
        if k == "c":       # additive constants: the very common ones as gate constants, the rest as (shared) constant targets
            return (v[1], self.b.one_ext()) if v[1] <= 4 or v[1] >= P - 4 else (1, self.b.constant_ext(v[1]))
        return 1, self.mat(v)

    def mul(self, x, y):
        if x[0] == "c":
            return self.scale(x[1], y)
        if y[0] == "c":
            return self.scale(y[1], x)
        cx, ex = self._lin(x)
        cy, ey = self._lin(y)
        return ("p", cx * cy % P, ex, ey)

    def add(self, x, y):
        if x[0] == "c" and y[0] == "c":
            return ("c", (x[1] + y[1]) % P)
        if x[0] == "c" and x[1] == 0:
            return y
        if y[0] == "c" and y[1] == 0:
            return x
        b = self.b
        if y[0] == "p" and x[0] != "p":
            x, y = y, x
        if x[0] == "p":
            c1, d = self._lin(y)
            return ("t", b.arithmetic_ext(x[1], c1, x[2], x[3], d))
        c0, a = self._lin(x)
        c1, d = self._lin(y)
        return ("t", b.arithmetic_ext(c0, c1, a, b.one_ext(), d))

    def sub(self, x, y):
        return self.add(x, self.scale(P - 1, y))

    def mds(self, state):
        return [("t", e) for e in self.b.mds_ext([self.mat(v) for v in state])]


class LiteralK(CircuitK):
    """DIAGNOSTIC adapter (tools/wrap_instance.py --literal, profiles/r05_wrap_instance_hypotheses.txt): every K operation becomes
    ONE builder operation at once -- mul -> mul_extension, add / sub -> add_extension / sub_extension, a constant factor ->
    constant_extension + mul_extension (plonky2's `mul_const_extension`) -- with no fusing of a product into the following addition.
    Not used by any prover path: it answers "how many rows would an unfused evaluator take" in the hypothesis log."""

    def mat(self, v):
        return self.b.constant_ext(v[1]) if v[0] == "c" else v[1]

    def scale(self, c, v):
        c %= P
        if v[0] == "c":
            return ("c", c * v[1] % P)
        return ("t", self.b.mul_ext(self.b.constant_ext(c), self.mat(v)))

    def mul(self, x, y):
        if x[0] == "c" and y[0] == "c":
            return ("c", x[1] * y[1] % P)
        return ("t", self.b.mul_ext(self.mat(x), self.mat(y)))

    def add(self, x, y):
        if x[0] == "c" and y[0] == "c":
            return ("c", (x[1] + y[1]) % P)
        return ("t", self.b.add_ext(self.mat(x), self.mat(y)))

    def sub(self, x, y):
        if x[0] == "c" and y[0] == "c":
            return ("c", (x[1] - y[1]) % P)
        return ("t", self.b.sub_ext(self.mat(x), self.mat(y)))


# ---- the degree-2 extension algebra over K (pairs of K values, X^2 = 7): quadratic_extension_algebra.go
def _alg(w, start):
    return (w[start], w[start + 1])


def _alg_add(K, a, b):
    return (K.add(a[0], b[0]), K.add(a[1], b[1]))


def _alg_sub(K, a, b):
    return (K.sub(a[0], b[0]), K.sub(a[1], b[1]))


def _alg_mul(K, a, b):
    return (K.add(K.mul(a[0], b[0]), K.scale(W, K.mul(a[1], b[1]))), K.add(K.mul(a[0], b[1]), K.mul(a[1], b[0])))


def _alg_scalar(K, s, a):
    return (K.mul(s, a[0]), K.mul(s, a[1]))


def _reduce_with_powers(K, terms, base):
    acc = K.zero
    for t in reversed(terms):
        acc = K.add(K.scale(base, acc), t)
    return acc


def _range_product(K, x, base):
    acc = K.one
    for k in range(base):
        acc = K.mul(acc, K.sub(x, K.const(k)))
    return acc


def _sbox(K, x):
    x2 = K.mul(x, x)
    x4 = K.mul(x2, x2)
    return K.mul(x4, K.mul(x, x2))


def _eval_poseidon(K, w):
    pc = poseidon_constants()
    rc = pc["rc"]
    out = []
    swap = w[24]
    out.append(K.mul(swap, K.sub(swap, K.one)))
    for i in range(4):
        out.append(K.sub(K.mul(swap, K.sub(w[i + 4], w[i])), w[25 + i]))
    st = [None] * 12
    for i in range(4):
        st[i] = K.add(w[i], w[25 + i])
        st[i + 4] = K.sub(w[i + 4], w[25 + i])
    for i in range(8, 12):
        st[i] = w[i]
    rnd = 0
    for r in range(4):
        st = [K.add(st[i], K.const(rc[12 * rnd + i])) for i in range(12)]
        if r:
            for i in range(12):
                sin = w[29 + 12 * (r - 1) + i]
                out.append(K.sub(st[i], sin))
                st[i] = sin
        st = K.mds([_sbox(K, x) for x in st])
        rnd += 1
    # partial rounds: plonky2's PoseidonGate::eval_unfiltered_circuit takes the NAIVE form whenever a PoseidonMdsGate row fits the
    # routed wires ("the naive method is more efficient if we have enough routed wires for PoseidonMdsGate"): the full constant
    # layer (twelve constant targets per round), the S-box on word 0, one dense MDS row.  As polynomials in the wires it equals the
    # fast form the evaluators use (poseidon_gate.go:118-150): the S-box inputs are the same values.
    for r in range(22):
        st = [K.add(st[i], K.const(rc[12 * rnd + i])) for i in range(12)]
        sin = w[65 + r]
        out.append(K.sub(st[0], sin))
        st = K.mds([_sbox(K, sin)] + st[1:])
        rnd += 1
    for r in range(4):
        st = [K.add(st[i], K.const(rc[12 * rnd + i])) for i in range(12)]
        for i in range(12):
            sin = w[87 + 12 * r + i]
            out.append(K.sub(st[i], sin))
            st[i] = sin
        st = K.mds([_sbox(K, x) for x in st])
        rnd += 1
    for i in range(12):
        out.append(K.sub(st[i], w[12 + i]))
    return out


def _eval_coset_interpolation(K, g, w):
    np_, d, ni = 1 << g.subgroup_bits, g.degree, g.num_intermediates
    start_pt = 1 + 2 * np_
    start_val, start_inter = start_pt + 2, start_pt + 4
    shift = w[0]
    point = _alg(w, start_pt)
    shifted = _alg(w, start_inter + 4 * ni)
    out = list(_alg_sub(K, point, _alg_scalar(K, shift, shifted)))
    gen = root_of_unity(g.subgroup_bits)
    dom = [pow(gen, i, P) for i in range(np_)]
    vals = [_alg(w, 1 + 2 * i) for i in range(np_)]

    def partial(s, e, ev, prod):
        for i in range(s, e):
            term = (K.sub(shifted[0], K.const(dom[i])), shifted[1])
            wv = (K.scale(g.weights[i], vals[i][0]), K.scale(g.weights[i], vals[i][1]))
            ev = _alg_add(K, _alg_mul(K, ev, term), _alg_mul(K, wv, prod))
            prod = _alg_mul(K, prod, term)
        return ev, prod
    ev, prod = partial(0, d, (K.zero, K.zero), (K.one, K.zero))
    for i in range(ni):
        iev, ipr = _alg(w, start_inter + 2 * i), _alg(w, start_inter + 2 * (ni + i))
        out.extend(_alg_sub(K, iev, ev))
        out.extend(_alg_sub(K, ipr, prod))
        s = 1 + (d - 1) * (i + 1)
        ev, prod = partial(s, min(s + d - 1, np_), iev, ipr)
    out.extend(_alg_sub(K, _alg(w, start_val), ev))
    return out


def _eval_comparison(K, g, w):
    nc, cb = g.num_chunks, g.chunk_bits
    size = 1 << cb
    first = [w[4 + i] for i in range(nc)]
    second = [w[4 + nc + i] for i in range(nc)]
    out = [K.sub(_reduce_with_powers(K, first, size), w[0]), K.sub(_reduce_with_powers(K, second, size), w[1])]
    msd = K.zero
    for i in range(nc):
        out.append(_range_product(K, first[i], size))
        out.append(_range_product(K, second[i], size))
        diff = K.lift(K.mat(K.sub(second[i], first[i])))
        dummy, eq = w[4 + 2 * nc + i], w[4 + 3 * nc + i]
        out.append(K.sub(K.mul(diff, dummy), K.sub(K.one, eq)))
        out.append(K.mul(eq, diff))
        inter = w[4 + 4 * nc + i]
        out.append(K.sub(inter, K.mul(eq, msd)))
        msd = K.add(inter, K.mul(K.sub(K.one, eq), diff))
    out.append(K.sub(w[3], msd))
    bits = [w[4 + 5 * nc + i] for i in range(cb + 1)]
    for b_ in bits:
        out.append(K.mul(b_, K.sub(K.one, b_)))
    out.append(K.sub(K.add(K.const(size), w[3]), _reduce_with_powers(K, bits, 2)))
    out.append(K.sub(w[2], bits[cb]))
    return out


def eval_gate_circuit(K, g, c, w, pih):
    """constraints of gate `g` (a gates.py descriptor) on the K-values: c = gate-local constants, w = wires, pih = 4 values"""
    code = g.code
    if code == G.NOOP:
        return []
    if code == G.CONSTANT:
        return [K.sub(c[i], w[i]) for i in range(g.num_consts)]
    if code == G.PUBLIC_INPUT:
        return [K.sub(w[i], pih[i]) for i in range(4)]
    if code == G.ARITHMETIC:
        return [K.sub(w[4 * i + 3], K.add(K.mul(K.mul(w[4 * i], w[4 * i + 1]), c[0]), K.mul(w[4 * i + 2], c[1])))
                for i in range(g.num_ops)]
    if code == G.ARITHMETIC_EXT:
        out = []
        for i in range(g.num_ops):
            m0, m1, a, o = (_alg(w, 8 * i + 2 * k) for k in range(4))
            comp = _alg_add(K, _alg_scalar(K, c[1], a), _alg_scalar(K, c[0], _alg_mul(K, m0, m1)))
            out.extend(_alg_sub(K, o, comp))
        return out
    if code == G.MUL_EXT:
        out = []
        for i in range(g.num_ops):
            m0, m1, o = (_alg(w, 6 * i + 2 * k) for k in range(3))
            out.extend(_alg_sub(K, o, _alg_scalar(K, c[0], _alg_mul(K, m0, m1))))
        return out
    if code == G.BASE_SUM:
        limbs = w[1:1 + g.num_limbs]
        return [K.sub(_reduce_with_powers(K, limbs, g.base), w[0])] + [_range_product(K, l, g.base) for l in limbs]
    if code == G.POSEIDON:
        return _eval_poseidon(K, w)
    if code == G.POSEIDON_MDS:
        pc = poseidon_constants()
        ins = [_alg(w, 2 * i) for i in range(12)]
        out = []
        for r in range(12):
            acc = (K.zero, K.zero)
            for i in range(12):
                v = ins[(i + r) % 12]
                acc = (K.add(acc[0], K.scale(pc["circ"][i], v[0])), K.add(acc[1], K.scale(pc["circ"][i], v[1])))
            if pc["diag"][r]:
                acc = (K.add(acc[0], K.scale(pc["diag"][r], ins[r][0])), K.add(acc[1], K.scale(pc["diag"][r], ins[r][1])))
            out.extend(_alg_sub(K, _alg(w, 2 * (12 + r)), acc))
        return out
    if code == G.RANDOM_ACCESS:
        vs = 1 << g.bits
        out = []
        for cp in range(g.num_copies):
            base = (2 + vs) * cp
            idx, claimed = w[base], w[base + 1]
            items = list(w[base + 2:base + 2 + vs])
            bits = [w[g.num_routed + cp * g.bits + i] for i in range(g.bits)]
            for b_ in bits:
                out.append(K.sub(K.mul(b_, b_), b_))
            out.append(K.sub(_reduce_with_powers(K, bits, 2), idx))
            for b_ in bits:
                items = [K.add(items[i], K.mul(b_, K.sub(items[i + 1], items[i]))) for i in range(0, len(items), 2)]
            out.append(K.sub(items[0], claimed))
        for i in range(g.num_extra_constants):
            out.append(K.sub(c[i], w[(2 + vs) * g.num_copies + i]))
        return out
    if code in (G.REDUCING, G.REDUCING_EXT):
        n = g.num_coeffs
        ext = code == G.REDUCING_EXT
        alpha, acc = _alg(w, 2), _alg(w, 4)
        start_accs = 6 + (2 * n if ext else n)
        out = []
        for i in range(n):
            nxt = _alg(w, 0) if i == n - 1 else _alg(w, start_accs + 2 * i)
            coeff = _alg(w, 6 + 2 * i) if ext else (w[6 + i], K.zero)
            out.extend(_alg_sub(K, _alg_add(K, _alg_mul(K, acc, alpha), coeff), nxt))
            acc = nxt
        return out
    if code == G.EXPONENTIATION:
        n = g.num_power_bits
        base, bits, outp, inter = w[0], w[1:1 + n], w[1 + n], w[2 + n:2 + 2 * n]
        out = []
        for i in range(n):
            prev = K.one if i == 0 else K.mul(inter[i - 1], inter[i - 1])
            b_ = bits[n - 1 - i]
            mul_by = K.lift(K.mat(K.sub(K.mul(b_, base), K.sub(b_, K.one))))
            out.append(K.sub(K.mul(prev, mul_by), inter[i]))
        out.append(K.sub(outp, inter[n - 1]))
        return out
    if code == G.COSET_INTERPOLATION:
        return _eval_coset_interpolation(K, g, w)
    if code == G.U32_ARITHMETIC:
        out, n = [], g.num_ops
        for i in range(n):
            m0, m1, add, lo, hi, inv = w[6 * i:6 * i + 6]
            computed = K.add(K.mul(m0, m1), add)
            diff = K.sub(K.const(0xFFFFFFFF), hi)
            out.append(K.mul(K.lift(K.mat(K.sub(K.mul(inv, diff), K.one))), lo))
            out.append(K.sub(K.add(K.scale(1 << 32, hi), lo), computed))
            limbs = [w[6 * n + 32 * i + j] for j in range(32)]
            for j in reversed(range(32)):
                out.append(_range_product(K, limbs[j], 4))
            out.append(K.sub(_reduce_with_powers(K, limbs[:16], 4), lo))
            out.append(K.sub(_reduce_with_powers(K, limbs[16:], 4), hi))
        return out
    if code == G.U32_ADD_MANY:
        na, n = g.num_addends, g.num_ops
        per, out = na + 3, []
        for i in range(n):
            comp = w[per * i + na]
            for j in range(na):
                comp = K.add(comp, w[per * i + j])
            res, carry = w[per * i + na + 1], w[per * i + na + 2]
            out.append(K.sub(K.add(K.scale(1 << 32, carry), res), comp))
            limbs = [w[per * n + 18 * i + j] for j in range(18)]
            for j in reversed(range(18)):
                out.append(_range_product(K, limbs[j], 4))
            out.append(K.sub(_reduce_with_powers(K, limbs[:16], 4), res))
            out.append(K.sub(_reduce_with_powers(K, limbs[16:], 4), carry))
        return out
    if code == G.U32_SUBTRACTION:
        out, n = [], g.num_ops
        for i in range(n):
            x, y, bin_, res, bout = w[5 * i:5 * i + 5]
            out.append(K.sub(res, K.add(K.sub(K.sub(x, y), bin_), K.scale(1 << 32, bout))))
            limbs = [w[5 * n + 16 * i + j] for j in range(16)]
            for j in reversed(range(16)):
                out.append(_range_product(K, limbs[j], 4))
            out.append(K.sub(_reduce_with_powers(K, limbs, 4), res))
            out.append(K.mul(bout, K.sub(K.one, bout)))
        return out
    if code == G.U32_RANGE_CHECK:
        out, n = [], g.num_input_limbs
        for i in range(n):
            aux = [w[n + 16 * i + j] for j in range(16)]
            out.append(K.sub(_reduce_with_powers(K, aux, 4), w[i]))
            out.extend(_range_product(K, a, 4) for a in aux)
        return out
    if code == G.COMPARISON:
        return _eval_comparison(K, g, w)
    if code == G.U32_INTERLEAVE:
        out, n = [], g.num_ops
        for i in range(n):
            bits = [w[2 * n + 32 * i + j] for j in range(32)]
            out.append(K.sub(_reduce_with_powers(K, bits[::-1], 2), w[2 * i]))
            out.append(K.sub(_reduce_with_powers(K, bits[::-1], 4), w[2 * i + 1]))
            out.extend(_range_product(K, b_, 2) for b_ in bits)
        return out
    if code in (G.UNINTERLEAVE_TO_U32, G.UNINTERLEAVE_TO_B32):
        out, n = [], g.num_ops
        base = 4 if code == G.UNINTERLEAVE_TO_B32 else 2
        for i in range(n):
            bits = [w[3 * n + 64 * i + j] for j in range(64)]
            out.append(K.sub(_reduce_with_powers(K, bits[::-1], 2), w[3 * i]))
            out.append(K.sub(_reduce_with_powers(K, bits[0::2][::-1], base), w[3 * i + 1]))
            out.append(K.sub(_reduce_with_powers(K, bits[1::2][::-1], base), w[3 * i + 2]))
            out.extend(_range_product(K, b_, 2) for b_ in bits)
        return out
    raise ValueError("no in-circuit evaluator for " + g.id())
