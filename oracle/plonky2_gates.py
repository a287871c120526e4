"""plonky2 gate-constraint evaluators and the vanishing polynomial.  TEST INFRASTRUCTURE.

Restates the reference's gate evaluators:
  gnark-plonky2-verifier/plonk/gates/evaluate_gates.go:34-105   filters (selector groups) and the sum over gates
  plonk/gates/arithmetic_gate.go:64-84, arithmetic_extension_gate.go:62-86, multiplication_extension_gate.go:56-76,
  base_sum_gate.go:66-96, constant_gate.go:59-69, noop_gate.go, public_input_gate.go:32-51,
  poseidon_gate.go:84-181, poseidon_mds_gate.go:44-99, random_access_gate.go:127-190,
  reducing_gate.go:76-110, reducing_extension_gate.go:75-109, exponentiation_gate.go:85-128,
  coset_interpolation_gate.go:152-226 (+ goldilocks/quadratic_extension_algebra.go:47-131)
and the in-tree custom gates of the Ed25519/SHA circuits:
  crypto/plonky2_u32/src/gates/arithmetic_u32.rs:110-170   U32ArithmeticGate
  crypto/plonky2_u32/src/gates/add_many_u32.rs              U32AddManyGate
  crypto/plonky2_u32/src/gates/subtraction_u32.rs           U32SubtractionGate
  crypto/plonky2_u32/src/gates/range_check_u32.rs           U32RangeCheckGate
  crypto/plonky2_u32/src/gates/comparison.rs                ComparisonGate
  crypto/plonky2_u32/src/gates/{interleave_u32,uninterleave_to_u32,uninterleave_to_b32}.rs  (SHA-256 circuits)
and the vanishing-polynomial combination of plonk/plonk.go:60-250.

Every evaluator is written once over a field adapter K (BaseK = Goldilocks integers, used by the prover
restatement at the points of the LDE coset; ExtK = quadratic-extension pairs, used by the verifier at zeta),
exactly like plonky2's eval_unfiltered / eval_unfiltered_base pair.
Pinned: the four golden proofs of the reference satisfy vanishing(zeta) == Z_H(zeta) * t(zeta)
(tests/test_oracle_plonky2.py), which exercises 13 of the gate types; the u32 gates are pinned only by
self-consistency with their witness generators (no golden proof of an inner circuit is in the tree).
"""
import re

from . import goldilocks as gl
from . import poseidon_gl as pgl

P = gl.P
UNUSED_SELECTOR = (1 << 32) - 1


class BaseK:
    zero, one = 0, 1

    @staticmethod
    def add(a, b):
        return (a + b) % P

    @staticmethod
    def sub(a, b):
        return (a - b) % P

    @staticmethod
    def mul(a, b):
        return a * b % P

    @staticmethod
    def const(c):
        return c % P


class ExtK:
    zero, one = (0, 0), (1, 0)
    add = staticmethod(gl.ext_add)
    sub = staticmethod(gl.ext_sub)
    mul = staticmethod(gl.ext_mul)

    @staticmethod
    def const(c):
        return (c % P, 0)


# ---- the degree-2 "extension algebra" over K (pairs of K; X^2 = 7) -- quadratic_extension_algebra.go
def alg(K, wires, start):
    return (wires[start], wires[start + 1])


def alg_add(K, a, b):
    return (K.add(a[0], b[0]), K.add(a[1], b[1]))


def alg_sub(K, a, b):
    return (K.sub(a[0], b[0]), K.sub(a[1], b[1]))


def alg_mul(K, a, b):
    w = K.const(gl.W)
    return (K.add(K.mul(a[0], b[0]), K.mul(w, K.mul(a[1], b[1]))), K.add(K.mul(a[0], b[1]), K.mul(a[1], b[0])))


def alg_scalar(K, s, a):
    return (K.mul(s, a[0]), K.mul(s, a[1]))


def reduce_with_powers(K, terms, base):
    acc = K.zero
    for t in reversed(terms):
        acc = K.add(K.mul(acc, base), t)
    return acc


# ---- gates.  Each class: id regex, num_constants, degree, num_constraints, eval(K, consts, wires, pi_hash)
class Gate:
    num_constants = 0

    def eval(self, K, c, w, pih):
        raise NotImplementedError


class NoopGate(Gate):
    degree, num_constraints = 0, 0

    def eval(self, K, c, w, pih):
        return []


class ConstantGate(Gate):
    def __init__(self, num_consts):
        self.n = self.num_constants = self.num_constraints = num_consts
        self.degree = 1

    def eval(self, K, c, w, pih):
        return [K.sub(c[i], w[i]) for i in range(self.n)]


class PublicInputGate(Gate):
    degree, num_constraints = 1, 4

    def eval(self, K, c, w, pih):
        return [K.sub(w[i], K.const(pih[i])) for i in range(4)]


class ArithmeticGate(Gate):
    num_constants, degree = 2, 3

    def __init__(self, num_ops):
        self.n = self.num_constraints = num_ops

    def eval(self, K, c, w, pih):
        out = []
        for i in range(self.n):
            m0, m1, a, o = w[4 * i:4 * i + 4]
            out.append(K.sub(o, K.add(K.mul(K.mul(m0, m1), c[0]), K.mul(a, c[1]))))
        return out


class ArithmeticExtensionGate(Gate):
    num_constants, degree = 2, 3

    def __init__(self, num_ops):
        self.n = num_ops
        self.num_constraints = 2 * num_ops

    def eval(self, K, c, w, pih):
        out = []
        for i in range(self.n):
            m0, m1, a, o = (alg(K, w, 8 * i + 2 * k) for k in range(4))
            comp = alg_add(K, alg_scalar(K, c[1], a), alg_scalar(K, c[0], alg_mul(K, m0, m1)))
            out.extend(alg_sub(K, o, comp))
        return out


class MulExtensionGate(Gate):
    num_constants, degree = 1, 3

    def __init__(self, num_ops):
        self.n = num_ops
        self.num_constraints = 2 * num_ops

    def eval(self, K, c, w, pih):
        out = []
        for i in range(self.n):
            m0, m1, o = (alg(K, w, 6 * i + 2 * k) for k in range(3))
            out.extend(alg_sub(K, o, alg_scalar(K, c[0], alg_mul(K, m0, m1))))
        return out


class BaseSumGate(Gate):
    def __init__(self, num_limbs, base):
        self.n, self.base = num_limbs, base
        self.degree, self.num_constraints = base, 1 + num_limbs

    def eval(self, K, c, w, pih):
        limbs = w[1:1 + self.n]
        out = [K.sub(reduce_with_powers(K, limbs, K.const(self.base)), w[0])]
        for l in limbs:
            acc = K.one
            for i in range(self.base):
                acc = K.mul(acc, K.sub(l, K.const(i)))
            out.append(acc)
        return out


class PoseidonGate(Gate):
    degree, num_constraints = 7, 12 + 12 * 3 + 22 + 12 * 4 + 1 + 4
    SWAP, DELTA, FULL0, PARTIAL, FULL1, END = 24, 25, 29, 65, 87, 135

    def eval(self, K, c, w, pih):
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

        def sbox(x):
            x2 = K.mul(x, x)
            x4 = K.mul(x2, x2)
            return K.mul(x4, K.mul(x, x2))

        def mds(v):
            res = []
            for r in range(12):
                acc = K.zero
                for i in range(12):
                    acc = K.add(acc, K.mul(v[(i + r) % 12], K.const(pgl.CIRC[i])))
                res.append(K.add(acc, K.mul(v[r], K.const(pgl.DIAG[r]))))
            return res

        rnd = 0
        for r in range(4):
            st = [K.add(st[i], K.const(pgl.RC[12 * rnd + i])) for i in range(12)]
            if r:
                for i in range(12):
                    sin = w[29 + 12 * (r - 1) + i]
                    out.append(K.sub(st[i], sin))
                    st[i] = sin
            st = mds([sbox(x) for x in st])
            rnd += 1
        st = [K.add(st[i], K.const(pgl.FP_FIRST[i])) for i in range(12)]
        res = [K.zero] * 12
        res[0] = st[0]
        for r in range(1, 12):
            for d in range(1, 12):
                res[d] = K.add(res[d], K.mul(st[r], K.const(pgl.FP_INIT[r - 1][d - 1])))
        st = res
        for r in range(22):
            sin = w[65 + r]
            out.append(K.sub(st[0], sin))
            s0 = sbox(sin)
            if r < 21:
                s0 = K.add(s0, K.const(pgl.FP_RC[r]))
            d = K.mul(s0, K.const(25))
            for i in range(1, 12):
                d = K.add(d, K.mul(st[i], K.const(pgl.FP_WHATS[r][i - 1])))
            st = [d] + [K.add(K.mul(s0, K.const(pgl.FP_VS[r][i - 1])), st[i]) for i in range(1, 12)]
        rnd += 22
        for r in range(4):
            st = [K.add(st[i], K.const(pgl.RC[12 * rnd + i])) for i in range(12)]
            for i in range(12):
                sin = w[87 + 12 * r + i]
                out.append(K.sub(st[i], sin))
                st[i] = sin
            st = mds([sbox(x) for x in st])
            rnd += 1
        for i in range(12):
            out.append(K.sub(st[i], w[12 + i]))
        return out


class PoseidonMdsGate(Gate):
    degree, num_constraints = 1, 24

    def eval(self, K, c, w, pih):
        ins = [alg(K, w, 2 * i) for i in range(12)]
        out = []
        for r in range(12):
            acc = (K.zero, K.zero)
            for i in range(12):
                acc = alg_add(K, acc, alg_scalar(K, K.const(pgl.CIRC[i]), ins[(i + r) % 12]))
            acc = alg_add(K, acc, alg_scalar(K, K.const(pgl.DIAG[r]), ins[r]))
            out.extend(alg_sub(K, alg(K, w, 2 * (12 + r)), acc))
        return out


class RandomAccessGate(Gate):
    def __init__(self, bits, num_copies, num_extra_constants):
        self.bits, self.copies, self.extra = bits, num_copies, num_extra_constants
        self.num_constants = num_extra_constants
        self.degree = bits + 1
        self.num_constraints = num_copies * (bits + 2) + num_extra_constants

    def eval(self, K, c, w, pih):
        vs = 1 << self.bits
        routed = (2 + vs) * self.copies + self.extra
        out = []
        for cp in range(self.copies):
            base = (2 + vs) * cp
            idx, claimed = w[base], w[base + 1]
            items = list(w[base + 2:base + 2 + vs])
            bits = [w[routed + cp * self.bits + i] for i in range(self.bits)]
            for b in bits:
                out.append(K.sub(K.mul(b, b), b))
            out.append(K.sub(reduce_with_powers(K, bits, K.const(2)), idx))
            for b in bits:
                items = [K.add(items[i], K.mul(b, K.sub(items[i + 1], items[i]))) for i in range(0, len(items), 2)]
            out.append(K.sub(items[0], claimed))
        for i in range(self.extra):
            out.append(K.sub(c[i], w[(2 + vs) * self.copies + i]))
        return out


class ReducingGate(Gate):
    degree = 2

    def __init__(self, num_coeffs):
        self.n = num_coeffs
        self.num_constraints = 2 * num_coeffs

    def _acc(self, K, w, i, start_accs):
        return alg(K, w, 0) if i == self.n - 1 else alg(K, w, start_accs + 2 * i)

    def eval(self, K, c, w, pih):
        alpha, acc = alg(K, w, 2), alg(K, w, 4)
        start_accs = 6 + self.n
        out = []
        for i in range(self.n):
            nxt = self._acc(K, w, i, start_accs)
            t = alg_add(K, alg_mul(K, acc, alpha), (w[6 + i], K.zero))
            out.extend(alg_sub(K, t, nxt))
            acc = nxt
        return out


class ReducingExtensionGate(ReducingGate):
    def eval(self, K, c, w, pih):
        alpha, acc = alg(K, w, 2), alg(K, w, 4)
        start_accs = 6 + 2 * self.n
        out = []
        for i in range(self.n):
            nxt = self._acc(K, w, i, start_accs)
            t = alg_add(K, alg_mul(K, acc, alpha), alg(K, w, 6 + 2 * i))
            out.extend(alg_sub(K, t, nxt))
            acc = nxt
        return out


class ExponentiationGate(Gate):
    degree = 4

    def __init__(self, num_power_bits):
        self.n = num_power_bits
        self.num_constraints = num_power_bits + 1

    def eval(self, K, c, w, pih):
        n = self.n
        base, bits, outp, inter = w[0], w[1:1 + n], w[1 + n], w[2 + n:2 + 2 * n]
        out = []
        for i in range(n):
            prev = K.one if i == 0 else K.mul(inter[i - 1], inter[i - 1])
            b = bits[n - 1 - i]
            mul_by = K.sub(K.mul(b, base), K.sub(b, K.one))   # b*base + 1 - b
            out.append(K.sub(K.mul(prev, mul_by), inter[i]))
        out.append(K.sub(outp, inter[n - 1]))
        return out


class CosetInterpolationGate(Gate):
    def __init__(self, subgroup_bits, degree, weights):
        self.sb, self.deg, self.weights = subgroup_bits, degree, weights
        self.np = 1 << subgroup_bits
        self.n_inter = (self.np - 2) // (degree - 1)
        self.degree = degree
        self.num_constraints = 2 + 2 + 4 * self.n_inter

    def _partial(self, K, dom, vals, wts, point, ev, prod):
        for x, v, wt in zip(dom, vals, wts):
            term = alg_sub(K, point, (K.const(x), K.zero))
            wv = alg_scalar(K, K.const(wt), v)
            ev = alg_add(K, alg_mul(K, ev, term), alg_mul(K, wv, prod))
            prod = alg_mul(K, prod, term)
        return ev, prod

    def eval(self, K, c, w, pih):
        np_, d = self.np, self.deg
        start_pt = 1 + 2 * np_
        start_val = start_pt + 2
        start_inter = start_val + 2
        shift = w[0]
        point = alg(K, w, start_pt)
        shifted = alg(K, w, start_inter + 4 * self.n_inter)
        out = list(alg_add(K, alg_scalar(K, K.sub(K.zero, shift), shifted), point))
        g = gl.root_of_unity(self.sb)
        dom = [pow(g, i, P) for i in range(np_)]
        vals = [alg(K, w, 1 + 2 * i) for i in range(np_)]
        ev, prod = self._partial(K, dom[:d], vals[:d], self.weights[:d], shifted, (K.zero, K.zero), (K.one, K.zero))
        for i in range(self.n_inter):
            iev = alg(K, w, start_inter + 2 * i)
            ipr = alg(K, w, start_inter + 2 * (self.n_inter + i))
            out.extend(alg_sub(K, iev, ev))
            out.extend(alg_sub(K, ipr, prod))
            s = 1 + (d - 1) * (i + 1)
            e = min(s + d - 1, np_)
            ev, prod = self._partial(K, dom[s:e], vals[s:e], self.weights[s:e], shifted, iev, ipr)
        out.extend(alg_sub(K, alg(K, w, start_val), ev))
        return out


# ---- in-tree u32 gates (crypto/plonky2_u32/src/gates)
class U32ArithmeticGate(Gate):
    """arithmetic_u32.rs:110-170: m0*m1 + addend = hi*2^32 + lo, hi/lo range-checked by 2-bit limbs,
    canonicity of (hi, lo) through an inverse wire"""
    degree = 4   # limb products: prod_{k<4}(limb - k)

    def __init__(self, num_ops):
        self.n = num_ops
        self.num_constraints = num_ops * (4 + 32)

    def eval(self, K, c, w, pih):
        out = []
        n = self.n
        for i in range(n):
            m0, m1, add, lo, hi, inv = w[6 * i:6 * i + 6]
            computed = K.add(K.mul(m0, m1), add)
            diff = K.sub(K.const(0xFFFFFFFF), hi)
            hi_not_max = K.sub(K.mul(inv, diff), K.one)
            out.append(K.mul(hi_not_max, lo))
            combined = K.add(K.mul(hi, K.const(1 << 32)), lo)
            out.append(K.sub(combined, computed))
            comb_lo, comb_hi = K.zero, K.zero
            limbs = [w[6 * n + 32 * i + j] for j in range(32)]
            for j in reversed(range(32)):
                l = limbs[j]
                prod = K.one
                for k in range(4):
                    prod = K.mul(prod, K.sub(l, K.const(k)))
                out.append(prod)
            for j in reversed(range(16)):
                comb_lo = K.add(K.mul(comb_lo, K.const(4)), limbs[j])
            for j in reversed(range(16, 32)):
                comb_hi = K.add(K.mul(comb_hi, K.const(4)), limbs[j])
            out.append(K.sub(comb_lo, lo))
            out.append(K.sub(comb_hi, hi))
        return out


def _range_product(K, x, base):
    acc = K.one
    for k in range(base):
        acc = K.mul(acc, K.sub(x, K.const(k)))
    return acc


class U32AddManyGate(Gate):
    """add_many_u32.rs `eval_unfiltered`"""
    degree = 4

    def __init__(self, num_addends, num_ops):
        self.na, self.n = num_addends, num_ops
        self.num_constraints = num_ops * 21

    def eval(self, K, c, w, pih):
        per, out = self.na + 3, []
        for i in range(self.n):
            comp = w[per * i + self.na]
            for j in range(self.na):
                comp = K.add(comp, w[per * i + j])
            res, carry = w[per * i + self.na + 1], w[per * i + self.na + 2]
            out.append(K.sub(K.add(K.mul(carry, K.const(1 << 32)), res), comp))
            cr, cc = K.zero, K.zero
            for j in reversed(range(18)):
                l = w[per * self.n + 18 * i + j]
                out.append(_range_product(K, l, 4))
                if j < 16:
                    cr = K.add(K.mul(K.const(4), cr), l)
                else:
                    cc = K.add(K.mul(K.const(4), cc), l)
            out.append(K.sub(cr, res))
            out.append(K.sub(cc, carry))
        return out


class U32SubtractionGate(Gate):
    """subtraction_u32.rs `eval_unfiltered`"""
    degree = 4

    def __init__(self, num_ops):
        self.n = num_ops
        self.num_constraints = num_ops * 19

    def eval(self, K, c, w, pih):
        out = []
        for i in range(self.n):
            x, y, bin_, res, bout = w[5 * i:5 * i + 5]
            initial = K.sub(K.sub(x, y), bin_)
            out.append(K.sub(res, K.add(initial, K.mul(K.const(1 << 32), bout))))
            comb = K.zero
            for j in reversed(range(16)):
                l = w[5 * self.n + 16 * i + j]
                out.append(_range_product(K, l, 4))
                comb = K.add(K.mul(K.const(4), comb), l)
            out.append(K.sub(comb, res))
            out.append(K.mul(bout, K.sub(K.one, bout)))
        return out


class U32RangeCheckGate(Gate):
    """range_check_u32.rs `eval_unfiltered`"""
    degree = 4

    def __init__(self, num_input_limbs):
        self.n = num_input_limbs
        self.num_constraints = num_input_limbs * 17

    def eval(self, K, c, w, pih):
        out = []
        for i in range(self.n):
            aux = [w[self.n + 16 * i + j] for j in range(16)]
            out.append(K.sub(reduce_with_powers(K, aux, K.const(4)), w[i]))
            for a in aux:
                out.append(_range_product(K, a, 4))
        return out


class ComparisonGate(Gate):
    """comparison.rs:106-190 `eval_unfiltered`"""

    def __init__(self, num_bits, num_chunks):
        self.nb, self.nc = num_bits, num_chunks
        self.cb = -(-num_bits // num_chunks)
        self.degree = 1 << self.cb
        self.num_constraints = 6 + 5 * num_chunks + self.cb

    def eval(self, K, c, w, pih):
        nc, cb = self.nc, self.cb
        size = 1 << cb
        first = [w[4 + i] for i in range(nc)]
        second = [w[4 + nc + i] for i in range(nc)]
        out = [K.sub(reduce_with_powers(K, first, K.const(size)), w[0]), K.sub(reduce_with_powers(K, second, K.const(size)), w[1])]
        msd = K.zero
        for i in range(nc):
            out.append(_range_product(K, first[i], size))
            out.append(_range_product(K, second[i], size))
            diff = K.sub(second[i], first[i])
            dummy, eq = w[4 + 2 * nc + i], w[4 + 3 * nc + i]
            out.append(K.sub(K.mul(diff, dummy), K.sub(K.one, eq)))
            out.append(K.mul(eq, diff))
            inter = w[4 + 4 * nc + i]
            out.append(K.sub(inter, K.mul(eq, msd)))
            msd = K.add(inter, K.mul(K.sub(K.one, eq), diff))
        out.append(K.sub(w[3], msd))
        bits = [w[4 + 5 * nc + i] for i in range(cb + 1)]
        for b in bits:
            out.append(K.mul(b, K.sub(K.one, b)))
        out.append(K.sub(K.add(K.const(size), w[3]), reduce_with_powers(K, bits, K.const(2))))
        out.append(K.sub(w[2], bits[cb]))
        return out


class U32InterleaveGate(Gate):
    """interleave_u32.rs:103-139 `eval_unfiltered`"""
    degree = 2

    def __init__(self, num_ops):
        self.n = num_ops
        self.num_constraints = num_ops * 34

    def eval(self, K, c, w, pih):
        out = []
        for i in range(self.n):
            bits = [w[2 * self.n + 32 * i + j] for j in range(32)]           # big-endian
            out.append(K.sub(reduce_with_powers(K, bits[::-1], K.const(2)), w[2 * i]))
            out.append(K.sub(reduce_with_powers(K, bits[::-1], K.const(4)), w[2 * i + 1]))
            out.extend(_range_product(K, b, 2) for b in bits)
        return out


class UninterleaveToU32Gate(Gate):
    """uninterleave_to_u32.rs:112-159 `eval_unfiltered`; with to_b32 the weights are 4^(31-j) (uninterleave_to_b32.rs)"""
    degree, to_b32 = 2, False

    def __init__(self, num_ops):
        self.n = num_ops
        self.num_constraints = num_ops * 67

    def eval(self, K, c, w, pih):
        out = []
        for i in range(self.n):
            bits = [w[3 * self.n + 64 * i + j] for j in range(64)]
            out.append(K.sub(reduce_with_powers(K, bits[::-1], K.const(2)), w[3 * i]))
            ev, od = K.zero, K.zero
            for j in range(32):
                coeff = K.const(1 << (2 * (31 - j)) if self.to_b32 else 1 << (31 - j))
                ev = K.add(ev, K.mul(coeff, bits[2 * j]))
                od = K.add(od, K.mul(coeff, bits[2 * j + 1]))
            out.append(K.sub(ev, w[3 * i + 1]))
            out.append(K.sub(od, w[3 * i + 2]))
            out.extend(_range_product(K, b, 2) for b in bits)
        return out


class UninterleaveToB32Gate(UninterleaveToU32Gate):
    to_b32 = True


GATE_PATTERNS = [
    (re.compile(r"^NoopGate"), lambda m: NoopGate()),
    (re.compile(r"^ConstantGate \{ num_consts: (\d+) \}"), lambda m: ConstantGate(int(m[1]))),
    (re.compile(r"^PublicInputGate"), lambda m: PublicInputGate()),
    (re.compile(r"^ArithmeticGate \{ num_ops: (\d+) \}"), lambda m: ArithmeticGate(int(m[1]))),
    (re.compile(r"^ArithmeticExtensionGate \{ num_ops: (\d+) \}"), lambda m: ArithmeticExtensionGate(int(m[1]))),
    (re.compile(r"^MulExtensionGate \{ num_ops: (\d+) \}"), lambda m: MulExtensionGate(int(m[1]))),
    (re.compile(r"^BaseSumGate \{ num_limbs: (\d+) \} \+ Base: (\d+)"), lambda m: BaseSumGate(int(m[1]), int(m[2]))),
    (re.compile(r"^PoseidonGate"), lambda m: PoseidonGate()),
    (re.compile(r"^PoseidonMdsGate"), lambda m: PoseidonMdsGate()),
    (re.compile(r"^RandomAccessGate \{ bits: (\d+), num_copies: (\d+), num_extra_constants: (\d+)"),
     lambda m: RandomAccessGate(int(m[1]), int(m[2]), int(m[3]))),
    (re.compile(r"^ReducingGate \{ num_coeffs: (\d+) \}"), lambda m: ReducingGate(int(m[1]))),
    (re.compile(r"^ReducingExtensionGate \{ num_coeffs: (\d+) \}"), lambda m: ReducingExtensionGate(int(m[1]))),
    (re.compile(r"^ExponentiationGate \{ num_power_bits: (\d+)"), lambda m: ExponentiationGate(int(m[1]))),
    (re.compile(r"^CosetInterpolationGate \{ subgroup_bits: (\d+), degree: (\d+), barycentric_weights: \[([0-9, ]+)\]"),
     lambda m: CosetInterpolationGate(int(m[1]), int(m[2]), [int(x) for x in m[3].split(",")])),
    (re.compile(r"^U32ArithmeticGate \{ num_ops: (\d+)"), lambda m: U32ArithmeticGate(int(m[1]))),
    (re.compile(r"^U32AddManyGate \{ num_addends: (\d+), num_ops: (\d+)"), lambda m: U32AddManyGate(int(m[1]), int(m[2]))),
    (re.compile(r"^U32SubtractionGate \{ num_ops: (\d+)"), lambda m: U32SubtractionGate(int(m[1]))),
    (re.compile(r"^U32RangeCheckGate \{ num_input_limbs: (\d+)"), lambda m: U32RangeCheckGate(int(m[1]))),
    (re.compile(r"^ComparisonGate \{ num_bits: (\d+), num_chunks: (\d+)"), lambda m: ComparisonGate(int(m[1]), int(m[2]))),
    (re.compile(r"^U32InterleaveGate \{ num_ops: (\d+)"), lambda m: U32InterleaveGate(int(m[1]))),
    (re.compile(r"^UninterleaveToU32Gate \{ num_ops: (\d+)"), lambda m: UninterleaveToU32Gate(int(m[1]))),
    (re.compile(r"^UninterleaveToB32Gate \{ num_ops: (\d+)"), lambda m: UninterleaveToB32Gate(int(m[1]))),
]


def gate_from_id(gid):
    for rx, mk in GATE_PATTERNS:
        m = rx.match(gid)
        if m:
            return mk(m)
    raise ValueError("unknown gate id " + gid)


# ---- evaluate_gates.go:34-105
def compute_filter(K, row, group, s, many_selectors):
    prod = K.one
    for i in range(group[0], group[1]):
        if i != row:
            prod = K.mul(prod, K.sub(K.const(i), s))
    if many_selectors:
        prod = K.mul(prod, K.sub(K.const(UNUSED_SELECTOR), s))
    return prod


def evaluate_gate_constraints(K, gates, selectors_info, num_gate_constraints, consts, wires, pih):
    idx = selectors_info["selector_indices"]
    groups = [(g["start"], g["end"]) for g in selectors_info["groups"]]
    nsel = len(groups)
    out = [K.zero] * num_gate_constraints
    for row, g in enumerate(gates):
        f = compute_filter(K, row, groups[idx[row]], consts[idx[row]], nsel > 1)
        cs = g.eval(K, consts[nsel:], wires, pih)
        assert len(cs) == g.num_constraints <= num_gate_constraints, (type(g).__name__, len(cs), g.num_constraints)
        for i, cst in enumerate(cs):
            out[i] = K.add(out[i], K.mul(cst, f))
    return out


def vanishing_terms(K, common, gates, x, x_pow_n_minus_1_over_l0_den, consts, sigmas, wires, zs, zs_next, pps, betas, gammas,
                    pih):
    """plonk.go:121-207 -- the list [Z1 terms | partial-product terms | gate constraints] at one point.
    `x_pow_n_minus_1_over_l0_den` is L_0(x) = (x^n - 1) / (n (x - 1)), supplied by the caller."""
    cfg = common["config"]
    routed, nch = cfg["num_routed_wires"], cfg["num_challenges"]
    npp, qdf = common["num_partial_products"], common["quotient_degree_factor"]
    l0 = x_pow_n_minus_1_over_l0_den
    constraint_terms = evaluate_gate_constraints(K, gates, common["selectors_info"], common["num_gate_constraints"], consts, wires,
                                                 pih)
    s_ids = [K.mul(x, K.const(k)) for k in common["k_is"][:routed]]
    z1, ppt = [], []
    for i in range(nch):
        z1.append(K.mul(l0, K.sub(zs[i], K.one)))
        b, g = K.const(betas[i]), K.const(gammas[i])
        nums = [K.add(K.mul(b, s_ids[j]), K.add(wires[j], g)) for j in range(routed)]
        dens = [K.add(K.mul(b, sigmas[j]), K.add(wires[j], g)) for j in range(routed)]
        accs = [zs[i]] + list(pps[i * npp:(i + 1) * npp]) + [zs_next[i]]
        for k in range(npp + 1):
            n_, d_ = K.one, K.one
            for j in range(k * qdf, min((k + 1) * qdf, routed)):
                n_ = K.mul(n_, nums[j])
                d_ = K.mul(d_, dens[j])
            ppt.append(K.sub(K.mul(accs[k], n_), K.mul(accs[k + 1], d_)))
    return z1 + ppt + constraint_terms
