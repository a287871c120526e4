"""In-circuit plonky2 verifier: `recursive_proof` of the reference.

The reference folds proofs with near_bft_finality/src/prove_crypto/recursion.rs:16-97:
    builder = CircuitBuilder::new(standard_recursion_config())
    pt = builder.add_virtual_proof_with_pis(inner_common);  vd = { add_virtual_cap(cap_height), add_virtual_hash() }
    pw.set_proof_with_pis_target(pt, proof); pw.set_cap_target(..); pw.set_hash_target(..)
    builder.verify_proof(pt, vd, inner_common)            (once or twice)
    register optional public inputs;  data = builder.build();  proof = data.prove(pw)
`verify_proof` itself lives in the un-vendored plonky2 fork (wormhole-foundation/plonky2-near@2244a9d,
plonk/recursive_verifier.rs, fri/recursive_verifier.rs, plonk/vanishing_poly.rs).  What it has to check is stated in-tree by
the Go verifier this package's oracle restates -- gnark-plonky2-verifier/verifier/verifier.go:41-82,143-170 (public-input hash,
challenges, Verify), challenger/challenger.go:42-166, plonk/plonk.go:60-250 (vanishing identity), fri/fri.go:40-497 (proof of
work, Merkle paths to the cap, combine-initial, coset interpolation folds, final polynomial) -- and the gadgets used are the
ones whose gates appear in the reference's recursion circuits (near_bft_finality/proofs/*/common_data.json "gates":
ArithmeticGate, ArithmeticExtensionGate, MulExtensionGate, BaseSumGate{63}, ConstantGate, CosetInterpolationGate{4, 6},
ExponentiationGate{66}, PoseidonGate, PoseidonMdsGate, RandomAccessGate{4}, ReducingGate{43}, ReducingExtensionGate{32},
PublicInputGate, NoopGate).  This module builds that verifier on the host circuit builder; the outer proof is produced by
the GPU prover like any other circuit.
"""
from . import gates as G
from .builder import GENERATOR, OP_SPLIT, P, CircuitBuilder, Target, root_of_unity, standard_recursion_config  # noqa: F401
from .gate_circuits import CircuitK, eval_gate_circuit

W = 7
UNUSED_SELECTOR = (1 << 32) - 1


# ---------------------------------------------------------------------------------- extension-field values (witness side)
def e_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def e_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def e_mul(a, b):
    return ((a[0] * b[0] + W * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def e_scalar(c, a):
    return (c * a[0] % P, c * a[1] % P)


def e_inv(a):
    d = pow((a[0] * a[0] - W * a[1] * a[1]) % P, P - 2, P)
    return (a[0] * d % P, (-a[1]) * d % P)


def barycentric_weights(points):
    """w_i = 1 / prod_{j != i} (x_i - x_j)"""
    out = []
    for i, x in enumerate(points):
        d = 1
        for j, y in enumerate(points):
            if i != j:
                d = d * (x - y) % P
        out.append(pow(d, P - 2, P))
    return out


def coset_interpolation_gate(subgroup_bits, max_degree):
    """CosetInterpolationGate::with_max_degree: the smallest degree in 2..=max_degree with the fewest wires"""
    n_pts = 1 << subgroup_bits
    g = root_of_unity(subgroup_bits)
    wts = barycentric_weights([pow(g, i, P) for i in range(n_pts)])
    best = None
    for d in range(2, max_degree + 1):
        gate = G.CosetInterpolationGate(subgroup_bits, d, wts)
        if best is None or gate.num_wires < best.num_wires:
            best = gate
    return best


class RecursiveCircuitBuilder(CircuitBuilder):
    """CircuitBuilder + the extension-field, reducing, interpolation, exponentiation, Merkle and challenger gadgets the
    in-circuit verifier needs."""

    def __init__(self, config=None):
        super().__init__(config or standard_recursion_config())
        self._aext_slot, self._mext_slot, self._ext_memo = {}, {}, {}
        self.k_adapter = None

    # ---- extension targets: pairs of targets
    def add_virtual_ext(self):
        return (self.add_virtual_target(), self.add_virtual_target())

    def constant_ext(self, c):
        c = (c, 0) if isinstance(c, int) else c
        return (self.constant(c[0]), self.constant(c[1]))

    def zero_ext(self):
        return self.constant_ext(0)

    def one_ext(self):
        return self.constant_ext(1)

    def convert_to_ext(self, t):
        return (t, self.zero())

    def ext_as_constant(self, e):
        a, b = self.target_as_constant(e[0]), self.target_as_constant(e[1])
        return None if a is None or b is None else (a, b)

    def connect_ext(self, a, b):
        self.connect(a[0], b[0])
        self.connect(a[1], b[1])

    def arithmetic_ext(self, c0, c1, m0, m1, addend):
        """c0*m0*m1 + c1*addend (ArithmeticExtensionGate; MulExtensionGate when the addend vanishes)"""
        c0 %= P
        c1 %= P
        k0, k1, ka = self.ext_as_constant(m0), self.ext_as_constant(m1), self.ext_as_constant(addend)
        if k0 is not None and k1 is not None and ka is not None:
            return self.constant_ext(e_add(e_scalar(c0, e_mul(k0, k1)), e_scalar(c1, ka)))
        if c0 == 0 or k0 == (0, 0) or k1 == (0, 0):
            if c1 == 0 or ka == (0, 0):
                return self.zero_ext()
            if c1 == 1:
                return addend
            c0, m0, m1 = 0, self.zero_ext(), self.zero_ext()
        if c0 == 1 and (c1 == 0 or ka == (0, 0)):
            if k0 == (1, 0):
                return m1
            if k1 == (1, 0):
                return m0
        no_addend = c1 == 0 or ka == (0, 0)
        key = (c0, 0 if no_addend else c1, m0[0].key(), m0[1].key(), m1[0].key(), m1[1].key(),
               None if no_addend else (addend[0].key(), addend[1].key()))
        hit = self._ext_memo.get(key)
        if hit is not None:
            return hit
        if no_addend:
            gate = G.MulExtensionGate.new_from_config(self.config)
            slot = self._mext_slot.get(c0)
            if slot is None or slot[1] == gate.num_ops:
                slot = self._mext_slot[c0] = [self.add_gate(gate, [c0]), 0]
            row, i = slot
            slot[1] += 1
            w = [Target(row, 6 * i + k) for k in range(6)]
            ins = [m0[0], m0[1], m1[0], m1[1]]
            for a, t in zip(ins, w[:4]):
                self.connect(a, t)

            def gen(v, w=w, c0=c0):
                o = e_scalar(c0, e_mul((v[0], v[1]), (v[2], v[3])))
                return [(w[4], o[0]), (w[5], o[1])]
            self.add_generator(w[:4], gen, OP_EXT_MUL, (c0,), outs=w[4:6])
            out = (w[4], w[5])
        else:
            gate = G.ArithmeticExtensionGate.new_from_config(self.config)
            slot = self._aext_slot.get((c0, c1))
            if slot is None or slot[1] == gate.num_ops:
                slot = self._aext_slot[(c0, c1)] = [self.add_gate(gate, [c0, c1]), 0]
            row, i = slot
            slot[1] += 1
            w = [Target(row, 8 * i + k) for k in range(8)]
            ins = [m0[0], m0[1], m1[0], m1[1], addend[0], addend[1]]
            for a, t in zip(ins, w[:6]):
                self.connect(a, t)

            def gen(v, w=w, c0=c0, c1=c1):
                o = e_add(e_scalar(c0, e_mul((v[0], v[1]), (v[2], v[3]))), e_scalar(c1, (v[4], v[5])))
                return [(w[6], o[0]), (w[7], o[1])]
            self.add_generator(w[:6], gen, OP_EXT_ARITH, (c0, c1), outs=w[6:8])
            out = (w[6], w[7])
        self._ext_memo[key] = out
        return out

    def mul_ext(self, a, b):
        return self.arithmetic_ext(1, 0, a, b, self.zero_ext())

    def square_ext(self, a):
        return self.mul_ext(a, a)

    def add_ext(self, a, b):
        return self.arithmetic_ext(1, 1, self.one_ext(), a, b)

    def sub_ext(self, a, b):
        return self.arithmetic_ext(1, P - 1, self.one_ext(), a, b)

    def mul_add_ext(self, a, b, c):
        return self.arithmetic_ext(1, 1, a, b, c)

    def mul_sub_ext(self, a, b, c):
        return self.arithmetic_ext(1, P - 1, a, b, c)

    def scalar_mul_ext(self, t, e):
        return self.mul_ext(self.convert_to_ext(t), e)

    def mul_many_ext(self, terms):
        terms = list(terms)
        acc = terms[0] if terms else self.one_ext()
        for t in terms[1:]:
            acc = self.mul_ext(acc, t)
        return acc

    def div_add_ext(self, x, y, z):
        """x / y + z: the inverse of y is a witness, constrained by y * inv == 1"""
        inv = self.add_virtual_ext()
        self.add_generator([y[0], y[1]], lambda v, inv=inv: list(zip(inv, e_inv((v[0], v[1])))), OP_EXT_INV, outs=list(inv))
        self.connect_ext(self.mul_ext(y, inv), self.one_ext())
        return self.mul_add_ext(x, inv, z)

    def div_ext(self, x, y):
        return self.div_add_ext(x, y, self.zero_ext())

    def exp_power_of_2_ext(self, base, power_log):
        for _ in range(power_log):
            base = self.square_ext(base)
        return base

    def exp_u64_ext(self, base, exponent):
        if exponent == 0:
            return self.one_ext()
        cur, prod = base, None
        for j in range(exponent.bit_length()):
            if j:
                cur = self.square_ext(cur)
            if (exponent >> j) & 1:
                prod = cur if prod is None else self.mul_ext(prod, cur)
        return prod

    def exp_power_of_2(self, base, power_log):
        for _ in range(power_log):
            base = self.mul(base, base)
        return base

    # ---- bit decompositions as the verifier circuits use them (BaseSumGate{63} rows)
    def split_le_63(self, x, num_bits):
        """plonky2 `split_le`: ceil(num_bits / 63) BaseSumGate{63} rows, unused limbs forced to zero"""
        if num_bits == 0:
            return []
        nl = min(63, self.config["num_routed_wires"] - 1)
        k = -(-num_bits // nl)
        rows = [self.add_gate(G.BaseSumGate(nl, 2)) for _ in range(k)]
        bits = [Target(r, 1 + i) for r in rows for i in range(nl)]
        for t in bits[num_bits:]:
            self.assert_zero(t)
        sums = [Target(r, 0) for r in rows]
        if k > 1:       # generators first, in dependency order (the program compiler then keeps the creation order)
            self.add_generator([x], lambda v, sums=sums, nl=nl: [(s, (v[0] >> (nl * i)) & ((1 << nl) - 1)) for i, s in enumerate(sums)],
                               OP_SPLIT, (1 << nl, k), outs=sums)
        for r, s in zip(rows, sums):
            limbs = [Target(r, 1 + i) for i in range(nl)]
            self.add_generator([s], lambda v, limbs=limbs: [(t, (v[0] >> i) & 1) for i, t in enumerate(limbs)], OP_SPLIT, (2, nl), outs=limbs)
        if k == 1:
            self.connect(sums[0], x)
        else:
            acc = sums[-1]
            for s in reversed(sums[:-1]):
                acc = self.arithmetic(1 << nl, acc, self.one(), 1, s)
            self.connect(acc, x)
        return bits[:num_bits]

    def range_check(self, x, num_bits):
        self.split_le_63(x, num_bits)

    def le_sum_small(self, bits):
        """sum of bits * 2^i by Horner with arithmetic operations (plonky2 `le_sum` for short inputs)"""
        bits = list(bits)
        if not bits:
            return self.zero()
        acc = bits[-1]
        two = self.constant(2)
        for b_ in reversed(bits[:-1]):
            acc = self.mul_add(two, acc, b_)
        return acc

    def random_access_ext(self, index, items):
        return (self.random_access(index, [e[0] for e in items]), self.random_access(index, [e[1] for e in items]))

    # ---- ExponentiationGate
    def exp_from_bits(self, base, bits):
        """base^(sum bits_i 2^i)"""
        gate = G.ExponentiationGate.new_from_config(self.config)
        n = gate.num_power_bits
        bits = list(bits)
        assert len(bits) <= n
        bits = bits + [self.zero()] * (n - len(bits))
        row = self.add_gate(gate)
        self.connect(base, Target(row, 0))
        wb = [Target(row, 1 + i) for i in range(n)]
        for a, t in zip(bits, wb):
            self.connect(a, t)
        out = Target(row, 1 + n)

        def gen(v, row=row, n=n, out=out):
            base_v, res, cur = v[0], [], 1
            for i in range(n):
                prev = 1 if i == 0 else cur * cur % P
                cur = prev * (base_v if v[1 + n - 1 - i] else 1) % P
                res.append((Target(row, 2 + n + i), cur))
            return res + [(out, cur)]
        self.add_generator([Target(row, 0)] + wb, gen, OP_EXPONENTIATION, outs=[Target(row, 2 + n + i) for i in range(n)] + [out])
        return out

    def exp_from_bits_const_base(self, base, bits):
        """base^(sum bits_i 2^i) for a constant base with ArithmeticGate operations: product *= 1 + bit (base^(2^i) - 1)
        (the reference's recursion circuits contain no ExponentiationGate: near_bft_finality/proofs/*/common_data.json)"""
        product = self.one()
        for i, bit in enumerate(bits):
            product = self.arithmetic(pow(base, 1 << i, P) - 1, product, bit, 1, product)
        return product

    # ---- CosetInterpolationGate
    def interpolate_coset(self, gate, shift, values, point):
        np_, d, ni = 1 << gate.subgroup_bits, gate.degree, gate.num_intermediates
        assert len(values) == np_
        row = self.add_gate(gate)
        start_pt = 1 + 2 * np_
        start_val, start_inter = start_pt + 2, start_pt + 4
        self.connect(shift, Target(row, 0))
        ins = [Target(row, 0)]
        for i, v in enumerate(values):
            t = (Target(row, 1 + 2 * i), Target(row, 2 + 2 * i))
            self.connect_ext(v, t)
            ins += list(t)
        tp = (Target(row, start_pt), Target(row, start_pt + 1))
        self.connect_ext(point, tp)
        ins += list(tp)
        dom = [pow(root_of_unity(gate.subgroup_bits), i, P) for i in range(np_)]
        wts = gate.weights

        def gen(v, row=row):
            shift_v = v[0]
            vals = [(v[1 + 2 * i], v[2 + 2 * i]) for i in range(np_)]
            pt = (v[1 + 2 * np_], v[2 + 2 * np_])
            shifted = e_scalar(pow(shift_v, P - 2, P), pt)
            res = [(Target(row, start_inter + 4 * ni), shifted[0]), (Target(row, start_inter + 4 * ni + 1), shifted[1])]

            def partial(s, e, ev, prod):
                for i in range(s, e):
                    term = e_sub(shifted, (dom[i], 0))
                    ev = e_add(e_mul(ev, term), e_mul(e_scalar(wts[i], vals[i]), prod))
                    prod = e_mul(prod, term)
                return ev, prod
            ev, prod = partial(0, d, (0, 0), (1, 0))
            for i in range(ni):
                res += [(Target(row, start_inter + 2 * i), ev[0]), (Target(row, start_inter + 2 * i + 1), ev[1]),
                        (Target(row, start_inter + 2 * (ni + i)), prod[0]), (Target(row, start_inter + 2 * (ni + i) + 1), prod[1])]
                s = 1 + (d - 1) * (i + 1)
                ev, prod = partial(s, min(s + d - 1, np_), ev, prod)
            return res + [(Target(row, start_val), ev[0]), (Target(row, start_val + 1), ev[1])]
        outs = [Target(row, start_inter + 4 * ni), Target(row, start_inter + 4 * ni + 1)]
        for i in range(ni):
            outs += [Target(row, start_inter + 2 * i), Target(row, start_inter + 2 * i + 1),
                     Target(row, start_inter + 2 * (ni + i)), Target(row, start_inter + 2 * (ni + i) + 1)]
        outs += [Target(row, start_val), Target(row, start_val + 1)]
        self.add_generator(ins, gen, OP_COSET_INTERP, (gate.subgroup_bits, gate.degree) + tuple(gate.weights), outs=outs)
        return (Target(row, start_val), Target(row, start_val + 1))

    # ---- PoseidonMdsGate on extension targets
    def mds_ext(self, state):
        from .gate_circuits import poseidon_constants
        circ, diag = poseidon_constants()["circ"], poseidon_constants()["diag"]
        row = self.add_gate(G.PoseidonMdsGate())
        ins = []
        for i, e in enumerate(state):
            t = (Target(row, 2 * i), Target(row, 2 * i + 1))
            self.connect_ext(e, t)
            ins += list(t)

        def gen(v, row=row):
            res = []
            for r in range(12):
                a0 = a1 = 0
                for i in range(12):
                    j = (i + r) % 12
                    a0 += v[2 * j] * circ[i]
                    a1 += v[2 * j + 1] * circ[i]
                a0 += v[2 * r] * diag[r]
                a1 += v[2 * r + 1] * diag[r]
                res += [(Target(row, 24 + 2 * r), a0 % P), (Target(row, 25 + 2 * r), a1 % P)]
            return res
        self.add_generator(ins, gen, OP_POSEIDON_MDS, outs=[Target(row, 24 + k) for k in range(24)])
        return [(Target(row, 24 + 2 * r), Target(row, 25 + 2 * r)) for r in range(12)]

    # ---- hashing
    def hash_or_noop(self, inputs):
        inputs = list(inputs)
        if len(inputs) <= 4:
            return inputs + [self.zero()] * (4 - len(inputs))
        return self.hash_n_to_hash_no_pad(inputs)

    def verify_merkle_proof_to_cap(self, leaf, index_bits, cap_index, cap, siblings):
        """fri.go:97-160: hash the leaf, walk up with (swap = index bit), compare with cap[cap_index]"""
        z = self.zero()
        state = self.hash_or_noop(leaf)
        assert len(index_bits) == len(siblings)
        for bit, sib in zip(index_bits, siblings):
            state = self.permute(list(state) + list(sib) + [z] * 4, swap=bit)[:4]
        for i in range(4):
            self.connect(self.random_access(cap_index, [h[i] for h in cap]), state[i])


# opcodes of the native witness interpreter for the gadgets above (csrc/plonky2_witness.cpp)

OP_EXT_ARITH, OP_EXT_MUL, OP_EXT_INV, OP_EXPONENTIATION, OP_COSET_INTERP, OP_POSEIDON_MDS, OP_REDUCING, OP_REDUCING_EXT = range(18, 26)


class ReducingFactor:
    """plonky2 `ReducingFactorTarget`: sum_i terms[i] * base^i with ReducingGate / ReducingExtensionGate rows"""

    def __init__(self, base):
        self.base, self.count = base, 0

    def _gate_rows(self, b, gate, terms, ext):
        n = gate.num_coeffs
        self.count += len(terms)
        pad = self._zero_term(b, ext)
        rev = list(terms) + [pad] * (-len(terms) % n)
        rev.reverse()
        acc = b.zero_ext()
        step = 2 if ext else 1
        for c0 in range(0, len(rev), n):
            chunk = rev[c0:c0 + n]
            row = b.add_gate(gate)
            alpha = (Target(row, 2), Target(row, 3))
            old = (Target(row, 4), Target(row, 5))
            b.connect_ext(self.base, alpha)
            b.connect_ext(acc, old)
            ins = list(alpha) + list(old)
            for i, t in enumerate(chunk):
                if ext:
                    w = (Target(row, 6 + 2 * i), Target(row, 7 + 2 * i))
                    b.connect_ext(t, w)
                    ins += list(w)
                else:
                    w = Target(row, 6 + i)
                    b.connect(t, w)
                    ins.append(w)
            start_accs = 6 + step * n

            def gen(v, row=row, n=n, ext=ext, start_accs=start_accs):
                alpha_v, acc_v = (v[0], v[1]), (v[2], v[3])
                res = []
                for i in range(n):
                    coeff = (v[4 + 2 * i], v[5 + 2 * i]) if ext else (v[4 + i], 0)
                    acc_v = e_add(e_mul(acc_v, alpha_v), coeff)
                    c = 0 if i == n - 1 else start_accs + 2 * i
                    res += [(Target(row, c), acc_v[0]), (Target(row, c + 1), acc_v[1])]
                return res
            outs = []
            for i in range(n):
                c = 0 if i == n - 1 else start_accs + 2 * i
                outs += [Target(row, c), Target(row, c + 1)]
            b.add_generator(ins, gen, OP_REDUCING_EXT if ext else OP_REDUCING, (n,), outs=outs)
            acc = (Target(row, 0), Target(row, 1))
        return acc

    @staticmethod
    def _zero_term(b, ext):
        return b.zero_ext() if ext else b.zero()

    def reduce_arithmetic(self, terms, b):
        self.count += len(terms)
        acc = b.zero_ext()
        for t in reversed(list(terms)):
            acc = b.mul_add_ext(self.base, acc, t)
        return acc

    def reduce_base(self, terms, b):
        terms = list(terms)
        if len(terms) <= G.ArithmeticGate.new_from_config(b.config).num_ops + 1:
            return self.reduce_arithmetic([b.convert_to_ext(t) for t in terms], b)
        cfg = b.config
        n = min(cfg["num_routed_wires"] - 6, (cfg["num_wires"] - 4) // 3)
        return self._gate_rows(b, G.ReducingGate(n), terms, False)

    def reduce(self, terms, b):
        terms = list(terms)
        if len(terms) <= G.ArithmeticExtensionGate.new_from_config(b.config).num_ops + 1:
            return self.reduce_arithmetic(terms, b)
        cfg = b.config
        n = min((cfg["num_routed_wires"] - 6) // 2, (cfg["num_wires"] - 4) // 4)
        return self._gate_rows(b, G.ReducingExtensionGate(n), terms, True)

    def shift(self, x, b):
        count, self.count = self.count, 0
        if b.ext_as_constant(x) == (0, 0):
            return b.zero_ext()
        return b.mul_ext(b.exp_u64_ext(self.base, count), x)


class RecursiveChallenger:
    """challenger.go:42-166 on targets (plonky2 `RecursiveChallenger`): overwrite-mode duplex sponge, rate 8"""

    def __init__(self, b):
        self.b = b
        self.state = [b.zero()] * 12
        self.inp, self.out = [], []

    def observe(self, t):
        self.out = []
        self.inp.append(t)

    def observe_many(self, ts):
        for t in ts:
            self.observe(t)

    def observe_cap(self, cap):
        for h in cap:
            self.observe_many(h)

    def observe_ext(self, e):
        self.observe_many(e)

    def _absorb(self):
        if not self.inp:
            return
        for i in range(0, len(self.inp), 8):
            chunk = self.inp[i:i + 8]
            self.state = self.b.permute(list(chunk) + self.state[len(chunk):])
        self.out = list(self.state[:8])
        self.inp = []

    def challenge(self):
        self._absorb()
        if not self.out:
            self.state = self.b.permute(self.state)
            self.out = list(self.state[:8])
        return self.out.pop()

    def challenges(self, n):
        return [self.challenge() for _ in range(n)]

    def ext_challenge(self):
        a, b_ = self.challenges(2)
        return (a, b_)


# ---------------------------------------------------------------------------------- proof targets
def _oracle_widths(common):
    cfg = common["config"]
    nch = cfg["num_challenges"]
    return [common["num_constants"] + cfg["num_routed_wires"], cfg["num_wires"], nch * (1 + common["num_partial_products"]),
            nch * common["quotient_degree_factor"]]


def add_virtual_proof_with_pis(b, common):
    """plonky2 `add_virtual_proof_with_pis`: one target per field element of a proof of the circuit described by `common`
    (the schema of the reference's proof_with_public_inputs.json)"""
    cfg = common["config"]
    fp = common["fri_params"]
    fc = fp["config"]
    nch, routed = cfg["num_challenges"], cfg["num_routed_wires"]
    cap_n = 1 << fc["cap_height"]
    n_log = fp["degree_bits"] + fc["rate_bits"]

    def vhash():
        return b.add_virtual_targets(4)

    def vcap():
        return [vhash() for _ in range(cap_n)]

    def vext(n):
        return [b.add_virtual_ext() for _ in range(n)]
    widths = _oracle_widths(common)
    rounds = []
    for _ in range(fc["num_query_rounds"]):
        init = [(b.add_virtual_targets(w), [vhash() for _ in range(n_log - fc["cap_height"])]) for w in widths]
        steps, bits = [], n_log
        for ab in fp["reduction_arity_bits"]:
            bits -= ab
            steps.append((vext(1 << ab), [vhash() for _ in range(bits - fc["cap_height"])]))
        rounds.append({"init": init, "steps": steps})
    return {
        "public_inputs": b.add_virtual_targets(common["num_public_inputs"]),
        "wires_cap": vcap(), "zs_pp_cap": vcap(), "quotient_cap": vcap(),
        "openings": {"constants": vext(common["num_constants"]), "plonk_sigmas": vext(routed), "wires": vext(cfg["num_wires"]),
                     "plonk_zs": vext(nch), "plonk_zs_next": vext(nch), "partial_products": vext(nch * common["num_partial_products"]),
                     "quotient_polys": vext(nch * common["quotient_degree_factor"])},
        "commit_caps": [vcap() for _ in fp["reduction_arity_bits"]],
        "rounds": rounds,
        "final_poly": vext(1 << (fp["degree_bits"] - sum(fp["reduction_arity_bits"]))),
        "pow_witness": b.add_virtual_target(),
    }


def add_virtual_verifier_data(b, common):
    """VerifierCircuitTarget { constants_sigmas_cap, circuit_digest } (recursion.rs:38-43)"""
    cap_n = 1 << common["config"]["fri_config"]["cap_height"]
    return {"constants_sigmas_cap": [b.add_virtual_targets(4) for _ in range(cap_n)], "circuit_digest": b.add_virtual_targets(4)}


def _hash_elems(h):
    return [int(x) for x in (h["elements"] if isinstance(h, dict) else h)]


def set_proof_with_pis_target(pw, pt, proof_json):
    """pw.set_proof_with_pis_target (recursion.rs:45): proof_json in the proof_with_public_inputs.json schema, Poseidon hasher"""
    def put(t, v):
        pw[t] = int(v) % P

    def put_hash(ts, h):
        for t, v in zip(ts, _hash_elems(h)):
            put(t, v)

    def put_cap(ts, cap):
        assert len(ts) == len(cap)
        for t, h in zip(ts, cap):
            put_hash(t, h)

    def put_ext(ts, vs):
        assert len(ts) == len(vs), (len(ts), len(vs))
        for t, v in zip(ts, vs):
            put(t[0], v[0])
            put(t[1], v[1])
    pr = proof_json["proof"]
    assert len(pt["public_inputs"]) == len(proof_json["public_inputs"])
    for t, v in zip(pt["public_inputs"], proof_json["public_inputs"]):
        put(t, v)
    put_cap(pt["wires_cap"], pr["wires_cap"])
    put_cap(pt["zs_pp_cap"], pr["plonk_zs_partial_products_cap"])
    put_cap(pt["quotient_cap"], pr["quotient_polys_cap"])
    for k, ts in pt["openings"].items():
        put_ext(ts, pr["openings"][k])
    op = pr["opening_proof"]
    assert len(pt["commit_caps"]) == len(op["commit_phase_merkle_caps"])
    for ts, cap in zip(pt["commit_caps"], op["commit_phase_merkle_caps"]):
        put_cap(ts, cap)
    assert len(pt["rounds"]) == len(op["query_round_proofs"])
    for rt, q in zip(pt["rounds"], op["query_round_proofs"]):
        for (leaf_t, sib_t), ep in zip(rt["init"], q["initial_trees_proof"]["evals_proofs"]):
            assert len(leaf_t) == len(ep[0]) and len(sib_t) == len(ep[1]["siblings"])
            for t, v in zip(leaf_t, ep[0]):
                put(t, v)
            for t, h in zip(sib_t, ep[1]["siblings"]):
                put_hash(t, h)
        for (ev_t, sib_t), st in zip(rt["steps"], q["steps"]):
            put_ext(ev_t, st["evals"])
            assert len(sib_t) == len(st["merkle_proof"]["siblings"])
            for t, h in zip(sib_t, st["merkle_proof"]["siblings"]):
                put_hash(t, h)
    put_ext(pt["final_poly"], op["final_poly"]["coeffs"])
    put(pt["pow_witness"], op["pow_witness"])


def set_verifier_data_target(pw, vt, verifier_json):
    for ts, h in zip(vt["constants_sigmas_cap"], verifier_json["constants_sigmas_cap"]):
        for t, v in zip(ts, _hash_elems(h)):
            pw[t] = int(v) % P
    for t, v in zip(vt["circuit_digest"], _hash_elems(verifier_json["circuit_digest"])):
        pw[t] = int(v) % P


# ---------------------------------------------------------------------------------- the verifier
def _get_challenges(b, pt, vt, common, pih):
    """verifier.go:58-82 (GetChallenges) on targets"""
    cfg = common["config"]
    nch = cfg["num_challenges"]
    ch = RecursiveChallenger(b)
    ch.observe_many(vt["circuit_digest"])
    ch.observe_many(pih)
    ch.observe_cap(pt["wires_cap"])
    betas, gammas = ch.challenges(nch), ch.challenges(nch)
    ch.observe_cap(pt["zs_pp_cap"])
    alphas = ch.challenges(nch)
    ch.observe_cap(pt["quotient_cap"])
    zeta = ch.ext_challenge()
    o = pt["openings"]
    batch0 = o["constants"] + o["plonk_sigmas"] + o["wires"] + o["plonk_zs"] + o["partial_products"] + o["quotient_polys"]
    batch1 = o["plonk_zs_next"]
    for e in batch0 + batch1:
        ch.observe_ext(e)
    fri_alpha = ch.ext_challenge()
    fri_betas = []
    for cap in pt["commit_caps"]:
        ch.observe_cap(cap)
        fri_betas.append(ch.ext_challenge())
    for c in pt["final_poly"]:
        ch.observe_ext(c)
    ch.observe(pt["pow_witness"])
    pow_response = ch.challenge()
    indices = ch.challenges(cfg["fri_config"]["num_query_rounds"])
    return {"betas": betas, "gammas": gammas, "alphas": alphas, "zeta": zeta, "fri_alpha": fri_alpha, "fri_betas": fri_betas,
            "pow_response": pow_response, "query_indices": indices, "batches": [batch0, batch1]}


def _eval_vanishing_poly(b, common, pt, ch, pih, zeta_pow_n):
    """plonk.go:121-207 on targets: [L_0 (Z - 1) | partial-product checks | filtered gate constraints] reduced by each alpha"""
    cfg = common["config"]
    routed, nch = cfg["num_routed_wires"], cfg["num_challenges"]
    npp, qdf = common["num_partial_products"], common["quotient_degree_factor"]
    o = pt["openings"]
    zeta = ch["zeta"]
    n = 1 << common["fri_params"]["degree_bits"]
    one = b.one_ext()
    # gate constraints
    K = (b.k_adapter or CircuitK)(b)       # k_adapter: a diagnostic hook of tools/wrap_instance.py (never set by a prover)
    sel = common["selectors_info"]
    groups = [(g["start"], g["end"]) for g in sel["groups"]]
    nsel = len(groups)
    consts = [K.lift(e) for e in o["constants"]]
    wires = [K.lift(e) for e in o["wires"]]
    pih_k = [K.lift(b.convert_to_ext(t)) for t in pih]
    terms_gate = [K.zero] * common["num_gate_constraints"]
    for row, gid in enumerate(common["gates"]):
        gate = G.gate_from_id(gid)
        si = sel["selector_indices"][row]
        s = consts[si]
        f = K.one
        for i in range(groups[si][0], groups[si][1]):
            if i != row:
                f = K.mul(f, K.sub(K.const(i), s))
        if nsel > 1:
            f = K.mul(f, K.sub(K.const(UNUSED_SELECTOR), s))
        f = K.lift(K.mat(f))
        cs = eval_gate_circuit(K, gate, consts[nsel:], wires, pih_k)
        assert len(cs) == gate.num_constraints
        for i, c in enumerate(cs):
            terms_gate[i] = K.add(K.mul(f, c), terms_gate[i])
    terms_gate = [K.mat(t) for t in terms_gate]
    # permutation argument
    zh = b.sub_ext(zeta_pow_n, one)
    l0 = b.div_ext(zh, b.arithmetic_ext(n, P - n, zeta, one, one))
    s_ids = [b.scalar_mul_ext(b.constant(k), zeta) for k in common["k_is"][:routed]]
    z1, ppt = [], []
    for i in range(nch):
        z_x, z_gx = o["plonk_zs"][i], o["plonk_zs_next"][i]
        z1.append(b.mul_sub_ext(l0, z_x, l0))
        beta, gamma = b.convert_to_ext(ch["betas"][i]), b.convert_to_ext(ch["gammas"][i])
        nums, dens = [], []
        for j in range(routed):
            wg = b.add_ext(o["wires"][j], gamma)
            nums.append(b.mul_add_ext(beta, s_ids[j], wg))
            dens.append(b.mul_add_ext(beta, o["plonk_sigmas"][j], wg))
        accs = [z_x] + list(o["partial_products"][i * npp:(i + 1) * npp]) + [z_gx]
        for k in range(npp + 1):
            lo, hi = k * qdf, min((k + 1) * qdf, routed)
            np_ = b.mul_many_ext(nums[lo:hi])
            dp = b.mul_many_ext(dens[lo:hi])
            ppt.append(b.mul_sub_ext(accs[k], np_, b.mul_ext(accs[k + 1], dp)))
    terms = z1 + ppt + terms_gate
    return [ReducingFactor(b.convert_to_ext(a)).reduce(terms, b) for a in ch["alphas"]], zh


def _fri_combine_initial(b, common, init, alpha, subgroup_x, reduced_openings, points):
    """fri.go:208-251"""
    nch = common["config"]["num_challenges"]
    widths = _oracle_widths(common)
    all_polys = [(k, i) for k in range(4) for i in range(widths[k])]
    zs_polys = [(2, i) for i in range(nch)]
    rf = ReducingFactor(alpha)
    s = b.zero_ext()
    x_ext = b.convert_to_ext(subgroup_x)
    for polys, red_open, point in zip([all_polys, zs_polys], reduced_openings, points):
        evals = [init[k][0][i] for (k, i) in polys]
        red = rf.reduce_base(evals, b)
        num = b.sub_ext(red, red_open)
        den = b.sub_ext(x_ext, point)
        s = rf.shift(s, b)
        s = b.div_add_ext(num, den, s)
    return s


def _compute_evaluation(b, x, within_bits, arity_bits, evals, beta, max_degree):
    """fri.go:314-384: interpolate the coset evaluations (bit-reversed order) and evaluate at beta"""
    arity = 1 << arity_bits
    g = root_of_unity(arity_bits)
    g_inv = pow(g, arity - 1, P)
    ys = [None] * arity
    for j in range(arity):
        ys[int(format(j, "0%db" % arity_bits)[::-1], 2)] = evals[j]
    start = b.exp_from_bits_const_base(g_inv, list(reversed(within_bits)))
    coset_start = b.mul(start, x)
    return b.interpolate_coset(coset_interpolation_gate(arity_bits, max_degree), coset_start, ys, beta)


def verify_proof(b, pt, vt, common):
    """builder.verify_proof(pt, vt, inner_common) (recursion.rs:55-59): constrains `pt` to be a valid proof of the circuit
    whose verifier data is `vt`.  Poseidon (Goldilocks) hasher for the inner proof."""
    cfg = common["config"]
    fp = common["fri_params"]
    fc = fp["config"]
    assert not fp["hiding"] and common.get("num_lookup_polys", 0) == 0
    degree_bits, rate_bits, cap_h = fp["degree_bits"], fc["rate_bits"], fc["cap_height"]
    n_log = degree_bits + rate_bits
    nch, qdf = cfg["num_challenges"], common["quotient_degree_factor"]
    pih = b.hash_n_to_hash_no_pad(pt["public_inputs"])
    ch = _get_challenges(b, pt, vt, common, pih)
    zeta = ch["zeta"]
    # vanishing identity (plonk.go:209-250)
    zeta_pow_n = b.exp_power_of_2_ext(zeta, degree_bits)
    vanishing, zh = _eval_vanishing_poly(b, common, pt, ch, pih, zeta_pow_n)
    for i in range(nch):
        t = ReducingFactor(zeta_pow_n).reduce(pt["openings"]["quotient_polys"][i * qdf:(i + 1) * qdf], b)
        b.connect_ext(vanishing[i], b.mul_ext(zh, t))
    # FRI (fri.go:75-80: proof of work; :386-497 query rounds)
    b.range_check(ch["pow_response"], 64 - fc["proof_of_work_bits"])
    g = root_of_unity(degree_bits)
    points = [zeta, b.scalar_mul_ext(b.constant(g), zeta)]
    alpha = ch["fri_alpha"]
    reduced_openings = [ReducingFactor(alpha).reduce(batch, b) for batch in ch["batches"]]
    caps = [vt["constants_sigmas_cap"], pt["wires_cap"], pt["zs_pp_cap"], pt["quotient_cap"]]
    for rnd, rt in enumerate(pt["rounds"]):
        x_bits = b.split_le_63(ch["query_indices"][rnd], 64)[:n_log]
        cap_index = b.le_sum_small(x_bits[n_log - cap_h:])
        for k in range(4):
            leaf, sib = rt["init"][k]
            b.verify_merkle_proof_to_cap(leaf, x_bits[:n_log - cap_h], cap_index, caps[k], sib)
        phi = b.exp_from_bits_const_base(root_of_unity(n_log), list(reversed(x_bits)))
        x = b.mul(b.constant(GENERATOR), phi)
        old = _fri_combine_initial(b, common, rt["init"], alpha, x, reduced_openings, points)
        bits = x_bits
        for i, ab in enumerate(fp["reduction_arity_bits"]):
            evals, sib = rt["steps"][i]
            within, coset = bits[:ab], bits[ab:]
            b.connect_ext(b.random_access_ext(b.le_sum_small(within), evals), old)
            old = _compute_evaluation(b, x, within, ab, evals, ch["fri_betas"][i], cfg["max_quotient_degree_factor"])
            flat = [c for e in evals for c in e]
            b.verify_merkle_proof_to_cap(flat, coset[:len(coset) - cap_h], cap_index, pt["commit_caps"][i], sib)
            x = b.exp_power_of_2(x, ab)
            bits = coset
        ev = ReducingFactor(b.convert_to_ext(x)).reduce(pt["final_poly"], b)
        b.connect_ext(ev, old)
    return ch


def recursive_circuit(inners, num_public_inputs=0, builder=None):
    """The circuit of `recursive_proof` (recursion.rs:16-97) for one or two inner circuits given by their common data.
    Returns (CircuitData, targets): targets["proofs"][i], targets["verifier_data"][i], targets["public_inputs"].
    `builder`: a RecursiveCircuitBuilder to build into (the layout experiments of tools/wrap_instance.py pass modified ones)."""
    b = builder or RecursiveCircuitBuilder(standard_recursion_config())
    pts, vts = [], []
    for common in inners:
        pt = add_virtual_proof_with_pis(b, common)
        vt = add_virtual_verifier_data(b, common)
        verify_proof(b, pt, vt, common)
        pts.append(pt)
        vts.append(vt)
    pis = b.add_virtual_targets(num_public_inputs)
    for t in pis:
        b.register_public_input(t)
    return b.build(), {"proofs": pts, "verifier_data": vts, "public_inputs": pis}


def target_leaves(tree):
    """every Target of a nested dict / list / tuple, in traversal order"""
    from .builder import Target
    if isinstance(tree, Target):
        yield tree
    elif isinstance(tree, dict):
        for v in tree.values():
            yield from target_leaves(v)
    elif isinstance(tree, (list, tuple)):
        for v in tree:
            yield from target_leaves(v)


def recursive_witness(targets, inner_proofs, public_inputs=()):
    """the PartialWitness of recursion.rs:44-92: inner_proofs = [(proof_json, verifier_only_json), ...]"""
    pw = {}
    for pt, vt, (proof_json, verifier_json) in zip(targets["proofs"], targets["verifier_data"], inner_proofs):
        set_proof_with_pis_target(pw, pt, proof_json)
        set_verifier_data_target(pw, vt, verifier_json)
    for t, v in zip(targets["public_inputs"], public_inputs):
        pw[t] = int(v) % P
    return pw


class RecursiveCircuit:
    """what `recursive_proof` returns first in the reference (CircuitData): `.common`, `.verifier_only`, and the prover"""

    def __init__(self, data, targets, prover, inner_commons, inner_hasher):
        self.data, self.targets, self.prover = data, targets, prover
        self.common = data.common_data()
        self.verifier_only = prover.verifier_data() if prover is not None else None
        self.inner_commons, self.inner_hasher = inner_commons, inner_hasher
        self._pos = None
        self._maps = None

    def compile(self, example_pw, example_raws):
        """first proof of this shape: compile the generators into the native program and index the program's inputs, so that
        later inner proofs go from their bytes to the input vector by one gather.
        example_raws: the inner proofs of the example as bytes (layout templates)"""
        import numpy as np
        from . import serialization as S
        self.data.witness_program(example_pw)
        self._pos = {t: i for i, t in enumerate(self.data._program["input_targets"])}
        self._maps = []
        for pt, common, raw in zip(self.targets["proofs"], self.inner_commons, example_raws):
            tmp = {}
            set_proof_with_pis_target(tmp, pt, S.proof_offsets(raw, common, self.inner_hasher))
            pos = np.array([self._pos[t] for t in tmp], dtype=np.int64)
            off = np.array(list(tmp.values()), dtype=np.int64)
            self._maps.append((pos, off[:, None] + np.arange(8, dtype=np.int64)[None, :], len(raw)))

    def wire_buffer(self):
        """the host wire matrix of this circuit, allocated once (pinned when a GPU is present: the prover uploads it with one
        DMA); the witness interpreter rewrites the same cells on every run"""
        if getattr(self, "_wbuf", None) is None:
            import numpy as np
            shape = (1, self.data.config["num_wires"], self.data.n)
            try:
                import torch
                self._wpin = torch.zeros(shape, dtype=torch.int64)
                if torch.cuda.is_available():
                    self._wpin = self._wpin.pin_memory()
                self._wbuf = self._wpin.numpy().view(np.uint64)
            except ImportError:
                self._wbuf = np.zeros(shape, dtype=np.uint64)
        return self._wbuf

    def device_witness_state(self, ctx):
        """(DeviceWitness, zero-initialised wire matrix in HBM) of this circuit on `ctx`, created on first use"""
        if getattr(self, "_dw", None) is None:
            import torch
            self._dw = self.data.device_witness(ctx)
            self._dbuf = torch.zeros((1, self.data.config["num_wires"], self.data.n), dtype=torch.int64, device="cuda:%d" % ctx.device_id)
            torch.cuda.synchronize(ctx.device_id)      # the fill ran on torch's stream, the kernels run on the context's
        return self._dw, self._dbuf

    def close_device_witness(self):
        if getattr(self, "_dw", None) is not None:
            self._dw.close()
            self._dw = self._dbuf = None

    def input_vector(self, inner_proofs, public_inputs):
        """inner_proofs: [(verifier_only_json, proof bytes or proof json)]"""
        import numpy as np
        vals = np.zeros(len(self._pos), dtype=np.uint64)
        tmp = {}
        for k, (vd, proof) in enumerate(inner_proofs):
            if isinstance(proof, (bytes, bytearray, memoryview)):
                pos, gather, size = self._maps[k]
                if len(proof) != size:
                    raise ValueError("inner proof %d: %d bytes, expected %d" % (k, len(proof), size))
                vals[pos] = np.ascontiguousarray(np.frombuffer(proof, dtype=np.uint8)[gather]).view("<u8")[:, 0]
            else:
                set_proof_with_pis_target(tmp, self.targets["proofs"][k], proof)
            set_verifier_data_target(tmp, self.targets["verifier_data"][k], vd)
        for t, v in zip(self.targets["public_inputs"], public_inputs):
            tmp[t] = int(v) % P
        for t, v in tmp.items():
            vals[self._pos[t]] = v
        return vals


class RecursionProver:
    """`recursive_proof` (near_bft_finality/src/prove_crypto/recursion.rs:16-97) on one GPU context.

    The reference rebuilds the verifier circuit on every call (recursion.rs:36,94); the circuit depends only on the inner
    circuits' common data and the number of public inputs, so it is built and uploaded once per distinct shape and reused:
    a fold (prove_block_data/signatures.rs:97-105) settles on two shapes plus the closing one."""

    def __init__(self, ctx, hasher=0, threads=None, inner_hasher=0, device_witness=False, witness_threads=4):
        """device_witness=True: the generators of the recursion circuit run on the GPU (csrc/plonky2_witness_dev.hip: ~12 k coarse
        instructions in ~150 dependence levels, one single-workgroup launch) and the proof is made from the matrix in HBM.
        Measured inside a block proof (profiles/r02_bench_block_v4_recursion_witness_on_gpu.json) this is SLOWER than the host
        interpreter -- 45 ms against 37 ms per fold step: its launch queues behind the signature proofs' kernels -- so the default
        stays the host interpreter, run level-parallel on `witness_threads` host threads; the batch form (64 signatures) is where
        the device interpreter pays."""
        self.ctx, self.hasher, self.threads, self.inner_hasher = ctx, hasher, threads, inner_hasher
        self.device_witness = device_witness
        # host threads of ONE recursion witness (csrc/plonky2_witness.cpp, the levelled form): the fold is a serial chain of
        # witness -> proof -> witness, so the latency of a single witness is on the critical path of a block
        self.witness_threads = max(1, int(witness_threads))
        assert inner_hasher == 0, "the in-circuit verifier handles Poseidon-Goldilocks inner proofs"
        self._cache = {}

    def circuit_for(self, commons, num_public_inputs=0):
        import json
        key = json.dumps([commons, num_public_inputs], sort_keys=True)
        rc = self._cache.get(key)
        if rc is None:
            # through the circuit cache (round 6): a verifier circuit is ~1-2.5 s of host Python to build and compile, a block's DAG
            # has ~20 distinct shapes, and a container entry loads in 0.1 s.  The program's inputs are every target of the tree
            # (what `recursive_witness` assigns), in tree order.
            import hashlib
            from .circuit_cache import load_or_build

            def build():
                data, targets = recursive_circuit(commons, num_public_inputs)
                data.witness_program(list(target_leaves(targets)))
                return data, targets
            data, targets, _ = load_or_build("recursion", hashlib.sha256(key.encode()).hexdigest(), build)
            rc = self._cache[key] = RecursiveCircuit(data, targets, data.prover(self.ctx, self.hasher), list(commons), self.inner_hasher)
        return rc

    def recursive_proof(self, first, second=None, public_inputs=None, raw=False):
        """first / second: (common_data, verifier_only_data, proof) of the inner proofs -- the proof either in the JSON schema
        of the reference's proof_with_public_inputs.json or as `ProofWithPublicInputs::to_bytes` bytes (signatures.rs:225-230);
        Poseidon-Goldilocks config.  Returns (RecursiveCircuit, proof) -- recursion.rs:95-96 -- with the proof as JSON, or as
        bytes if `raw`.  Raises AssertionError if an inner proof does not verify (no witness exists)."""
        from . import serialization as S
        inners = [first] + ([second] if second is not None else [])
        pis = [int(x) for x in (public_inputs or [])]
        rc = self.circuit_for([c for c, _, _ in inners], len(pis))
        if rc._pos is None:
            is_raw = [isinstance(p, (bytes, bytearray, memoryview)) for _, _, p in inners]
            as_json = [(S.proof_from_bytes(bytes(p), c, self.inner_hasher) if r else p, v) for (c, v, p), r in zip(inners, is_raw)]
            raws = [bytes(p) if r else S.proof_to_bytes(p, c, self.inner_hasher) for (c, v, p), r in zip(inners, is_raw)]
            rc.compile(recursive_witness(rc.targets, as_json, pis), raws)
        import time
        t0 = time.perf_counter()
        vals = rc.input_vector([(v, p) for _, v, p in inners], pis)
        t1 = time.perf_counter()
        if self.device_witness:
            dw, dbuf = rc.device_witness_state(self.ctx)
            st = self.ctx.stream_ptr()
            wpis = dw.run(dbuf.data_ptr(), input_values=vals[None, :], stream=st)
            t2 = time.perf_counter()
            out = rc.prover.prove_dev(dbuf.data_ptr(), [int(x) for x in wpis[0]], stream=st)
        else:
            wires, wpis = rc.data.generate_witness_native(None, out=rc.wire_buffer(), threads=self.witness_threads,
                                                          input_values=vals[None, :])
            t2 = time.perf_counter()
            out = rc.prover.prove_host_ptr(wires.ctypes.data, [int(x) for x in wpis[0]])
        t3 = time.perf_counter()
        self.last_host_ms = {"inputs": (t1 - t0) * 1e3, "witness": (t2 - t1) * 1e3, "prove": (t3 - t2) * 1e3}
        return rc, (out if raw else S.proof_from_bytes(out, rc.common, self.hasher))

    def close(self):
        for rc in self._cache.values():
            rc.close_device_witness()
            rc.prover.close()
        self._cache = {}
