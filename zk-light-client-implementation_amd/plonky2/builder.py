"""Host-side circuit builder: the subset of plonky2's `CircuitBuilder` the signature-aggregation circuits use.

Mirrors the API the reference calls on the un-vendored plonky2 fork (wormhole-foundation/plonky2-near@2244a9d):
  CircuitBuilder::new(config)           near_bft_finality/src/prove_crypto/ed25519.rs:29-33, recursion.rs:36
  add_virtual_target / register_public_input / connect / constant / add / mul / mul_add / arithmetic
  hash_n_to_hash_no_pad                 (PoseidonGate rows)
  split_le (BaseSumGate), add_u32/mul_u32 via U32ArithmeticGate (crypto/plonky2_u32/src/gadgets/arithmetic_u32.rs)
  build() -> CircuitData                ed25519.rs:33,75, recursion.rs:94
`CircuitData.prove(inputs)` (ed25519.rs:60,100, recursion.rs:95) runs the witness generators on the host and
hands the wire matrix to the GPU prover (prover.py -> zklc_plonky2_prove).

What `build()` produces is the prover's view of the circuit (plonky2 `ProverOnlyCircuitData` + `CommonCircuitData`):
selector + constant polynomials, sigma polynomials from the copy constraints, k_is, the gate list sorted by
(degree, id) and the selector groups -- the same construction as the fork's `selector_polynomials` /
`WirePartition::get_sigma_polys`, so `common_data()` has the schema of the reference's common_data.json.
"""
import numpy as np

from . import gates as G

P = 2**64 - 2**32 + 1
# instruction set of the native witness interpreter (csrc/plonky2_witness.cpp)
OP_CONST, OP_ARITH, OP_SPLIT, OP_LE_SUM, OP_U32_MULADD, OP_ADD_MANY, OP_SUB_U32, OP_RANGE_CHECK, OP_COMPARISON, OP_IS_EQUAL, \
    OP_RANDOM_ACCESS, OP_NN_ADD, OP_NN_SUB, OP_NN_MUL, OP_NN_INV, OP_DIV_REM, OP_DECOMPRESS, OP_POSEIDON = range(18)
OP_INTERLEAVE, OP_UNINTERLEAVE = 26, 27      # 18..25: the recursion gadgets (recursion.py)
UNUSED_SELECTOR = (1 << 32) - 1
GENERATOR = 7
POWER_OF_TWO_GENERATOR = 1753635133440165772


def standard_recursion_config():
    """CircuitConfig::standard_recursion_config() as serialised in the reference's common_data.json "config"."""
    fri = {"rate_bits": 3, "cap_height": 4, "proof_of_work_bits": 16, "reduction_strategy": {"ConstantArityBits": [4, 5]},
           "num_query_rounds": 28}
    return {"num_wires": 135, "num_routed_wires": 80, "num_constants": 2, "use_base_arithmetic_gate": True, "security_bits": 100,
            "num_challenges": 2, "zero_knowledge": False, "max_quotient_degree_factor": 8, "fri_config": fri}


def wide_ecc_config():
    """CircuitConfig::wide_ecc_config() (near_bft_finality/src/prove_crypto/ed25519.rs:29): 234 wires."""
    c = standard_recursion_config()
    c["num_wires"] = 234
    return c


def root_of_unity(log_n):
    return pow(POWER_OF_TWO_GENERATOR, 1 << (32 - log_n), P)


def gl_mul_np(a, b):
    """elementwise a * b mod p on uint64 arrays of canonical values (host-side preprocessing only: sigma values)"""
    a, b = np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64)
    m32 = np.uint64(0xFFFFFFFF)
    s32 = np.uint64(32)
    a0, a1, b0, b1 = a & m32, a >> s32, b & m32, b >> s32
    p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
    mid = (p00 >> s32) + (p01 & m32) + (p10 & m32)                       # < 3 * 2^32
    lo = (p00 & m32) | ((mid & m32) << s32)
    hi = p11 + (p01 >> s32) + (p10 >> s32) + (mid >> s32)                # the product is hi * 2^64 + lo
    # 2^64 = 2^32 - 1, 2^96 = -1 (mod p):  lo - hi_hi + hi_lo * (2^32 - 1)
    hi_hi, hi_lo = hi >> s32, hi & m32
    t0 = lo - hi_hi
    t0 = np.where(lo < hi_hi, t0 - m32, t0)                              # borrow: -2^64 = -(2^32 - 1)
    t1 = hi_lo * m32
    r = t0 + t1
    r = np.where(r < t1, r + m32, r)                                     # carry: +2^64 = +(2^32 - 1)
    return np.where(r >= np.uint64(P), r - np.uint64(P), r)


VIRTUAL_BASE = 1 << 40        # integer keys: wire (row, col) -> row * 256 + col; virtual target idx -> VIRTUAL_BASE + idx


def key_is_wire(k):
    return k < VIRTUAL_BASE


class Target:
    __slots__ = ("row", "col", "idx", "k")

    def __init__(self, row=None, col=None, idx=None):
        self.row, self.col, self.idx = row, col, idx
        self.k = (row << 8) | col if idx is None else VIRTUAL_BASE + idx

    def key(self):
        return self.k

    def __repr__(self):
        return "Wire(%d,%d)" % (self.row, self.col) if self.idx is None else "Virtual(%d)" % self.idx


class TargetRange:
    """the wires (row, col) .. (row, col + n - 1) without one Python object per wire: the limb / bit cells a gate's generator
    fills (tens per operation, millions per circuit) are only ever named in bulk"""
    __slots__ = ("row", "col", "n")

    def __init__(self, row, col, n):
        self.row, self.col, self.n = row, col, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [Target(self.row, self.col + j) for j in range(*i.indices(self.n))]
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        return Target(self.row, self.col + i)

    def __iter__(self):
        return (Target(self.row, self.col + j) for j in range(self.n))

    def keys(self):
        k0 = (self.row << 8) | self.col
        return range(k0, k0 + self.n)


class CircuitBuilder:
    def __init__(self, config=None):
        self.config = config or standard_recursion_config()
        assert self.config["num_wires"] <= 256, "target keys pack the column into 8 bits"
        self.rows = []            # (gate, constants)
        self.n_virtual = 0
        self._conn_a, self._conn_b = [], []      # copy constraints as recorded: pairs of target keys (append only)
        self._classes = None      # (number of pairs resolved, {key: root} of the non-root connected keys, sorted keys, their roots)
        self.generators = []      # (input targets, fn(values) -> [(target, value)])
        self.public_inputs = []
        self._const_targets = {}
        self._arith_slot = {}     # (c0, c1) -> (row, next op)
        self._u32_slot = None
        self._ra_slot = {}
        self._addmany_slot = {}
        self._target_const = {}
        self._sub_slot = None
        self._il_slot, self._ul_slot = None, {}

    # ---- targets / copy constraints
    def add_virtual_target(self):
        t = Target(idx=self.n_virtual)
        self.n_virtual += 1
        return t

    def add_virtual_targets(self, n):
        return [self.add_virtual_target() for _ in range(n)]

    def add_virtual_public_input(self):
        t = self.add_virtual_target()
        self.register_public_input(t)
        return t

    def register_public_input(self, t):
        self.public_inputs.append(t)

    # Copy constraints (plonky2 `connect` -> the disjoint-set forest of `wire_partition`).  Round 5: `connect` only RECORDS the pair
    # (the Ed25519 circuit makes 4.4 M of them; a Python union-find per call was a quarter of its construction time); the classes
    # are resolved once, when something first asks for them, as the connected components of the recorded graph
    # (native union-find, zklc_host_copy_classes).  The root of a class is its smallest key -- nothing depends on which member it is.
    def connect(self, a, b):
        self._conn_a.append(a.k)
        self._conn_b.append(b.k)

    def _resolve_classes(self):
        n_pairs = len(self._conn_a)
        if self._classes is not None and self._classes[0] == n_pairs:
            return self._classes
        if n_pairs == 0:
            self._classes = (0, {}, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
            return self._classes
        a = np.array(self._conn_a, dtype=np.int64)
        b = np.array(self._conn_b, dtype=np.int64)
        keys, inv = np.unique(np.concatenate([a, b]), return_inverse=True)
        ia, ib = np.ascontiguousarray(inv[:n_pairs], dtype=np.int64), np.ascontiguousarray(inv[n_pairs:], dtype=np.int64)
        # native union-find (csrc/plonky2_host.cpp): root index = the smallest index of the class; keys are sorted, so that is the
        # class's smallest key
        from .. import _lib
        root_idx = np.empty(len(keys), dtype=np.int64)
        rc = _lib.load().zklc_host_copy_classes(ia.ctypes.data, ib.ctypes.data, n_pairs, len(keys), root_idx.ctypes.data)
        assert rc == 0, "copy classes: index out of range"
        roots = keys[root_idx]
        nonroot = roots != keys
        parent = dict(zip(keys[nonroot].tolist(), roots[nonroot].tolist()))
        self._classes = (n_pairs, parent, keys, roots)
        return self._classes

    @property
    def parent(self):
        """{key: class root} for every connected key that is not itself the root of its class (the shape the union-find had)"""
        return self._resolve_classes()[1]

    def _find(self, k):
        return self._resolve_classes()[1].get(k, k)

    def class_arrays(self):
        """(keys, roots) of the non-root connected keys as sorted int64 arrays (what CircuitData and the program compiler need)"""
        _, _, keys, roots = self._resolve_classes()
        nonroot = roots != keys
        return keys[nonroot], roots[nonroot]

    def add_gate(self, gate, constants=()):
        assert gate.num_wires <= self.config["num_wires"], "gate %s needs %d wires" % (gate.id(), gate.num_wires)
        self.rows.append((gate, [c % P for c in constants]))
        return len(self.rows) - 1

    def add_generator(self, inputs, fn, op=None, params=(), outs=None):
        """fn(values of inputs) -> [(target, value)]; (op, params) name the same computation for the native interpreter
        (csrc/plonky2_witness.cpp), which must emit its outputs in the order fn returns them.  `outs` = the targets fn sets, in
        that order (Targets and TargetRanges): when every generator declares them, the interpreter program is compiled without
        running the Python generators (CircuitData.witness_program)."""
        self.generators.append((list(inputs), fn, op, tuple(int(x) for x in params), None if outs is None else list(outs)))

    # ---- constants
    def constant(self, c):
        """plonky2 `CircuitBuilder::constant`: a virtual target per distinct value (`constants_to_targets`).  No row is spent
        here: `build()` hands the values, sorted, to the circuit's constant generators -- first the extra constant wires that gates
        with spare constant columns expose (`Gate::extra_constant_wires`: RandomAccessGate's `num_extra_constants`, the reason the
        reference's gate id reads `RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2 }`), then ConstantGate rows
        appended for the overflow."""
        c %= P
        t = self._const_targets.get(c)
        if t is None:
            t = self._const_targets[c] = self.add_virtual_target()
            self._target_const[t.key()] = c
        return t

    def _place_constants(self):
        """the tail of plonky2's `build()`: 'make sure we have enough constant generators, if not add a ConstantGate', then zip the
        constants (sorted by value: deterministic) with the generators (in row order)"""
        nc = self.config["num_constants"]
        slots = []                      # (row, constant index, wire)
        for r, (g, _) in enumerate(self.rows):
            slots += [(r, i, w) for i, w in g.extra_constant_wires()]
        while len(self._const_targets) > len(slots):
            g = G.ConstantGate(nc)
            r = self.add_gate(g, [0] * nc)
            slots += [(r, i, w) for i, w in g.extra_constant_wires()]
        for (c, t), (r, i, w) in zip(sorted(self._const_targets.items()), slots):
            cs = self.rows[r][1]
            cs.extend([0] * (i + 1 - len(cs)))
            cs[i] = c
            wt = Target(r, w)
            self.connect(wt, t)
            self.add_generator([], lambda v, wt=wt, c=c: [(wt, c)], OP_CONST, (c,), outs=[wt])

    def target_as_constant(self, t):
        """the value if `t` is a constant target created by `constant()` (plonky2 `target_as_constant`)"""
        return self._target_const.get(t.key())

    def zero(self):
        return self.constant(0)

    def one(self):
        return self.constant(1)

    # ---- arithmetic (ArithmeticGate: out = c0*m0*m1 + c1*addend)
    def arithmetic(self, c0, m0, m1, c1, addend):
        c0 %= P
        c1 %= P
        gate = G.ArithmeticGate.new_from_config(self.config)
        slot = self._arith_slot.get((c0, c1))
        if slot is None or slot[1] == gate.num_ops:
            slot = [self.add_gate(gate, [c0, c1]), 0]
            self._arith_slot[(c0, c1)] = slot
        row, i = slot
        slot[1] += 1
        w = [Target(row, 4 * i + k) for k in range(4)]
        self.connect(m0, w[0])
        self.connect(m1, w[1])
        self.connect(addend, w[2])
        self.add_generator([w[0], w[1], w[2]],
                           lambda v, w=w, c0=c0, c1=c1: [(w[3], (c0 * v[0] * v[1] + c1 * v[2]) % P)], OP_ARITH, (c0, c1), outs=[w[3]])
        return w[3]

    def mul(self, a, b):
        return self.arithmetic(1, a, b, 0, self.zero())

    def add(self, a, b):
        return self.arithmetic(1, a, self.one(), 1, b)

    def sub(self, a, b):
        return self.arithmetic(1, a, self.one(), P - 1, b)

    def mul_add(self, a, b, c):
        return self.arithmetic(1, a, b, 1, c)

    def mul_sub(self, a, b, c):
        return self.arithmetic(1, a, b, P - 1, c)

    def select(self, b, x, y):
        """plonky2 `select`: b ? x : y as b*x - (b*y - y)"""
        return self.mul_sub(b, x, self.mul_sub(b, y, y))

    def neg_one(self):
        return self.constant(P - 1)

    def two(self):
        return self.constant(2)

    def mul_const(self, c, a):
        return self.arithmetic(c, a, self.one(), 0, self.zero())

    # ---- BaseSumGate: little-endian bit decomposition (split_le)
    def split_le(self, x, num_bits):
        gate = G.BaseSumGate(num_bits, 2)
        row = self.add_gate(gate)
        s = Target(row, 0)
        self.connect(x, s)
        bits = [Target(row, 1 + i) for i in range(num_bits)]

        def gen(v, bits=bits, num_bits=num_bits):
            assert v[0] < (1 << num_bits), "split_le: value does not fit"
            return [(bits[i], (v[0] >> i) & 1) for i in range(num_bits)]
        self.add_generator([s], gen, OP_SPLIT, (2, num_bits), outs=bits)
        return bits

    def le_sum(self, bits):
        """BaseSumGate row whose limbs are connected to `bits` (little-endian); returns the sum wire
        (plonky2 `CircuitBuilder::le_sum`, used by crypto/plonky2_sha512/src/circuit.rs:74-80)"""
        bits = list(bits)
        if not bits:
            return self.zero()
        row = self.add_gate(G.BaseSumGate(len(bits), 2))
        limbs = [Target(row, 1 + i) for i in range(len(bits))]
        for b, l in zip(bits, limbs):
            self.connect(b, l)
        s = Target(row, 0)
        self.add_generator(limbs, lambda v, s=s: [(s, sum(x << i for i, x in enumerate(v)) % P)], OP_LE_SUM, outs=[s])
        return s

    # ---- U32AddManyGate (crypto/plonky2_u32/src/gadgets/arithmetic_u32.rs:157-183 `add_many_u32`)
    def add_many_u32(self, to_add):
        to_add = list(to_add)
        if len(to_add) == 0:
            return self.zero(), self.zero()
        if len(to_add) == 1:
            return to_add[0], self.zero()
        if len(to_add) == 2:
            return self.mul_add_u32(to_add[0], self.one(), to_add[1])
        return self._add_many_gate(to_add, self.zero())

    def add_u32s_with_carry(self, to_add, carry):
        """arithmetic_u32.rs:185-214: U32AddManyGate with an explicit carry-in target"""
        to_add = list(to_add)
        if len(to_add) == 1:
            return self.mul_add_u32(to_add[0], self.one(), carry)
        return self._add_many_gate(to_add, carry)

    def _add_many_gate(self, to_add, carry_t):
        na = len(to_add)
        gate = G.U32AddManyGate.new_from_config(self.config, na)
        slot = self._addmany_slot.get(na)
        if slot is None or slot[1] == gate.num_ops:
            slot = [self.add_gate(gate), 0]
            self._addmany_slot[na] = slot
        row, i = slot
        slot[1] += 1
        per = na + 3
        ins = [Target(row, per * i + j) for j in range(na + 1)]
        for a, w in zip(list(to_add) + [carry_t], ins):
            self.connect(a, w)
        res, carry = Target(row, per * i + na + 1), Target(row, per * i + na + 2)
        limbs = TargetRange(row, per * gate.num_ops + 18 * i, 18)

        def gen(v, res=res, carry=carry, limbs=limbs):
            s = sum(v)
            lo, hi = s & 0xFFFFFFFF, s >> 32
            assert hi < 16, "U32AddManyGate: carry does not fit in 4 bits"
            out = [(res, lo), (carry, hi)]
            out += [(limbs[j], (lo >> (2 * j)) & 3) for j in range(16)]
            out += [(limbs[16 + j], (hi >> (2 * j)) & 3) for j in range(2)]
            return out
        self.add_generator(ins, gen, OP_ADD_MANY, outs=[res, carry, limbs])
        return res, carry

    def sub_u32(self, x, y, borrow):
        """arithmetic_u32.rs:221-238 (U32SubtractionGate): (x - y - borrow mod 2^32, borrow out)"""
        gate = G.U32SubtractionGate.new_from_config(self.config)
        if self._sub_slot is None or self._sub_slot[1] == gate.num_ops:
            self._sub_slot = [self.add_gate(gate), 0]
        row, i = self._sub_slot
        self._sub_slot[1] += 1
        w = [Target(row, 5 * i + k) for k in range(5)]
        for a, t in zip((x, y, borrow), w):
            self.connect(a, t)
        limbs = TargetRange(row, 5 * gate.num_ops + 16 * i, 16)

        def gen(v, w=w, limbs=limbs):
            d = v[0] - v[1] - v[2]
            bout = 1 if d < 0 else 0
            res = d + (bout << 32)
            assert 0 <= res < (1 << 32)
            return [(w[3], res), (w[4], bout)] + [(limbs[j], (res >> (2 * j)) & 3) for j in range(16)]
        self.add_generator(w[:3], gen, OP_SUB_U32, outs=[w[3], w[4], limbs])
        return w[3], w[4]

    def range_check_u32(self, vals):
        """crypto/plonky2_u32/src/gadgets/range_check.rs: one U32RangeCheckGate row for the list"""
        vals = list(vals)
        gate = G.U32RangeCheckGate(len(vals))
        row = self.add_gate(gate)
        n = len(vals)
        ins = [Target(row, i) for i in range(n)]
        for a, t in zip(vals, ins):
            self.connect(a, t)

        def gen(v, row=row, n=n):
            out = []
            for i, x in enumerate(v):
                assert x < (1 << 32), "range_check_u32: value exceeds 32 bits"
                out += [(Target(row, n + 16 * i + j), (x >> (2 * j)) & 3) for j in range(16)]
            return out
        self.add_generator(ins, gen, OP_RANGE_CHECK, outs=[TargetRange(row, n, 16 * n)])

    def _comparison(self, a, b, num_bits=32):
        """one ComparisonGate row: result = (a <= b)  (crypto/plonky2_u32/src/gates/comparison.rs generator)"""
        num_chunks = -(-num_bits // 2)
        gate = G.ComparisonGate(num_bits, num_chunks)
        row = self.add_gate(gate)
        wa, wb = Target(row, 0), Target(row, 1)
        self.connect(a, wa)
        self.connect(b, wb)
        nc, cb = num_chunks, gate.chunk_bits

        def gen(v, row=row, nc=nc, cb=cb):
            x, y = v
            out = []
            msd = 0
            size = 1 << cb
            for i in range(nc):
                ca, cy = (x >> (cb * i)) & (size - 1), (y >> (cb * i)) & (size - 1)
                diff = (cy - ca) % P
                eq = 1 if ca == cy else 0
                out += [(Target(row, 4 + i), ca), (Target(row, 4 + nc + i), cy),
                        (Target(row, 4 + 2 * nc + i), 1 if eq else pow(diff, P - 2, P)), (Target(row, 4 + 3 * nc + i), eq)]
                inter = eq * msd % P
                out.append((Target(row, 4 + 4 * nc + i), inter))
                msd = (inter + (1 - eq) * diff) % P
            out.append((Target(row, 3), msd))
            top = (size + msd) % P
            assert top < 2 * size
            for i in range(cb + 1):
                out.append((Target(row, 4 + 5 * nc + i), (top >> i) & 1))
            out.append((Target(row, 2), (top >> cb) & 1))
            return out
        outs = []
        for i in range(nc):
            outs += [Target(row, 4 + i), Target(row, 4 + nc + i), Target(row, 4 + 2 * nc + i), Target(row, 4 + 3 * nc + i),
                     Target(row, 4 + 4 * nc + i)]
        outs += [Target(row, 3)] + [Target(row, 4 + 5 * nc + i) for i in range(cb + 1)] + [Target(row, 2)]
        self.add_generator([wa, wb], gen, OP_COMPARISON, (nc, cb), outs=outs)
        return Target(row, 2)

    def list_le(self, a, b, num_bits=32):
        """multiple_comparison.rs:17-64 `list_le_circuit`: a <= b as little-endian lists of num_bits-bit limbs"""
        assert len(a) == len(b)
        one = self.one()
        result = one
        for x, y in zip(a, b):
            a_le_b = self._comparison(x, y, num_bits)
            b_le_a = self._comparison(y, x, num_bits)
            equal = self.mul(a_le_b, b_le_a)
            less = self.sub(one, b_le_a)
            result = self.mul_add(equal, result, less)
        return result

    def not_(self, b):
        return self.sub(self.one(), b)

    def assert_zero(self, x):
        self.connect(x, self.zero())

    def assert_one(self, x):
        self.connect(x, self.one())

    def assert_bool(self, b):
        """b * b - b == 0"""
        self.assert_zero(self.arithmetic(1, b, b, P - 1, b))

    def is_equal(self, x, y):
        """plonky2 `is_equal`: equal bit + inverse-of-difference witness"""
        equal, inv = self.add_virtual_target(), self.add_virtual_target()
        self.add_generator([x, y], lambda v, equal=equal, inv=inv: [(equal, 1 if v[0] == v[1] else 0),
                                                                    (inv, 0 if v[0] == v[1] else pow((v[0] - v[1]) % P, P - 2, P))],
                           OP_IS_EQUAL, outs=[equal, inv])
        not_equal = self.not_(equal)
        diff = self.sub(x, y)
        self.connect(self.mul(diff, equal), self.zero())
        self.connect(not_equal, self.mul(diff, inv))
        return equal

    def split_le_base(self, x, num_limbs, base):
        """plonky2 `split_le_base::<B>`: BaseSumGate row, little-endian base-B limbs"""
        if base == 2:
            return self.split_le(x, num_limbs)
        gate = G.BaseSumGate(num_limbs, base)
        row = self.add_gate(gate)
        s = Target(row, 0)
        self.connect(x, s)
        limbs = [Target(row, 1 + i) for i in range(num_limbs)]

        def gen(v, limbs=limbs, base=base, num_limbs=num_limbs):
            assert v[0] < base ** num_limbs, "split_le_base: value does not fit"
            return [(limbs[i], (v[0] // base ** i) % base) for i in range(num_limbs)]
        self.add_generator([s], gen, OP_SPLIT, (base, num_limbs), outs=limbs)
        return limbs

    def random_access(self, index, items):
        """plonky2 `random_access` (RandomAccessGate): items[index], len(items) a power of two"""
        items = list(items)
        bits = len(items).bit_length() - 1
        assert 1 << bits == len(items)
        if bits == 0:
            return items[0]
        gate = G.RandomAccessGate.new_from_config(self.config, bits)
        slot = self._ra_slot.get(bits)
        if slot is None or slot[1] == gate.num_copies:
            consts = [0] * gate.num_extra_constants
            slot = [self.add_gate(gate, consts), 0]
            self._ra_slot[bits] = slot
        row, cp = slot
        slot[1] += 1
        vs = 1 << bits
        base = (2 + vs) * cp
        w_idx, w_claim = Target(row, base), Target(row, base + 1)
        w_items = [Target(row, base + 2 + i) for i in range(vs)]
        self.connect(index, w_idx)
        for a, t in zip(items, w_items):
            self.connect(a, t)
        w_bits = [Target(row, gate.num_routed + cp * bits + i) for i in range(bits)]

        def gen(v, w_claim=w_claim, w_bits=w_bits, bits=bits):
            idx = v[0]
            assert idx < (1 << bits), "random_access: index out of range"
            return [(w_claim, v[1 + idx])] + [(w_bits[i], (idx >> i) & 1) for i in range(bits)]
        self.add_generator([w_idx] + w_items, gen, OP_RANDOM_ACCESS, (bits,), outs=[w_claim] + w_bits)
        return w_claim

    # ---- U32ArithmeticGate: (lo, hi) = m0*m1 + addend on 32-bit values
    def mul_add_u32(self, m0, m1, addend):
        c0, c1, c2 = self.target_as_constant(m0), self.target_as_constant(m1), self.target_as_constant(addend)
        if c0 is not None and c1 is not None and c2 is not None:   # arithmetic_u32.rs:107-130 special case
            s_ = (c0 * c1 + c2) % P
            return self.constant(s_ & 0xFFFFFFFF), self.constant(s_ >> 32)
        gate = G.U32ArithmeticGate.new_from_config(self.config)
        if self._u32_slot is None or self._u32_slot[1] == gate.num_ops:
            self._u32_slot = [self.add_gate(gate), 0]
        row, i = self._u32_slot
        self._u32_slot[1] += 1
        w = [Target(row, 6 * i + k) for k in range(6)]
        limbs = TargetRange(row, 6 * gate.num_ops + 32 * i, 32)
        self.connect(m0, w[0])
        self.connect(m1, w[1])
        self.connect(addend, w[2])

        def gen(v, w=w, limbs=limbs):
            out = v[0] * v[1] + v[2]
            assert out < (1 << 64) and out < P
            lo, hi = out & 0xFFFFFFFF, out >> 32
            diff = (0xFFFFFFFF - hi) % P
            inv = pow(diff, P - 2, P) if diff else 0   # arithmetic_u32.rs generator: inverse of (u32::MAX - hi), or 0
            res = [(w[3], lo), (w[4], hi), (w[5], inv)]
            res += [(limbs[j], (out >> (2 * j)) & 3) for j in range(32)]
            return res
        self.add_generator(w[:3], gen, OP_U32_MULADD, outs=[w[3], w[4], w[5], limbs])
        return w[3], w[4]

    # ---- interleaved ("B32") representation: crypto/plonky2_u32/src/gadgets/interleaved_u32.rs
    def interleave_u32(self, x):
        """:75-83 -- the bits of x spread over the even positions of a 64-bit word (U32InterleaveGate)"""
        gate = G.U32InterleaveGate.new_from_config(self.config)
        if self._il_slot is None or self._il_slot[1] == gate.num_ops:
            self._il_slot = [self.add_gate(gate), 0]
        row, i = self._il_slot
        self._il_slot[1] += 1
        wx, wi = Target(row, 2 * i), Target(row, 2 * i + 1)
        self.connect(wx, x)
        bits = TargetRange(row, 2 * gate.num_ops + 32 * i, 32)

        def gen(v, wi=wi, bits=bits):
            xv = v[0]
            assert xv < (1 << 32), "interleave_u32: value exceeds 32 bits"
            bv = [(xv >> (31 - j)) & 1 for j in range(32)]                  # big-endian
            return [(wi, sum(b << (2 * (31 - j)) for j, b in enumerate(bv)))] + list(zip(bits, bv))
        self.add_generator([wx], gen, OP_INTERLEAVE, outs=[wi, bits])
        return wi

    def _uninterleave(self, x, to_b32):
        cls = G.UninterleaveToB32Gate if to_b32 else G.UninterleaveToU32Gate
        gate = cls.new_from_config(self.config)
        slot = self._ul_slot.get(to_b32)
        if slot is None or slot[1] == gate.num_ops:
            slot = self._ul_slot[to_b32] = [self.add_gate(gate), 0]
        row, i = slot
        slot[1] += 1
        wx, we, wo = (Target(row, 3 * i + k) for k in range(3))
        self.connect(wx, x)
        bits = TargetRange(row, 3 * gate.num_ops + 64 * i, 64)

        def gen(v, we=we, wo=wo, bits=bits, to_b32=to_b32):
            bv = [(v[0] >> (63 - j)) & 1 for j in range(64)]               # big-endian
            step = 2 if to_b32 else 1
            ev = sum(bv[2 * j] << (step * (31 - j)) for j in range(32))
            od = sum(bv[2 * j + 1] << (step * (31 - j)) for j in range(32))
            return [(we, ev), (wo, od)] + list(zip(bits, bv))
        self.add_generator([wx], gen, OP_UNINTERLEAVE, (1 if to_b32 else 0,), outs=[we, wo, bits])
        return we, wo

    def uninterleave_to_u32(self, x):
        """:85-99 -- (bits at the even positions, bits at the odd positions) of a 64-bit word as two u32"""
        return self._uninterleave(x, False)

    def uninterleave_to_b32(self, x):
        """:101-115 -- the same halves, left in interleaved form"""
        return self._uninterleave(x, True)

    def and_xor_b32(self, x, y):
        """:145-148: the sum of two interleaved words has (a AND b) at the even and (a XOR b) at the odd positions"""
        return self.uninterleave_to_b32(self.add(x, y))

    def and_xor_u32(self, x, y):
        return self.and_xor_b32(self.interleave_u32(x), self.interleave_u32(y))

    def and_xor_b32_to_u32(self, x, y):
        return self.uninterleave_to_u32(self.add(x, y))

    def and_xor_u32_to_u32(self, x, y):
        return self.and_xor_b32_to_u32(self.interleave_u32(x), self.interleave_u32(y))

    def and_u32(self, x, y):
        return self.and_xor_u32_to_u32(x, y)[0]

    def xor_u32(self, x, y):
        return self.and_xor_u32_to_u32(x, y)[1]

    def unsafe_xor_many_u32(self, xs):
        """:117-143"""
        xs = list(xs)
        if len(xs) == 0:
            return self.zero()
        if len(xs) == 1:
            return xs[0]
        if len(xs) <= 3:
            t = self.xor_u32(xs[0], xs[1])
            return t if len(xs) == 2 else self.xor_u32(t, xs[2])
        r = self.interleave_u32(xs[0])
        for i in range((len(xs) - 3) // 2):
            a, c = self.interleave_u32(xs[1 + 2 * i]), self.interleave_u32(xs[2 + 2 * i])
            r = self.uninterleave_to_b32(self.add(self.add(r, a), c))[1]
        if len(xs) % 2 == 0:
            r = self.and_xor_b32(r, self.interleave_u32(xs[-3]))[1]
        a, c = self.interleave_u32(xs[-2]), self.interleave_u32(xs[-1])
        return self.uninterleave_to_u32(self.add(self.add(r, a), c))[1]

    def mul_u32(self, a, b):
        return self.mul_add_u32(a, b, self.zero())

    def add_u32(self, a, b):
        """arithmetic_u32.rs:139-142"""
        return self.mul_add_u32(a, self.one(), b)

    def not_u32(self, a):
        return self.sub_u32(self.constant(0xFFFFFFFF), a, self.zero())[0]

    def lsh_u32(self, a, n):
        return self.mul_u32(a, self.constant(1 << n))[0]

    def rsh_u32(self, a, n):
        return a if n == 0 else self.mul_u32(a, self.constant(1 << (32 - n)))[1]

    def lrot_u32(self, a, n):
        lo, hi = self.mul_u32(a, self.constant(1 << n))
        return self.add_u32(lo, hi)[0]

    def rrot_u32(self, a, n):
        return self.lrot_u32(a, 32 - n)

    # ---- Poseidon (PoseidonGate rows); witness rows come from the library (zklc_poseidon_gl_gate_rows)
    def permute(self, state12, swap=None):
        """one PoseidonGate row; `swap` (a boolean target) exchanges inputs 0..4 and 4..8 first (plonky2 `permute_swapped`)"""
        row = self.add_gate(G.PoseidonGate())
        ins = [Target(row, i) for i in range(12)]
        for a, b in zip(state12, ins):
            self.connect(a, b)
        sw = Target(row, G.PoseidonGate.WIRE_SWAP)
        self.connect(sw, self.zero() if swap is None else swap)

        def gen(v, row=row):
            from .prover import poseidon_gate_rows
            r = poseidon_gate_rows(np.array([v[:12]], dtype=np.uint64), np.array([v[12]], dtype=np.uint64))[0]
            return [(Target(row, c), int(r[c])) for c in range(12, 135) if c != 24]
        self.add_generator(ins + [sw], gen, OP_POSEIDON, outs=[TargetRange(row, 12, 12), TargetRange(row, 25, 110)])
        return [Target(row, 12 + i) for i in range(12)]

    def hash_n_to_hash_no_pad(self, inputs):
        """overwrite-mode sponge, rate 8 (poseidon/goldilocks.go:41-68)"""
        z = self.zero()
        state = [z] * 12
        for i in range(0, len(inputs), 8):
            chunk = inputs[i:i + 8]
            state = list(chunk) + state[len(chunk):]
            state = self.permute(state)
        return state[:4]

    # ---- build
    def build(self):
        cfg = self.config
        # public inputs hash wired into a PublicInputGate (plonky2 `CircuitBuilder::build`)
        pi_hash = self.hash_n_to_hash_no_pad(self.public_inputs)
        pi_row = self.add_gate(G.PublicInputGate())
        for i in range(4):
            self.connect(pi_hash[i], Target(pi_row, i))
        self._place_constants()
        # pad to a power of two (and to a degree FRI can handle: lde size >= 2^cap_height)
        min_rows = max(1 << max(0, cfg["fri_config"]["cap_height"] - cfg["fri_config"]["rate_bits"]), 2)
        while len(self.rows) < min_rows or len(self.rows) & (len(self.rows) - 1):
            self.add_gate(G.NoopGate())
        return CircuitData(self)


def fri_reduction_arity_bits(cfg, degree_bits):
    """FriReductionStrategy::ConstantArityBits(arity, final_poly_bits).reduction_arity_bits (plonky2 fri/reduction_strategies.rs)"""
    arity_bits, final_poly_bits = cfg["fri_config"]["reduction_strategy"]["ConstantArityBits"]
    rate_bits, cap_height = cfg["fri_config"]["rate_bits"], cfg["fri_config"]["cap_height"]
    out = []
    while degree_bits > final_poly_bits and degree_bits + rate_bits - arity_bits >= cap_height:
        out.append(arity_bits)
        degree_bits -= arity_bits
    return out


class CircuitData:
    """Prover-side circuit: sorted gates, selector groups, constants, sigmas (numpy u64, poly-major) + witness program."""

    def __init__(self, b):
        n = len(b.rows)
        uniq = sorted({g for g, _ in b.rows}, key=lambda g: (g.degree, g.id()))
        index = {g: i for i, g in enumerate(uniq)}
        row_gate = np.array([index[g] for g, _ in b.rows], dtype=np.int64)
        ngc = max(g.num_constants for g in uniq)
        row_consts = np.zeros((ngc, n), dtype=np.uint64)
        for r, (_, cs) in enumerate(b.rows):
            for k, c in enumerate(cs):
                row_consts[k, r] = c
        routed = b.config["num_routed_wires"]
        # sigma: every routed wire maps to the next wire of its copy class, cyclically; members in (row, column) order -- the order in
        # which plonky2's `wire_partition` (plonk/permutation_argument.rs) walks the wires: for row { for column }
        sig_col = np.tile(np.arange(routed, dtype=np.int64)[:, None], (1, n))
        sig_row = np.tile(np.arange(n, dtype=np.int64)[None, :], (routed, 1))
        pk, pr = b.class_arrays()
        wk, wr = pk[pk < VIRTUAL_BASE], pr[pk < VIRTUAL_BASE]
        root_wires = np.unique(pr[pr < VIRTUAL_BASE])          # a class root has no parent entry: add it to its own class
        wk, wr = np.concatenate([wk, root_wires]), np.concatenate([wr, root_wires])
        if len(wk):
            col, row = wk & 255, wk >> 8
            assert int(col.max()) < routed, "copy constraint on a non-routed wire"
            o = np.lexsort((col, row, wr))
            col, row, wr = col[o], row[o], wr[o]
            first = np.ones(len(wr), dtype=bool)
            first[1:] = wr[1:] != wr[:-1]
            start = np.maximum.accumulate(np.where(first, np.arange(len(wr)), 0))       # index of the first member of my class
            nxt = np.arange(len(wr)) + 1
            last = np.ones(len(wr), dtype=bool)
            last[:-1] = first[1:]
            nxt[last] = start[last]
            sig_col[col, row], sig_row[col, row] = col[nxt], row[nxt]
        self.builder = b
        self._plan = None
        self._trace = None
        self._program = None
        self._init_arrays(b.config, uniq, row_gate, row_consts, sig_col, sig_row, len(b.public_inputs))

    _container = None         # set by plonky2/container.py `read_circuit`: the mapped file this circuit's arrays are views of

    @property
    def subgroup(self):
        """[w^i] for the circuit's 2^degree_bits subgroup as Python integers (tests and tools only; computed on first use)"""
        if getattr(self, "_subgroup", None) is None:
            w, sub = root_of_unity(self.degree_bits), [1] * self.n
            for i in range(1, self.n):
                sub[i] = sub[i - 1] * w % P
            self._subgroup = sub
        return self._subgroup

    def save(self, path, aux=None, note=None):
        """write this circuit (and its compiled witness program) as a circuit container (plonky2/container.py, include/zklc.h b'')"""
        from .container import write_circuit
        write_circuit(path, self, aux, note)

    @classmethod
    def load(cls, path, verify=True):
        """-> (CircuitData, aux) from a circuit container; the arrays are views of the mapped file"""
        from .container import read_circuit
        return read_circuit(path, verify)

    @classmethod
    def from_arrays(cls, config, gates, row_gate, row_consts, sig_col, sig_row, num_public_inputs):
        """gates: list sorted by (degree, id); row_gate[n]: index into gates; row_consts[k, n]: gate-local constants;
        sigma as (column, row) index arrays of shape [routed, n]"""
        self = cls.__new__(cls)
        self.builder = None
        self._plan = None
        self._trace = None
        self._program = None
        assert gates == sorted(gates, key=lambda g: (g.degree, g.id()))
        self._init_arrays(config, gates, row_gate, row_consts, sig_col, sig_row, num_public_inputs)
        return self

    def _init_arrays(self, cfg, uniq, row_gate, row_consts, sig_col, sig_row, num_public_inputs):
        self.config = cfg
        n = self.n = len(row_gate)
        assert n & (n - 1) == 0
        self.degree_bits = n.bit_length() - 1
        qdf = self.quotient_degree_factor = cfg["max_quotient_degree_factor"]
        # selector groups (plonky2 gates/selectors.rs `selector_polynomials`)
        self.gates = uniq
        max_degree = qdf + 1
        if uniq[-1].degree + len(uniq) - 1 <= max_degree:
            groups = [(0, len(uniq))]
        else:
            assert uniq[-1].degree < max_degree, "gate degree too high for the quotient degree factor"
            groups, start = [], 0
            while start < len(uniq):
                size = 0
                while start + size < len(uniq) and size + uniq[start + size].degree < max_degree:
                    size += 1
                groups.append((start, start + size))
                start += size
        self.groups = groups
        group_of = np.array([next(j for j, (s, e) in enumerate(groups) if s <= i < e) for i in range(len(uniq))])
        self.selector_indices = [int(x) for x in group_of]
        nsel = len(groups)
        ngc = max(g.num_constants for g in uniq)
        assert row_consts.shape[0] >= ngc
        self.num_constants = nsel + ngc
        consts = np.zeros((self.num_constants, n), dtype=np.uint64)
        row_group = group_of[row_gate]
        for s_ in range(nsel):
            consts[s_] = np.where(row_group == s_, row_gate, UNUSED_SELECTOR).astype(np.uint64)
        consts[nsel:] = row_consts[:ngc]
        self.constants = consts
        self.num_gate_constraints = max(g.num_constraints for g in uniq)
        routed = cfg["num_routed_wires"]
        self.num_partial_products = -(-routed // qdf) - 1
        self.k_is = [pow(GENERATOR, i, P) for i in range(routed)]
        sub = self.subgroup
        # sigma_j(w^i) = k_is[col'] * w^(row'): one field multiplication per cell
        sub_arr = np.array(sub, dtype=np.uint64)
        k_arr = np.array(self.k_is, dtype=np.uint64)
        from .. import _lib
        ka, sa = np.ascontiguousarray(k_arr[sig_col]), np.ascontiguousarray(sub_arr[sig_row])
        sig = np.empty_like(ka)
        _lib.load().zklc_gl_mul_vec(ka.ctypes.data, sa.ctypes.data, sig.ctypes.data, ka.size)
        self.sigmas = sig
        self.fri_arity_bits = fri_reduction_arity_bits(cfg, self.degree_bits)
        self.num_public_inputs = num_public_inputs

    # ---- native witness generation (csrc/plonky2_witness.cpp)
    def witness_program(self, example_inputs, trace_python=False):
        """Compile the circuit's generators into the interpreter's program.  `example_inputs`: a partial witness {Target: value}
        (or just the list of its targets) naming the circuit's inputs.  When every generator declares its outputs the program is
        assembled from the dependency structure alone (numpy, no generator is run); otherwise (or with trace_python) one run
        of the Python generators on the example -- which must then be satisfiable -- fixes the order and the output slots."""
        if self._program is not None:
            return self._program
        b = self.builder
        assert self.config["num_wires"] * self.n < (1 << 32)
        if not trace_python and all(g[4] is not None for g in b.generators):
            self._program = self._compile_program(list(example_inputs))
            return self._program
        self._trace = []
        try:
            self.generate_witness(example_inputs)
            trace = self._trace
        finally:
            self._trace = None
        _, find = self._plan
        slot_of = {}

        def slot(k):
            s_ = slot_of.get(k)
            if s_ is None:
                s_ = slot_of[k] = len(slot_of)
            return s_
        code, pvals = [], []
        for gi, out_keys in trace:
            ins, _, op, params, _ = b.generators[gi]
            assert op is not None, "generator without a native opcode"
            code += [op, len(params), len(ins), len(out_keys)]
            pvals += [x - (1 << 64) if x >= (1 << 63) else x for x in params]
            code += [slot(find(t)) for t in ins] + [slot(k) for k in out_keys]
        in_targets = list(example_inputs.keys()) if isinstance(example_inputs, dict) else list(example_inputs)
        ws, wi = [], []
        n_rows = self.n
        for k in list(b.parent) + [k for k in slot_of if k < VIRTUAL_BASE and k not in b.parent]:
            if k < VIRTUAL_BASE:
                r = b._find(k)
                if r in slot_of:
                    ws.append(slot_of[r])
                    wi.append((k & 255) * n_rows + (k >> 8))
        order = np.argsort(np.array(wi, dtype=np.int64), kind="stable")      # scatter in address order
        self._program = {
            "code": np.array(code, dtype=np.uint32), "params": np.array(pvals + [0], dtype=np.int64),
            "input_targets": in_targets, "input_slots": np.array([slot(find(t)) for t in in_targets], dtype=np.uint32),
            "wire_slot": np.array(ws, dtype=np.uint32)[order], "wire_index": np.array(wi, dtype=np.uint32)[order],
            "pi_slots": np.array([slot(find(t)) for t in b.public_inputs], dtype=np.uint32),
        }
        self._program["n_slots"] = len(slot_of) + 1
        return self._program

    def _compile_program(self, in_targets):
        """the interpreter program from the generators' declared inputs / outputs: copy classes -> dense slots, a topological
        order (creation order when it already is one), instruction words and the slot -> wire-cell scatter list, all as array
        operations (the Ed25519 circuit has ~2 M generators over ~40 M targets)"""
        b = self.builder
        gens = b.generators
        G_ = len(gens)
        assert all(g[2] is not None for g in gens), "generator without a native opcode"
        ins_len = np.fromiter((len(g[0]) for g in gens), dtype=np.int64, count=G_)
        outs_len = np.fromiter((sum(o.n if type(o) is TargetRange else 1 for o in g[4]) for g in gens), dtype=np.int64, count=G_)
        par_len = np.fromiter((len(g[3]) for g in gens), dtype=np.int64, count=G_)
        ops = np.fromiter((g[2] for g in gens), dtype=np.int64, count=G_)
        ins_k = np.fromiter((t.k for g in gens for t in g[0]), dtype=np.int64, count=int(ins_len.sum()))
        flat = []
        for g in gens:
            for o in g[4]:
                if type(o) is TargetRange:
                    flat.extend(o.keys())
                else:
                    flat.append(o.k)
        outs_k = np.array(flat, dtype=np.int64)
        del flat
        params = np.fromiter((x - (1 << 64) if x >= (1 << 63) else x for g in gens for x in g[3]), dtype=np.int64,
                             count=int(par_len.sum()))
        in_k = np.array([t.k for t in in_targets], dtype=np.int64)
        pi_k = np.array([t.k for t in b.public_inputs], dtype=np.int64)
        # copy classes: key -> root for the keys that were ever connected, identity for the rest
        pk, pr = b.class_arrays()                      # sorted by key

        def roots(keys):
            if len(pk) == 0 or len(keys) == 0:
                return keys.copy()
            idx = np.minimum(np.searchsorted(pk, keys), len(pk) - 1)
            hit = pk[idx] == keys
            r = keys.copy()
            r[hit] = pr[idx[hit]]
            return r
        parts = [roots(ins_k), roots(outs_k), roots(in_k), roots(pi_k)]
        uniq, inv = np.unique(np.concatenate(parts), return_inverse=True)
        # wire cells to fill: every connected wire, every wire a generator names, and the class roots that are wires
        wire_k = np.unique(np.concatenate([pk[pk < VIRTUAL_BASE], ins_k[ins_k < VIRTUAL_BASE], outs_k[outs_k < VIRTUAL_BASE],
                                           in_k[in_k < VIRTUAL_BASE], uniq[uniq < VIRTUAL_BASE]]))
        n_slots = len(uniq)
        cuts = np.cumsum([len(p) for p in parts])
        in_slots, out_slots, input_slots, pi_slots = inv[:cuts[0]], inv[cuts[0]:cuts[1]], inv[cuts[1]:cuts[2]], inv[cuts[2]:]
        # order: creation order if every input is a circuit input or produced by an earlier generator; else multi-pass scheduling
        in_off, out_off = np.cumsum(ins_len) - ins_len, np.cumsum(outs_len) - outs_len
        gen_of_in = np.repeat(np.arange(G_, dtype=np.int64), ins_len)
        producer = np.full(n_slots, G_, dtype=np.int64)
        np.minimum.at(producer, out_slots, np.repeat(np.arange(G_, dtype=np.int64), outs_len))
        producer[input_slots] = -1
        if np.all(producer[in_slots] < gen_of_in):
            order = np.arange(G_, dtype=np.int64)
        else:
            avail = np.zeros(n_slots, dtype=bool)
            avail[input_slots] = True
            done, pending = [], range(G_)
            il, ol = ins_len.tolist(), outs_len.tolist()
            io, oo = in_off.tolist(), out_off.tolist()
            while len(pending):
                rest = []
                for g in pending:
                    if il[g] == 0 or avail[in_slots[io[g]:io[g] + il[g]]].all():
                        done.append(g)
                        avail[out_slots[oo[g]:oo[g] + ol[g]]] = True
                    else:
                        rest.append(g)
                assert len(rest) < len(pending), "witness program: %d generators can never run" % len(rest)
                pending = rest
            order = np.array(done, dtype=np.int64)
        # instruction words: [opcode, n_params, n_in, n_out, ins..., outs...]
        il, ol = ins_len[order], outs_len[order]
        length = 4 + il + ol
        starts = np.cumsum(length) - length
        code = np.empty(int(length.sum()), dtype=np.uint32)
        code[starts], code[starts + 1], code[starts + 2], code[starts + 3] = ops[order], par_len[order], il, ol

        def gather(dst0, lens, src_off, src):
            tot = int(lens.sum())
            within = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
            code[np.repeat(dst0, lens) + within] = src[np.repeat(src_off, lens) + within]
        gather(starts + 4, il, in_off[order], in_slots)
        gather(starts + 4 + il, ol, out_off[order], out_slots)
        pl = par_len[order]
        tot = int(pl.sum())
        within = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(pl) - pl, pl)
        pvals = params[np.repeat((np.cumsum(par_len) - par_len)[order], pl) + within]
        # every wire cell whose copy class has a slot receives that slot's value
        wr = roots(wire_k)
        idx = np.minimum(np.searchsorted(uniq, wr), n_slots - 1)
        has = uniq[idx] == wr
        wire_k, idx = wire_k[has], idx[has]
        wi = (wire_k & 255) * self.n + (wire_k >> 8)
        o = np.argsort(wi, kind="stable")
        return {"code": code, "params": np.concatenate([pvals, np.zeros(1, dtype=np.int64)]), "input_targets": in_targets,
                "input_slots": input_slots.astype(np.uint32), "wire_slot": idx[o].astype(np.uint32),
                "wire_index": wi[o].astype(np.uint32), "pi_slots": pi_slots.astype(np.uint32), "n_slots": n_slots + 1}

    def generate_witness_native(self, inputs_list, out=None, threads=None, input_values=None):
        """inputs_list: partial witnesses ({Target: value} with the same keys as the example given to witness_program), or
        `input_values`: the same as a uint64 array [k, n_inputs] in the order of witness_program()["input_targets"].
        Returns (wires uint64 [k, num_wires, n], public inputs uint64 [k, n_pi]); raises if a witness does not exist.
        `out` may be a zero-initialised buffer of that shape reused across calls (only the circuit's wire cells are written)."""
        import ctypes
        import os
        from .. import _lib
        pr = self._program
        assert pr is not None, "call witness_program(example_inputs) first"
        if input_values is not None:
            vals = np.ascontiguousarray(input_values, dtype=np.uint64).reshape(-1, len(pr["input_targets"]))
            k = vals.shape[0]
        else:
            k = len(inputs_list)
            vals = np.array([[int(w[t]) for t in pr["input_targets"]] for w in inputs_list], dtype=np.uint64).reshape(k, -1)
        nw = self.config["num_wires"]
        wires = out if out is not None else np.zeros((k, nw, self.n), dtype=np.uint64)
        assert wires.shape == (k, nw, self.n) and wires.dtype == np.uint64 and wires.flags["C_CONTIGUOUS"]
        npi = len(pr["pi_slots"])
        pis = np.zeros((k, max(npi, 1)), dtype=np.uint64)
        status = np.zeros(k, dtype=np.int32)
        err = ctypes.create_string_buffer(200 * k)
        threads = threads or min(k, len(os.sched_getaffinity(0)))
        rc = _lib.load().zklc_plonky2_witness_run(
            pr["code"].ctypes.data, len(pr["code"]), pr["params"].ctypes.data, pr["n_slots"], pr["input_slots"].ctypes.data,
            vals.shape[1], vals.ctypes.data, k, pr["wire_slot"].ctypes.data, pr["wire_index"].ctypes.data, len(pr["wire_slot"]), nw,
            self.n, wires.ctypes.data, pr["pi_slots"].ctypes.data, npi, pis.ctypes.data, status.ctypes.data, err, threads)
        if rc != 0:
            raise ValueError("zklc_plonky2_witness_run: invalid argument")
        for i in range(k):
            if status[i]:
                raise AssertionError("witness %d: %s" % (i, err.raw[200 * i:200 * i + 200].split(b"\0")[0].decode()))
        return wires, pis[:, :npi]

    def device_witness(self, ctx):
        """the witness program resident on `ctx`'s GPU (zklc_plonky2_witness_program_create): `DeviceWitness.run` generates a
        batch of witnesses straight into HBM (csrc/plonky2_witness_dev.hip)"""
        assert self._program is not None, "call witness_program(example_inputs) first"
        return DeviceWitness(ctx, self)

    def prover(self, ctx, hasher=0):
        """upload + preprocess the circuit on `ctx`'s GPU (zklc_plonky2_circuit_create)"""
        from .prover import Prover
        return Prover(ctx, self, hasher)

    def common_data(self):
        """same schema as the reference's common_data.json (near_bft_finality/src/bin/prove_block.rs:320-458 writes it)"""
        cfg = self.config
        return {
            "config": cfg,
            "fri_params": {"config": cfg["fri_config"], "hiding": False, "degree_bits": self.degree_bits,
                           "reduction_arity_bits": self.fri_arity_bits},
            "gates": [g.id() for g in self.gates],
            "selectors_info": {"selector_indices": self.selector_indices,
                               "groups": [{"start": s, "end": e} for s, e in self.groups]},
            "quotient_degree_factor": self.quotient_degree_factor,
            "num_gate_constraints": self.num_gate_constraints,
            "num_constants": self.num_constants,
            "num_public_inputs": self.num_public_inputs,
            "k_is": self.k_is,
            "num_partial_products": self.num_partial_products,
            "num_lookup_polys": 0, "num_lookup_selectors": 0, "luts": [],
        }

    def _ensure_plan(self):
        if self._plan is None:
            # the copy classes are final once the circuit is built: resolve every target to its class once
            b = self.builder
            rep = {}

            def find(t):
                k = t.k
                r = rep.get(k)
                if r is None:
                    r = rep[k] = b._find(k)
                return r
            self._plan = ([([find(t) for t in ins], fn, gi) for gi, (ins, fn, _, _, _) in enumerate(b.generators)], find)

    # ---- witness generation (plonky2 `generate_partial_witness`): run generators until no progress
    def generate_witness(self, inputs):
        """inputs: {Target: value}.  Returns (wires u64[num_wires, n], public input values)."""
        b = self.builder
        self._ensure_plan()
        plan, find = self._plan
        vals = {}

        def setv(t, v):
            k = find(t)
            v %= P
            old = vals.get(k)
            if old is None:
                vals[k] = v
            elif old != v:
                raise AssertionError("copy constraint violated at %r: %d != %d" % (t, old, v))

        for t, v in inputs.items():
            setv(t, v)
        pending = plan
        trace = self._trace
        while pending:
            rest = []
            for ks, fn, gi in pending:
                try:
                    args = [vals[k] for k in ks]
                except KeyError:
                    rest.append((ks, fn, gi))
                    continue
                outs = fn(args)
                if trace is not None:
                    trace.append((gi, [find(t) for t, _ in outs]))
                for t, v in outs:
                    setv(t, v)
            assert len(rest) < len(pending), "witness generation stuck: %d generators without inputs" % len(rest)
            pending = rest
        wires = np.zeros((self.config["num_wires"], self.n), dtype=np.uint64)
        for k, v in vals.items():
            if k < VIRTUAL_BASE:
                wires[k & 255, k >> 8] = v
        for k in b.parent:
            if k < VIRTUAL_BASE:
                r = b._find(k)
                if r in vals:
                    wires[k & 255, k >> 8] = vals[r]
        pis = [vals[find(t)] for t in b.public_inputs]
        return wires, pis


class DeviceWitness:
    """A circuit's generator program on one GPU: witnesses are produced in HBM in the prover's layout, up to 64 per call.
    The reference runs these generators on the CPU inside `CircuitData::prove` (prove_crypto/ed25519.rs:60,100)."""

    def __init__(self, ctx, data):
        import ctypes
        from .. import _lib
        self.ctx, self.data, self._lib = ctx, data, _lib.load()
        pr = data._program
        self.n_inputs, self.n_pi = len(pr["input_slots"]), len(pr["pi_slots"])
        self.num_wires, self.n_rows = data.config["num_wires"], data.n
        h = ctypes.c_void_p()
        if data._container is not None:      # loaded from a circuit container: the library reads the program's sections itself
            rc = self._lib.zklc_plonky2_witness_program_create_from_container(ctx._h, data._container._h, ctypes.byref(h))
            data._container.release_pages()      # program tables uploaded: the mapping stays valid, its pages are given back
        else:
            rc = self._lib.zklc_plonky2_witness_program_create(
                ctx._h, pr["code"].ctypes.data, len(pr["code"]), pr["params"].ctypes.data, len(pr["params"]), pr["n_slots"],
                pr["input_slots"].ctypes.data, self.n_inputs, pr["wire_slot"].ctypes.data, pr["wire_index"].ctypes.data, len(pr["wire_slot"]),
                self.num_wires, self.n_rows, pr["pi_slots"].ctypes.data, self.n_pi, ctypes.byref(h))
        ctx._check(rc)
        self._h = h

    def info(self, n_witnesses=1):
        import ctypes
        a, b, c = ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_uint32()
        self.ctx._check(self._lib.zklc_plonky2_witness_program_info(self._h, n_witnesses, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"instructions": a.value, "levels": b.value, "launches": c.value}

    def input_matrix(self, inputs_list):
        pr = self.data._program
        return np.array([[int(w[t]) for t in pr["input_targets"]] for w in inputs_list], dtype=np.uint64).reshape(len(inputs_list), -1)

    def run(self, d_wires_ptr, inputs_list=None, input_values=None, stream=None, capacity=64):
        """d_wires_ptr: device address of a ZERO-INITIALISED uint64 [capacity, num_wires, n] buffer (reusable: the same cells are
        written every time); `capacity` = the witnesses the buffer holds, a larger batch raises instead of writing past it.
        Returns the public inputs uint64 [k, n_pi]; raises AssertionError if a witness does not exist."""
        import ctypes
        vals = self.input_matrix(inputs_list) if input_values is None else np.ascontiguousarray(input_values, dtype=np.uint64).reshape(-1, self.n_inputs)
        k = vals.shape[0]
        assert vals.shape[1] == self.n_inputs
        if not 1 <= k <= min(64, int(capacity)):
            raise ValueError("witness batch of %d for a buffer of %d witnesses (<= 64 per call)" % (k, min(64, int(capacity))))
        pis = np.zeros((k, max(self.n_pi, 1)), dtype=np.uint64)
        status = np.zeros(k, dtype=np.int32)
        err = ctypes.create_string_buffer(200 * k)
        if stream is None:
            stream = self.ctx.stream_ptr()
        rc = self._lib.zklc_plonky2_witness_run_dev(self.ctx._h, stream, self._h, vals.ctypes.data, k, d_wires_ptr, pis.ctypes.data,
                                                    status.ctypes.data, err)
        self.ctx._check(rc)
        for i in range(k):
            if status[i]:
                raise AssertionError("witness %d: %s" % (i, err.raw[200 * i:200 * i + 200].split(b"\0")[0].decode()))
        return pis[:, :self.n_pi]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.zklc_plonky2_witness_program_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
