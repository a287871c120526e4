"""Gate descriptors of the plonky2 circuits on the signature-aggregation path (host side).

A descriptor carries what the circuit builder and the GPU prover need to know about a gate type:
its id string (the exact `Gate::id()` text that appears in the reference's common_data.json, e.g.
near_bft_finality/proofs/random/CGZP.../common_data.json "gates"), wire/constant/constraint counts,
constraint degree, and the (type code, parameters) pair handed to the C ABI
(include/zklc.h `zklc_plonky2_gate`).  The constraint evaluators themselves are HIP device code
(csrc/plonky2_gates.cuh); nothing here evaluates a constraint.

Standard gates: plonky2-near@2244a9d `plonky2/src/gates/*` (un-vendored; layouts as restated in
gnark-plonky2-verifier/plonk/gates/*.go).  u32 gates: crypto/plonky2_u32/src/gates/*.rs (in-tree).
"""
import re

# type codes shared with csrc/plonky2_gates.cuh
NOOP, CONSTANT, PUBLIC_INPUT, ARITHMETIC, ARITHMETIC_EXT, MUL_EXT, BASE_SUM, POSEIDON, POSEIDON_MDS, RANDOM_ACCESS, \
    REDUCING, REDUCING_EXT, EXPONENTIATION, COSET_INTERPOLATION, U32_ARITHMETIC, U32_ADD_MANY, U32_SUBTRACTION, \
    U32_RANGE_CHECK, COMPARISON, U32_INTERLEAVE, UNINTERLEAVE_TO_U32, UNINTERLEAVE_TO_B32 = range(22)

_PH = "PhantomData<plonky2_field::goldilocks_field::GoldilocksField>"


class Gate:
    num_constants = 0
    params = (0, 0, 0, 0)
    extra = ()          # u64 table handed to the device (CosetInterpolationGate weights)

    def id(self):
        raise NotImplementedError

    def extra_constant_wires(self):
        """plonky2 `Gate::extra_constant_wires`: (constant index, routed wire) pairs through which a row of this gate can
        serve `CircuitBuilder::constant` targets (the wire is constrained to equal the row's constant)"""
        return []

    def __eq__(self, o):
        return isinstance(o, Gate) and self.id() == o.id()

    def __hash__(self):
        return hash(self.id())

    def __repr__(self):
        return self.id()


class NoopGate(Gate):
    code, degree, num_constraints, num_wires = NOOP, 0, 0, 0

    def id(self):
        return "NoopGate"


class ConstantGate(Gate):
    code, degree = CONSTANT, 1

    def __init__(self, num_consts):
        self.num_consts = self.num_constants = self.num_constraints = self.num_wires = num_consts
        self.params = (num_consts, 0, 0, 0)

    def id(self):
        return "ConstantGate { num_consts: %d }" % self.num_consts

    def extra_constant_wires(self):
        return [(i, i) for i in range(self.num_consts)]


class PublicInputGate(Gate):
    code, degree, num_constraints, num_wires = PUBLIC_INPUT, 1, 4, 4

    def id(self):
        return "PublicInputGate"


class ArithmeticGate(Gate):
    """num_ops x (m0, m1, addend, out): out = c0*m0*m1 + c1*addend (arithmetic_gate.go:48-84)"""
    code, degree, num_constants = ARITHMETIC, 3, 2

    def __init__(self, num_ops):
        self.num_ops = self.num_constraints = num_ops
        self.num_wires = 4 * num_ops
        self.params = (num_ops, 0, 0, 0)

    @staticmethod
    def new_from_config(cfg):
        return ArithmeticGate(cfg["num_routed_wires"] // 4)

    def id(self):
        return "ArithmeticGate { num_ops: %d }" % self.num_ops


class ArithmeticExtensionGate(Gate):
    code, degree, num_constants = ARITHMETIC_EXT, 3, 2

    def __init__(self, num_ops):
        self.num_ops = num_ops
        self.num_constraints = 2 * num_ops
        self.num_wires = 8 * num_ops
        self.params = (num_ops, 0, 0, 0)

    @staticmethod
    def new_from_config(cfg):
        return ArithmeticExtensionGate(cfg["num_routed_wires"] // 8)

    def id(self):
        return "ArithmeticExtensionGate { num_ops: %d }" % self.num_ops


class MulExtensionGate(Gate):
    code, degree, num_constants = MUL_EXT, 3, 1

    def __init__(self, num_ops):
        self.num_ops = num_ops
        self.num_constraints = 2 * num_ops
        self.num_wires = 6 * num_ops
        self.params = (num_ops, 0, 0, 0)

    @staticmethod
    def new_from_config(cfg):
        return MulExtensionGate(cfg["num_routed_wires"] // 6)

    def id(self):
        return "MulExtensionGate { num_ops: %d }" % self.num_ops


class BaseSumGate(Gate):
    """wire 0 = sum, wires 1..=num_limbs = limbs in base B (base_sum_gate.go:66-96)"""
    code = BASE_SUM

    def __init__(self, num_limbs, base=2):
        self.num_limbs, self.base = num_limbs, base
        self.degree = base
        self.num_constraints = 1 + num_limbs
        self.num_wires = 1 + num_limbs
        self.params = (num_limbs, base, 0, 0)

    def id(self):
        return "BaseSumGate { num_limbs: %d } + Base: %d" % (self.num_limbs, self.base)


class PoseidonGate(Gate):
    """poseidon_gate.go:27-82: inputs 0..12, outputs 12..24, swap 24, delta 25..29, S-box inputs 29..135"""
    code, degree, num_constraints, num_wires = POSEIDON, 7, 123, 135
    WIRE_SWAP, START_DELTA, START_FULL_0, START_PARTIAL, START_FULL_1 = 24, 25, 29, 65, 87

    def id(self):
        return "PoseidonGate(%s)<WIDTH=12>" % _PH


class PoseidonMdsGate(Gate):
    code, degree, num_constraints, num_wires = POSEIDON_MDS, 1, 24, 48

    def id(self):
        return "PoseidonMdsGate(%s)<WIDTH=12>" % _PH


class RandomAccessGate(Gate):
    code = RANDOM_ACCESS

    def __init__(self, bits, num_copies, num_extra_constants):
        self.bits, self.num_copies, self.num_extra_constants = bits, num_copies, num_extra_constants
        self.num_constants = num_extra_constants
        self.degree = bits + 1
        self.num_constraints = num_copies * (bits + 2) + num_extra_constants
        self.num_routed = (2 + (1 << bits)) * num_copies + num_extra_constants
        self.num_wires = self.num_routed + bits * num_copies
        self.params = (bits, num_copies, num_extra_constants, 0)

    @staticmethod
    def new_from_config(cfg, bits):
        vec = 1 << bits
        max_copies = min(cfg["num_routed_wires"] // (2 + vec), cfg["num_wires"] // (2 + vec + bits))
        extra = min(cfg["num_routed_wires"] - (2 + vec) * max_copies, cfg["num_constants"])
        return RandomAccessGate(bits, max_copies, extra)

    def id(self):
        return "RandomAccessGate { bits: %d, num_copies: %d, num_extra_constants: %d, _phantom: %s }<D=2>" % (
            self.bits, self.num_copies, self.num_extra_constants, _PH)

    def extra_constant_wires(self):
        base = (2 + (1 << self.bits)) * self.num_copies        # `wire_extra_constant(i)`: after the copies' routed wires
        return [(i, base + i) for i in range(self.num_extra_constants)]


class ReducingGate(Gate):
    code, degree = REDUCING, 2

    def __init__(self, num_coeffs):
        self.num_coeffs = num_coeffs
        self.num_constraints = 2 * num_coeffs
        self.num_wires = 6 + num_coeffs + 2 * (num_coeffs - 1)
        self.params = (num_coeffs, 0, 0, 0)

    def id(self):
        return "ReducingGate { num_coeffs: %d }" % self.num_coeffs


class ReducingExtensionGate(Gate):
    code, degree = REDUCING_EXT, 2

    def __init__(self, num_coeffs):
        self.num_coeffs = num_coeffs
        self.num_constraints = 2 * num_coeffs
        self.num_wires = 6 + 2 * num_coeffs + 2 * (num_coeffs - 1)
        self.params = (num_coeffs, 0, 0, 0)

    def id(self):
        return "ReducingExtensionGate { num_coeffs: %d }" % self.num_coeffs


class ExponentiationGate(Gate):
    code, degree = EXPONENTIATION, 4

    def __init__(self, num_power_bits):
        self.num_power_bits = num_power_bits
        self.num_constraints = num_power_bits + 1
        self.num_wires = 2 + 2 * num_power_bits
        self.params = (num_power_bits, 0, 0, 0)

    @staticmethod
    def new_from_config(cfg):
        return ExponentiationGate(min(cfg["num_routed_wires"] - 2, (cfg["num_wires"] - 2) // 2))

    def id(self):
        return "ExponentiationGate { num_power_bits: %d, _phantom: %s }<D=2>" % (self.num_power_bits, _PH)


class CosetInterpolationGate(Gate):
    code = COSET_INTERPOLATION

    def __init__(self, subgroup_bits, degree, barycentric_weights):
        self.subgroup_bits, self.degree, self.weights = subgroup_bits, degree, list(barycentric_weights)
        n_pts = 1 << subgroup_bits
        self.num_intermediates = (n_pts - 2) // (degree - 1)
        self.num_constraints = 4 + 4 * self.num_intermediates
        self.num_wires = 1 + 2 * n_pts + 4 + 4 * self.num_intermediates + 2
        self.params = (subgroup_bits, degree, 0, 0)
        self.extra = tuple(self.weights)

    def id(self):
        return "CosetInterpolationGate { subgroup_bits: %d, degree: %d, barycentric_weights: [%s], _phantom: %s }<D=2>" % (
            self.subgroup_bits, self.degree, ", ".join(str(w) for w in self.weights), _PH)


class U32ArithmeticGate(Gate):
    """crypto/plonky2_u32/src/gates/arithmetic_u32.rs:36-94: per op 6 routed wires (m0, m1, addend, lo, hi, inverse)
    then 32 two-bit limbs per op after all routed wires"""
    code, degree = U32_ARITHMETIC, 4

    def __init__(self, num_ops):
        self.num_ops = num_ops
        self.num_constraints = num_ops * 36
        self.num_wires = num_ops * 38
        self.params = (num_ops, 0, 0, 0)

    @staticmethod
    def new_from_config(cfg):
        return U32ArithmeticGate(min(cfg["num_wires"] // 38, cfg["num_routed_wires"] // 6))

    def id(self):
        return "U32ArithmeticGate { num_ops: %d, _phantom: %s }" % (self.num_ops, _PH)


class U32AddManyGate(Gate):
    """crypto/plonky2_u32/src/gates/add_many_u32.rs: per op num_addends addends, carry in, result, carry out (routed),
    then 16 result + 2 carry two-bit limbs per op after all routed wires"""
    code, degree = U32_ADD_MANY, 4

    def __init__(self, num_addends, num_ops):
        self.num_addends, self.num_ops = num_addends, num_ops
        self.num_constraints = num_ops * (3 + 18)
        self.num_wires = num_ops * (num_addends + 3 + 18)
        self.params = (num_addends, num_ops, 0, 0)

    @staticmethod
    def new_from_config(cfg, num_addends):
        per, routed = num_addends + 3 + 18, num_addends + 3
        return U32AddManyGate(num_addends, min(cfg["num_wires"] // per, cfg["num_routed_wires"] // routed))

    def id(self):
        return "U32AddManyGate { num_addends: %d, num_ops: %d, _phantom: %s }" % (self.num_addends, self.num_ops, _PH)


class U32SubtractionGate(Gate):
    """subtraction_u32.rs: per op (x, y, borrow_in, result, borrow_out) routed, then 16 two-bit limbs of the result"""
    code, degree = U32_SUBTRACTION, 4

    def __init__(self, num_ops):
        self.num_ops = num_ops
        self.num_constraints = num_ops * (3 + 16)
        self.num_wires = num_ops * (5 + 16)
        self.params = (num_ops, 0, 0, 0)

    @staticmethod
    def new_from_config(cfg):
        return U32SubtractionGate(min(cfg["num_wires"] // 21, cfg["num_routed_wires"] // 5))

    def id(self):
        return "U32SubtractionGate { num_ops: %d, _phantom: %s }" % (self.num_ops, _PH)


class U32RangeCheckGate(Gate):
    """range_check_u32.rs: num_input_limbs inputs, each followed (after all inputs) by 16 two-bit aux limbs"""
    code, degree = U32_RANGE_CHECK, 4

    def __init__(self, num_input_limbs):
        self.num_input_limbs = num_input_limbs
        self.num_constraints = num_input_limbs * 17
        self.num_wires = num_input_limbs * 17
        self.params = (num_input_limbs, 0, 0, 0)

    def id(self):
        return "U32RangeCheckGate { num_input_limbs: %d, _phantom: %s }" % (self.num_input_limbs, _PH)


class ComparisonGate(Gate):
    """comparison.rs:36-80: first <= second on num_bits-bit values split into num_chunks chunks"""
    code = COMPARISON

    def __init__(self, num_bits, num_chunks):
        self.num_bits, self.num_chunks = num_bits, num_chunks
        self.chunk_bits = -(-num_bits // num_chunks)
        self.degree = 1 << self.chunk_bits
        self.num_constraints = 6 + 5 * num_chunks + self.chunk_bits
        self.num_wires = 4 + 5 * num_chunks + self.chunk_bits + 1
        self.params = (num_bits, num_chunks, 0, 0)

    def id(self):
        return "ComparisonGate { num_bits: %d, num_chunks: %d, _phantom: %s }<D=2>" % (self.num_bits, self.num_chunks, _PH)


class U32InterleaveGate(Gate):
    """crypto/plonky2_u32/src/gates/interleave_u32.rs:37-95: per op (x, x_interleaved) routed, then 32 big-endian bits per op"""
    code, degree = U32_INTERLEAVE, 2

    def __init__(self, num_ops):
        self.num_ops = num_ops
        self.num_constraints = num_ops * 34
        self.num_wires = num_ops * 34
        self.params = (num_ops, 0, 0, 0)

    @staticmethod
    def new_from_config(cfg):
        return U32InterleaveGate(min(cfg["num_wires"] // 34, cfg["num_routed_wires"] // 2))

    def id(self):
        return "U32InterleaveGate { num_ops: %d }" % self.num_ops


class UninterleaveToU32Gate(Gate):
    """uninterleave_to_u32.rs:32-94: per op (x_interleaved, evens, odds) routed, then 64 big-endian bits per op"""
    code, degree, NAME = UNINTERLEAVE_TO_U32, 2, "UninterleaveToU32Gate"

    def __init__(self, num_ops):
        self.num_ops = num_ops
        self.num_constraints = num_ops * 67
        self.num_wires = num_ops * 67
        self.params = (num_ops, 0, 0, 0)

    @classmethod
    def new_from_config(cls, cfg):
        return cls(min(cfg["num_wires"] // 67, cfg["num_routed_wires"] // 3))

    def id(self):
        return "%s { num_ops: %d }" % (self.NAME, self.num_ops)


class UninterleaveToB32Gate(UninterleaveToU32Gate):
    """uninterleave_to_b32.rs: same layout, the two halves stay in interleaved (base-4) form"""
    code, NAME = UNINTERLEAVE_TO_B32, "UninterleaveToB32Gate"


_PATTERNS = [
    (r"^NoopGate", lambda m: NoopGate()),
    (r"^ConstantGate \{ num_consts: (\d+) \}", lambda m: ConstantGate(int(m[1]))),
    (r"^PublicInputGate", lambda m: PublicInputGate()),
    (r"^ArithmeticGate \{ num_ops: (\d+) \}", lambda m: ArithmeticGate(int(m[1]))),
    (r"^ArithmeticExtensionGate \{ num_ops: (\d+) \}", lambda m: ArithmeticExtensionGate(int(m[1]))),
    (r"^MulExtensionGate \{ num_ops: (\d+) \}", lambda m: MulExtensionGate(int(m[1]))),
    (r"^BaseSumGate \{ num_limbs: (\d+) \} \+ Base: (\d+)", lambda m: BaseSumGate(int(m[1]), int(m[2]))),
    (r"^PoseidonGate", lambda m: PoseidonGate()),
    (r"^PoseidonMdsGate", lambda m: PoseidonMdsGate()),
    (r"^RandomAccessGate \{ bits: (\d+), num_copies: (\d+), num_extra_constants: (\d+)",
     lambda m: RandomAccessGate(int(m[1]), int(m[2]), int(m[3]))),
    (r"^ReducingGate \{ num_coeffs: (\d+) \}", lambda m: ReducingGate(int(m[1]))),
    (r"^ReducingExtensionGate \{ num_coeffs: (\d+) \}", lambda m: ReducingExtensionGate(int(m[1]))),
    (r"^ExponentiationGate \{ num_power_bits: (\d+)", lambda m: ExponentiationGate(int(m[1]))),
    (r"^CosetInterpolationGate \{ subgroup_bits: (\d+), degree: (\d+), barycentric_weights: \[([0-9, ]+)\]",
     lambda m: CosetInterpolationGate(int(m[1]), int(m[2]), [int(x) for x in m[3].split(",")])),
    (r"^U32ArithmeticGate \{ num_ops: (\d+)", lambda m: U32ArithmeticGate(int(m[1]))),
    (r"^U32AddManyGate \{ num_addends: (\d+), num_ops: (\d+)", lambda m: U32AddManyGate(int(m[1]), int(m[2]))),
    (r"^U32SubtractionGate \{ num_ops: (\d+)", lambda m: U32SubtractionGate(int(m[1]))),
    (r"^U32RangeCheckGate \{ num_input_limbs: (\d+)", lambda m: U32RangeCheckGate(int(m[1]))),
    (r"^ComparisonGate \{ num_bits: (\d+), num_chunks: (\d+)", lambda m: ComparisonGate(int(m[1]), int(m[2]))),
    (r"^U32InterleaveGate \{ num_ops: (\d+)", lambda m: U32InterleaveGate(int(m[1]))),
    (r"^UninterleaveToU32Gate \{ num_ops: (\d+)", lambda m: UninterleaveToU32Gate(int(m[1]))),
    (r"^UninterleaveToB32Gate \{ num_ops: (\d+)", lambda m: UninterleaveToB32Gate(int(m[1]))),
]


def gate_from_id(gid):
    """Parse a `Gate::id()` string as written to common_data.json (gates.go:27-54 does the same with regexes)."""
    for rx, mk in _PATTERNS:
        m = re.match(rx, gid)
        if m:
            return mk(m)
    raise ValueError("unknown gate id: " + gid)
