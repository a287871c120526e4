"""Host-side mirror of the small comparison circuits of the reference.

Reference: near_bft_finality/src/prove_block_data/primitives.rs
  prove_consecutive_heights  :32-124   two 8-byte little-endian heights with height1 = height2 + 1
  prove_eq_array             :126-174  two byte arrays are equal
  two_thirds                 :176-334  3 * value1 >= 2 * value2 on 17-byte little-endian values (stake sums)
Each returns the circuit and a witness map; `PrimitiveProver` proves them on the GPU like the reference functions do
(`data.prove(pw)`).  The byte-wise "is negative" test of the reference -- bytes 1..7 of a field difference compared with the
pattern of a small negative number -- is restated as written (`upper_bytes_match`), it is shared with keys_stakes.py.
"""
from .plonky2.recursion import RecursiveCircuitBuilder

BLOCK_HEIGHT_BYTES, STAKE_BYTES = 8, 16
CONSTANT1, CONSTANT2 = 0xFFFFFFFEFFFFFF00, 0xFFFFFFFF00000000


def upper_bytes_match(b, sub):
    """primitives.rs:206-218 / keys_stakes.rs:119-131: how many of the bytes 1..7 of `sub` equal those of p - 2^32 - 256 (of
    p - 2^32 when sub == -1): 7 exactly when sub is a small negative number"""
    chs = b.select(b.is_equal(sub, b.neg_one()), b.constant(CONSTANT2), b.constant(CONSTANT1))
    cb, sb = b.split_le_63(chs, 64), b.split_le_63(sub, 64)
    s = b.zero()
    for j in range(8, 64, 8):
        s = b.add(s, b.is_equal(b.le_sum_small(cb[j:j + 8]), b.le_sum_small(sb[j:j + 8])))
    return s


def byte_and_carry(b, t):
    bits = b.split_le_63(t, 64)
    return b.le_sum_small(bits[0:8]), b.le_sum_small(bits[8:16])


def times_small(b, limbs, k):
    """little-endian bytes times a small constant with byte carries: len(limbs) + 1 bytes (primitives.rs:236-256)"""
    out, crr = [], b.zero()
    kt = b.constant(k)
    for x in limbs:
        lo, crr = byte_and_carry(b, b.mul_add(x, kt, crr))
        out.append(lo)
    return out + [crr]


def select_not_smaller(b, lhs, rhs, count):
    """the loop of primitives.rs:257-305 / keys_stakes.rs:184-228 over the top `count` bytes, most significant first: res[i] is
    rhs[i] once lhs is known to be smaller, lhs[i] otherwise; the caller connects res to lhs, which rules "smaller" out"""
    seven, zero = b.constant(7), b.zero()
    res = [None] * count
    prev = (zero, zero)
    for i in range(count - 1, -1, -1):
        is_equal = b.is_equal(lhs[i], rhs[i])
        s = upper_bytes_match(b, b.sub(lhs[i], rhs[i]))
        is_negative = b.is_equal(s, seven)
        b.connect(s, b.select(is_negative, seven, zero))
        if i == count - 1:
            res[i] = b.select(is_negative, rhs[i], lhs[i])
            prev = (is_equal, is_negative)
        else:
            q = b.is_equal(prev[0], prev[1])
            prev = (b.select(q, prev[0], is_equal), b.select(q, prev[1], is_negative))
            res[i] = b.select(prev[1], rhs[i], lhs[i])
    return res


def two_thirds_circuit():
    """primitives.rs:176-334: public inputs = the 17 bytes of value1"""
    b = RecursiveCircuitBuilder()
    n = STAKE_BYTES + 1
    v1, v2 = b.add_virtual_targets(n), b.add_virtual_targets(n)
    seven, h = b.constant(7), b.constant(100)
    b.connect(upper_bytes_match(b, b.sub(v1[n - 1], h)), seven)
    b.connect(upper_bytes_match(b, b.sub(v2[n - 1], h)), seven)
    three_v1, two_v2 = times_small(b, v1, 3), times_small(b, v2, 2)
    res = select_not_smaller(b, three_v1, two_v2, n)
    for x, y in zip(three_v1, res):
        b.connect(x, y)
    for t in v1:
        b.register_public_input(t)
    return b.build(), v1, v2


def consecutive_heights_circuit():
    """primitives.rs:32-124: public inputs = height1 bytes then height2 bytes"""
    b = RecursiveCircuitBuilder()
    n = BLOCK_HEIGHT_BYTES
    h1, h2 = b.add_virtual_targets(n), b.add_virtual_targets(n)
    zero, one, tff = b.zero(), b.one(), b.constant(255)
    total, prev = zero, zero
    for i in range(n - 1, -1, -1):
        if i != n - 1:
            b.connect(b.sub(h1[i], b.select(prev, zero, h1[i])), zero)
            b.connect(b.sub(h2[i], b.select(prev, tff, h2[i])), zero)
        dif = b.select(b.is_equal(b.sub(h1[i], h2[i]), one), one, zero)
        prev = dif
        total = b.add(total, dif)
    b.connect(total, one)
    for t in h1 + h2:
        b.register_public_input(t)
    return b.build(), h1, h2


def eq_array_circuit(n):
    """primitives.rs:126-174: public inputs = array1"""
    b = RecursiveCircuitBuilder()
    a1, a2 = b.add_virtual_targets(n), b.add_virtual_targets(n)
    for x, y in zip(a1, a2):
        b.connect(x, y)
    for t in a1:
        b.register_public_input(t)
    return b.build(), a1, a2


class PrimitiveProver:
    """the three functions with the reference's signatures: byte strings in, (common, verifier_only, proof) out"""

    def __init__(self, ctx):
        self.ctx = ctx
        self._cache = {}

    def _prove(self, key, build, values):
        from .plonky2 import HASH_GL
        ent = self._cache.get(key)
        if ent is None:
            from .plonky2.circuit_cache import load_or_build

            def build_():
                data, t1, t2 = build()
                data.witness_program(t1 + t2)
                return data, {"t": t1 + t2}
            data, aux, _ = load_or_build("primitive", key, build_)
            prover = data.prover(self.ctx, HASH_GL)
            ent = self._cache[key] = (data, aux["t"], prover, data.common_data(), prover.verifier_data())
        data, targets, prover, common, vd = ent
        wires, pis = data.generate_witness_native([dict(zip(targets, values))])
        return common, vd, prover.prove(wires[0], [int(x) for x in pis[0]])

    def two_thirds(self, value1, value2):
        assert len(value1) == len(value2) == STAKE_BYTES + 1
        return self._prove("two_thirds", two_thirds_circuit, list(bytes(value1)) + list(bytes(value2)))

    def prove_consecutive_heights(self, height1, height2):
        assert len(height1) == len(height2) == BLOCK_HEIGHT_BYTES
        return self._prove("heights", consecutive_heights_circuit, list(bytes(height1)) + list(bytes(height2)))

    def prove_eq_array(self, array1, array2):
        assert len(array1) == len(array2)
        return self._prove(("eq", len(array1)), lambda: eq_array_circuit(len(array1)), list(bytes(array1)) + list(bytes(array2)))

    def close(self):
        for ent in self._cache.values():
            ent[2].close()
        self._cache = {}
