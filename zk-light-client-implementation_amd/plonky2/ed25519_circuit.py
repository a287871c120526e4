"""The per-signature Ed25519 circuit of the reference, restated gadget by gadget on zklc_amd.plonky2.CircuitBuilder:

  ed25519_circuit / fill_ecdsa_targets   crypto/plonky2_ed25519/src/gadgets/eddsa.rs:34-85, 87-117
  BigUint gadgets                        crypto/plonky2_ecdsa/src/gadgets/biguint.rs:88-372 (+ DivRem generator :417-470)
  non-native field gadgets + generators  crypto/plonky2_ed25519/src/gadgets/nonnative.rs:120-445, 447-705
  curve gadgets, point (de)compression   crypto/plonky2_ed25519/src/gadgets/curve.rs:107-324, 327-370
  4-bit windowed variable-base mul       crypto/plonky2_ed25519/src/gadgets/curve_windowed_mul.rs:38-149
  fixed-base mul                         crypto/plonky2_ed25519/src/gadgets/curve_fixed_base.rs:16-64
  nibble splitting                       crypto/plonky2_ed25519/src/gadgets/split_nonnative.rs:36-66
  curve / field parameters               crypto/plonky2_ed25519/src/curve/ed25519.rs:19-51, field/ed25519_{base,scalar}.rs
It is built with `wide_ecc_config()` (near_bft_finality/src/prove_crypto/ed25519.rs:29) and proves
[s]B == R + [h]A with h = SHA-512(R || A || M) mod l in affine twisted-Edwards coordinates over non-native 2^255 - 19
arithmetic on u32 limbs.  Public inputs: the message bits then the public-key bits (eddsa.rs:45-56).

Two deliberate differences from the Rust builder, neither of which changes the statement:
  * the blinding points that the reference draws with `C::ScalarField::rand()` at circuit-construction time
    (curve_windowed_mul.rs:47; SURVEY fact 7: the reference circuit is not reproducible across builds) come from a seeded
    generator here, so the circuit (and its digest) is deterministic;
  * the "nothing-up-my-sleeve" starting points derived from Keccak-256 of a zero field element
    (curve_windowed_mul.rs:115-121, curve_fixed_base.rs:30-34) use SHA3-256 (hashlib has no legacy Keccak padding).
Witness generation is the host-side generators below (Python big integers), not yet the GPU.
"""
import hashlib
import random

from . import sha512
from .builder import OP_NN_ADD, OP_NN_SUB, OP_NN_MUL, OP_NN_INV, OP_DIV_REM, OP_DECOMPRESS

P25519 = 2**255 - 19
L25519 = 2**252 + 27742317777372353535851937790883648493
D25519 = (-121665 * pow(121666, P25519 - 2, P25519)) % P25519
BASE_Y = 4 * pow(5, P25519 - 2, P25519) % P25519
WINDOW = 4


def _recover_x(y, sign):
    xx = (y * y - 1) * pow(D25519 * y * y + 1, P25519 - 2, P25519) % P25519
    x = pow(xx, (P25519 + 3) // 8, P25519)
    if (x * x - xx) % P25519:
        x = x * pow(2, (P25519 - 1) // 4, P25519) % P25519
    if (x * x - xx) % P25519:
        raise ValueError("not a curve point")
    if (x & 1) != sign:
        x = P25519 - x
    return x


BASE = (_recover_x(BASE_Y, 0), BASE_Y)


# ---- native curve arithmetic for the constants (curve/curve_adds.rs, affine twisted Edwards, a = -1)
def pt_add(p, q):
    x1, y1 = p
    x2, y2 = q
    t = D25519 * x1 * x2 * y1 * y2 % P25519
    x3 = (x1 * y2 + y1 * x2) * pow(1 + t, P25519 - 2, P25519) % P25519
    y3 = (y1 * y2 + x1 * x2) * pow(1 - t, P25519 - 2, P25519) % P25519
    return (x3, y3)


def pt_mul(k, p):
    r = (0, 1)
    while k:
        if k & 1:
            r = pt_add(r, p)
        p = pt_add(p, p)
        k >>= 1
    return r


def pt_neg(p):
    return ((-p[0]) % P25519, p[1])


def limbs_of(v, n=None):
    out = []
    while v:
        out.append(v & 0xFFFFFFFF)
        v >>= 32
    if n is not None:
        out += [0] * (n - len(out))
    return out


def value_of(vals):
    return sum(int(x) << (32 * i) for i, x in enumerate(vals))


class Gadgets:
    """BigUintTarget = list of u32 limb targets (little-endian); NonNativeTarget = the same list (8 limbs)."""

    def __init__(self, builder, seed=0):
        self.b = builder
        self.rng = random.Random(seed)

    # ------------------------------------------------------------------ BigUint (biguint.rs)
    def constant_biguint(self, v):
        return [self.b.constant(l) for l in limbs_of(v)]

    def virtual_biguint(self, n):
        return self.b.add_virtual_targets(n)

    def connect_biguint(self, lhs, rhs):
        m = min(len(lhs), len(rhs))
        for x, y in zip(lhs[:m], rhs[:m]):
            self.b.connect(x, y)
        for x in lhs[m:] + rhs[m:]:
            self.b.assert_zero(x)

    def pad(self, a, b):
        n = max(len(a), len(b))
        z = self.b.zero()
        return a + [z] * (n - len(a)), b + [z] * (n - len(b))

    def cmp_biguint(self, a, b):
        a, b = self.pad(a, b)
        return self.b.list_le(a, b, 32)

    def add_biguint(self, a, b):
        n = max(len(a), len(b))
        z = self.b.zero()
        out, carry = [], z
        for i in range(n):
            lo, carry = self.b.add_many_u32([carry, a[i] if i < len(a) else z, b[i] if i < len(b) else z])
            out.append(lo)
        return out + [carry]

    def sub_biguint(self, a, b):
        a, b = self.pad(a, b)
        out, borrow = [], self.b.zero()
        for x, y in zip(a, b):
            r, borrow = self.b.sub_u32(x, y, borrow)
            out.append(r)
        return out

    def mul_biguint(self, a, b):
        total = len(a) + len(b)
        to_add = [[] for _ in range(total)]
        z = self.b.zero()
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                lo, hi = self.b.mul_add_u32(x, y, z)
                to_add[i + j].append(lo)
                to_add[i + j + 1].append(hi)
        out, carry = [], z
        for col in to_add:
            if not col:                       # only when one operand has no limbs
                out.append(carry)
                carry = z
                continue
            r, carry = self.b.add_u32s_with_carry(col, carry)
            out.append(r)
        return out + [carry]

    def mul_biguint_by_bool(self, a, bit):
        return [self.b.mul(l, bit) for l in a]

    def div_rem_biguint(self, a, bv):
        """biguint.rs:302-334 with the generator of :417-470"""
        a_len, b_len = len(a), len(bv)
        div = self.virtual_biguint(0 if b_len > a_len + 1 else a_len - b_len + 1)
        rem = self.virtual_biguint(b_len)

        def gen(v, div=div, rem=rem, a_len=a_len):
            av, bvv = value_of(v[:a_len]), value_of(v[a_len:])
            q, r = divmod(av, bvv)
            return list(zip(div, limbs_of(q, len(div)))) + list(zip(rem, limbs_of(r, len(rem))))
        self.b.add_generator(a + bv, gen, OP_DIV_REM, (a_len, len(div), len(rem)), outs=div + rem)
        div_b = self.mul_biguint(div, bv)
        self.connect_biguint(a, self.add_biguint(div_b, rem))
        self.b.assert_one(self.cmp_biguint(rem, bv))
        return div, rem

    # ------------------------------------------------------------------ non-native field (nonnative.rs), modulus m
    def constant_nonnative(self, v):
        return self.constant_biguint(v)

    def add_nonnative(self, a, bv, m=P25519):
        s = self.virtual_biguint(8)
        overflow = self.b.add_virtual_target()

        def gen(v, s=s, overflow=overflow, na=len(a), m=m):
            x, y = value_of(v[:na]) % m, value_of(v[na:]) % m
            t = x + y
            ov = 1 if t > m else 0                                    # nonnative.rs:487 (strict: t == m stays unreduced)
            return list(zip(s, limbs_of(t - m * ov, 8))) + [(overflow, ov)]
        self.b.add_generator(a + bv, gen, OP_NN_ADD, [len(a)] + limbs_of(m, 8), outs=s + [overflow])
        expected = self.add_biguint(a, bv)
        modulus = self.constant_biguint(m)
        actual = self.add_biguint(s, self.mul_biguint_by_bool(modulus, overflow))
        self.connect_biguint(expected, actual)
        self.b.assert_one(self.cmp_biguint(s, modulus))
        return s

    def sub_nonnative(self, a, bv, m=P25519):
        diff = self.virtual_biguint(8)
        overflow = self.b.add_virtual_target()

        def gen(v, diff=diff, overflow=overflow, na=len(a), m=m):
            x, y = value_of(v[:na]) % m, value_of(v[na:]) % m
            return list(zip(diff, limbs_of((x - y) % m, 8))) + [(overflow, 1 if x < y else 0)]
        self.b.add_generator(a + bv, gen, OP_NN_SUB, [len(a)] + limbs_of(m, 8), outs=diff + [overflow])
        self.b.range_check_u32(diff)
        self.b.assert_bool(overflow)
        diff_plus_b = self.add_biguint(diff, bv)
        modulus = self.constant_biguint(m)
        reduced = self.sub_biguint(diff_plus_b, self.mul_biguint_by_bool(modulus, overflow))
        self.connect_biguint(a, reduced)
        return diff

    def mul_nonnative(self, a, bv, m=P25519):
        prod = self.virtual_biguint(8)
        overflow = self.virtual_biguint(len(a) + len(bv) - 8)

        def gen(v, prod=prod, overflow=overflow, na=len(a), m=m):
            x, y = value_of(v[:na]) % m, value_of(v[na:]) % m
            q, r = divmod(x * y, m)
            return list(zip(prod, limbs_of(r, 8))) + list(zip(overflow, limbs_of(q, len(overflow))))
        self.b.add_generator(a + bv, gen, OP_NN_MUL, [len(a), len(overflow)] + limbs_of(m, 8), outs=prod + overflow)
        self.b.range_check_u32(prod)
        self.b.range_check_u32(overflow)
        expected = self.mul_biguint(a, bv)
        modulus = self.constant_biguint(m)
        actual = self.add_biguint(prod, self.mul_biguint(modulus, overflow))
        self.connect_biguint(expected, actual)
        return prod

    def neg_nonnative(self, x, m=P25519):
        return self.sub_nonnative(self.constant_biguint(0), x, m)

    def inv_nonnative(self, x, m=P25519):
        n = len(x)
        inv, div = self.virtual_biguint(n), self.virtual_biguint(n)

        def gen(v, inv=inv, div=div, m=m, n=n):
            xv = value_of(v) % m
            iv = pow(xv, m - 2, m)
            return list(zip(inv, limbs_of(iv, n))) + list(zip(div, limbs_of((xv * iv - 1) // m, n)))
        self.b.add_generator(x, gen, OP_NN_INV, [n] + limbs_of(m, 8), outs=inv + div)
        product = self.mul_biguint(x, inv)
        modulus = self.constant_biguint(m)
        expected = self.add_biguint(self.mul_biguint(modulus, div), self.constant_biguint(1))
        self.connect_biguint(product, expected)
        return inv

    def reduce(self, x, m):
        return self.div_rem_biguint(x, self.constant_biguint(m))[1]

    def mul_nonnative_by_bool(self, a, bit):
        return self.mul_biguint_by_bool(a, bit)

    def nonnative_conditional_neg(self, x, bit):
        neg = self.neg_nonnative(x)
        return self.add_nonnative(self.mul_nonnative_by_bool(neg, bit), self.mul_nonnative_by_bool(x, self.b.not_(bit)))

    def split_u32_to_4_bit_limbs(self, limb):
        two = self.b.split_le_base(limb, 16, 4)
        four = self.b.constant(4)
        return [self.b.mul_add(two[2 * i + 1], four, two[2 * i]) for i in range(8)]

    def split_nonnative_to_4_bit_limbs(self, x):
        return [n for l in x for n in self.split_u32_to_4_bit_limbs(l)]

    # ------------------------------------------------------------------ curve (curve.rs); points = (x limbs, y limbs)
    def constant_affine_point(self, pt):
        return (self.constant_nonnative(pt[0]), self.constant_nonnative(pt[1]))

    def connect_affine_point(self, p, q):
        self.connect_biguint(p[0], q[0])
        self.connect_biguint(p[1], q[1])

    def curve_neg(self, p):
        return (self.neg_nonnative(p[0]), p[1])

    def curve_double(self, p):
        x, y = p
        one, d = self.constant_nonnative(1), self.constant_nonnative(D25519)
        xx, yy, xy = self.mul_nonnative(x, x), self.mul_nonnative(y, y), self.mul_nonnative(x, y)
        xy2 = self.add_nonnative(xy, xy)
        xxyy_sum = self.add_nonnative(xx, yy)
        dxxyy = self.mul_nonnative(d, self.mul_nonnative(xx, yy))
        neg = self.neg_nonnative(dxxyy)
        inv_plus = self.inv_nonnative(self.add_nonnative(one, dxxyy))
        inv_minus = self.inv_nonnative(self.add_nonnative(one, neg))
        return (self.mul_nonnative(xy2, inv_plus), self.mul_nonnative(xxyy_sum, inv_minus))

    def curve_repeated_double(self, p, n):
        for _ in range(n):
            p = self.curve_double(p)
        return p

    def curve_add(self, p1, p2):
        (x1, y1), (x2, y2) = p1, p2
        one, d = self.constant_nonnative(1), self.constant_nonnative(D25519)
        x1y2, y1x2 = self.mul_nonnative(x1, y2), self.mul_nonnative(y1, x2)
        y1y2, x1x2 = self.mul_nonnative(y1, y2), self.mul_nonnative(x1, x2)
        sx = self.add_nonnative(x1y2, y1x2)
        sy = self.add_nonnative(y1y2, x1x2)
        dt = self.mul_nonnative(d, self.mul_nonnative(x1y2, y1x2))
        neg = self.neg_nonnative(dt)
        inv_plus = self.inv_nonnative(self.add_nonnative(one, dt))
        inv_minus = self.inv_nonnative(self.add_nonnative(one, neg))
        return (self.mul_nonnative(sx, inv_plus), self.mul_nonnative(sy, inv_minus))

    def curve_conditional_add(self, p1, p2, bit):
        not_b = self.b.not_(bit)
        s = self.curve_add(p1, p2)
        x = self.add_nonnative(self.mul_nonnative_by_bool(s[0], bit), self.mul_nonnative_by_bool(p1[0], not_b))
        y = self.add_nonnative(self.mul_nonnative_by_bool(s[1], bit), self.mul_nonnative_by_bool(p1[1], not_b))
        return (x, y)

    def point_compress(self, p):
        """curve.rs:295-307: y bits (MSB first) with the top bit OR-ed with the parity of x"""
        bits = sha512._to_bits(self.b, p[1])
        x_low = self.b.split_le(p[0][0], 32)
        a, bb = bits[0], x_low[0]
        bits[0] = self.b.sub(self.b.add(a, bb), self.b.mul(a, bb))
        return bits

    def point_decompress(self, pv):
        """curve.rs:309-324 with CurvePointDecompressionGenerator :327-370 (curve25519-dalek decompression)"""
        assert len(pv) == 256
        p = (self.virtual_biguint(8), self.virtual_biguint(8))

        def gen(v, p=p):
            val = 0
            for bit in v:
                val = (val << 1) | bit
            sign, y = val >> 255, val & ((1 << 255) - 1)
            x = _recover_x(y % P25519, sign)
            return list(zip(p[0], limbs_of(x, 8))) + list(zip(p[1], limbs_of(y, 8)))
        self.b.add_generator(pv, gen, OP_DECOMPRESS, outs=p[0] + p[1])
        pv2 = self.point_compress(p)
        for a, bb in zip(pv, pv2):
            self.b.connect(a, bb)
        return p

    # ------------------------------------------------------------------ scalar multiplications
    def random_access_curve_points(self, index, pts):
        x = [self.b.random_access(index, [pt[0][i] for pt in pts]) for i in range(8)]
        y = [self.b.random_access(index, [pt[1][i] for pt in pts]) for i in range(8)]
        return (x, y)

    def precompute_window(self, p):
        """curve_windowed_mul.rs:43-62 (the random blinding point g comes from the seeded generator)"""
        g = pt_mul(self.rng.randrange(1, L25519), BASE)
        neg = self.constant_affine_point(pt_neg(g))
        multiples = [self.constant_affine_point(g)]
        for i in range(1, 1 << WINDOW):
            multiples.append(self.curve_add(p, multiples[i - 1]))
        for i in range(1, 1 << WINDOW):
            multiples[i] = self.curve_add(neg, multiples[i])
        return multiples

    def _hash0_scalar(self, nbytes):
        return int.from_bytes(hashlib.sha3_256(bytes(8)).digest()[:nbytes], "little") % L25519

    def curve_scalar_mul_windowed(self, p, n):
        """curve_windowed_mul.rs:110-149"""
        start = pt_mul(self._hash0_scalar(25), BASE)
        start_mul = start
        for _ in range(256):                                   # Ed25519Scalar::BITS = 256 doublings (field/ed25519_scalar.rs:95)
            start_mul = pt_add(start_mul, start_mul)
        result = self.constant_affine_point(start)
        pre = self.precompute_window(p)
        zero = self.b.zero()
        windows = self.split_nonnative_to_4_bit_limbs(n)
        for i in reversed(range(len(windows))):
            result = self.curve_repeated_double(result, WINDOW)
            w = windows[i]
            to_add = self.random_access_curve_points(w, pre)
            should_add = self.b.not_(self.b.is_equal(w, zero))
            result = self.curve_conditional_add(result, to_add, should_add)
        to_sub = self.constant_affine_point(start_mul)
        return self.curve_add(result, self.curve_neg(to_sub))

    def fixed_base_curve_mul(self, base, scalar):
        """curve_fixed_base.rs:16-64"""
        limbs = self.split_nonnative_to_4_bit_limbs(scalar)
        rando = pt_mul(self._hash0_scalar(32), BASE)
        zero = self.b.zero()
        result = self.constant_affine_point(rando)
        point = base
        for limb in limbs:
            muls, acc = [], point
            for _ in range(15):
                muls.append(acc)
                acc = pt_add(acc, point)
            consts = [self.constant_affine_point(m) for m in muls]
            consts.insert(0, consts[0])
            should_add = self.b.not_(self.b.is_equal(limb, zero))
            r = self.random_access_curve_points(limb, consts)
            result = self.curve_conditional_add(result, r, should_add)
            for _ in range(4):
                point = pt_add(point, point)
        return self.curve_add(result, self.constant_affine_point(pt_neg(rando)))


def bits_in_le(bits):
    """eddsa.rs:23-32"""
    out = []
    for i in range(len(bits) // 8):
        out += [bits[i * 8 + 7 - j] for j in range(8)]
    return out[::-1]


def ed25519_circuit(b, msg_len_bits, seed=0):
    """-> dict(msg=[...], sig=[512], pk=[256]) bit targets (EDDSATargets, eddsa.rs:17-21, 34-85)"""
    g = Gadgets(b, seed)
    message, digest = sha512.sha512_circuit(b, msg_len_bits + 512)
    msg = list(message[512:])
    for t in msg:
        b.register_public_input(t)
    sig = b.add_virtual_targets(512)
    pk = b.add_virtual_targets(256)
    for t in pk:
        b.register_public_input(t)
    for i in range(256):
        b.connect(message[i], sig[i])
        b.connect(message[256 + i], pk[i])
    h = g.reduce(sha512._from_bits(b, bits_in_le(digest)), L25519)
    s = sha512._from_bits(b, bits_in_le(sig[256:]))
    a = g.point_decompress(bits_in_le(pk))
    ha = g.curve_scalar_mul_windowed(a, h)
    r = g.point_decompress(bits_in_le(sig[:256]))
    sb = g.fixed_base_curve_mul(BASE, s)
    g.connect_affine_point(sb, g.curve_add(r, ha))
    return {"msg": msg, "sig": sig, "pk": pk}


def fill_ecdsa_targets(targets, msg, sig, pk):
    """eddsa.rs:87-117: partial witness {target: bit}"""
    assert len(sig) == 64 and len(pk) == 32 and len(msg) * 8 == len(targets["msg"])
    w = dict(zip(targets["msg"], sha512.array_to_bits(msg)))
    w.update(zip(targets["sig"], sha512.array_to_bits(sig)))
    w.update(zip(targets["pk"], sha512.array_to_bits(pk)))
    return w


def build_cached(msg_len_bytes):
    """The reference's per-signature circuit for one message length (get_ed25519_circuit_targets, prove_crypto/ed25519.rs:18-42)
    with its witness program compiled, through the circuit cache: -> (CircuitData, targets, from_cache).  One definition for the
    sequential driver (signatures.ApprovalProver), the pipeline and the out-of-process prewarm (circuit_cache.prewarm)."""
    from .builder import CircuitBuilder, wide_ecc_config
    from .circuit_cache import load_or_build

    def build():
        b = CircuitBuilder(wide_ecc_config())
        targets = ed25519_circuit(b, 8 * msg_len_bytes)
        data = b.build()
        # the inputs in the order fill_ecdsa_targets names them: the program is the one an example witness would fix
        data.witness_program(list(targets["msg"]) + list(targets["sig"]) + list(targets["pk"]))
        return data, targets
    return load_or_build("ed25519", (msg_len_bytes, sorted(wide_ecc_config().items(), key=str)), build)
