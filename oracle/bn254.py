"""BN254 (alt_bn128) G1 arithmetic and MSM definition.  TEST INFRASTRUCTURE.

Parameters as pinned by the reference's Solidity verifier
(contracts/hardhat/contracts/Verifier.sol:28-40: P, R; curve y^2 = x^3 + 3, generator (1, 2)).
The MSM itself lives in gnark-crypto (un-vendored, gnark-plonky2-verifier/go.mod:9;
call site cmd/web-api.go:77): restated from its definition sum_i s_i * P_i.  PARITY UNPINNED
(the reference holds no MSM vector) -- the group law is pinned through the Groth16
known-answer proof instead (tests/test_oracle_bn254.py).
Memory format of the C ABI = gnark-crypto's: Fp element = x * 2^256 mod p as 4 little-endian u64.
"""
P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
B = 3
G1 = (1, 2)
MONT_R = 1 << 256


def inv(a):
    return pow(a, P - 2, P)


def is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - B) % P == 0


def add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * inv(2 * y1) % P
    else:
        lam = (y2 - y1) * inv(x2 - x1) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def neg(a):
    return None if a is None else (a[0], (-a[1]) % P)


def mul(k, a):
    r = None
    k %= R
    while k:
        if k & 1:
            r = add(r, a)
        a = add(a, a)
        k >>= 1
    return r


def msm(scalars, points):
    r = None
    for s, pt in zip(scalars, points):
        r = add(r, mul(s, pt))
    return r


def to_mont_words(x):
    """gnark-crypto memory image of an Fp element: 4 little-endian u64 of x * 2^256 mod p"""
    m = x * MONT_R % P
    return [(m >> (64 * i)) & (2**64 - 1) for i in range(4)]


def from_mont_words(w):
    m = sum(int(w[i]) << (64 * i) for i in range(4))
    return m * inv(MONT_R) % P


def hash_to_curve(seed, i):
    """try-and-increment on x, even y"""
    import hashlib
    ctr = 0
    while True:
        h = hashlib.sha256(b"zklc/bn254/g1" + seed.to_bytes(8, "little") + i.to_bytes(8, "little") + ctr.to_bytes(4, "little")).digest()
        x = int.from_bytes(h, "little") % P
        rhs = (x * x * x + B) % P
        y = pow(rhs, (P + 1) // 4, P)
        if y * y % P == rhs:
            if y & 1:
                y = P - y
            return (x, y)
        ctr += 1


# ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2 + 1) and G2 (twist y^2 = x^3 + 3/(9 + u))
# Parameters of the alt_bn128 pairing precompile the reference's Solidity verifier calls
# (contracts/hardhat/contracts/Verifier.sol: PRECOMPILE_VERIFY input = (G1, G2) pairs with G2 coordinates as
# (x_imaginary, x_real, y_imaginary, y_real) words).  G2 MSM itself is gnark-crypto's (un-vendored): PARITY UNPINNED,
# restated from the definition sum_i s_i Q_i.
def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_inv(a):
    d = inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * d % P, (-a[1]) * d % P)


def f2_scalar(a, k):
    return (a[0] * k % P, a[1] * k % P)


B2 = f2_mul((3, 0), f2_inv((9, 1)))
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
       11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930,
       4082367875863433681332203403145435568316851327593401208105741076214120093531))


def g2_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), B2)) == (0, 0)


def g2_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return None
        lam = f2_mul(f2_scalar(f2_mul(x1, x1), 3), f2_inv(f2_scalar(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))


def g2_neg(a):
    return None if a is None else (a[0], ((-a[1][0]) % P, (-a[1][1]) % P))


def g2_mul(k, a):
    r = None
    k %= R
    while k:
        if k & 1:
            r = g2_add(r, a)
        a = g2_add(a, a)
        k >>= 1
    return r


def g2_msm(scalars, points):
    r = None
    for s, pt in zip(scalars, points):
        r = g2_add(r, g2_mul(s, pt))
    return r


def g2_to_words(pt):
    """gnark-crypto G2Affine memory image: X.A0, X.A1, Y.A0, Y.A1 (16 u64); infinity = zeros"""
    if pt is None:
        return [0] * 16
    (x0, x1), (y0, y1) = pt
    return to_mont_words(x0) + to_mont_words(x1) + to_mont_words(y0) + to_mont_words(y1)


def g2_from_words(w, inf=False):
    if inf:
        return None
    c = [from_mont_words(w[4 * i:4 * i + 4]) for i in range(4)]
    return ((c[0], c[1]), (c[2], c[3]))


assert g2_is_on_curve(G2)
