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
