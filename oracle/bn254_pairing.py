"""BN254 optimal-ate pairing and the Groth16 verification equation.  TEST INFRASTRUCTURE.

The pairing the reference relies on is (a) the EVM precompile 0x08 called by its Solidity verifier
(contracts/hardhat/contracts/Verifier.sol:503-548: e(A,B) e(C,-delta) e(alpha,-beta) e(L_pub,-gamma) == 1 with
L_pub = CONSTANT + sum_i input_i PUB_i, :367-420) and (b) gnark-crypto's `bn254.Pair / PairingCheck` behind
`groth16.Verify` (gnark-plonky2-verifier/cmd/web-api.go:84; un-vendored).  Restated from the published definition of the
optimal ate pairing on BN curves (Vercauteren; loop count 6x+2 with x = 4965661367192848881, the `t` of Verifier.sol:29-33),
with gnark-crypto's tower: Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3 - (9+u)), Fp12 = Fp6[w]/(w^2 - v), D-type twist.
Pinned by the reference's Groth16 known-answer proof: contracts/hardhat/test/proof_with_witness.json must verify against
the verifying key of Verifier.sol:57-83, the tampered vectors of test/verify.ts:9-27 must not
(tests/test_oracle_pairing.py), plus bilinearity.  The reduced pairing is computed with the exact exponent (p^12-1)/r.
"""
from . import bn254 as B

P, R = B.P, B.R
X = 4965661367192848881
ATE_LOOP = 6 * X + 2
XI = (9, 1)

f2_add, f2_sub, f2_mul, f2_inv = B.f2_add, B.f2_sub, B.f2_mul, B.f2_inv
F2_ZERO, F2_ONE = (0, 0), (1, 0)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_conj(a):
    return (a[0], (-a[1]) % P)


def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_mul(a, a)
        e >>= 1
    return r


# ---- Fp6 = Fp2[v] / (v^3 - xi): (c0, c1, c2)
F6_ZERO, F6_ONE = (F2_ZERO, F2_ZERO, F2_ZERO), (F2_ONE, F2_ZERO, F2_ZERO)


def f6_add(a, b):
    return tuple(f2_add(x, y) for x, y in zip(a, b))


def f6_sub(a, b):
    return tuple(f2_sub(x, y) for x, y in zip(a, b))


def f6_neg(a):
    return tuple(f2_neg(x) for x in a)


def f6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    c0 = f2_add(f2_mul(a0, b0), f2_mul(XI, f2_add(f2_mul(a1, b2), f2_mul(a2, b1))))
    c1 = f2_add(f2_add(f2_mul(a0, b1), f2_mul(a1, b0)), f2_mul(XI, f2_mul(a2, b2)))
    c2 = f2_add(f2_add(f2_mul(a0, b2), f2_mul(a1, b1)), f2_mul(a2, b0))
    return (c0, c1, c2)


def f6_mul_by_v(a):
    return (f2_mul(XI, a[2]), a[0], a[1])


def f6_inv(a):
    a0, a1, a2 = a
    t0 = f2_sub(f2_mul(a0, a0), f2_mul(XI, f2_mul(a1, a2)))
    t1 = f2_sub(f2_mul(XI, f2_mul(a2, a2)), f2_mul(a0, a1))
    t2 = f2_sub(f2_mul(a1, a1), f2_mul(a0, a2))
    d = f2_inv(f2_add(f2_mul(a0, t0), f2_mul(XI, f2_add(f2_mul(a2, t1), f2_mul(a1, t2)))))
    return (f2_mul(t0, d), f2_mul(t1, d), f2_mul(t2, d))


# ---- Fp12 = Fp6[w] / (w^2 - v): (c0, c1)
F12_ONE = (F6_ONE, F6_ZERO)


def f12_mul(a, b):
    t0, t1 = f6_mul(a[0], b[0]), f6_mul(a[1], b[1])
    c1 = f6_sub(f6_sub(f6_mul(f6_add(a[0], a[1]), f6_add(b[0], b[1])), t0), t1)
    return (f6_add(t0, f6_mul_by_v(t1)), c1)


def f12_sqr(a):
    return f12_mul(a, a)


def f12_conj(a):
    return (a[0], f6_neg(a[1]))


def f12_inv(a):
    d = f6_inv(f6_sub(f6_mul(a[0], a[0]), f6_mul_by_v(f6_mul(a[1], a[1]))))
    return (f6_mul(a[0], d), f6_neg(f6_mul(a[1], d)))


def f12_pow(a, e):
    r = F12_ONE
    while e:
        if e & 1:
            r = f12_mul(r, a)
        a = f12_sqr(a)
        e >>= 1
    return r


# Frobenius: conjugate every Fp2 coefficient, multiply the coefficient of v^i w^j by xi^((2i + j)(p^k - 1)/6)
def _gammas(k):
    e = (P**k - 1) // 6
    return [f2_pow(XI, e * i) for i in range(6)]


_G1, _G2, _G3 = _gammas(1), _gammas(2), _gammas(3)


def f12_frobenius(a, k=1):
    g = {1: _G1, 2: _G2, 3: _G3}[k]
    cj = (lambda x: f2_conj(x)) if k & 1 else (lambda x: x)
    c0 = tuple(f2_mul(cj(a[0][i]), g[2 * i]) for i in range(3))
    c1 = tuple(f2_mul(cj(a[1][i]), g[2 * i + 1]) for i in range(3))
    return (c0, c1)


# ---- Miller loop (affine lines; the line through T with slope lam evaluated at P = (xp, yp), D-type twist:
#      l = yp - lam xp w + (lam xT - yT) w^3, w^3 = v w)
def _line(lam, t, p):
    xp, yp = p
    c0 = ((yp % P, 0), F2_ZERO, F2_ZERO)
    c1 = (f2_neg(B.f2_scalar(lam, xp)), f2_sub(f2_mul(lam, t[0]), t[1]), F2_ZERO)
    return (c0, c1)


def _double_step(t, p):
    lam = f2_mul(B.f2_scalar(f2_mul(t[0], t[0]), 3), f2_inv(B.f2_scalar(t[1], 2)))
    line = _line(lam, t, p)
    x3 = f2_sub(f2_mul(lam, lam), B.f2_scalar(t[0], 2))
    return (x3, f2_sub(f2_mul(lam, f2_sub(t[0], x3)), t[1])), line


def _add_step(t, q, p):
    lam = f2_mul(f2_sub(q[1], t[1]), f2_inv(f2_sub(q[0], t[0])))
    line = _line(lam, t, p)
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), t[0]), q[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(t[0], x3)), t[1])), line


def _frob_q(q, k):
    """pi^k on the twist: (x, y) -> (conj^k(x) xi^((p^k-1)/3), conj^k(y) xi^((p^k-1)/2))"""
    g = {1: _G1, 2: _G2}[k]
    cj = (lambda x: f2_conj(x)) if k & 1 else (lambda x: x)
    return (f2_mul(cj(q[0]), g[2]), f2_mul(cj(q[1]), g[3]))


def miller_loop(p, q):
    """f_{6x+2,Q}(P) l_{[6x+2]Q, pi Q}(P) l_{., -pi^2 Q}(P); p in G1 (affine ints), q in G2 (affine Fp2); None = infinity"""
    if p is None or q is None:
        return F12_ONE
    f = F12_ONE
    t = q
    for bit in bin(ATE_LOOP)[3:]:
        t, l = _double_step(t, p)
        f = f12_mul(f12_sqr(f), l)
        if bit == "1":
            t, l = _add_step(t, q, p)
            f = f12_mul(f, l)
    q1 = _frob_q(q, 1)
    q2 = _frob_q(q, 2)
    q2 = (q2[0], f2_neg(q2[1]))
    t, l = _add_step(t, q1, p)
    f = f12_mul(f, l)
    t, l = _add_step(t, q2, p)
    f = f12_mul(f, l)
    return f


FINAL_EXP = (P**12 - 1) // R


def final_exponentiation(f):
    # easy part (p^6 - 1)(p^2 + 1), then the hard part (p^4 - p^2 + 1)/r by plain exponentiation
    f = f12_mul(f12_conj(f), f12_inv(f))
    f = f12_mul(f12_frobenius(f, 2), f)
    return f12_pow(f, (P**4 - P**2 + 1) // R)


def pairing(p, q):
    return final_exponentiation(miller_loop(p, q))


def pairing_check(pairs):
    """product of e(P_i, Q_i) == 1"""
    f = F12_ONE
    for p, q in pairs:
        f = f12_mul(f, miller_loop(p, q))
    return final_exponentiation(f) == F12_ONE


def f12_flat(a):
    """12 Fp coefficients in tower order: c0.b0.a0, c0.b0.a1, c0.b1.a0, ... (gnark-crypto E12 memory order)"""
    return [x for c in a for b in c for x in b]


# ---- Groth16 verification with the EVM conventions of the reference's verifier
def groth16_verify(vk, proof8, inputs):
    """vk: dict with alpha (G1), beta_neg / gamma_neg / delta_neg (G2, already negated as in Verifier.sol:57-75),
    ic (list of G1: constant + one per input); proof8: the 8 uint256 words [A.x, A.y, B.x1, B.x0, B.y1, B.y0, C.x, C.y]"""
    a = (proof8[0], proof8[1])
    b = ((proof8[3], proof8[2]), (proof8[5], proof8[4]))
    c = (proof8[6], proof8[7])
    if not (B.is_on_curve(a) and B.is_on_curve(c) and B.g2_is_on_curve(b)) or any(x >= R for x in inputs):
        return False
    l = vk["ic"][0]
    for s, pt in zip(inputs, vk["ic"][1:]):
        l = B.add(l, B.mul(s, pt))
    return pairing_check([(a, b), (c, vk["delta_neg"]), (vk["alpha"], vk["beta_neg"]), (l, vk["gamma_neg"])])
