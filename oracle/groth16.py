"""Groth16 over BN254 as gnark v0.9.1 runs it (backend/groth16/bn254/{setup,prove,verify}.go -- un-vendored: gnark-plonky2-verifier/
go.mod:8; called at cmd/compile.go:40, cmd/web-api.go:77,84).  TEST INFRASTRUCTURE: pure Python on oracle/bn254*.py; only tests/,
smoke() and bench.py's checker role may import it.

What pins it: the verification equation is Verifier.sol's (contracts/hardhat/contracts/Verifier.sol:503-548:
e(A, B) e(C, -delta) e(alpha, -beta) e(L_pub, -gamma) = 1), which the reference's own proof satisfies under the reference's key
(tests/test_oracle_pairing.py); proofs made by `prove` below must satisfy it under keys made by `setup`.  The quotient step
(computeH: three inverse FFTs, three coset FFTs, (a b - c) / Z on the coset, one coset inverse FFT) is restated from gnark's prover;
the PROOF BYTES of the Go prover cannot be pinned (they are randomised by r, s and the reference ships no proving key):
**parity unpinned** beyond the verification equation.

R1CS: rows of {wire: coefficient}; wires = [1, public.., private..] (gnark's ordering).
"""
from . import bn254 as B
from . import bn254_fr as FR

R = B.R


def domain_size(n_constraints):
    n = 1
    while n < max(2, n_constraints):
        n *= 2
    return n


def lagrange_at(tau, n):
    """L_j(tau) for the subgroup of size n: (tau^n - 1) w^j / (n (tau - w^j))"""
    w = FR.root(n.bit_length() - 1)
    z = (pow(tau, n, R) - 1) % R
    ninv = pow(n, R - 2, R)
    out, wj = [], 1
    for _ in range(n):
        out.append(z * wj % R * ninv % R * pow((tau - wj) % R, R - 2, R) % R)
        wj = wj * w % R
    return out


def setup(r1cs, n_public, toxic):
    """r1cs = (A, B, C) lists of sparse rows over n_wires wires; toxic = (tau, alpha, beta, gamma, delta).
    -> (pk, vk) with every point as an affine tuple (None = infinity)"""
    A, Bm, C = r1cs
    tau, alpha, beta, gamma, delta = [x % R for x in toxic]
    n_wires = 1 + max(max((max(row) if row else 0) for M in (A, Bm, C) for row in M), n_public)
    n = domain_size(len(A))
    L = lagrange_at(tau, n)
    a_t, b_t, c_t = [0] * n_wires, [0] * n_wires, [0] * n_wires
    for M, acc in ((A, a_t), (Bm, b_t), (C, c_t)):
        for j, row in enumerate(M):
            for w, coef in row.items():
                acc[w] = (acc[w] + coef * L[j]) % R
    dinv, ginv = pow(delta, R - 2, R), pow(gamma, R - 2, R)
    k = [(beta * a_t[i] + alpha * b_t[i] + c_t[i]) % R for i in range(n_wires)]
    zt = (pow(tau, n, R) - 1) % R
    g1 = lambda s: B.mul(s % R, B.G1) if s % R else None
    g2 = lambda s: B.g2_mul(s % R, B.G2) if s % R else None
    pk = {"n": n, "n_wires": n_wires, "n_public": n_public,
          "A": [g1(x) for x in a_t], "B1": [g1(x) for x in b_t], "B2": [g2(x) for x in b_t],
          "K": [g1(k[i] * dinv) for i in range(1 + n_public, n_wires)],
          "Z": [g1(pow(tau, i, R) * zt % R * dinv) for i in range(n - 1)],
          "alpha1": g1(alpha), "beta1": g1(beta), "delta1": g1(delta), "beta2": g2(beta), "delta2": g2(delta)}
    vk = {"alpha1": g1(alpha), "beta2": g2(beta), "gamma2": g2(gamma), "delta2": g2(delta),
          "K": [g1(k[i] * ginv) for i in range(1 + n_public)]}
    return pk, vk


def abc_evaluations(r1cs, witness, n):
    out = []
    for M in r1cs:
        v = [sum(coef * witness[w] for w, coef in row.items()) % R for row in M]
        out.append(v + [0] * (n - len(v)))
    return out


def compute_h(a, b, c):
    """gnark `computeH`: coefficients of (A B - C) / Z, n - 1 of them used"""
    n = len(a)
    ca, cb, cc = FR.ntt(a, inverse=True), FR.ntt(b, inverse=True), FR.ntt(c, inverse=True)
    ea, eb, ec = FR.ntt(ca, coset=True), FR.ntt(cb, coset=True), FR.ntt(cc, coset=True)
    den = pow((pow(FR.GENERATOR, n, R) - 1) % R, R - 2, R)
    q = [(x * y - z) % R * den % R for x, y, z in zip(ea, eb, ec)]
    return FR.ntt(q, inverse=True, coset=True)


def _msm1(scalars, points):
    acc = None
    for s, p in zip(scalars, points):
        if p is not None and s % R:
            acc = B.add(acc, B.mul(s % R, p))
    return acc


def _msm2(scalars, points):
    acc = None
    for s, p in zip(scalars, points):
        if p is not None and s % R:
            acc = B.g2_add(acc, B.g2_mul(s % R, p))
    return acc


def prove(pk, r1cs, witness, r, s):
    """-> (Ar, Bs, Krs) affine; witness = values of all wires, witness[0] = 1"""
    n, npub = pk["n"], pk["n_public"]
    a, b, c = abc_evaluations(r1cs, witness, n)
    h = compute_h(a, b, c)
    ar = B.add(B.add(pk["alpha1"], _msm1(witness, pk["A"])), B.mul(r % R, pk["delta1"]))
    bs2 = B.g2_add(B.g2_add(pk["beta2"], _msm2(witness, pk["B2"])), B.g2_mul(s % R, pk["delta2"]))
    bs1 = B.add(B.add(pk["beta1"], _msm1(witness, pk["B1"])), B.mul(s % R, pk["delta1"]))
    krs = B.add(_msm1(witness[1 + npub:], pk["K"]), _msm1(h[:n - 1], pk["Z"]))
    krs = B.add(krs, B.mul(s % R, ar))
    krs = B.add(krs, B.mul(r % R, bs1))
    krs = B.add(krs, B.neg(B.mul(r * s % R, pk["delta1"])))
    return ar, bs2, krs


def verify(vk, proof, public_inputs):
    """through oracle.bn254_pairing.groth16_verify -- the function the reference's own proof / tampered vectors pin
    (Verifier.sol:503-548: e(A, B) e(C, -delta) e(alpha, -beta) e(L, -gamma) == 1)"""
    from . import bn254_pairing as PR
    v = {"alpha": vk["alpha1"], "beta_neg": B.g2_neg(vk["beta2"]), "gamma_neg": B.g2_neg(vk["gamma2"]),
         "delta_neg": B.g2_neg(vk["delta2"]), "ic": vk["K"]}
    return PR.groth16_verify(v, proof_to_uint256x8(proof), [x % R for x in public_inputs])


def proof_to_uint256x8(proof):
    """the order of gnark's WriteRawTo / Verifier.sol: A.x, A.y, B.x1, B.x0, B.y1, B.y0, C.x, C.y"""
    ar, bs, krs = proof
    (bx0, bx1), (by0, by1) = bs
    return [ar[0], ar[1], bx1, bx0, by1, by0, krs[0], krs[1]]


def square_chain_r1cs(n_constraints, n_public=2):
    """a small satisfiable system for the tests: wires [1, p1..p_np, x0, x1, ...]; constraint j: x_j * x_j = x_{j+1} - p_(j mod np) - 3
    i.e. x_{j+1} = x_j^2 + p + 3.  Returns (r1cs, witness_fn(publics, x0))"""
    base = 1 + n_public
    A, Bm, C = [], [], []
    for j in range(n_constraints):
        A.append({base + j: 1})
        Bm.append({base + j: 1})
        C.append({base + j + 1: 1, 1 + (j % n_public): R - 1, 0: R - 3})

    def witness(publics, x0):
        w = [1] + [p % R for p in publics] + [x0 % R]
        for j in range(n_constraints):
            w.append((w[base + j] ** 2 + publics[j % n_public] + 3) % R)
        return w
    return (A, Bm, C), witness
