"""Poseidon over Goldilocks (width 12, rate 8, x^7, 4+22+4 rounds).  TEST INFRASTRUCTURE.

Follows the reference's Go restatement gnark-plonky2-verifier/poseidon/goldilocks.go:
  Poseidon :30-37, HashNToMNoPad :41-68 (overwrite-mode sponge, no padding),
  fullRounds :92-100, partialRounds :102-115, constantLayer :117-125, sBox :138-145,
  mdsRowShf/mdsLayer :172-216, partialFirstConstantLayer :231-238,
  mdsPartialLayerInit :251-275, mdsPartialLayerFast :300-331;
parameters from poseidon/goldilocks_constants.go (tests/golden/poseidon_goldilocks.json).
`permute_naive` is the textbook definition (constants, S-box on lane 0, full MDS in
every partial round); the reference's "fast" partial rounds must equal it -- checked
in tests, which also validates the extracted FAST_* tables.
Pinned by tests/goldilocks_test.go:47-53 and tests/public_inputs_hash_test.go:54-55.
Merkle tree / hash_or_noop / two_to_one follow plonky2 (un-vendored) as restated by
gnark-plonky2-verifier/fri/fri.go:97-144 and poseidon/goldilocks.go:72-86.
"""
import json
import os

from .goldilocks import P

_J = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "poseidon_goldilocks.json")))
RC = _J["all_round_constants"]
CIRC = _J["mds_circ"]
DIAG = _J["mds_diag"]
FP_FIRST = _J["fast_partial_first_round_constant"]
FP_RC = _J["fast_partial_round_constants"]
FP_VS = _J["fast_partial_round_vs"]
FP_WHATS = _J["fast_partial_round_w_hats"]
FP_INIT = _J["fast_partial_round_initial_matrix"]
WIDTH, RATE, HALF_FULL, N_PARTIAL = 12, 8, 4, 22


def sbox(x):
    x2 = x * x % P
    x3 = x2 * x % P
    return x3 * x3 % P * x % P


def mds(v):
    return [(sum(v[(i + r) % 12] * CIRC[i] for i in range(12)) + v[r] * DIAG[r]) % P for r in range(12)]


def permute_naive(state):
    s = list(state)
    rnd = 0
    for _ in range(HALF_FULL):
        s = mds([sbox((s[i] + RC[12 * rnd + i]) % P) for i in range(12)])
        rnd += 1
    for _ in range(N_PARTIAL):
        s = [(s[i] + RC[12 * rnd + i]) % P for i in range(12)]
        s[0] = sbox(s[0])
        s = mds(s)
        rnd += 1
    for _ in range(HALF_FULL):
        s = mds([sbox((s[i] + RC[12 * rnd + i]) % P) for i in range(12)])
        rnd += 1
    return s


def permute(state):
    """the reference's formulation (fast partial rounds), goldilocks.go:30-37"""
    s = list(state)
    rnd = 0
    for _ in range(HALF_FULL):
        s = mds([sbox((s[i] + RC[12 * rnd + i]) % P) for i in range(12)])
        rnd += 1
    s = [(s[i] + FP_FIRST[i]) % P for i in range(12)]
    res = [0] * 12
    res[0] = s[0]
    for r in range(1, 12):
        for d in range(1, 12):
            res[d] = (res[d] + s[r] * FP_INIT[r - 1][d - 1]) % P
    s = res
    for i in range(N_PARTIAL):
        s[0] = (sbox(s[0]) + FP_RC[i]) % P
        d = (s[0] * 25 + sum(s[j] * FP_WHATS[i][j - 1] for j in range(1, 12))) % P   # MDS0TO0 = 25
        s = [d] + [(s[0] * FP_VS[i][j - 1] + s[j]) % P for j in range(1, 12)]
    rnd += N_PARTIAL
    for _ in range(HALF_FULL):
        s = mds([sbox((s[i] + RC[12 * rnd + i]) % P) for i in range(12)])
        rnd += 1
    return s


_perm = permute


def use_c_port():
    """route the sponge / Merkle helpers below through oracle/c/goldilocks_oracle.c (checked equal to `permute`
    in tests/test_oracle_c.py); used by the plonky2 prover restatement, where pure Python is too slow"""
    global _perm
    from . import cport
    cport.load()
    _perm = cport.poseidon_gl_permute


def hash_n_to_m_no_pad(inp, n_out):
    s = [0] * 12
    for i in range(0, len(inp), RATE):
        chunk = inp[i:i + RATE]
        s[:len(chunk)] = chunk
        s = _perm(s)
    out = []
    while True:
        for i in range(RATE):
            out.append(s[i])
            if len(out) == n_out:
                return out
        s = _perm(s)


def hash_no_pad(inp):
    return hash_n_to_m_no_pad([x % P for x in inp], 4)


def hash_or_noop(inp):
    if len(inp) <= 4:
        return list(inp) + [0] * (4 - len(inp))
    return hash_no_pad(inp)


def two_to_one(left, right):
    return _perm(list(left) + list(right) + [0] * 4)[:4]


def merkle_tree(leaves, cap_height):
    """-> (cap: list of 2^cap_height digests, layers: list of digest layers from the leaves up, excluding the cap)"""
    layer = [hash_or_noop(l) for l in leaves]
    n = len(layer)
    assert n & (n - 1) == 0 and (1 << cap_height) <= n
    layers = []
    while len(layer) > (1 << cap_height):
        layers.append(layer)
        layer = [two_to_one(layer[2 * i], layer[2 * i + 1]) for i in range(len(layer) // 2)]
    return layer, layers


def merkle_prove(layers, index):
    sib = []
    for layer in layers:
        sib.append(layer[index ^ 1])
        index >>= 1
    return sib


def merkle_verify(leaf, index, siblings, cap):
    cur = hash_or_noop(leaf)
    for s in siblings:
        cur = two_to_one(s, cur) if index & 1 else two_to_one(cur, s)
        index >>= 1
    return cur == cap[index]
