"""Poseidon over BN254 Fr (iden3 parameters, t = 4, 8 full + 56 partial rounds, x^5) and
the plonky2 hasher built on it.  TEST INFRASTRUCTURE.

Follows crypto/plonky2_bn128/src/poseidon_bn128.rs:18-108 (permution, ark, exp5,
full_rounds, partial_rounds, mix) and crypto/plonky2_bn128/src/config.rs:132-199
(hash_no_pad: three Goldilocks elements per Fr limb as little-endian u64s, three limbs per
permutation into state[1..4], digest = state[0]; hash_pad; hash_or_noop; two_to_one) and
:36-70 (to_bytes / to_vec with 7-byte chunks).  Pinned by the four permutation KATs of
poseidon_bn128.rs:133-180 (tests/golden/poseidon_bn254.json).
"""
import json
import os

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_J = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "poseidon_bn254.json")))
C = [int(x) for x in _J["C"]]
S = [int(x) for x in _J["S"]]
M = [[int(x) for x in row] for row in _J["M"]]
PM = [[int(x) for x in row] for row in _J["P"]]
KATS = [{"in": [int(x) for x in k["in"]], "out": [int(x) for x in k["out"]]} for k in _J["kats"]]
WIDTH, RATE, FULL, PARTIAL = 4, 3, 8, 56


def _exp5(x):
    x2 = x * x % R
    return x2 * x2 % R * x % R


def _mix(s, m):
    return [sum(m[j][i] * s[j] for j in range(4)) % R for i in range(4)]


def permute(state):
    s = [(state[i] + C[i]) % R for i in range(4)]
    # first half of the full rounds
    for i in range(FULL // 2 - 1):
        s = [_exp5(x) for x in s]
        s = [(s[k] + C[(i + 1) * 4 + k]) % R for k in range(4)]
        s = _mix(s, M)
    s = [_exp5(x) for x in s]
    s = [(s[k] + C[(FULL // 2) * 4 + k]) % R for k in range(4)]
    s = _mix(s, PM)
    for i in range(PARTIAL):
        s[0] = (_exp5(s[0]) + C[(FULL // 2 + 1) * 4 + i]) % R
        new0 = sum(S[7 * i + j] * s[j] for j in range(4)) % R
        for k in range(1, 4):
            s[k] = (s[k] + s[0] * S[7 * i + 4 + k - 1]) % R
        s[0] = new0
    for i in range(FULL // 2 - 1):
        s = [_exp5(x) for x in s]
        s = [(s[k] + C[(FULL // 2 + 1) * 4 + PARTIAL + i * 4 + k]) % R for k in range(4)]
        s = _mix(s, M)
    s = [_exp5(x) for x in s]
    return _mix(s, M)


def hash_no_pad(gl_elems):
    s = [0, 0, 0, 0]
    for i in range(0, len(gl_elems), 9):
        chunk = gl_elems[i:i + 9]
        for j in range(0, len(chunk), 3):
            limb = 0
            for k, e in enumerate(chunk[j:j + 3]):
                limb |= e << (64 * k)
            assert limb < R
            s[j // 3 + 1] = limb
        s = permute(s)
    return s[0]


def hash_or_noop(gl_elems):
    if len(gl_elems) <= 3:
        v = 0
        for k, e in enumerate(gl_elems):
            v |= e << (64 * k)
        return v
    return hash_no_pad(gl_elems)


def two_to_one(left, right):
    return permute([0, 0, left, right])[0]


def hash_to_vec(h):
    b = h.to_bytes(32, "little")
    return [int.from_bytes(b[i:i + 7], "little") for i in range(0, 32, 7)]
