"""The two-pass NTT over BN254 Fr (csrc/bn254_fr_ntt_tile.cuh: table block, tile butterflies, inter-pass twiddles, coset / 1/n
scaling) walked sequentially on the CPU against the oracle's definition of the transform (oracle/bn254_fr.py), all four modes."""
import ctypes

import numpy as np
import pytest

from oracle import bn254_fr as FR

R = FR.R


def _to_words(vals):
    mont = [v * (1 << 256) % R for v in vals]
    return np.array([[(m >> (64 * k)) & (2**64 - 1) for k in range(4)] for m in mont], dtype=np.uint64)


def _from_words(a):
    rinv = pow(1 << 256, R - 2, R)
    return [sum(int(row[k]) << (64 * k) for k in range(4)) * rinv % R for row in a]


@pytest.mark.parametrize("log_n", [12, 13])
@pytest.mark.parametrize("inverse,coset", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_two_pass_walk_matches_definition(hostsim, log_n, inverse, coset):
    rng = np.random.default_rng(100 * log_n + 2 * inverse + coset)
    n = 1 << log_n
    vals = [int(rng.integers(0, 2**63)) * int(rng.integers(0, 2**63)) * int(rng.integers(0, 2**63)) * 977 % R for _ in range(n)]
    vals[0], vals[1], vals[n - 1] = 0, R - 1, 1
    a = _to_words(vals)
    hostsim.hostsim_fr_ntt_two_pass(a.ctypes.data_as(ctypes.c_void_p), log_n, inverse, coset)
    assert _from_words(a) == FR.ntt(vals, inverse=bool(inverse), coset=bool(coset))
