"""The C restatement (oracle/c) against the Python oracle and the reference fixtures."""
import hashlib
import os
import time

import numpy as np
import pytest

from conftest import load_golden, near_sets, near_set_arrays
from edcases import edge_cases, synthetic_set
from oracle import cport
from oracle import ed25519_ref as ref


@pytest.mark.parametrize("name", near_sets())
def test_near_fixtures(name):
    j = load_golden(name)
    msg, approvals, validators = near_set_arrays(j)
    ok = sum(cport.ed25519_verify(va[-48:-16], ap[2:], msg) for ap, va in zip(approvals, validators) if len(ap) == 66)
    assert ok == j["expect_valid"]


def test_edge_cases_match_python_oracle():
    for pk, sig, msg, label in edge_cases():
        assert cport.ed25519_verify(pk, sig, msg) == ref.verify(pk, sig, msg), label


def test_sha512():
    for n in [0, 1, 111, 112, 113, 127, 128, 129, 1000]:
        m = os.urandom(n)
        assert cport.sha512(m) == hashlib.sha512(m).digest()


def test_batch_threads():
    pks, sigs, msg = synthetic_set(64, seed=9, corrupt_every=5)
    pk = np.frombuffer(b"".join(pks), np.uint8)
    sg = np.frombuffer(b"".join(sigs), np.uint8)
    m = np.frombuffer(msg, np.uint8)
    ok1, _ = cport.ed25519_verify_batch(pk, sg, m, len(msg), 0, 64, nthreads=1)
    ok2, used = cport.ed25519_verify_batch(pk, sg, m, len(msg), 0, 64, nthreads=2)
    assert ok1.tolist() == ok2.tolist() == [int(i % 5 != 4) for i in range(64)]
    assert used == 2
