"""Pins oracle/ed25519_ref.py to the reference's own fixtures."""
import hashlib

import pytest

from conftest import load_golden, near_sets, near_set_arrays
from oracle import ed25519_ref as ref


@pytest.mark.parametrize("name", near_sets())
def test_near_fixture_accepts(name):
    j = load_golden(name)
    msg, approvals, validators = near_set_arrays(j)
    assert msg == ref.generate_signed_message(j["current_height"], j["next_height"], bytes.fromhex(j["next_prev_hash"]))
    assert len(msg) in (41, 17)
    n_ok = 0
    for ap, va in zip(approvals, validators):
        if len(ap) == 66:
            pk, sig = va[-48:-16], ap[2:]
            assert ref.verify(pk, sig, msg)
            assert ref.verify_message_intree(msg, sig, pk)  # eddsa.rs:33-58 agrees on honest input
            n_ok += 1
    assert n_ok == j["expect_valid"]


def test_c2_counts():
    j = load_golden("ed25519_near_c2_100.json")
    assert len(j["entries"]) == 100 and j["expect_valid"] == 66


def test_fixed_triple():
    t = load_golden("ed25519_fixed_triple.json")
    assert ref.verify(bytes.fromhex(t["pk"]), bytes.fromhex(t["sig"]), bytes.fromhex(t["msg"]))


def test_rfc8032_vector_1():
    sk = bytes.fromhex("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60")
    pk = bytes.fromhex("d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a")
    sig = bytes.fromhex("e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46b"
                        "d25bf5f0595bbe24655141438e7a100b")
    assert ref.keypair(sk)[2] == pk
    assert ref.sign(sk, b"") == sig
    assert ref.verify(pk, sig, b"")
    assert not ref.verify(pk, sig, b"x")


def test_reject_classes():
    from edcases import edge_cases
    seen = {}
    for pk, sig, msg, label in edge_cases():
        seen.setdefault(label.split(" #")[0], []).append(ref.verify(pk, sig, msg))
    assert seen["honest"] == [True]
    assert seen["s+l"] == [False] and seen["s=l"] == [False]
    assert seen["undecodable A"] == [False]
    assert seen["R sign flipped"] == [False]
    assert any(seen["small-order A"])  # cofactor-less equation accepts crafted small-order keys
