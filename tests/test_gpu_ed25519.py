"""GPU parity: the HIP Ed25519/SHA-512 kernels through the C ABI vs the oracle."""
import hashlib
import os

import numpy as np
import pytest

from conftest import load_golden, near_sets, near_set_arrays
from edcases import edge_cases, synthetic_set
from oracle import ed25519_ref as ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", near_sets())
def test_near_fixtures_verify_approvals(zctx, name):
    from zklc_amd import signatures as S
    j = load_golden(name)
    msg, approvals, validators = near_set_arrays(j)
    valid_keys, valid_pos, valid_stake, total_stake = S.verify_approvals(zctx, msg, approvals, validators)
    assert len(valid_pos) == j["expect_valid"]
    assert len(valid_keys) == 33 * j["expect_valid"]
    exp_pos = [i for i, a in enumerate(approvals) if len(a) == 66]
    assert valid_pos == exp_pos
    assert valid_keys[:1] == bytes([exp_pos[0]]) and valid_keys[1:33] == validators[exp_pos[0]][-48:-16]
    assert 0 < valid_stake <= total_stake


def test_invalid_signature_raises_like_reference_panic(zctx):
    from zklc_amd import signatures as S
    j = load_golden("ed25519_near_c1_small.json")
    msg, approvals, validators = near_set_arrays(j)
    bad = bytearray(approvals[1])
    bad[10] ^= 1
    approvals[1] = bytes(bad)
    with pytest.raises(S.InvalidSignature) as e:
        S.verify_approvals(zctx, msg, approvals, validators)
    assert e.value.positions == [1]
    _, pos, _, _ = S.verify_approvals(zctx, msg, approvals, validators, strict=False)
    assert pos == [0, 2]


def test_fixed_triple(zctx):
    t = load_golden("ed25519_fixed_triple.json")
    ok = zctx.ed25519_verify_batch(bytes.fromhex(t["pk"]), bytes.fromhex(t["sig"]), bytes.fromhex(t["msg"]))
    assert ok.tolist() == [1]


def test_edge_cases_match_oracle(zctx):
    cases = edge_cases()
    # group by message so each group is one shared-message batch
    by_msg = {}
    for pk, sig, msg, label in cases:
        by_msg.setdefault(msg, []).append((pk, sig, label))
    for msg, items in by_msg.items():
        ok = zctx.ed25519_verify_batch(b"".join(i[0] for i in items), b"".join(i[1] for i in items), msg)
        for o, (pk, sig, label) in zip(ok, items):
            assert int(o) == int(ref.verify(pk, sig, msg)), label


def test_per_signature_messages(zctx):
    rng = np.random.default_rng(5)
    n, stride, mlen = 70, 48, 37
    msgs = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
    pks, sigs, exp = [], [], []
    for i in range(n):
        sd = ref.synthetic_seed(21, i)
        _, _, pk = ref.keypair(sd)
        m = msgs[i, :mlen].tobytes()
        sg = bytearray(ref.sign(sd, m))
        if i % 9 == 3:
            sg[40] ^= 2
        pks.append(pk)
        sigs.append(bytes(sg))
        exp.append(int(ref.verify(pk, bytes(sg), m)))
    ok = zctx.ed25519_verify_batch(b"".join(pks), b"".join(sigs), msgs, msg_stride=stride, msg_len=mlen)
    assert ok.tolist() == exp


@pytest.mark.parametrize("n", [1, 63, 64, 65, 300])
def test_ragged_batches(zctx, n):
    pks, sigs, msg = synthetic_set(n, seed=2, corrupt_every=11)
    ok = zctx.ed25519_verify_batch(b"".join(pks), b"".join(sigs), msg)
    assert ok.tolist() == [int(i % 11 != 10) for i in range(n)]


def test_empty_batch(zctx):
    assert zctx.ed25519_verify_batch(b"", b"", b"abc").size == 0


def test_device_pointer_path_large_tiled(zctx):
    """Full-size property test: tile a verified 256-signature set to 2^17 signatures
    on the device, flip one byte in every 1000th -> exactly those are rejected."""
    import torch
    base_n, reps = 256, 512
    pks, sigs, msg = synthetic_set(base_n, seed=4)
    pk = torch.tensor(np.frombuffer(b"".join(pks), np.uint8).copy(), device="cuda").view(base_n, 32).repeat(reps, 1).contiguous()
    sg = torch.tensor(np.frombuffer(b"".join(sigs), np.uint8).copy(), device="cuda").view(base_n, 64).repeat(reps, 1).contiguous()
    n = base_n * reps
    bad = torch.arange(0, n, 1000, device="cuda")
    sg[bad, 7] ^= 1
    m = torch.tensor(np.frombuffer(msg, np.uint8).copy(), device="cuda")
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    zctx.ed25519_verify_batch_dev(pk, sg, m, len(msg), 0, n, ok, stream=torch.cuda.current_stream())
    torch.cuda.synchronize()
    exp = torch.ones(n, dtype=torch.uint8, device="cuda")
    exp[bad] = 0
    assert torch.equal(ok, exp)


def test_second_context_gives_the_same_answers(zctx):
    """contexts are independent (each builds its own base-point table on the device)"""
    import zklc_amd
    pks, sigs, msg = synthetic_set(130, seed=8, corrupt_every=6)
    want = [int(i % 6 != 5) for i in range(130)]
    with zklc_amd.Context(0) as c:
        assert c.ed25519_verify_batch(b"".join(pks), b"".join(sigs), msg).tolist() == want
    assert zctx.ed25519_verify_batch(b"".join(pks), b"".join(sigs), msg).tolist() == want


@pytest.mark.parametrize("length", [0, 1, 41, 105, 111, 112, 127, 128, 129, 240, 1000])
def test_sha512_batch(zctx, length):
    rng = np.random.default_rng(length)
    n, stride = 97, length + 5
    data = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
    out = zctx.sha512_batch(data, stride, length, n)
    for i in range(n):
        assert out[i].tobytes() == hashlib.sha512(data[i, :length].tobytes()).digest()
