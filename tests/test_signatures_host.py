"""Host logic of the approval pre-check mirror (no GPU): message format and borsh slicing."""
import numpy as np
import pytest

from conftest import load_golden, near_sets, near_set_arrays


def test_generate_signed_message_matches_fixture():
    from zklc_amd import signatures as S
    for name in near_sets():
        j = load_golden(name)
        msg = S.generate_signed_message(j["current_height"], j["next_height"], bytes.fromhex(j["next_prev_hash"]))
        assert msg.hex() == j["msg"]
    # signatures.rs:299-305 / :317-323 : Endorsement = 0x00||hash||le64, Skip = 0x01||le64||le64
    assert S.generate_signed_message(10, 11, bytes(32)) == b"\x00" + bytes(32) + (11).to_bytes(8, "little")
    assert S.generate_signed_message(10, 12, bytes(32)) == b"\x01" + (10).to_bytes(8, "little") + (12).to_bytes(8, "little")


def test_slice_approvals():
    from zklc_amd import signatures as S
    j = load_golden("ed25519_near_c2_100.json")
    msg, approvals, validators = near_set_arrays(j)
    pos, pks, sigs = S.slice_approvals(approvals, validators)
    assert len(pos) == 66 and pks.shape == (66, 32) and sigs.shape == (66, 64)
    assert pos == [i for i, a in enumerate(approvals) if len(a) == 66]
    assert pks[0].tobytes() == validators[pos[0]][-48:-16] and sigs[0].tobytes() == approvals[pos[0]][2:]
    with pytest.raises(ValueError):
        S.slice_approvals(approvals[:-1], validators)
