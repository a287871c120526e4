"""`prove_valid_keys_stakes_in_valiators_list` (near_bft_finality/src/prove_block_data/keys_stakes.rs:18-266, its test :282-343
runs on data/*_small.json = fixture C1) -- the circuit on the CPU: witness for the reference's fixtures (3 validators; 100
validators with 66 approvals = 67.07 % of the stake), public inputs = valid_keys bytes then the valid stake sum, every gate
constraint satisfied, and no witness when the listed keys hold less than two thirds of the stake or a key byte is wrong."""
import numpy as np
import pytest

import zklc_amd  # noqa: F401
from zklc_amd import keys_stakes as KS
from conftest import load_golden, near_set_arrays
from p2_witness_check import gate_constraint_failures


def _case(name, drop=0):
    msg, approvals, validators = near_set_arrays(load_golden(name))
    present = [pos for pos, a in enumerate(approvals) if len(a) == 66]
    present = present[:len(present) - drop]
    valid_keys = b"".join(bytes([pos]) + validators[pos][-48:-16] for pos in present)
    stake = lambda v: int.from_bytes(v[-16:], "little")
    return valid_keys, validators, sum(stake(validators[p]) for p in present), sum(stake(v) for v in validators)


@pytest.mark.parametrize("name", ["ed25519_near_c1_small.json", "ed25519_near_c2_100.json"])
def test_keys_stakes_circuit_on_the_reference_fixtures(hostsim, name):
    valid_keys, validators, vs, al = _case(name)
    assert 3 * vs >= 2 * al
    data, vt, kt = KS.keys_stakes_circuit(valid_keys, [len(v) for v in validators])
    pw = {t: x for ts, v in zip(vt, validators) for t, x in zip(ts, v)}
    pw.update(zip(kt, valid_keys))
    data.witness_program(list(pw))
    wn, pn = data.generate_witness_native([pw])
    pis = [int(x) for x in pn[0]]
    assert bytes(pis[:len(valid_keys)]) == valid_keys
    assert int.from_bytes(bytes(pis[len(valid_keys):]), "little") == vs and len(pis) == len(valid_keys) + 17
    if len(validators) < 10:
        w, pp = data.generate_witness(pw)
        assert np.array_equal(w, wn[0]) and pp == pis
    assert not gate_constraint_failures(hostsim, data, wn[0], pis)
    # a key byte that is not the validator's
    bad = dict(pw)
    bad[kt[5]] ^= 1
    with pytest.raises(AssertionError):
        data.generate_witness_native([bad])


def test_less_than_two_thirds_of_the_stake_has_no_witness():
    valid_keys, validators, vs, al = _case("ed25519_near_c2_100.json", drop=6)
    assert 3 * vs < 2 * al
    data, vt, kt = KS.keys_stakes_circuit(valid_keys, [len(v) for v in validators])
    pw = {t: x for ts, v in zip(vt, validators) for t, x in zip(ts, v)}
    pw.update(zip(kt, valid_keys))
    data.witness_program(list(pw))
    with pytest.raises(AssertionError):
        data.generate_witness_native([pw])
