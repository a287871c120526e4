"""The comparison circuits of near_bft_finality/src/prove_block_data/primitives.rs on the CPU (witness existence = the circuit
accepts), following the reference's own tests :336-456: two_thirds accepts 2/3 + 5 and exactly 2/3, rejects 2/3 - 5;
consecutive heights accept h + 1 / h (with byte borrows) and reject anything else; prove_eq_array."""
import random

import pytest

import zklc_amd  # noqa: F401
from zklc_amd import primitives as PR
from p2_witness_check import gate_constraint_failures


def _run(data, targets, values):
    if data._program is None:
        data.witness_program(list(targets))
    w, p = data.generate_witness_native([dict(zip(targets, values))])
    return w[0], [int(x) for x in p[0]]


def test_two_thirds(hostsim):
    data, v1, v2 = PR.two_thirds_circuit()
    rng = random.Random(5)
    for trial in range(6):
        third = rng.getrandbits(64) if trial else (1 << 120) // 3
        v = 3 * third
        for delta, ok in ((5, True), (0, True), (-5, False), (third, True)):
            x = (v // 3) * 2 + delta
            vals = list(x.to_bytes(17, "little")) + list(v.to_bytes(17, "little"))
            if ok:
                w, pis = _run(data, v1 + v2, vals)
                assert pis == vals[:17]
                if trial == 0 and delta == 5:
                    assert not gate_constraint_failures(hostsim, data, w, pis)
            else:
                with pytest.raises(AssertionError):
                    _run(data, v1 + v2, vals)


def test_consecutive_heights_and_eq_array(hostsim):
    data, h1, h2 = PR.consecutive_heights_circuit()
    for h in (105971806, 255, 256, 65535, (1 << 40) - 1, 0xFFFFFFFFFFFFFFFE):
        vals = list((h + 1).to_bytes(8, "little")) + list(h.to_bytes(8, "little"))
        w, pis = _run(data, h1 + h2, vals)
        assert pis == vals
    assert not gate_constraint_failures(hostsim, data, w, pis)
    # (as in the reference, only the byte right below the differing byte is checked for the 00 / ff borrow pattern: :84-101)
    for a, c in ((100, 100), (100, 101), (102, 100), (512, 255), (0x10000, 0xFE00)):
        with pytest.raises(AssertionError):
            _run(data, h1 + h2, list(a.to_bytes(8, "little")) + list(c.to_bytes(8, "little")))
    data, a1, a2 = PR.eq_array_circuit(32)
    x = bytes(range(32))
    assert _run(data, a1 + a2, list(x) + list(x))[1] == list(x)
    with pytest.raises(AssertionError):
        _run(data, a1 + a2, list(x) + list(x[:31] + b"\x00"))
