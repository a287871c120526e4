"""zklc_amd/plonky2/circuit_cache.py: a circuit loaded from the on-disk cache is the circuit that was built -- same matrices, same
witness program, and its input targets are still the objects the witness program names (one pickle per entry)."""
import hashlib

import numpy as np

from zklc_amd.plonky2 import sha256 as SHA
from zklc_amd.plonky2 import circuit_cache as CC


def _build(msg_len):
    def build():
        data, words = SHA.sha256_circuit(msg_len)
        data.witness_program(list(words))
        return data, words
    return build


def test_cache_round_trip(tmp_path, monkeypatch):
    msg = bytes(range(64))
    monkeypatch.delenv("ZKLC_CIRCUIT_CACHE", raising=False)
    d0, w0, hit = CC.load_or_build("sha256", 2, _build(len(msg)))
    assert not hit and not list(tmp_path.iterdir())                 # disabled: nothing is written anywhere
    monkeypatch.setenv("ZKLC_CIRCUIT_CACHE", str(tmp_path))
    d1, w1, hit = CC.load_or_build("sha256", 2, _build(len(msg)))
    assert not hit and len(list(tmp_path.glob("sha256-*.circuit"))) == 1
    d2, w2, hit = CC.load_or_build("sha256", 2, lambda: (_ for _ in ()).throw(AssertionError("must come from the cache")))
    assert hit and d2.builder is None
    for name in ("constants", "sigmas"):
        assert np.array_equal(getattr(d1, name), getattr(d2, name))
    assert [g.id() for g in d1.gates] == [g.id() for g in d2.gates] and d1.common_data() == d2.common_data()
    for k in ("code", "params", "input_slots", "wire_slot", "wire_index", "pi_slots"):
        assert np.array_equal(d1._program[k], d2._program[k])
    assert all(a is b for a, b in zip(w2, d2._program["input_targets"]))    # shared objects survive the round trip
    want = [int.from_bytes(hashlib.sha256(msg).digest()[4 * i:4 * i + 4], "big") for i in range(8)]
    for d, w in ((d1, w1), (d2, w2)):
        wires, pis = d.generate_witness_native([SHA.sha256_witness(w, msg)])
        assert [int(x) for x in pis[0]] == want
    wa, _ = d1.generate_witness_native([SHA.sha256_witness(w1, msg)])
    wb, _ = d2.generate_witness_native([SHA.sha256_witness(w2, msg)])
    assert np.array_equal(wa, wb)
    # another key is another entry; a damaged entry is rebuilt
    _, _, hit = CC.load_or_build("sha256", 1, _build(10))
    assert not hit and len(list(tmp_path.glob("sha256-*.circuit"))) == 2
    path = sorted(tmp_path.glob("sha256-*.circuit"))[0]
    path.write_bytes(b"not a pickle")
    for key, n in ((2, 64), (1, 10)):
        d, w, _ = CC.load_or_build("sha256", key, _build(n))
        assert d._program is not None
