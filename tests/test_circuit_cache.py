"""zklc_amd/plonky2/circuit_cache.py: a circuit loaded from the on-disk cache is the circuit that was built -- same matrices, same
witness program, and its input targets are still the objects the witness program names (one circuit container per entry: plonky2/container.py)."""
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
    path.write_bytes(b"not a container")
    for key, n in ((2, 64), (1, 10)):
        d, w, _ = CC.load_or_build("sha256", key, _build(n))
        assert d._program is not None


def test_concurrent_misses_build_once(tmp_path):
    """the ranks of a multi-GPU job start together on a cold cache: one of them builds an entry, the others wait and load it"""
    import subprocess
    import sys
    import textwrap
    script = textwrap.dedent("""
        import os, sys, time
        sys.path.insert(0, %r)
        from zklc_amd.plonky2 import sha256 as SHA
        from zklc_amd.plonky2 import circuit_cache as CC
        def build():
            open(os.path.join(%r, "built-%%d" %% os.getpid()), "w").close()
            time.sleep(1.0)                      # the others arrive while this one is building
            data, words = SHA.sha256_circuit(10)
            data.witness_program(list(words))
            return data, words
        data, words, hit = CC.load_or_build("sha256", "conc", build)
        print("hit" if hit else "built", data.n)
    """) % (str(__import__("pathlib").Path(__file__).resolve().parents[1]), str(tmp_path))
    env = dict(__import__("os").environ, ZKLC_CIRCUIT_CACHE=str(tmp_path / "cache"))
    procs = [subprocess.Popen([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE, text=True) for _ in range(3)]
    outs = [p.communicate(timeout=300)[0].split() for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert sorted(o[0] for o in outs) == ["built", "hit", "hit"] and len({o[1] for o in outs}) == 1
    assert len(list(tmp_path.glob("built-*"))) == 1


def test_prewarm_builds_missing_entries_in_child_processes(tmp_path, monkeypatch):
    """circuit_cache.prewarm (cold start, round 5): missing entries are built by child interpreters side by side and found by
    build_cached afterwards; present entries are skipped; without a cache directory it is a no-op; an unknown kind is reported,
    never raised"""
    from zklc_amd.plonky2 import circuit_cache as CC
    from zklc_amd.plonky2 import sha256
    monkeypatch.setenv("ZKLC_CIRCUIT_CACHE", str(tmp_path))
    rep = CC.prewarm([("sha256", 64), ("sha256", 130), ("sha256", 64)], processes=2, timeout_s=300)
    assert rep["missing"] == 2 and rep["built"] == 2 and rep["failed"] == [], rep        # 64 and 130 bytes: two and three blocks
    data, words, from_cache = sha256.build_cached(64)
    assert from_cache and data._program is not None and len(words) == 16 * sha256.block_num_of(64)
    again = CC.prewarm([("sha256", 64), ("sha256", 130)])
    assert again["missing"] == 0 and again["built"] == 0
    bad = CC.prewarm([("nonsense", 1)])
    assert bad["failed"] and bad["built"] == 0
    monkeypatch.delenv("ZKLC_CIRCUIT_CACHE")
    assert "skipped" in CC.prewarm([("sha256", 64)])
