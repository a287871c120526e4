"""The drop-in seam exercised by a NATIVE caller: tests/c_abi/prove_from_file.c -- plain C against include/zklc.h
and libzklc_mi355.so, no Python, no HIP headers -- loads a circuit container + a witness-input file, creates circuit and witness
program on the GPU, generates the witness there, proves and writes proof.bin.  Its bytes must equal (i) what the Python host mirror
gets for the same circuit and inputs and (ii) what the oracle's C prover gets on the CPU, at the fold shape (2^14 x 135, the step
`prove_approvals` repeats per signature: signatures.rs:97-105 -> recursion.rs:95) and at the per-signature Ed25519 shape
(2^18 x 234: ed25519.rs:60)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("c_abi") / "prove_from_file"
    lib_dir = os.path.join(ROOT, "zk-light-client-implementation_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "prove_from_file.c"), "-o", str(out), "-L", lib_dir, "-lzklc_mi355",
                           "-Wl,-rpath," + lib_dir])
    return str(out)


def run(exe, *args):
    # a separate process with its own HIP runtime (the system one, /opt/rocm: the binary knows nothing of torch)
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    print(r.stderr.strip().splitlines()[-2])
    return r.stderr


def test_ed25519_shape_from_files(zctx, approval_prover, exe, tmp_path):
    """2^18 x 234, 20 gate types, the three real NEAR approval signatures of fixture C1 as ONE batch of device witnesses"""
    from oracle import cport
    from zklc_amd.plonky2 import container as C, ed25519_circuit as E, sha512
    j = load_golden("ed25519_near_c1_small.json")
    msg = bytes.fromhex(j["msg"])
    data, targets, prover, vd = approval_prover.ed25519_circuit(len(msg))
    assert data.n == 1 << 18 and data.config["num_wires"] == 234
    circ, inp, out = tmp_path / "ed25519.zkcc", tmp_path / "ed25519_inputs.zkcc", tmp_path / "proof.bin"
    data.save(circ, targets, note="ed25519_circuit, %d-byte message" % len(msg))
    fills = [E.fill_ecdsa_targets(targets, msg, bytes.fromhex(x["approval"])[2:], bytes.fromhex(x["validator_tail"])[1:33]) for x in j["entries"]]
    vals = np.array([[int(w[t]) for t in data._program["input_targets"]] for w in fills], dtype=np.uint64)
    C.write_input_values(inp, vals)
    log = run(exe, circ, inp, out, "--repeat", "3")
    assert "2^18 rows x 234 wires" in log and "3 witness(es)" in log
    raw = out.read_bytes()
    assert len(raw) == 3 * prover.proof_bytes
    got = [raw[i * prover.proof_bytes:(i + 1) * prover.proof_bytes] for i in range(3)]
    # (i) the Python path: host interpreter witness + the prover the session holds
    wn, pn = data.generate_witness_native(fills)
    for k in range(3):
        assert got[k] == prover.prove_bytes(wn[k], [int(x) for x in pn[k]]), "native caller's proof %d differs from the Python path's" % k
    assert [int(x) for x in pn[0]] == sha512.array_to_bits(msg) + sha512.array_to_bits(bytes.fromhex(j["entries"][0]["validator_tail"])[1:33])
    # the digest the binary printed is the circuit's
    assert "circuit_digest:" + "".join("%016x" % int.from_bytes(int(x).to_bytes(8, "little"), "big") for x in vd["circuit_digest"]["elements"]) in log
    # (ii) the oracle's C prover on the CPU (1-2 minutes of host cores)
    if not os.environ.get("ZKLC_FAST_TESTS"):
        want, _ = cport.plonky2_prove(data, wn[0], [int(x) for x in pn[0]])
        assert got[0] == want, "native caller's proof differs from the oracle C prover's"
    # the host-interpreter path of the same binary gives the same bytes
    out2 = tmp_path / "proof_host.bin"
    C.write_input_values(inp, vals[:1])
    run(exe, circ, inp, out2, "--host-witness")
    assert out2.read_bytes() == got[0]


def test_fold_shape_from_files(zctx, approval_prover, exe, tmp_path):
    """R(R, ed): the in-circuit verifier over a recursion proof and an Ed25519 proof -- 2^14 x 135, 13 gate types"""
    from oracle import cport, plonky2_verifier as V
    from zklc_amd.plonky2 import container as C, HASH_GL, serialization as S
    j = load_golden("ed25519_near_c1_small.json")
    msg = bytes.fromhex(j["msg"])
    sigs = [bytes.fromhex(x["approval"])[2:] for x in j["entries"]]
    pks = [bytes.fromhex(x["validator_tail"])[1:33] for x in j["entries"]]
    ed = approval_prover.ed25519_proofs(msg, sigs, pks)
    rp = approval_prover.recursion
    rc1, p1 = rp.recursive_proof(ed[0], ed[1], raw=True)
    r1 = (rc1.common, rc1.verifier_only, p1)
    rc, want_py = rp.recursive_proof(r1, ed[2], raw=True)
    assert rc.data.n == 1 << 14 and rc.data.config["num_wires"] == 135
    vals = rc.input_vector([(r1[1], r1[2]), (ed[2][1], ed[2][2])], [])
    circ, inp, out = tmp_path / "fold.zkcc", tmp_path / "fold_inputs.zkcc", tmp_path / "proof.bin"
    rc.data.save(circ, rc.targets, note="recursive_proof R(R, ed)")
    C.write_input_values(inp, vals[None, :])
    log = run(exe, circ, inp, out, "--repeat", "5")
    assert "2^14 rows x 135 wires" in log
    got = out.read_bytes()
    assert got == bytes(want_py), "native caller's fold proof differs from RecursionProver's"
    wires = rc.wire_buffer()[0].copy()           # the witness the host interpreter left there for want_py
    want_c, _ = cport.plonky2_prove(rc.data, wires, [])
    assert got == want_c, "native caller's fold proof differs from the oracle C prover's"
    V.verify(json.loads(json.dumps(S.proof_from_bytes(got, rc.common, HASH_GL))), rc.verifier_only, rc.common)
    # the container alone rebuilds the host mirror's view: same common data, and a prover created from the FILE gives the same bytes
    from zklc_amd.plonky2.builder import CircuitData
    d2, t2 = CircuitData.load(circ)
    assert d2.common_data() == rc.common
    p2 = d2.prover(zctx, HASH_GL)
    assert p2.verifier_data() == rc.verifier_only
    w2, pi2 = d2.generate_witness_native(None, input_values=vals[None, :])
    assert p2.prove_bytes(w2[0], []) == got
    p2.close()
    # a tampered inner proof has no witness: the binary says so and writes nothing
    bad = vals.copy()
    bad[len(bad) // 2] ^= np.uint64(1)
    C.write_input_values(inp, bad[None, :])
    r = subprocess.run([exe, str(circ), str(inp), str(tmp_path / "none.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and not (tmp_path / "none.bin").exists()
