"""GPU plonky2 prover (zklc_plonky2_prove) against the oracle: the proof bytes must equal the pure-Python prover
restatement's proof bit for bit, and the verifier restatement (pinned by the reference's golden proofs) must accept it."""
import json

import numpy as np
import pytest

import zklc_amd
from zklc_amd.plonky2 import CircuitBuilder, serialization as S, HASH_GL, HASH_BN128
from oracle import plonky2_prover as OP, plonky2_verifier as V

pytestmark = pytest.mark.gpu
HASHERS = {HASH_GL: V.HasherGL, HASH_BN128: V.HasherBN128}


def small_circuit(n_extra_rows=0):
    b = CircuitBuilder()
    x = b.add_virtual_public_input()
    y = b.add_virtual_target()
    z = b.mul(x, y)
    w = b.add(z, b.constant(5))
    b.split_le(x, 10)
    lo, hi = b.mul_add_u32(x, y, b.constant(77))
    s = b.sub(w, z)
    b.connect(s, b.constant(5))
    b.register_public_input(lo)
    acc = z
    for i in range(n_extra_rows * 20):
        acc = b.mul_add(acc, y, x)
    data = b.build()
    wires, pis = data.generate_witness({x: 1000, y: 4000000000})
    return data, wires, pis


@pytest.mark.parametrize("hasher", [HASH_GL, HASH_BN128])
@pytest.mark.parametrize("extra", [0, 40])
def test_proof_matches_oracle_bit_for_bit(zctx, hasher, extra):
    data, wires, pis = small_circuit(extra)
    common = data.common_data()
    H = HASHERS[hasher]
    prover = data.prover(zctx, hasher)
    proof_bytes = prover.prove_bytes(wires, pis)
    assert len(proof_bytes) == S.proof_size(common, hasher)
    proof = S.proof_from_bytes(proof_bytes, common, hasher)
    vd = prover.verifier_data()
    trace = {}
    oproof, ovd = OP.prove(common, data.constants, data.sigmas, wires, pis, H, trace=trace)
    assert vd == ovd, "constants_sigmas cap / circuit digest"
    ch = prover.last_challenges()
    nch = 2
    assert ch[:nch] == trace["betas"] and ch[nch:2 * nch] == trace["gammas"], "betas/gammas (wires commitment)"
    assert ch[2 * nch:3 * nch] == trace["alphas"], "alphas (Z / partial products commitment)"
    assert tuple(ch[3 * nch:3 * nch + 2]) == trace["zeta"], "zeta (quotient commitment)"
    assert tuple(ch[3 * nch + 2:3 * nch + 4]) == trace["fri_alpha"], "fri alpha (openings)"
    assert proof_bytes == S.proof_to_bytes(oproof, common, hasher), "proof bytes"
    V.verify(json.loads(json.dumps(proof)), vd, common)
    # proving twice gives the same bytes (no hidden state)
    assert prover.prove_bytes(wires, pis) == proof_bytes


def test_bad_witness_is_rejected(zctx):
    data, wires, pis = small_circuit()
    prover = data.prover(zctx, HASH_GL)
    from zklc_amd.plonky2 import gates as G
    row = next(r for r, (g, _) in enumerate(data.builder.rows) if isinstance(g, G.ArithmeticGate))
    bad = wires.copy()
    bad[3, row] ^= 1      # a copy-constrained output: the permutation product does not close -> ZKLC_ERR_INVALID_ARG
    with pytest.raises(zklc_amd.ZklcError):
        prover.prove_bytes(bad, pis)
    # EVERY connected wire is bound by the permutation argument, the union-find representatives of the copy classes included:
    # changing any single one of them (and nothing else) must break the grand product
    b = data.builder
    classes = {}
    for k in list(b.parent) + [b._find(k) for k in b.parent]:
        if k < (1 << 40):
            classes.setdefault(b._find(k), set()).add(k)
    connected = sorted(k for members in classes.values() if len(members) > 1 for k in members)
    assert len(connected) > 20
    for k in connected[::3]:
        bad = wires.copy()
        bad[k & 255, k >> 8] ^= 1
        with pytest.raises(zklc_amd.ZklcError):
            prover.prove_bytes(bad, pis)
    bad = wires.copy()
    bad[79, row] ^= 1     # only a gate constraint breaks: a proof comes out (as with plonky2) and the verifier rejects it
    proof = prover.prove(bad, pis)
    with pytest.raises(AssertionError, match="vanishing"):
        V.verify(json.loads(json.dumps(proof)), prover.verifier_data(), data.common_data())
    V.verify(json.loads(json.dumps(prover.prove(wires, pis))), prover.verifier_data(), data.common_data())


def _synthetic(shape, degree_bits, seed=3, npi=11):
    from zklc_amd.plonky2 import synthetic as SY, gates as G, standard_recursion_config, wide_ecc_config
    if shape == "recursion":
        cfg = standard_recursion_config()
        mix = SY.recursion_shape_mix(cfg) + [(G.ExponentiationGate(20), 3)]
    else:
        cfg = wide_ecc_config()
        mix = SY.ed25519_shape_mix(cfg)
    return SY.synthetic_circuit(degree_bits, cfg, mix, num_public_inputs=npi, seed=seed)


@pytest.mark.parametrize("shape,hasher", [("recursion", HASH_GL), ("recursion", HASH_BN128), ("ed25519", HASH_GL)])
def test_all_gate_types_match_oracle_bit_for_bit(zctx, shape, hasher):
    """every gate type of the two reference circuit shapes (19 in total), 2^6 rows: GPU proof == oracle proof"""
    data, wires, pis = _synthetic(shape, 6)
    common = data.common_data()
    prover = data.prover(zctx, hasher)
    proof_bytes = prover.prove_bytes(wires, pis)
    oproof, ovd = OP.prove(common, data.constants, data.sigmas, wires, pis, HASHERS[hasher])
    assert prover.verifier_data() == ovd
    assert proof_bytes == S.proof_to_bytes(oproof, common, hasher)
    V.verify(json.loads(json.dumps(S.proof_from_bytes(proof_bytes, common, hasher))), ovd, common)


@pytest.mark.parametrize("shape,degree_bits,hasher", [("recursion", 12, HASH_GL), ("recursion", 12, HASH_BN128), ("ed25519", 13, HASH_GL)])
def test_reference_sized_proofs_verify(zctx, shape, degree_bits, hasher):
    """proofs of the reference's recursion shape (2^12 x 135, both hashers) and a 2^13 x 234 slice of the Ed25519 shape are
    accepted by the verifier restatement (all 28 query rounds, vanishing identity, PoW)"""
    data, wires, pis = _synthetic(shape, degree_bits, seed=5, npi=16)
    common = data.common_data()
    prover = data.prover(zctx, hasher)
    proof = prover.prove(wires, pis)
    V.verify(json.loads(json.dumps(proof)), prover.verifier_data(), common)
    # tampering with one opened wire value must be caught
    proof["proof"]["openings"]["wires"][7][0] ^= 1
    with pytest.raises(AssertionError):
        V.verify(json.loads(json.dumps(proof)), prover.verifier_data(), common)


@pytest.mark.parametrize("shape,degree_bits", [("recursion", 12), ("ed25519", 13)])
def test_gpu_proof_bytes_equal_the_c_prover_at_reference_sizes(zctx, shape, degree_bits):
    """above ~2^8 rows the Python prover restatement is impractical and the tests relied on the verifier alone; the oracle's C
    prover (oracle/c/plonky2_prover_oracle.c, bytes equal to the Python restatement on small circuits: tests/test_oracle_cprover.py)
    gives byte-level parity at the reference's sizes: 2^12 x 135 (recursion shape) and a 2^13 x 234 slice of the Ed25519 shape"""
    from oracle import cport
    data, wires, pis = _synthetic(shape, degree_bits, seed=5, npi=16)
    prover = data.prover(zctx, HASH_GL)
    got = prover.prove_bytes(wires, pis)
    want, secs, vd = cport.plonky2_prove(data, wires, pis, verifier_data=True)
    assert prover.verifier_data() == vd
    assert got == want
    prover.close()


def test_reference_sha512_circuit_proof(zctx):
    """the reference's SHA-512 circuit (crypto/plonky2_sha512/src/circuit.rs, restated in zklc_amd/plonky2/sha512.py) with a
    real witness -- SHA-512 of a 105-byte NEAR Ed25519 preimage -- proven on the GPU; the verifier restatement accepts the
    proof and its 512 public inputs are the hashlib digest bits"""
    import hashlib
    from zklc_amd.plonky2 import sha512
    msg = bytes((11 * i + 5) & 0xFF for i in range(105))
    b = CircuitBuilder()
    message, digest = sha512.sha512_circuit(b, 8 * len(msg))
    for t in digest:
        b.register_public_input(t)
    data = b.build()
    wires, pis = data.generate_witness(dict(zip(message, sha512.array_to_bits(msg))))
    prover = data.prover(zctx, HASH_GL)
    proof = prover.prove(wires, pis)
    assert proof["public_inputs"] == sha512.array_to_bits(hashlib.sha512(msg).digest())
    V.verify(json.loads(json.dumps(proof)), prover.verifier_data(), data.common_data())
    print("sha512 circuit: 2^14 rows, proof stages", prover.last_timings())


def test_reference_ed25519_circuit_proof_of_a_near_mainnet_signature(zctx, approval_prover):
    """the reference's per-signature circuit (crypto/plonky2_ed25519/src/gadgets/eddsa.rs:34-85, restated in
    zklc_amd/plonky2/ed25519_circuit.py: SHA-512 + two point decompressions + windowed [h]A + fixed-base [s]B over
    non-native 2^255-19 arithmetic) on a real approval signature of the NEAR fixture data/*_small.json: host witness
    generation, GPU proof, verifier restatement accepts; public inputs = message bits then public-key bits.
    Building the 181k-row circuit and its witness in Python takes a few minutes of host time."""
    from conftest import load_golden
    from zklc_amd.plonky2 import ed25519_circuit as E, sha512
    j = load_golden("ed25519_near_c1_small.json")
    msg = bytes.fromhex(j["msg"])
    e = j["entries"][0]
    pk, sig = bytes.fromhex(e["validator_tail"])[1:33], bytes.fromhex(e["approval"])[2:]
    data, targets, prover, vd = approval_prover.ed25519_circuit(len(msg))
    assert data.n == 1 << 18 and data.num_public_inputs == 8 * len(msg) + 256
    import os
    import time
    # native witness generation (csrc/plonky2_witness.cpp): all three signatures of the fixture
    if data._program is None:
        data.witness_program(E.fill_ecdsa_targets(targets, msg, sig, pk))
    sigs = [(bytes.fromhex(x["approval"])[2:], bytes.fromhex(x["validator_tail"])[1:33]) for x in j["entries"]]
    t0 = time.time()
    wn, pn = data.generate_witness_native([E.fill_ecdsa_targets(targets, msg, s_, p_) for s_, p_ in sigs])
    print("native witness generation: %.2f s for %d signatures" % (time.time() - t0, len(sigs)))
    wires, pis = wn[0], [int(x) for x in pn[0]]
    assert pis == sha512.array_to_bits(msg) + sha512.array_to_bits(pk)
    if os.environ.get("ZKLC_SLOW_TESTS"):     # ~1 min of host Python: the Python generators give the same matrix cell for cell
        wp, pp = data.generate_witness(E.fill_ecdsa_targets(targets, msg, sig, pk))
        assert np.array_equal(wp, wires) and pp == pis
    # a corrupted signature has no witness (the reference's generators / prover fail the same way)
    bad = bytearray(sig)
    bad[40] ^= 1
    with pytest.raises(AssertionError):
        data.generate_witness_native([E.fill_ecdsa_targets(targets, msg, bytes(bad), pk)])
    V.verify(json.loads(json.dumps(prover.prove(wn[2], [int(x) for x in pn[2]]))), vd, data.common_data())
    if not os.environ.get("ZKLC_FAST_TESTS"):     # 1-2 minutes of host cores: byte parity at the real 2^18 x 234 circuit (20 gate types)
        from oracle import cport
        want, secs = cport.plonky2_prove(data, wires, pis)
        assert prover.prove_bytes(wires, pis) == want, "GPU proof of the Ed25519 circuit differs from the C prover's"
        print("C prover, Ed25519 circuit: %.1f s (+ %.1f s preprocessing); bytes equal" % (secs["proof"], secs["preprocess"]))
    proof = prover.prove(wires, pis)
    print("ed25519 circuit: 2^18 rows x 234 wires, 20 gate types; proof stages", prover.last_timings())
    V.verify(json.loads(json.dumps(proof)), vd, data.common_data())
    assert proof["public_inputs"] == pis


def test_prove_approvals_on_the_reference_small_fixture(zctx, approval_prover):
    """`prove_approvals` (near_bft_finality/src/prove_block_data/signatures.rs:43-141; its own test :287-367 runs it on
    data/*_small.json = BASELINE configs[0]): 3 approvals -> 3 Ed25519-circuit proofs, the left fold of two `recursive_proof`
    calls (in-circuit verification of an Ed25519 proof and of a recursion proof), and the closing recursion whose 32 public
    inputs are sha256(valid_keys).  Every recursion proof is checked by the verifier restatement."""
    import hashlib
    from conftest import load_golden, near_set_arrays
    from oracle import cport
    j = load_golden("ed25519_near_c1_small.json")
    msg, approvals, validators = near_set_arrays(j)
    rec = approval_prover.recursion
    seen = []
    orig = rec.recursive_proof

    def checked(first, second=None, public_inputs=None, **kw):
        # prove_approvals hands proofs over as `to_bytes` bytes between fold steps (raw=True): check what it really passes on
        rc, proof = orig(first, second, public_inputs, **kw)
        is_raw = isinstance(proof, (bytes, bytearray))
        as_json = S.proof_from_bytes(proof, rc.common, HASH_GL) if is_raw else proof
        V.verify(json.loads(json.dumps(as_json)), rc.verifier_only, rc.common)
        # byte-level parity of the REAL fold circuits -- R(ed, ed), R(R, ed) at 2^14 rows and the closing R(R) + 32 public inputs --
        # against the oracle's C prover on the very witness the GPU proved (the host interpreter left it in rc.wire_buffer())
        want, _ = cport.plonky2_prove(rc.data, rc.wire_buffer()[0].copy(), [int(x) for x in as_json["public_inputs"]])
        got = bytes(proof) if is_raw else S.proof_to_bytes(proof, rc.common, HASH_GL)
        assert got == want, "GPU proof of the %d-row recursion circuit differs from the C prover's" % rc.data.n
        seen.append((rc.data.n, len(rc.common["gates"]), rc.prover.last_timings()["total"]))
        return rc, proof
    rec.recursive_proof = checked
    try:
        (rc, proof), valid_keys = approval_prover.prove_approvals(msg, approvals, validators)
    finally:
        rec.recursive_proof = orig
    n_present = sum(1 for a in approvals if len(a) == 66)
    assert len(seen) == n_present                      # n - 1 folds + the closing proof
    assert len(valid_keys) == 33 * n_present
    assert proof["public_inputs"] == list(hashlib.sha256(valid_keys).digest())
    print("prove_approvals: recursion proofs (rows, gate types, GPU ms):", seen)
    # a tampered inner proof has no witness: recursion.rs:127-158 (test_recursive_proof_invalid)
    bad = json.loads(json.dumps(proof))
    bad["public_inputs"][-1] = 10000
    with pytest.raises(AssertionError):
        rec.recursive_proof((rc.common, rc.verifier_only, bad))
    # the last recursion of the reference (bin/prove_block.rs:279-287): the same verifier circuit, Poseidon-BN128 Merkle caps
    from zklc_amd.plonky2.recursion import RecursionProver
    wrap = RecursionProver(zctx, HASH_BN128)
    wrc, wproof = wrap.recursive_proof((rc.common, rc.verifier_only, proof))
    V.verify(json.loads(json.dumps(wproof)), wrc.verifier_only, wrc.common)
    golden = load_golden("plonky2_near_random_CGZP.json")["common_data"]
    assert wrc.common["gates"] == golden["gates"]      # the wrap circuit has the gate list of the reference's final proofs
    wrap.close()


def test_tree_aggregation_matches_the_left_fold(zctx, approval_prover):
    """SURVEY 8f.4 on ONE GPU: `prove_approvals(.., tree=True)` aggregates the signature proofs pairwise instead of by the serial chain
    of signatures.rs:97-105.  Five real approvals of the 100-validator mainnet fixture (tree: ((p0 p1)(p2 p3)) p4 -- the shapes R(ed, ed),
    R(R, R), R(R, ed)): the tree-folded closing proof is accepted by the verifier restatement and carries the SAME public inputs
    (sha256(valid_keys)) and valid_keys as the left fold's, which is proven and verified beside it."""
    import hashlib
    from conftest import load_golden, near_set_arrays
    j = load_golden("ed25519_near_c2_100.json")
    msg, approvals, validators = near_set_arrays(j)
    present = [i for i, a in enumerate(approvals) if len(a) == 66][:5]
    keep = sorted(present)
    approvals5 = [approvals[i] if i in keep else b"\x00" for i in range(len(approvals))]
    results = {}
    for tree in (False, True):
        (rc, proof), valid_keys = approval_prover.prove_approvals(msg, approvals5, validators, tree=tree)
        V.verify(json.loads(json.dumps(proof)), rc.verifier_only, rc.common)
        results[tree] = (proof["public_inputs"], valid_keys, rc.data.n)
    assert results[True][0] == results[False][0] == list(hashlib.sha256(results[True][1]).digest())
    assert results[True][1] == results[False][1] and len(results[True][1]) == 33 * 5


def test_two_message_lengths_on_one_approval_prover(zctx, approval_prover):
    """the Endorsement (41-byte) and Skip (17-byte) circuits have the same shape (2^18 x 234): an ApprovalProver that alternates between
    them keeps one device wire matrix PER circuit (cells one program writes and the other does not would otherwise go stale) -- the
    proofs made from the device witnesses must be byte-identical to the ones made from the host interpreter's matrices.
    (The second circuit is another minute of host Python on a cold circuit cache; ZKLC_FAST_TESTS=1 skips.)"""
    import os
    if os.environ.get("ZKLC_FAST_TESTS"):
        pytest.skip("ZKLC_FAST_TESTS: builds the second 2^18-row Ed25519 circuit")
    from conftest import load_golden
    from zklc_amd.plonky2 import ed25519_circuit as E
    sets = []
    for name in ("ed25519_near_c1_small.json", "ed25519_near_c1_small_skip.json"):
        j = load_golden(name)
        e = j["entries"][0]
        sets.append((bytes.fromhex(j["msg"]), bytes.fromhex(e["approval"])[2:], bytes.fromhex(e["validator_tail"])[1:33]))
    assert len(sets[0][0]) != len(sets[1][0])
    for msg, sig, pk in sets + sets:          # A, B, A, B on the same prover
        (common, vd, got), = approval_prover.ed25519_proofs(msg, [sig], [pk])
        data, targets, prover, _ = approval_prover.ed25519_circuit(len(msg))
        wn, pn = data.generate_witness_native([E.fill_ecdsa_targets(targets, msg, sig, pk)])
        assert got == prover.prove_bytes(wn[0], [int(x) for x in pn[0]])


def test_full_block_proof_on_a_mainnet_window(zctx, block_prover):
    """`prove_block_bft` (near_bft_finality/src/prove_bft/bft.rs:38-500, the path of bin/prove_random.rs) on the reference's own
    data set -- NEAR mainnet blocks 121798939..43 of epoch HPi5.., its 100 block producers, Block_0 of the previous epoch and the
    last block of the one before (tests/golden/block_window_HPi5.json: borsh bytes rebuilt from the JSON views and pinned by the
    block hashes): 73 Ed25519-circuit proofs and their fold, keys / stakes, seven header-hash chains, bp_hash, equalities,
    consecutive heights and the recursions that join them.  The final proof is accepted by the verifier restatement and its
    public inputs are [0, hash(Block_i), hash(Block_n-1(epoch i-2)), hash(Block_0(epoch i-1))]."""
    import time
    from conftest import load_golden
    from zklc_amd.prove_bft import BlockProver
    w = load_golden("block_window_HPi5.json")
    hx = bytes.fromhex
    blocks = []
    for blk in w["blocks"]:
        f = {k: hx(blk[k]) for k in ("hash", "prev_hash", "epoch_id", "last_ds_final_hash", "last_final_hash")}
        f["height"] = blk["height"]
        f["approvals"] = [hx(a) for a in blk["approvals"]]
        blocks.append((f, hx(blk["bytes"])))
    validators = [hx(v) for v in w["validators"]]
    bp = block_prover
    bp.counts, bp.seconds = {}, {}
    t0 = time.time()
    bi, none = bp.prove_block_bft(hx(w["ep2_last_block"]["bytes"]), hx(w["ep2_last_block"]["hash"]), hx(w["ep1_first_block"]["bytes"]),
                                  hx(w["ep1_first_block"]["hash"]), blocks, validators)
    dt = time.time() - t0
    assert none is None
    import conftest
    conftest.STASH["HPi5"] = (bi, None)
    V.verify(json.loads(json.dumps(bi[2])), bi[1], bi[0])
    want = [0] + list(hx(w["blocks"][4]["hash"])) + list(hx(w["ep2_last_block"]["hash"])) + list(hx(w["ep1_first_block"]["hash"]))
    assert bi[2]["public_inputs"] == want
    # bin/prove_block.rs:279-287: the last recursion, Poseidon-BN128 config, carries the block proof's public inputs.  Its shape must be
    # the shape of the reference's final proofs (near_bft_finality/proofs/random/CGZP.../{common_data.json, proof.bin}: degree_bits 12,
    # 97 public inputs, 13 gate types, 127 968 bytes): a gnark verifier circuit compiled from that common_data accepts exactly this
    from zklc_amd.plonky2.recursion import RecursionProver
    import os as _os
    golden = load_golden("plonky2_near_random_CGZP.json")["common_data"]
    wrap = RecursionProver(zctx, HASH_BN128)
    wrc, wraw = wrap.recursive_proof(bi, None, list(bi[2]["public_inputs"]), raw=True)
    assert {k: wrc.common[k] for k in golden} == golden, "wrap circuit common_data differs from the reference's final proof"
    golden_bin = _os.path.join(_os.path.dirname(__file__), "golden", "plonky2_near_random_CGZP_proof.bin")
    assert len(wraw) == _os.path.getsize(golden_bin) == 127968
    wjson = S.proof_from_bytes(wraw, wrc.common, HASH_BN128)
    V.verify(json.loads(json.dumps(wjson)), wrc.verifier_only, wrc.common)
    assert wjson["public_inputs"] == want
    wrap.close()
    n_present = sum(1 for a in blocks[3][0]["approvals"] if len(a) == 66)
    print("full Block_i proof, first call (circuits of %d SHA-256 sizes built in Python): %.1f s; %d approvals; counts %s; seconds %s"
          % (len(bp.hashes.sha._circuits), dt, n_present, bp.counts, {k: round(v, 1) for k, v in bp.seconds.items()}))
    import os
    if not os.environ.get("ZKLC_SLOW_TESTS"):
        return
    # second call: every circuit except the keys / stakes one is resident
    bp.counts, bp.seconds = {}, {}
    t0 = time.time()
    bi2, _ = bp.prove_block_bft(hx(w["ep2_last_block"]["bytes"]), hx(w["ep2_last_block"]["hash"]), hx(w["ep1_first_block"]["bytes"]),
                                hx(w["ep1_first_block"]["hash"]), blocks, validators)
    assert bi2[2] == bi[2]
    print("full Block_i proof, circuits resident, sequential host driver: %.1f s; seconds %s"
          % (time.time() - t0, {k: round(v, 1) for k, v in bp.seconds.items()}))


QUOTIENT_VARIANTS = [{}, {"ZKLC_P2_POSEIDON_GATE": "plain"}, {"ZKLC_P2_POSEIDON_GATE": "lazy"}, {"ZKLC_P2_POSEIDON_GATE": "lazy1"},
                     {"ZKLC_P2_ADDMANY": "pergate"}, {"ZKLC_P2_ADDMANY": "multi"}, {"ZKLC_P2_GATE_LAUNCH": "single"},
                     {"ZKLC_P2_QUOTIENT": "fused"}, {"ZKLC_MERKLE_FUSED": "0"},
                     {"ZKLC_NTT_ZSKIP": "0"}, {"ZKLC_SETTLED_COPIES": "0"}]        # round 6: the general first LDE group, parked read-backs

_VARIANT_CHILD = r'''
import hashlib, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(tests)r)
import zklc_amd
from zklc_amd.plonky2 import HASH_GL
from test_gpu_plonky2 import _synthetic
data, wires, pis = _synthetic("ed25519", 13, seed=5, npi=16)
with zklc_amd.Context(0) as ctx:
    prover = data.prover(ctx, HASH_GL)
    print("DIGEST " + hashlib.sha256(prover.prove_bytes(wires, pis)).hexdigest())
    prover.close()
'''


def test_every_quotient_evaluator_variant_gives_the_same_proof_bytes(zctx):
    """The library reads its A/B switches once per process (ZKLC_P2_POSEIDON_GATE = loose (default) / plain / lazy / lazy1,
    ZKLC_P2_ADDMANY = tile (default) / pergate / multi, per-gate launches, the fused quotient kernel, the per-level Merkle form):
    one child process per setting proves the 2^13 x 234 slice of the Ed25519 shape (all 20 gate types) and every one must give
    the bytes of the C prover (crypto/plonky2_u32/src/gates/add_many_u32.rs:270-284 and the Poseidon gate are the evaluators that
    have variants)."""
    import hashlib
    import os
    import subprocess
    import sys
    from oracle import cport
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data, wires, pis = _synthetic("ed25519", 13, seed=5, npi=16)
    want, _ = cport.plonky2_prove(data, wires, pis)
    want = hashlib.sha256(want).hexdigest()
    code = _VARIANT_CHILD % {"root": root, "tests": os.path.join(root, "tests")}
    keys = sorted({k for v in QUOTIENT_VARIANTS for k in v})
    procs = []
    for var in QUOTIENT_VARIANTS:
        env = {k: v for k, v in os.environ.items() if k not in keys}
        env.update(var)
        procs.append((var, subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                            env=env, cwd=root)))
    for var, p in procs:
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, (var, se[-2000:])
        got = next(ln for ln in so.splitlines() if ln.startswith("DIGEST "))[7:]
        assert got == want, "proof bytes under %r differ from the C prover's" % (var,)
