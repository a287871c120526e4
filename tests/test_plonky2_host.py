"""Host side of the plonky2 path without a GPU: proof serialisation pinned by the reference's golden proof.bin / proof.json
pair, the circuit builder + prover restatement + verifier restatement chain on small circuits (both hash configurations),
synthetic circuits of both reference shapes (every gate type), and rejection of bad witnesses / tampered proofs."""
import copy
import json
import os

import numpy as np
import pytest

from conftest import load_golden, GOLDEN
import zklc_amd
from zklc_amd.plonky2 import CircuitBuilder, serialization as S, synthetic as SY, gates as G, standard_recursion_config, wide_ecc_config
from zklc_amd.plonky2 import poseidon_gate_rows
from oracle import plonky2_prover as OP, plonky2_verifier as V, poseidon_gl as pgl, plonky2_gates as OG


def test_proof_bin_format_is_pinned_by_the_golden_pair():
    j = load_golden("plonky2_near_random_CGZP.json")
    raw = open(os.path.join(GOLDEN, "plonky2_near_random_CGZP_proof.bin"), "rb").read()
    common = j["common_data"]
    assert len(raw) == 127968 == S.proof_size(common, S.HASH_BN128)
    pj = S.proof_from_bytes(raw, common, S.HASH_BN128)
    gold = j["proof"]
    assert pj["public_inputs"] == gold["public_inputs"]
    for k in ["wires_cap", "plonk_zs_partial_products_cap", "quotient_polys_cap", "openings"]:
        assert pj["proof"][k] == gold["proof"][k], k
    gop, pop = gold["proof"]["opening_proof"], pj["proof"]["opening_proof"]
    assert pop["commit_phase_merkle_caps"] == gop["commit_phase_merkle_caps"] and pop["final_poly"] == gop["final_poly"]
    assert pop["pow_witness"] == gop["pow_witness"]
    kept = j["kept_query_rounds"]
    assert pop["query_round_proofs"][:kept] == gop["query_round_proofs"]
    assert S.proof_to_bytes(pj, common, S.HASH_BN128) == raw
    # and the parsed binary verifies (all 28 rounds)
    V.verify(pj, j["verifier_data"], common)


def test_poseidon_gate_rows_match_the_permutation_and_the_gate_constraints():
    rng = np.random.default_rng(3)
    ins = rng.integers(0, pgl.P, (5, 12), dtype=np.uint64)
    swap = np.array([0, 1, 0, 1, 1], dtype=np.uint64)
    rows = poseidon_gate_rows(ins, swap)
    gate = OG.PoseidonGate()
    for k in range(5):
        st = [int(x) for x in ins[k]]
        if swap[k]:
            st = st[4:8] + st[:4] + st[8:]
        assert [int(x) for x in rows[k][12:24]] == pgl.permute(st)
        assert all(c == 0 for c in gate.eval(OG.BaseK, [], [int(x) for x in rows[k]], [0] * 4))


def _small():
    b = CircuitBuilder()
    x = b.add_virtual_public_input()
    y = b.add_virtual_target()
    z = b.mul(x, y)
    w = b.add(z, b.constant(5))
    b.split_le(x, 10)
    lo, hi = b.mul_add_u32(x, y, b.constant(77))
    b.connect(b.sub(w, z), b.constant(5))
    b.register_public_input(lo)
    b.register_public_input(hi)
    data = b.build()
    wires, pis = data.generate_witness({x: 1000, y: 4000000000})
    return data, wires, pis


@pytest.mark.parametrize("H,hasher", [(V.HasherGL, S.HASH_GL), (V.HasherBN128, S.HASH_BN128)])
def test_builder_prover_verifier_chain(H, hasher):
    data, wires, pis = _small()
    assert pis == [1000, (1000 * 4000000000 + 77) & 0xFFFFFFFF, (1000 * 4000000000 + 77) >> 32]
    common = data.common_data()
    proof, vd = OP.prove(common, data.constants, data.sigmas, wires, pis, H)
    V.verify(json.loads(json.dumps(proof)), vd, common)
    raw = S.proof_to_bytes(proof, common, hasher)
    assert len(raw) == S.proof_size(common, hasher)
    assert json.loads(json.dumps(S.proof_from_bytes(raw, common, hasher))) == json.loads(json.dumps(proof))
    bad = copy.deepcopy(proof)
    bad["public_inputs"][1] ^= 1
    with pytest.raises(AssertionError):
        V.verify(json.loads(json.dumps(bad)), vd, common)


def test_unsatisfied_witness_is_rejected_by_the_prover_restatement():
    data, wires, pis = _small()
    row = next(r for r, (g, _) in enumerate(data.builder.rows) if isinstance(g, G.ArithmeticGate))
    common = data.common_data()
    bad = wires.copy()
    bad[3, row] ^= 1      # a copy-constrained output: the permutation product does not close, the prover refuses
    with pytest.raises(AssertionError):
        OP.prove(common, data.constants, data.sigmas, bad, pis, V.HasherGL)
    bad = wires.copy()
    bad[79, row] ^= 1     # the free output of an unused op: only a gate constraint breaks -> like plonky2, the prover still
    proof, vd = OP.prove(common, data.constants, data.sigmas, bad, pis, V.HasherGL)   # emits a proof, the verifier rejects it
    with pytest.raises(AssertionError, match="vanishing"):
        V.verify(json.loads(json.dumps(proof)), vd, common)


@pytest.mark.parametrize("shape", ["recursion", "ed25519"])
def test_synthetic_reference_shapes_prove_and_verify(shape):
    if shape == "recursion":
        cfg = standard_recursion_config()
        mix = SY.recursion_shape_mix(cfg) + [(G.ExponentiationGate(20), 3)]
    else:
        cfg = wide_ecc_config()
        mix = SY.ed25519_shape_mix(cfg)
    data, wires, pis = SY.synthetic_circuit(6, cfg, mix, num_public_inputs=11, seed=7)
    common = data.common_data()
    assert len(common["gates"]) == len(mix) + 3 - (1 if shape == "recursion" else 0)   # + Noop, PublicInput (+ Poseidon)
    # every row satisfies its gate (evaluated with the oracle, on the subgroup)
    gates = [OG.gate_from_id(g) for g in common["gates"]]
    nsel = len(common["selectors_info"]["groups"])
    pih = pgl.hash_no_pad(pis)
    for r in range(data.n):
        consts = [int(x) for x in data.constants[:, r]]
        cs = OG.evaluate_gate_constraints(OG.BaseK, gates, common["selectors_info"], common["num_gate_constraints"], consts,
                                          [int(x) for x in wires[:, r]], pih)
        assert not any(cs), "row %d" % r
    proof, vd = OP.prove(common, data.constants, data.sigmas, wires, pis, V.HasherGL)
    V.verify(json.loads(json.dumps(proof)), vd, common)


def test_selector_groups_reproduce_the_reference_common_data():
    """the gate order and selector groups computed by the builder equal those of the reference's golden common_data"""
    j = load_golden("plonky2_near_random_CGZP.json")["common_data"]
    cfg = standard_recursion_config()
    data, _, _ = SY.synthetic_circuit(6, cfg, SY.recursion_shape_mix(cfg), num_public_inputs=16, seed=1)
    c = data.common_data()
    assert c["gates"] == j["gates"]
    assert c["selectors_info"] == j["selectors_info"]
    for k in ["num_constants", "num_partial_products", "quotient_degree_factor", "num_gate_constraints", "k_is", "config"]:
        assert c[k] == j[k], k
    from zklc_amd.plonky2.builder import fri_reduction_arity_bits
    assert fri_reduction_arity_bits(cfg, 12) == j["fri_params"]["reduction_arity_bits"] == [4, 4]
    assert fri_reduction_arity_bits(cfg, 17) == [4, 4, 4]


def test_sha512_circuit_witness_matches_hashlib():
    """crypto/plonky2_sha512/src/circuit.rs restated on the host builder: the generated witness carries SHA-512 of NEAR's
    105-byte Ed25519 preimage (R || A || M, one block) in its public inputs, and sampled rows satisfy their gates"""
    import hashlib
    import random
    from zklc_amd.plonky2 import sha512
    msg = bytes((7 * i + 3) & 0xFF for i in range(105))
    b = CircuitBuilder()
    message, digest = sha512.sha512_circuit(b, 8 * len(msg))
    assert len(message) == 840 and len(digest) == 512
    for t in digest:
        b.register_public_input(t)
    data = b.build()
    assert data.n == 1 << 14
    wires, pis = data.generate_witness(dict(zip(message, sha512.array_to_bits(msg))))
    assert pis == sha512.array_to_bits(hashlib.sha512(msg).digest())
    common = data.common_data()
    gates = [OG.gate_from_id(g) for g in common["gates"]]
    pih = pgl.hash_no_pad(pis)
    rng = random.Random(1)
    for r in rng.sample(range(data.n), 60) + [data.n - 1]:
        cs = OG.evaluate_gate_constraints(OG.BaseK, gates, common["selectors_info"], common["num_gate_constraints"],
                                          [int(x) for x in data.constants[:, r]], [int(x) for x in wires[:, r]], pih)
        assert not any(cs), "row %d" % r


def test_native_witness_interpreter_matches_the_python_generators():
    """csrc/plonky2_witness.cpp (zklc_plonky2_witness_run) against the builder's Python generators on a circuit that uses the
    non-native field, point decompression, curve doubling / conditional addition, random access, comparison, division and
    u32 gadgets of the Ed25519 circuit; an undecodable public key has no witness"""
    import random
    from zklc_amd.plonky2 import ed25519_circuit as E, sha512
    rng = random.Random(1)
    b = CircuitBuilder(wide_ecc_config())
    g = E.Gadgets(b)
    xa, ya = g.virtual_biguint(8), g.virtual_biguint(8)
    for f in (g.add_nonnative, g.sub_nonnative, g.mul_nonnative):
        f(xa, ya)
    g.inv_nonnative(xa)
    g.neg_nonnative(ya)
    pk_bits = b.add_virtual_targets(256)
    pt = g.point_decompress(pk_bits)
    dbl = g.curve_double(pt)
    w4 = g.split_nonnative_to_4_bit_limbs(xa)
    sel = g.random_access_curve_points(w4[0], [g.constant_affine_point(E.pt_mul(i + 1, E.BASE)) for i in range(16)])
    g.curve_conditional_add(dbl, sel, b.not_(b.is_equal(w4[1], b.zero())))
    for t in g.reduce(xa + ya, E.L25519):
        b.register_public_input(t)
    data = b.build()
    j = load_golden("ed25519_near_c2_100.json")
    pks = [bytes.fromhex(e["validator_tail"])[1:33] for e in j["entries"][:4]]

    def inputs(x, y, pk):
        d = dict(zip(xa, E.limbs_of(x, 8)))
        d.update(zip(ya, E.limbs_of(y, 8)))
        d.update(zip(pk_bits, E.bits_in_le(sha512.array_to_bits(pk))))
        return d
    ins = [inputs(rng.randrange(E.P25519), rng.randrange(E.P25519), pk) for pk in pks]
    ins.append(inputs(E.P25519 - 1, 1, pks[0]))
    # the program traced from a run of the Python generators and the one assembled from the declared outputs give one witness
    data.witness_program(ins[0], trace_python=True)
    wt, pt_ = data.generate_witness_native(ins[:2], threads=2)
    data._program = None
    data.witness_program(list(ins[0]))
    wn, pn = data.generate_witness_native(ins, threads=2)
    assert np.array_equal(wt, wn[:2]) and np.array_equal(pt_, pn[:2])
    for i, inp in enumerate(ins):
        wp, pp = data.generate_witness(inp)
        assert np.array_equal(wp, wn[i]) and pp == [int(x) for x in pn[i]], i
    x, y = E.value_of(ins[1][t] for t in xa), E.value_of(ins[1][t] for t in ya)
    assert E.value_of(pn[1]) == (x + (y << 256)) % E.L25519
    bad = dict(ins[0])
    bad[pk_bits[5]] ^= 1
    with pytest.raises(AssertionError, match="decompression|copy constraint"):
        data.generate_witness_native([bad])


def test_sigma_cycles_are_exactly_the_copy_classes():
    """every wire of a copy class -- including the class representative of the union-find -- lies on one cycle of sigma, and
    wires of different classes lie on different cycles (the permutation argument enforces every `connect`)"""
    from zklc_amd.plonky2.builder import P
    b = CircuitBuilder()
    x, y = b.add_virtual_target(), b.add_virtual_target()
    z = b.mul(x, y)
    z2 = b.mul(z, x)
    bits = b.split_le(z2, 64)
    b.connect(b.le_sum(bits[:20]), b.add(x, y))
    h = b.hash_n_to_hash_no_pad([x, y, z, z2])
    b.register_public_input(h[0])
    data = b.build()
    n, routed = data.n, b.config["num_routed_wires"]
    lookup = {data.k_is[j] * data.subgroup[i] % P: (j, i) for j in range(routed) for i in range(n)}
    sigma = {(j, i): lookup[int(data.sigmas[j, i])] for j in range(routed) for i in range(n)}
    assert sorted(sigma.values()) == sorted(sigma.keys())                 # a permutation
    cycle_of = {}
    for start in sigma:
        if start in cycle_of:
            continue
        cur = start
        while cur not in cycle_of:
            cycle_of[cur] = start
            cur = sigma[cur]
    classes = {}
    for k in list(b.parent) + [b._find(k) for k in b.parent]:
        if k < (1 << 40):
            classes.setdefault(b._find(k), set()).add((k & 255, k >> 8))
    assert len(classes) > 10
    seen = set()
    for members in classes.values():
        ids = {cycle_of[m] for m in members}
        assert len(ids) == 1, "a copy class is split over several sigma cycles"
        assert not (ids & seen), "two copy classes share a sigma cycle"
        seen |= ids
        cyc = {c for c, s_ in cycle_of.items() if s_ in ids}
        assert cyc == members


def test_proof_files_round_trip_with_the_golden_proof(tmp_path):
    """write_proof_files / read_proof_files (prove_block.rs:320-458 layout): the golden proof.bin of the reference goes through
    unchanged, the JSON side is the golden proof.json, and hash.json is the block hash carried in public inputs 1..33"""
    j = load_golden("plonky2_near_random_CGZP.json")
    raw = open(os.path.join(GOLDEN, "plonky2_near_random_CGZP_proof.bin"), "rb").read()
    common, vd = j["common_data"], j["verifier_data"]
    S.write_proof_files(str(tmp_path), common, vd, raw, S.HASH_BN128)
    assert open(tmp_path / "proof.bin", "rb").read() == raw
    c2, v2, p2 = S.read_proof_files(str(tmp_path), S.HASH_BN128)
    assert c2 == common and v2 == vd
    assert S.proof_to_bytes(p2, common, S.HASH_BN128) == raw
    assert p2["public_inputs"] == j["proof"]["public_inputs"]
    hash_hex = open(tmp_path / "hash.json").read()
    assert bytes.fromhex(hash_hex) == bytes(int(x) for x in j["proof"]["public_inputs"][1:33])
    import base58_min
    assert base58_min.encode(bytes.fromhex(hash_hex)).startswith("CGZP")      # the directory name of the golden proof
