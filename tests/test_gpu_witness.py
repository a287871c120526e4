"""Witness generation on the GPU (csrc/plonky2_witness_dev.hip) against the host interpreter (same instruction
semantics, csrc/plonky2_witness_ops.h; itself equal to the builder's Python generators, tests/test_plonky2_host.py): the wire
matrices produced in HBM must equal the host interpreter's cell for cell, for every batch size, and a partial witness with no
completion (undecodable key, corrupted signature, tampered inner proof) must fail on the device as it does on the host."""
import json

import numpy as np
import pytest
import torch

import zklc_amd
from zklc_amd.plonky2 import CircuitBuilder, HASH_GL, wide_ecc_config
from conftest import load_golden
from oracle import plonky2_verifier as V

pytestmark = pytest.mark.gpu


def device_witness(dw, inputs_list=None, input_values=None):
    k = len(inputs_list) if inputs_list is not None else len(input_values)
    d = torch.zeros((k, dw.num_wires, dw.n_rows), dtype=torch.int64, device="cuda")
    pis = dw.run(d.data_ptr(), inputs_list, input_values)
    torch.cuda.synchronize()
    return d, pis


def test_device_witness_equals_host_interpreter_on_the_nonnative_gadgets(zctx):
    """the circuit of test_native_witness_interpreter_matches_the_python_generators: non-native add / sub / mul / inverse,
    point decompression, curve doubling / conditional addition, random access, comparison, division, u32 gadgets"""
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
    pks = [bytes.fromhex(e["validator_tail"])[1:33] for e in j["entries"][:5]]

    def inputs(x, y, pk):
        d = dict(zip(xa, E.limbs_of(x, 8)))
        d.update(zip(ya, E.limbs_of(y, 8)))
        d.update(zip(pk_bits, E.bits_in_le(sha512.array_to_bits(pk))))
        return d
    ins = [inputs(rng.randrange(E.P25519), rng.randrange(E.P25519), pk) for pk in pks]
    ins.append(inputs(E.P25519 - 1, 1, pks[0]))
    data.witness_program(list(ins[0]))
    wn, pn = data.generate_witness_native(ins, threads=2)
    dw = data.device_witness(zctx)
    info = dw.info(len(ins))
    print("device witness program:", info)
    assert info["levels"] > 10 and info["launches"] <= info["levels"]
    for k in (len(ins), 1, 3):
        d, pis = device_witness(dw, ins[:k])
        assert np.array_equal(d.cpu().numpy().view(np.uint64), wn[:k]), "batch of %d" % k
        assert np.array_equal(pis, pn[:k])
    bad = dict(ins[0])
    bad[pk_bits[5]] ^= 1
    with pytest.raises(AssertionError, match="decompression|copy constraint"):
        device_witness(dw, [ins[1], bad])
    # a failed witness does not poison the next batch
    d, pis = device_witness(dw, ins[:2])
    assert np.array_equal(d.cpu().numpy().view(np.uint64), wn[:2])
    dw.close()


def test_device_witness_of_the_reference_ed25519_circuit_and_proof_from_hbm(zctx, approval_prover):
    """the reference's per-signature circuit (crypto/plonky2_ed25519/src/gadgets/eddsa.rs:34-85; 1.13 M generators, 2^18 x 234
    wires): the three real NEAR approval signatures of the C1 fixture, witnesses generated on the GPU == the host interpreter's,
    the proof is made from the matrix in HBM (zklc_plonky2_prove_dev) and accepted by the verifier restatement; a corrupted
    signature has no witness"""
    import time
    from zklc_amd.plonky2 import ed25519_circuit as E, sha512
    j = load_golden("ed25519_near_c1_small.json")
    msg = bytes.fromhex(j["msg"])
    data, targets, prover, _ = approval_prover.ed25519_circuit(len(msg))       # the session's circuit and its GPU prover
    sigs = [(bytes.fromhex(x["approval"])[2:], bytes.fromhex(x["validator_tail"])[1:33]) for x in j["entries"]]
    fills = [E.fill_ecdsa_targets(targets, msg, s_, p_) for s_, p_ in sigs]
    if data._program is None:
        data.witness_program(fills[0])
    wn, pn = data.generate_witness_native(fills)
    dw = data.device_witness(zctx)
    print("Ed25519 circuit witness program on the device:", dw.info(len(fills)))
    d, pis = device_witness(dw, fills)
    t0 = time.time()
    d2, _ = device_witness(dw, fills)
    print("device witness generation: %.3f s for %d signatures (second call)" % (time.time() - t0, len(fills)))
    got = d.cpu().numpy().view(np.uint64)
    assert np.array_equal(pis, pn)
    for k in range(len(fills)):
        assert np.array_equal(got[k], wn[k]), "signature %d" % k
    assert torch.equal(d, d2)
    assert [int(x) for x in pis[0]] == sha512.array_to_bits(msg) + sha512.array_to_bits(sigs[0][1])
    from zklc_amd.plonky2 import serialization as S
    raw = prover.prove_dev(d[1].data_ptr(), [int(x) for x in pis[1]])
    V.verify(json.loads(json.dumps(S.proof_from_bytes(raw, prover.common, HASH_GL))), prover.verifier_data(), data.common_data())
    assert raw == prover.prove_bytes(wn[1], [int(x) for x in pn[1]])
    bad = bytearray(sigs[0][0])
    bad[40] ^= 1
    with pytest.raises(AssertionError):
        device_witness(dw, [E.fill_ecdsa_targets(targets, msg, bytes(bad), sigs[0][1])])
    dw.close()


def test_device_witness_of_a_recursion_circuit(zctx):
    """the in-circuit verifier (recursive_proof, prove_crypto/recursion.rs:16-97): Poseidon rows, extension arithmetic,
    reducing / interpolation / exponentiation gadgets -- device witness == host witness, tampered inner proof rejected"""
    from zklc_amd.plonky2.recursion import RecursionProver
    b = CircuitBuilder()
    x = b.add_virtual_public_input()
    y = b.add_virtual_target()
    z = b.mul(x, y)
    b.split_le(x, 10)
    b.register_public_input(b.add(z, b.constant(5)))
    data = b.build()
    wires, pis = data.generate_witness({x: 1000, y: 4000000000})
    prover = data.prover(zctx, HASH_GL)
    raw = prover.prove_bytes(wires, pis)
    rp = RecursionProver(zctx, HASH_GL)
    inner = (data.common_data(), prover.verifier_data(), raw)
    rc, proof = rp.recursive_proof(inner, None, [7, 8])         # compiles the program of the recursion circuit
    vals = rc.input_vector([(inner[1], raw)], [7, 8])
    wn, pn = rc.data.generate_witness_native(None, input_values=vals[None, :])
    dw = rc.data.device_witness(zctx)
    print("recursion circuit witness program on the device:", dw.info(1))
    d, dpis = device_witness(dw, input_values=vals[None, :])
    assert np.array_equal(d.cpu().numpy().view(np.uint64), wn) and np.array_equal(dpis, pn)
    bad = bytearray(raw)
    bad[800] ^= 1
    with pytest.raises(AssertionError):
        device_witness(dw, input_values=rc.input_vector([(inner[1], bytes(bad))], [7, 8])[None, :])
    dw.close()
    rp.close()
    prover.close()
