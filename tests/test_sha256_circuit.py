"""The reference's SHA-256 circuit (crypto/plonky2_sha256_u32/src/sha256.rs, restated in zklc_amd/plonky2/sha256.py on the
interleave / uninterleave gates of crypto/plonky2_u32) on the CPU: digests against hashlib -- including the doc example of
near_bft_finality/src/prove_crypto/sha256.rs:51-58 -- every gate constraint of the wire matrix (device evaluators compiled for
the host), the native witness interpreter, and the u32 bit gadgets one by one."""
import hashlib

import numpy as np

import zklc_amd  # noqa: F401
from zklc_amd.plonky2 import CircuitBuilder, sha256 as S
from p2_witness_check import gate_constraint_failures

M = 0xFFFFFFFF


def _digest(pis):
    return b"".join(int(x).to_bytes(4, "big") for x in pis)


def test_reference_doc_example_and_block_boundaries(hostsim):
    msg = bytes.fromhex("60")
    assert hashlib.sha256(msg).hexdigest() == "8d33f520a3c4cef80d2453aef81b612bfe1cb44c8b2025630ad38662763f13d3"   # sha256.rs:53-54
    for n in (1, 0, 55, 56, 64, 119):
        m = msg if n == 1 else bytes((7 * i + n) & 0xFF for i in range(n))
        data, words = S.sha256_circuit(len(m))
        assert len(words) == 16 * ((8 * len(m) + 64 + 512) // 512) and data.num_public_inputs == 8
        pw = S.sha256_witness(words, m)
        data.witness_program(list(pw))
        wn, pn = data.generate_witness_native([pw])
        assert _digest(pn[0]) == hashlib.sha256(m).digest(), n
        if n in (1, 56):
            w, pis = data.generate_witness(pw)
            assert np.array_equal(w, wn[0]) and pis == [int(x) for x in pn[0]]
            assert not gate_constraint_failures(hostsim, data, w, pis)


def test_two_to_one_sha256(hostsim):
    b = CircuitBuilder()
    left, right = b.add_virtual_targets(8), b.add_virtual_targets(8)
    for t in S.two_to_one_sha256(b, left, right):
        b.register_public_input(t)
    data = b.build()
    l, r = hashlib.sha256(b"left").digest(), hashlib.sha256(b"right").digest()
    pw = {t: int.from_bytes(d[4 * i:4 * i + 4], "big") for ts, d in ((left, l), (right, r)) for i, t in enumerate(ts)}
    w, pis = data.generate_witness(pw)
    assert _digest(pis) == hashlib.sha256(l + r).digest()
    assert not gate_constraint_failures(hostsim, data, w, pis)


def test_u32_bit_gadgets():
    """crypto/plonky2_u32/src/gadgets/interleaved_u32.rs: rotations, shifts, not, and/xor, xor-many, ch, maj"""
    b = CircuitBuilder()
    x, y, z, u, v = b.add_virtual_targets(5)
    outs = [b.xor_u32(x, y), b.and_u32(x, y), b.rrot_u32(x, 7), b.lrot_u32(x, 13), b.rsh_u32(x, 3), b.rsh_u32(x, 0), b.lsh_u32(x, 5),
            b.not_u32(x), S._ch(b, x, y, z), S._maj(b, x, y, z), b.unsafe_xor_many_u32([x, y, z]), b.unsafe_xor_many_u32([x, y, z, u]),
            b.unsafe_xor_many_u32([x, y, z, u, v]), b.unsafe_xor_many_u32([x]), b.add_many_u32([x, y, z, u, v])[0]]
    for t in outs:
        b.register_public_input(t)
    data = b.build()
    rr = lambda a, n: ((a >> n) | (a << (32 - n))) & M
    for X, Y, Z, U, V in [(0x9b05688c, 0x1f83d9ab, 0xdeadbeef, 0x01234567, 0xFFFFFFFF), (0, M, 1, 0x80000000, 0x7FFFFFFF)]:
        _, pis = data.generate_witness({x: X, y: Y, z: Z, u: U, v: V})
        assert pis == [X ^ Y, X & Y, rr(X, 7), rr(X, 19), X >> 3, X, (X << 5) & M, X ^ M, (X & Y) ^ (~X & M & Z),
                       (X & Y) ^ (X & Z) ^ (Y & Z), X ^ Y ^ Z, X ^ Y ^ Z ^ U, X ^ Y ^ Z ^ U ^ V, X, (X + Y + Z + U + V) & M]
