"""The reference's SHA-256 circuit on the host builder (first piece).

Restates crypto/plonky2_sha256_u32/src/sha256.rs:
  sigma / big_sigma / ch / maj            :75-142 (rotations as u32 multiplications, XOR / AND through the interleaved form)
  hash_sha256                             :173-245 (message schedule in a 16-word ring, 64 rounds per 512-bit block)
  set_sha256_input_target (padding)       :19-32 with types.rs:51-66 (`set_biguint_u32_be_target`: the value's little-endian u32
                                          digits, each byte-swapped)
and the caller near_bft_finality/src/prove_crypto/sha256.rs:62-83 (`sha256_proof_u32`: one circuit per number of blocks,
public inputs = the eight 32-bit words of the digest).
Gadgets: zklc_amd/plonky2/builder.py (interleave / uninterleave gates of crypto/plonky2_u32, U32Arithmetic / U32AddMany /
U32Subtraction)."""
from .builder import CircuitBuilder, standard_recursion_config

H256 = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
K32 = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2,
]
SHA256_BLOCK = 512


def _sigma(b, a, r1, r2, s3):
    return b.unsafe_xor_many_u32([b.rrot_u32(a, r1), b.rrot_u32(a, r2), b.rsh_u32(a, s3)])


def _big_sigma(b, a, r1, r2, r3):
    return b.unsafe_xor_many_u32([b.rrot_u32(a, r1), b.rrot_u32(a, r2), b.rrot_u32(a, r3)])


def _ch(b, e, f, g):
    ef = b.and_xor_u32(e, f)[0]
    eg = b.and_xor_u32(b.not_u32(e), g)[0]
    return b.and_xor_b32_to_u32(ef, eg)[1]


def _maj(b, a, bb, c):
    b_and_c, b_xor_c = b.and_xor_u32(bb, c)
    abc = b.and_xor_b32(b.interleave_u32(a), b_xor_c)[0]
    return b.and_xor_b32_to_u32(abc, b_and_c)[1]


def _compress(b, state, w, k256):
    w = list(w)
    a, bb, c, d, e, f, g, h = state
    for i in range(64):
        if i >= 16:
            s0 = _sigma(b, w[(i + 1) & 15], 7, 18, 3)
            s1 = _sigma(b, w[(i + 14) & 15], 17, 19, 10)
            w[i & 15] = b.add_many_u32([s0, s1, w[(i + 9) & 15], w[i & 15]])[0]
        temp1 = b.add_many_u32([h, _big_sigma(b, e, 6, 11, 25), _ch(b, e, f, g), k256[i], w[i & 15]])[0]
        temp2 = b.add_u32(_big_sigma(b, a, 2, 13, 22), _maj(b, a, bb, c))[0]
        h, g, f = g, f, e
        e = b.add_u32(d, temp1)[0]
        d, c, bb = c, bb, a
        a = b.add_u32(temp1, temp2)[0]
    return [b.add_u32(s, x)[0] for s, x in zip(state, (a, bb, c, d, e, f, g, h))]


def hash_sha256(b, input_limbs):
    """`hash_sha256` (sha256.rs:173-245): input = 16 u32 words per block (already padded), returns the 8 digest words"""
    assert len(input_limbs) % 16 == 0
    state = [b.constant(x) for x in H256]
    k256 = [b.constant(x) for x in K32]
    for blk in range(len(input_limbs) // 16):
        state = _compress(b, state, input_limbs[16 * blk:16 * blk + 16], k256)
    return state


def two_to_one_sha256(b, left, right):
    """`two_to_one_sha256` (sha256.rs:247-390): sha256 of two 32-byte digests (one data block + the constant padding block)"""
    state = [b.constant(x) for x in H256]
    k256 = [b.constant(x) for x in K32]
    state = _compress(b, state, list(left) + list(right), k256)
    pad = [b.constant(0x80000000)] + [b.constant(0)] * 14 + [b.constant(512)]
    return _compress(b, state, pad, k256)


def padded_words(msg):
    """the u32 words `set_sha256_input_target` assigns (sha256.rs:19-32): message || 0x80 || zeros || be64(bit length)"""
    block_num = (8 * len(msg) + 64 + 512) // 512
    data = bytes(msg) + b"\x80" + b"\0" * (64 * block_num - len(msg) - 9) + (8 * len(msg)).to_bytes(8, "big")
    return [int.from_bytes(data[4 * i:4 * i + 4], "big") for i in range(len(data) // 4)]


def sha256_circuit(msg_len, config=None):
    """the circuit of `sha256_proof_u32` (near_bft_finality/src/prove_crypto/sha256.rs:62-78) for messages of msg_len bytes:
    returns (CircuitData, input word targets); public inputs = the 8 digest words"""
    b = CircuitBuilder(config or standard_recursion_config())
    block_num = (8 * msg_len + 64 + 512) // 512
    words = b.add_virtual_targets(16 * block_num)
    for t in hash_sha256(b, words):
        b.register_public_input(t)
    return b.build(), words


def sha256_witness(words_t, msg):
    return dict(zip(words_t, padded_words(msg)))


def block_num_of(msg_len):
    return (8 * msg_len + 64 + 512) // 512


def build_cached(msg_len):
    """the SHA-256 circuit for messages of `msg_len` bytes (one circuit per number of 512-bit blocks) with its witness program,
    through the circuit cache: -> (CircuitData, word targets, from_cache)"""
    from .circuit_cache import load_or_build

    def build():
        data, words = sha256_circuit(msg_len)
        data.witness_program(list(words))
        return data, words
    return load_or_build("sha256", (block_num_of(msg_len), sorted(standard_recursion_config().items(), key=str)), build)


class Sha256Prover:
    """`sha256_proof_u32` (near_bft_finality/src/prove_crypto/sha256.rs:62-83) on one GPU context: one circuit per number of
    512-bit blocks, built and uploaded once (the reference rebuilds it per call), native witness generation, GPU proof."""

    def __init__(self, ctx, hasher=0):
        self.ctx, self.hasher = ctx, hasher
        self._circuits = {}

    def circuit_for(self, msg_len):
        block_num = (8 * msg_len + 64 + 512) // 512
        ent = self._circuits.get(block_num)
        if ent is None:
            data, words, _ = build_cached(msg_len)
            prover = data.prover(self.ctx, self.hasher)
            ent = self._circuits[block_num] = (data, words, prover, data.common_data(), prover.verifier_data())
        return ent

    def sha256_proof_u32(self, msg, digest=None):
        """-> ((common_data, verifier_only), proof); public inputs = the digest as eight big-endian u32 words.  `digest`, when
        given, must be sha256(msg): the reference sets it as the output target's witness, so a wrong value fails the proof."""
        data, words, prover, common, vd = self.circuit_for(len(msg))
        wires, pis = data.generate_witness_native([sha256_witness(words, msg)])
        if digest is not None:
            want = [int.from_bytes(bytes(digest)[4 * i:4 * i + 4], "big") for i in range(8)]
            if [int(x) for x in pis[0]] != want:
                raise AssertionError("sha256_proof_u32: the given hash is not sha256(msg)")
        return (common, vd), prover.prove(wires[0], [int(x) for x in pis[0]])

    def close(self):
        for ent in self._circuits.values():
            ent[2].close()
        self._circuits = {}
