"""SHA-512 circuit of the reference, gate for gate: crypto/plonky2_sha512/src/circuit.rs:308-435 (`sha512_circuit`) with
its helpers (`xor3` :109-138, `big_sigma0/1` :141-181, `sigma0/1` :184-226, `ch` :231-249, `maj` :256-280,
`add_biguint_2limbs` :282-308, bit <-> BigUint conversion :59-95).  It is the hash inside the per-signature Ed25519 circuit
(crypto/plonky2_ed25519/src/gadgets/eddsa.rs:34-85: h = SHA-512(R || A || M), one 1024-bit block for NEAR's 105-byte preimage).

A 64-bit word is a BigUintTarget of two u32 limbs (little-endian limbs); bit vectors are MSB-first per word, exactly as
`biguint_to_bits_target` / `bits_to_biguint_target` produce them.  Built on zklc_amd.plonky2.CircuitBuilder, proven on the GPU.
"""
H512 = [0x6a09e667f3bcc908, 0xbb67ae8584caa73b, 0x3c6ef372fe94f82b, 0xa54ff53a5f1d36f1, 0x510e527fade682d1, 0x9b05688c2b3e6c1f,
        0x1f83d9abfb41bd6b, 0x5be0cd19137e2179]
K512 = [
    0x428a2f98d728ae22, 0x7137449123ef65cd, 0xb5c0fbcfec4d3b2f, 0xe9b5dba58189dbbc, 0x3956c25bf348b538, 0x59f111f1b605d019,
    0x923f82a4af194f9b, 0xab1c5ed5da6d8118, 0xd807aa98a3030242, 0x12835b0145706fbe, 0x243185be4ee4b28c, 0x550c7dc3d5ffb4e2,
    0x72be5d74f27b896f, 0x80deb1fe3b1696b1, 0x9bdc06a725c71235, 0xc19bf174cf692694, 0xe49b69c19ef14ad2, 0xefbe4786384f25e3,
    0x0fc19dc68b8cd5b5, 0x240ca1cc77ac9c65, 0x2de92c6f592b0275, 0x4a7484aa6ea6e483, 0x5cb0a9dcbd41fbd4, 0x76f988da831153b5,
    0x983e5152ee66dfab, 0xa831c66d2db43210, 0xb00327c898fb213f, 0xbf597fc7beef0ee4, 0xc6e00bf33da88fc2, 0xd5a79147930aa725,
    0x06ca6351e003826f, 0x142929670a0e6e70, 0x27b70a8546d22ffc, 0x2e1b21385c26c926, 0x4d2c6dfc5ac42aed, 0x53380d139d95b3df,
    0x650a73548baf63de, 0x766a0abb3c77b2a8, 0x81c2c92e47edaee6, 0x92722c851482353b, 0xa2bfe8a14cf10364, 0xa81a664bbc423001,
    0xc24b8b70d0f89791, 0xc76c51a30654be30, 0xd192e819d6ef5218, 0xd69906245565a910, 0xf40e35855771202a, 0x106aa07032bbd1b8,
    0x19a4c116b8d2d0c8, 0x1e376c085141ab53, 0x2748774cdf8eeb99, 0x34b0bcb5e19b48a8, 0x391c0cb3c5c95a63, 0x4ed8aa4ae3418acb,
    0x5b9cca4f7763e373, 0x682e6ff3d6b2b8a3, 0x748f82ee5defb2fc, 0x78a5636f43172f60, 0x84c87814a1f0ab72, 0x8cc702081a6439ec,
    0x90befffa23631e28, 0xa4506cebde82bde9, 0xbef9a3f7b2c67915, 0xc67178f2e372532b, 0xca273eceea26619c, 0xd186b8c721c0c207,
    0xeada7dd6cde0eb1e, 0xf57d4f7fee6ed178, 0x06f067aa72176fba, 0x0a637dc5a2c898a6, 0x113f9804bef90dae, 0x1b710b35131c471b,
    0x28db77f523047d84, 0x32caab7b40c72493, 0x3c9ebe0a15c9bebc, 0x431d67c49c100d4c, 0x4cc5d4becb3e42b6, 0x597f299cfc657e2a,
    0x5fcb6fab3ad6faec, 0x6c44198c4a475817]


def array_to_bits(data):
    """circuit.rs:46-56: MSB-first bits of every byte"""
    return [(b >> (7 - j)) & 1 for b in data for j in range(8)]


def _to_bits(b, word):
    """biguint_to_bits_target (:59-72): limbs high to low, each split little-endian and emitted MSB first"""
    out = []
    for limb in reversed(word):
        bits = b.split_le(limb, 32)
        out.extend(reversed(bits))
    return out


def _from_bits(b, bits):
    """bits_to_biguint_target (:74-92): 32-bit groups MSB first -> u32 limbs, little-endian limb order"""
    assert len(bits) % 32 == 0
    limbs = [b.le_sum(reversed(bits[i * 32:(i + 1) * 32])) for i in range(len(bits) // 32)]
    return list(reversed(limbs))


def _rotate64(y):
    return list(range(64 - y, 64)) + list(range(0, 64 - y))


def _shift64(y):
    return [64] * y + list(range(0, 64 - y))


def _xor3(b, x, y, z):
    """:109-138  a ^ b ^ c = a (1 - 2b - 2c + 4bc) + b + c - 2bc"""
    m = b.mul(y, z)
    two_b, two_c = b.add(y, y), b.add(z, z)
    two_m = b.add(m, m)
    four_m = b.add(two_m, two_m)
    t = b.sub(b.sub(b.one(), two_b), two_c)
    t = b.add(t, four_m)
    res = b.mul(x, t)
    res = b.add(res, y)
    res = b.add(res, z)
    return b.sub(res, two_m)


def _sigma(b, word, r0, r1, third, shift):
    bits = _to_bits(b, word)
    if shift:
        bits = bits + [b.constant(0)]
    i0, i1 = _rotate64(r0), _rotate64(r1)
    i2 = _shift64(third) if shift else _rotate64(third)
    return _from_bits(b, [_xor3(b, bits[i0[i]], bits[i1[i]], bits[i2[i]]) for i in range(64)])


def _ch(b, e, f, g):
    eb, fb, gb = _to_bits(b, e), _to_bits(b, f), _to_bits(b, g)
    return _from_bits(b, [b.add(b.mul(eb[i], b.sub(fb[i], gb[i])), gb[i]) for i in range(64)])


def _maj(b, x, y, z):
    xb, yb, zb = _to_bits(b, x), _to_bits(b, y), _to_bits(b, z)
    out = []
    two = b.constant(2)
    for i in range(64):
        m = b.mul(yb[i], zb[i])
        two_m = b.mul(two, m)
        t = b.sub(b.add(yb[i], zb[i]), two_m)
        out.append(b.add(b.mul(xb[i], t), m))
    return _from_bits(b, out)


def _add(b, x, y):
    """add_biguint_2limbs (:282-308): wrapping 64-bit addition through two add_many_u32"""
    carry = b.zero()
    limbs = []
    for i in range(2):
        lo, carry = b.add_many_u32([carry, x[i], y[i]])
        limbs.append(lo)
    return limbs


def _const64(b, v):
    return [b.constant(v & 0xFFFFFFFF), b.constant(v >> 32)]


def sha512_circuit(b, msg_len_in_bits):
    """-> (message bit targets, digest bit targets) as `Sha512Targets` (:40-43, :308-435)"""
    block_count = (msg_len_in_bits + 129 + 1023) // 1024
    padded = 1024 * block_count
    p = padded - 128 - msg_len_in_bits
    assert p > 1
    message = b.add_virtual_targets(msg_len_in_bits)
    msg = list(message) + [b.constant(1)] + [b.constant(0)] * (p - 1)
    msg += [b.constant((msg_len_in_bits >> (127 - i)) & 1) for i in range(128)]
    state = [_const64(b, h) for h in H512]
    k512 = [_const64(b, k) for k in K512]
    for blk in range(block_count):
        x = []
        a, bb, c, d, e, f, g, h = state
        for i in range(16):
            idx = blk * 1024 + i * 64
            u0 = b.le_sum(reversed(msg[idx:idx + 32]))
            u1 = b.le_sum(reversed(msg[idx + 32:idx + 64]))
            x.append([u1, u0])
            t1 = _add(b, h, _sigma(b, e, 14, 18, 41, False))
            t1 = _add(b, t1, _ch(b, e, f, g))
            t1 = _add(b, t1, k512[i])
            t1 = _add(b, t1, x[i])
            t2 = _add(b, _sigma(b, a, 28, 34, 39, False), _maj(b, a, bb, c))
            h, g, f, e, d, c, bb, a = g, f, e, _add(b, d, t1), c, bb, a, _add(b, t1, t2)
        for i in range(16, 80):
            s0 = _sigma(b, x[(i + 1) & 15], 1, 8, 7, True)
            s1 = _sigma(b, x[(i + 14) & 15], 19, 61, 6, True)
            x[i & 15] = _add(b, x[i & 15], _add(b, _add(b, s0, s1), x[(i + 9) & 15]))
            bs0 = _sigma(b, a, 28, 34, 39, False)
            bs1 = _sigma(b, e, 14, 18, 41, False)
            che, mj = _ch(b, e, f, g), _maj(b, a, bb, c)
            t1 = _add(b, x[i & 15], _add(b, _add(b, _add(b, h, bs1), che), k512[i]))
            t2 = _add(b, bs0, mj)
            h, g, f, e, d, c, bb, a = g, f, e, _add(b, d, t1), c, bb, a, _add(b, t1, t2)
        state = [_add(b, s, w) for s, w in zip(state, [a, bb, c, d, e, f, g, h])]
    digest = []
    for word in state:
        for j in (1, 0):
            bits = b.split_le(word[j], 32)
            digest.extend(reversed(bits))
    return message, digest
