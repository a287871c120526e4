"""Ed25519 verification oracle (pure-Python big integers).  TEST INFRASTRUCTURE.

Follows
  * call site  near_bft_finality/src/prove_block_data/signatures.rs:72-86
    (`sig.verify(msg, &pk)` = near-crypto 0.19.0 -> ed25519-dalek, UN-VENDORED:
    Cargo.toml:20-22) -- semantics restated from the published ed25519-dalek
    non-strict `verify`: reject s >= l, reject an undecodable A, compute
    R' = [s]B - [h]A and compare compress(R') with the 32 signature bytes of R
    (R itself is never decompressed; cofactor-less equation).
  * in-tree restatement  crypto/plonky2_ed25519/src/curve/eddsa.rs:33-58
    (`verify_message`): h = SHA512(R||A||M) mod l; decompress A and R;
    accept iff [s]B == R + [h]A in affine coordinates.  Implemented here as
    `verify_message_intree` and used to cross-check the accept path.
  * constants  crypto/plonky2_ed25519/src/curve/ed25519.rs:19-51,
    field/ed25519_base.rs:99-104, field/ed25519_scalar.rs:96-101.
  * message  near_bft_finality/src/prove_block_data/signatures.rs:24-39.

Parity pin: the accept path is pinned by the reference's mainnet fixtures
(tests/golden/near_*.json, generated from data/*.json) and the fixed triple of
crypto/plonky2_ed25519/src/main.rs:68-80.  The reject path is NOT pinned by
any reference test; the classes below are this repo's documented
choice (= ed25519-dalek non-strict).
"""
import hashlib

P = 2**255 - 19
L = 2**252 + 0x14def9dea2f79cd65812631a5cf5d3ed
D = 37095705934669439343138083508754565189542113879843219016388785533085940283555
SQRT_M1 = pow(2, (P - 1) // 4, P)
BX = 15112221349535400772501151409588531511454012693041857206046113283949847762202
BY = 46316835694926478169428394003475163141307993866256225615783033603165251855960
assert D == (-121665 * pow(121666, P - 2, P)) % P

IDENT = (0, 1, 1, 0)  # extended (X, Y, Z, T)
BASE = (BX, BY, 1, BX * BY % P)


def pt_add(a, b):
    # extended twisted-Edwards addition, a = -1 (complete)
    x1, y1, z1, t1 = a
    x2, y2, z2, t2 = b
    A = (y1 - x1) * (y2 - x2) % P
    B = (y1 + x1) * (y2 + x2) % P
    C = 2 * D * t1 * t2 % P
    Dd = 2 * z1 * z2 % P
    E, F, G, H = B - A, Dd - C, Dd + C, B + A
    return (E * F % P, G * H % P, F * G % P, E * H % P)


def pt_neg(a):
    x, y, z, t = a
    return ((-x) % P, y, z, (-t) % P)


def pt_mul(k, a):
    r = IDENT
    while k:
        if k & 1:
            r = pt_add(r, a)
        a = pt_add(a, a)
        k >>= 1
    return r


def pt_affine(a):
    x, y, z, _ = a
    zi = pow(z, P - 2, P)
    return (x * zi % P, y * zi % P)


def compress(a):
    x, y = pt_affine(a)
    return (y | ((x & 1) << 255)).to_bytes(32, "little")


def decompress(b):
    """curve25519-dalek `CompressedEdwardsY::decompress`: the top bit is the
    sign of x, the low 255 bits are y taken mod p WITHOUT a canonicity check;
    x = 0 with sign bit 1 is accepted (x stays 0).  None if x^2 is a
    non-residue."""
    n = int.from_bytes(b, "little")
    sign = n >> 255
    y = (n & ((1 << 255) - 1)) % P
    u = (y * y - 1) % P
    v = (D * y * y + 1) % P
    # sqrt_ratio_i(u, v)
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = u * v3 % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    if check == u:
        pass
    elif check == (-u) % P:
        r = r * SQRT_M1 % P
    else:
        return None
    if r & 1:           # dalek returns the non-negative root ...
        r = (-r) % P
    if sign:            # ... then conditionally negates by the sign bit
        r = (-r) % P
    return (r, y, 1, r * y % P)


def sha512_mod_l(*parts):
    h = hashlib.sha512()
    for p in parts:
        h.update(p)
    return int.from_bytes(h.digest(), "little") % L


def verify(pk: bytes, sig: bytes, msg: bytes) -> bool:
    """ed25519-dalek non-strict verify (what signatures.rs:79 calls)."""
    if len(pk) != 32 or len(sig) != 64:
        return False
    s = int.from_bytes(sig[32:], "little")
    if s >= L:
        return False
    A = decompress(pk)
    if A is None:
        return False
    h = sha512_mod_l(sig[:32], pk, msg)
    Rp = pt_add(pt_mul(s, BASE), pt_mul(h, pt_neg(A)))
    return compress(Rp) == sig[:32]


def verify_message_intree(msg: bytes, sig: bytes, pk: bytes) -> bool:
    """crypto/plonky2_ed25519/src/curve/eddsa.rs:33-58 (panics -> False)."""
    h = sha512_mod_l(sig[:32], pk, msg)
    A = decompress(pk)
    R = decompress(sig[:32])
    if A is None or R is None:
        return False
    s = int.from_bytes(sig[32:], "little") % L   # from_noncanonical_biguint
    sb = pt_affine(pt_mul(s, BASE))
    rhs = pt_affine(pt_add(R, pt_mul(h, A)))
    return sb == rhs


# ---- deterministic RFC 8032 signing: synthetic validator sets
def keypair(seed32: bytes):
    h = hashlib.sha512(seed32).digest()
    a = int.from_bytes(h[:32], "little")
    a &= (1 << 254) - 8
    a |= 1 << 254
    return a, h[32:], compress(pt_mul(a, BASE))


def sign(seed32: bytes, msg: bytes) -> bytes:
    a, prefix, pk = keypair(seed32)
    r = sha512_mod_l(prefix, msg)
    R = compress(pt_mul(r, BASE))
    h = sha512_mod_l(R, pk, msg)
    return R + ((r + h * a) % L).to_bytes(32, "little")


def synthetic_seed(seed: int, i: int) -> bytes:
    """sk_i = SHA512("zklc/ed25519/v1" || le64(seed) || le64(i))[:32]"""
    return hashlib.sha512(b"zklc/ed25519/v1" + seed.to_bytes(8, "little")
                          + i.to_bytes(8, "little")).digest()[:32]


# ---- NEAR message + borsh slicing (signatures.rs:24-39, 72-86; types.rs:7-17)
def generate_signed_message(ch_height: int, nb_height: int, nb_prev_hash: bytes) -> bytes:
    if ch_height + 1 == nb_height:
        inner = b"\x00" + nb_prev_hash          # ApprovalInner::Endorsement
    else:
        inner = b"\x01" + ch_height.to_bytes(8, "little")  # ApprovalInner::Skip
    return inner + nb_height.to_bytes(8, "little")
