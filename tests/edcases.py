"""Ed25519 test-vector builders shared by the CPU and GPU parity tests."""
import os
import random

from oracle import ed25519_ref as ref


def synthetic_set(n, seed=1, msg=b"zklc synthetic approval message...........", corrupt_every=0):
    """SURVEY 8(d) C2 synthetic scaling set: deterministic keys, one shared message."""
    pks, sigs = [], []
    for i in range(n):
        sd = ref.synthetic_seed(seed, i)
        _, _, pk = ref.keypair(sd)
        sig = bytearray(ref.sign(sd, msg))
        if corrupt_every and i % corrupt_every == corrupt_every - 1:
            sig[i % 64] ^= 1
        pks.append(pk)
        sigs.append(bytes(sig))
    return pks, sigs, msg


def edge_cases():
    """(pk, sig, msg, label) triples for the reject/accept classes of SURVEY 9.4.
    Expected results come from oracle.ed25519_ref.verify (dalek non-strict)."""
    rng = random.Random(99)
    out = []
    msg = b"edge case message"
    sd = ref.synthetic_seed(5, 0)
    a, _, pk = ref.keypair(sd)
    sig = ref.sign(sd, msg)
    out.append((pk, sig, msg, "honest"))
    out.append((pk, sig, b"", "honest sig, empty msg (reject)"))
    out.append((pk, ref.sign(sd, b""), b"", "honest empty msg"))
    # (i) s >= l : s + l still satisfies the group equation but must be rejected
    s = int.from_bytes(sig[32:], "little")
    out.append((pk, sig[:32] + (s + ref.L).to_bytes(32, "little"), msg, "s+l"))
    out.append((pk, sig[:32] + (ref.L).to_bytes(32, "little"), msg, "s=l"))
    out.append((pk, sig[:32] + (2**256 - 1).to_bytes(32, "little"), msg, "s=2^256-1"))
    # (ii) non-canonical y in A: y + p (only possible for y < 19); find keys? use small-order points instead
    # small-order points (order 1,2,4,8) in canonical and non-canonical encodings
    small = [
        (1).to_bytes(32, "little"),                                   # identity (0,1)
        (ref.P - 1).to_bytes(32, "little"),                           # (0,-1) order 2
        (0).to_bytes(32, "little"),                                   # (sqrt(-1),0)-ish order 4 (y=0)
        (1 | (1 << 255)).to_bytes(32, "little"),                      # identity with sign bit ("negative zero")
        (ref.P + 1).to_bytes(32, "little"),                           # y = p+1 non-canonical identity
        (ref.P).to_bytes(32, "little"),                               # y = p non-canonical 0
        bytes.fromhex("26e8958fc2b227b045c3f489f2ef98f0d5dfac05d3c63339b13802886d53fc05"),  # order 8
        bytes.fromhex("c7176a703d4dd84fba3c0b760d10670f2a2053fa2c39ccc64ec7fd7792ac037a"),  # order 8
    ]
    for k, A in enumerate(small):
        # with a small-order A and R = [s]B - [h]A the signature verifies cofactor-lessly
        for trial in range(2):
            sv = rng.randrange(ref.L)
            Ad = ref.decompress(A)
            for tweak in range(4):
                # choose R as the compression of [s]B - [h]A for the h that R itself induces: fixpoint search
                R = ref.compress(ref.pt_mul(sv, ref.BASE))
                if Ad is not None:
                    for _ in range(16):
                        h = ref.sha512_mod_l(R, A, msg)
                        R2 = ref.compress(ref.pt_add(ref.pt_mul(sv, ref.BASE), ref.pt_mul(h, ref.pt_neg(Ad))))
                        if R2 == R:
                            break
                        R = R2
                out.append((A, R + sv.to_bytes(32, "little"), msg + bytes([tweak]), "small-order A #%d" % k))
    # undecodable A (x^2 non-residue): y = 2 is not on the curve
    out.append(((2).to_bytes(32, "little"), sig, msg, "undecodable A"))
    for _ in range(6):
        A = bytes(rng.getrandbits(8) for _ in range(32))
        out.append((A, sig, msg, "random A"))
    # non-canonical R encoding of an honest signature: R.y + p impossible in general; flip the sign bit instead
    r2 = bytearray(sig)
    r2[31] ^= 0x80
    out.append((pk, bytes(r2), msg, "R sign flipped"))
    # R replaced by small-order encodings
    for A in small[:4]:
        out.append((pk, A + sig[32:], msg, "small-order R"))
    # all-zero everything
    out.append((bytes(32), bytes(64), msg, "zeros"))
    # honest signatures over various message lengths (1..3 SHA-512 blocks)
    for ln in [0, 1, 41, 47, 48, 49, 111, 112, 175, 176, 177, 300]:
        m = bytes(rng.getrandbits(8) for _ in range(ln))
        sdi = ref.synthetic_seed(6, ln)
        _, _, pki = ref.keypair(sdi)
        out.append((pki, ref.sign(sdi, m), m, "len %d" % ln))
    return out
