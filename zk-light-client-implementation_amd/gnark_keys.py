"""Readers / writers of gnark's Groth16 key files for BN254: `pk.bin` / `vk.bin` as `SaveVerifierCircuitGroth` writes them
(`pk.WriteRawTo`, `vk.WriteRawTo`: gnark-plonky2-verifier/verifier/util.go:172-216) and `LoadGroth16ProverData` /
`LoadGroth16VerifierKey` read them (`pk.ReadFrom`, `vk.ReadFrom`: util.go:337-389).  The result feeds `groth16.Groth16Prover` /
`groth16.Groth16Verifier` directly (the `*_words` arrays are gnark-crypto's in-memory layout, little-endian Montgomery limbs).

PARITY UNPINNED.  The byte layout lives in un-vendored dependencies (gnark v0.9.1 backend/groth16/bn254/marshal.go, gnark-crypto
v0.12.2 ecc/bn254/marshal.go and fr/fft/domain.go; go.mod:8-9) and the reference ships no pk.bin / vk.bin to check against (the
keys are ~GB files produced by its trusted setup run).  What IS pinned by the reference: the uncompressed point encoding (32-byte
big-endian coordinates, G2 as X.A1 | X.A0 | Y.A1 | Y.A0) through the 256 proof bytes that cmd/web-api.go:90-98 slices into the
eight words Verifier.sol consumes (tests/test_formats.py).  The rest -- field order, slice length prefixes, the 2-bit point flags,
the domain header -- is restated from the published marshal code; everything after the last field this module understands
(Pedersen commitment keys: the reference's circuit has none, Verifier.sol carries no commitment terms) is kept as an opaque
trailer and reported, not interpreted.  Tests: round trips and structure only (tests/test_gnark_keys.py).
"""
import struct

import numpy as np

from .formats import P_BN254 as P, ProofInvalid
from .groth16 import fp_to_mont_words

R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
_FLAG_MASK, _UNCOMPRESSED, _INFINITY, _SMALLEST, _LARGEST = 0xC0, 0x00, 0x40, 0x80, 0xC0
_HALF_P = (P - 1) // 2


class _Reader:
    def __init__(self, data):
        self.b, self.o = memoryview(bytes(data)), 0

    def take(self, n):
        if self.o + n > len(self.b):
            raise ProofInvalid("key file truncated at byte %d (+%d)" % (self.o, n))
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def u32(self):
        return struct.unpack(">I", self.take(4))[0]

    def u64(self):
        return struct.unpack(">Q", self.take(8))[0]

    def fr(self):
        v = int.from_bytes(self.take(32), "big")
        if v >= R:
            raise ProofInvalid("scalar field element not reduced")
        return v


def _sqrt_fp(a):
    x = pow(a, (P + 1) // 4, P)
    if x * x % P != a % P:
        raise ProofInvalid("compressed point: x is not on the curve")
    return x


def _fp2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def _fp2_sqrt(a):
    """a square root of a in Fp2 = Fp[i] / (i^2 + 1) (complex method); raises if a is not a square"""
    a0, a1 = a
    if a1 == 0:
        try:
            return (_sqrt_fp(a0), 0)
        except ProofInvalid:
            return (0, _sqrt_fp((P - a0) % P))
    d = _sqrt_fp((a0 * a0 + a1 * a1) % P)
    for dd in (d, (P - d) % P):
        h = (a0 + dd) * pow(2, P - 2, P) % P
        x0 = pow(h, (P + 1) // 4, P)
        if x0 * x0 % P != h or x0 == 0:
            continue
        x1 = a1 * pow(2 * x0, P - 2, P) % P
        if _fp2_mul((x0, x1), (x0, x1)) == (a0 % P, a1 % P):
            return (x0, x1)
    raise ProofInvalid("compressed point: x is not on the twist curve")


_B2 = None


def _twist_b():
    global _B2
    if _B2 is None:
        inv = pow(82, P - 2, P)
        _B2 = (27 * inv % P, (P - 3 * inv % P) % P)                   # 3 / (9 + i)
    return _B2


def _g1_lex_largest(y):
    return y > _HALF_P


def _g2_lex_largest(y):
    """gnark-crypto E2.LexicographicallyLargest: compares A1 first, A0 when A1 = 0"""
    return y[1] > _HALF_P if y[1] else y[0] > _HALF_P


def read_g1(rd):
    """one G1Affine as gnark-crypto's decoder accepts it: uncompressed (64 bytes) or compressed (32 bytes), by the flag bits"""
    first = rd.take(32)
    flag = first[0] & _FLAG_MASK
    x = int.from_bytes(first, "big") & ((1 << 254) - 1)
    if flag == _INFINITY:
        # compressed infinity is 32 bytes; the uncompressed form has 32 more zero bytes: gnark's writers never mix the two in a
        # file, so the caller says which one it expects (read_g1_raw)
        if x:
            raise ProofInvalid("infinity flag with a non-zero coordinate")
        return None
    if x >= P:
        raise ProofInvalid("G1 x not reduced")
    if flag == _UNCOMPRESSED:
        y = int.from_bytes(rd.take(32), "big")
        if y >= P:
            raise ProofInvalid("G1 y not reduced")
        if x == 0 and y == 0:
            return None
        if (y * y - x * x * x - 3) % P:
            raise ProofInvalid("G1 point not on the curve")
        return (x, y)
    y = _sqrt_fp((x * x * x + 3) % P)
    if _g1_lex_largest(y) != (flag == _LARGEST):
        y = (P - y) % P
    return (x, y)


def read_g2(rd):
    first = rd.take(64)
    flag = first[0] & _FLAG_MASK
    x1 = int.from_bytes(first[:32], "big") & ((1 << 254) - 1)
    x0 = int.from_bytes(first[32:], "big")
    if flag == _INFINITY:
        if x0 or x1:
            raise ProofInvalid("infinity flag with a non-zero coordinate")
        return None
    if x0 >= P or x1 >= P:
        raise ProofInvalid("G2 x not reduced")
    x = (x0, x1)
    rhs = _fp2_mul(_fp2_mul(x, x), x)
    b = _twist_b()
    rhs = ((rhs[0] + b[0]) % P, (rhs[1] + b[1]) % P)
    if flag == _UNCOMPRESSED:
        raw = rd.take(64)
        y = (int.from_bytes(raw[32:], "big"), int.from_bytes(raw[:32], "big"))
        if y[0] >= P or y[1] >= P:
            raise ProofInvalid("G2 y not reduced")
        if x == (0, 0) and y == (0, 0):
            return None
        if _fp2_mul(y, y) != rhs:
            raise ProofInvalid("G2 point not on the twist curve")
        return (x, y)
    y = _fp2_sqrt(rhs)
    if _g2_lex_largest(y) != (flag == _LARGEST):
        y = ((P - y[0]) % P, (P - y[1]) % P)
    return (x, y)


def _read_points(rd, one, raw_inf_pad):
    n = rd.u32()
    out = []
    for _ in range(n):
        start = rd.o
        pt = one(rd)
        if pt is None and raw_inf_pad and rd.o - start == raw_inf_pad:
            rd.take(raw_inf_pad)                           # RawBytes of infinity: the flag byte + zeros over the full uncompressed width
        out.append(pt)
    return out


def write_g1(pt, raw=True):
    if pt is None:
        return bytes([_INFINITY]) + bytes(63 if raw else 31)
    x, y = pt
    if raw:
        return x.to_bytes(32, "big") + y.to_bytes(32, "big")
    b = bytearray(x.to_bytes(32, "big"))
    b[0] |= _LARGEST if _g1_lex_largest(y) else _SMALLEST
    return bytes(b)


def write_g2(pt, raw=True):
    if pt is None:
        return bytes([_INFINITY]) + bytes(127 if raw else 63)
    (x0, x1), (y0, y1) = pt
    if raw:
        return b"".join(v.to_bytes(32, "big") for v in (x1, x0, y1, y0))
    b = bytearray(x1.to_bytes(32, "big") + x0.to_bytes(32, "big"))
    b[0] |= _LARGEST if _g2_lex_largest((y0, y1)) else _SMALLEST
    return bytes(b)


def _points(pts, one, raw):
    return struct.pack(">I", len(pts)) + b"".join(one(p, raw) for p in pts)


def _padded(rd, raw):
    """readers of single points: the raw (uncompressed) form of infinity is padded to the full width"""
    def g1():
        s = rd.o
        pt = read_g1(rd)
        if pt is None and raw and rd.o - s == 32:
            rd.take(32)
        return pt

    def g2():
        s = rd.o
        pt = read_g2(rd)
        if pt is None and raw and rd.o - s == 64:
            rd.take(64)
        return pt
    return g1, g2, (32 if raw else 0), (64 if raw else 0)


# ---------------------------------------------------------------------------------------------- verifying key
def vk_to_gnark_bytes(vk, raw=True, trailer=b""):
    """vk: dict alpha1, beta1, delta1 (G1), beta2, gamma2, delta2 (G2), K (G1 list).  gnark order: [alpha]1, [beta]1, [beta]2,
    [gamma]2, [delta]1, [delta]2, uint32 len(K), K..."""
    return (write_g1(vk["alpha1"], raw) + write_g1(vk["beta1"], raw) + write_g2(vk["beta2"], raw) + write_g2(vk["gamma2"], raw)
            + write_g1(vk["delta1"], raw) + write_g2(vk["delta2"], raw) + _points(vk["K"], write_g1, raw) + trailer)


def vk_from_gnark_bytes(data, raw=True):
    """-> dict as above + "trailer" (bytes after K: commitment fields, uninterpreted).  raw: the file was written by WriteRawTo
    (only matters for points at infinity, whose raw form is padded to the uncompressed width)."""
    rd = _Reader(data)
    g1, g2, pad1, pad2 = _padded(rd, raw)
    vk = {"alpha1": g1(), "beta1": g1(), "beta2": g2(), "gamma2": g2(), "delta1": g1(), "delta2": g2()}
    vk["K"] = _read_points(rd, read_g1, pad1)
    vk["trailer"] = bytes(rd.b[rd.o:])
    return vk


# ---------------------------------------------------------------------------------------------- proving key
def _domain_bytes(n):
    """fft.Domain.WriteTo: Cardinality (uint64), CardinalityInv, Generator, GeneratorInv, FrMultiplicativeGen, FrMultiplicativeGenInv
    (32-byte big-endian regular form)"""
    log_n = n.bit_length() - 1
    assert 1 << log_n == n and log_n <= 28
    gen = pow(5, (R - 1) >> log_n, R)
    vals = [pow(n, R - 2, R), gen, pow(gen, R - 2, R), 5, pow(5, R - 2, R)]
    return struct.pack(">Q", n) + b"".join(v.to_bytes(32, "big") for v in vals)


def pk_to_gnark_bytes(pk, raw=True, trailer=b""):
    """pk: the dict of oracle.groth16.setup / Groth16Prover (integer tuples, None = infinity; A, B1, B2 complete or compacted with
    infinity_a / infinity_b).  gnark order: domain, [alpha]1, [beta]1, [delta]1, A, B1, Z, K, [beta]2, [delta]2, B2, nbWires,
    NbInfinityA, NbInfinityB, InfinityA, InfinityB (one byte per bool)."""
    inf_a, inf_b = pk.get("infinity_a"), pk.get("infinity_b")
    A, B1, B2 = pk["A"], pk["B1"], pk["B2"]
    if inf_a is None:
        inf_a = [p is None for p in A]
        A = [p for p in A if p is not None]
    if inf_b is None:
        inf_b = [p is None for p in B1]
        assert [p is None for p in B2] == list(inf_b)
        B1 = [p for p in B1 if p is not None]
        B2 = [p for p in B2 if p is not None]
    out = [_domain_bytes(int(pk["n"])), write_g1(pk["alpha1"], raw), write_g1(pk["beta1"], raw), write_g1(pk["delta1"], raw),
           _points(A, write_g1, raw), _points(B1, write_g1, raw), _points(pk["Z"], write_g1, raw), _points(pk["K"], write_g1, raw),
           write_g2(pk["beta2"], raw), write_g2(pk["delta2"], raw), _points(B2, write_g2, raw),
           struct.pack(">QQQ", len(inf_a), int(sum(bool(x) for x in inf_a)), int(sum(bool(x) for x in inf_b))),
           bytes(1 if x else 0 for x in inf_a), bytes(1 if x else 0 for x in inf_b), trailer]
    return b"".join(out)


def pk_from_gnark_bytes(data, n_public, raw=True):
    """-> the dict `Groth16Prover` takes (compacted A / B1 / B2 + infinity masks).  n_public: public inputs WITHOUT the constant
    wire (gnark keeps it in the R1CS file, not in the key: len(K) = nbWires - nbPublic where nbPublic counts the constant one)."""
    rd = _Reader(data)
    n = rd.u64()
    if n == 0 or n & (n - 1):
        raise ProofInvalid("domain size is not a power of two")
    card_inv, gen = rd.fr(), rd.fr()
    rd.fr(), rd.fr(), rd.fr()
    if card_inv * n % R != 1 or pow(gen, n, R) != 1 or (n > 1 and pow(gen, n // 2, R) == 1):
        raise ProofInvalid("domain header inconsistent")
    g1, g2, pad1, pad2 = _padded(rd, raw)
    pk = {"n": n, "n_public": int(n_public), "alpha1": g1(), "beta1": g1(), "delta1": g1()}
    pk["A"] = _read_points(rd, read_g1, pad1)
    pk["B1"] = _read_points(rd, read_g1, pad1)
    pk["Z"] = _read_points(rd, read_g1, pad1)
    pk["K"] = _read_points(rd, read_g1, pad1)
    pk["beta2"], pk["delta2"] = g2(), g2()
    pk["B2"] = _read_points(rd, read_g2, pad2)
    n_wires, n_inf_a, n_inf_b = rd.u64(), rd.u64(), rd.u64()
    if n_wires > (1 << 32):
        raise ProofInvalid("wire count implausible")
    inf_a = np.frombuffer(rd.take(n_wires), dtype=np.uint8).astype(bool)
    inf_b = np.frombuffer(rd.take(n_wires), dtype=np.uint8).astype(bool)
    if int(inf_a.sum()) != n_inf_a or int(inf_b.sum()) != n_inf_b:
        raise ProofInvalid("infinity masks do not match their counts")
    if len(pk["A"]) != n_wires - n_inf_a or len(pk["B1"]) != n_wires - n_inf_b or len(pk["B2"]) != n_wires - n_inf_b:
        raise ProofInvalid("point arrays do not match the infinity masks")
    if len(pk["Z"]) != n - 1 and len(pk["Z"]) != n:
        raise ProofInvalid("Z has %d points for a domain of %d" % (len(pk["Z"]), n))
    pk["Z"] = pk["Z"][:n - 1]
    if len(pk["K"]) != n_wires - 1 - int(n_public):
        raise ProofInvalid("K has %d points, expected nbWires - nbPublic = %d" % (len(pk["K"]), n_wires - 1 - int(n_public)))
    pk["infinity_a"], pk["infinity_b"] = inf_a, inf_b
    pk["trailer"] = bytes(rd.b[rd.o:])
    return pk


def points_to_words(pts, g2=False):
    """affine integer tuples -> gnark-crypto's memory layout (uint64 [n, 8] / [n, 16]) for the `*_words` keys of Groth16Prover"""
    w = 16 if g2 else 8
    out = np.zeros((len(pts), w), dtype=np.uint64)
    for i, p in enumerate(pts):
        if p is None:
            continue
        coords = (p[0][0], p[0][1], p[1][0], p[1][1]) if g2 else p
        out[i] = np.array([v for c in coords for v in fp_to_mont_words(c)], dtype=np.uint64)
    return out
