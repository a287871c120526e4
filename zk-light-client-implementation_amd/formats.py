"""On-disk / wire formats either side of the proving path, beyond proof.bin / proof.json (plonky2/serialization.py):

  * `VerifierOnlyCircuitData::to_bytes` and `VerifierCircuitData::to_bytes(&DefaultGateSerializer)` -- what
    near_bft_finality/src/bin/prove_block.rs:320-458 writes as verifier_data.bin and what travels in OutputTask.verifier_data
    (near_bft_finality/src/prove_block_data/signatures.rs:223-230).  The byte layout is the un-vendored plonky2-near@2244a9d
    `util/serialization`; it is pinned here by the reference's three golden verifier_data.bin files
    (tests/test_formats.py writes them back byte for byte from the JSON next to them).
  * `InputTask` / `OutputTask` (near_bft_finality/src/types.rs:172-192): the NATS messages of prove_approvals_with_client
    (signatures.rs:144-274) -- serde_json of structs with Vec<u8> fields, i.e. JSON arrays of byte values.
  * the Groth16 proof encodings of the wrap: the 8 x uint256 form gnark-plonky2-verifier/cmd/web-api.go:90-98 returns
    (`proof.WriteRawTo`: A.X, A.Y, B.X.A1, B.X.A0, B.Y.A1, B.Y.A0, C.X, C.Y, big-endian) and the 4 x uint256 compressed form of
    contracts/hardhat/contracts/Verifier.sol:201-365,427-434 (`compressProof` / the decompression inside
    `verifyCompressedProof`), restated operation for operation.
Host-side byte shuffling only; no GPU involved.
"""
import json
import re
import struct

from .plonky2 import serialization as S

# ------------------------------------------------------------------------------------------- plonky2 circuit data
# tags of plonky2's DefaultGateSerializer (the gates in alphabetical order), confirmed by the golden files for the 13 gate types
# of the final proofs
GATE_TAGS = {"ArithmeticGate": 0, "ArithmeticExtensionGate": 1, "BaseSumGate": 2, "ConstantGate": 3, "CosetInterpolationGate": 4,
             "ExponentiationGate": 5, "MulExtensionGate": 8, "NoopGate": 9, "PoseidonMdsGate": 10, "PoseidonGate": 11,
             "PublicInputGate": 12, "RandomAccessGate": 13, "ReducingExtensionGate": 14, "ReducingGate": 15}


def _u64(x):
    return struct.pack("<Q", int(x))


def _gate_bytes(gid):
    name = re.match(r"[A-Za-z]+", gid).group(0)
    if name not in GATE_TAGS:
        raise ValueError("gate %r is not in plonky2's DefaultGateSerializer" % gid)
    out = struct.pack("<I", GATE_TAGS[name])

    def field(key):
        return int(re.search(key + r": (\d+)", gid).group(1))
    if name in ("ArithmeticGate", "ArithmeticExtensionGate", "MulExtensionGate"):
        out += _u64(field("num_ops"))
    elif name == "ConstantGate":
        out += _u64(field("num_consts"))
    elif name == "BaseSumGate":
        if not gid.endswith("Base: 2"):
            raise ValueError("only BaseSumGate<2> is in plonky2's DefaultGateSerializer: %r" % gid)
        out += _u64(field("num_limbs"))
    elif name in ("ReducingGate", "ReducingExtensionGate"):
        out += _u64(field("num_coeffs"))
    elif name == "ExponentiationGate":
        out += _u64(field("num_power_bits"))
    elif name == "RandomAccessGate":
        out += _u64(field("bits")) + _u64(field("num_copies")) + _u64(field("num_extra_constants"))
    elif name == "CosetInterpolationGate":
        w = [int(x) for x in re.search(r"barycentric_weights: \[([^\]]*)\]", gid).group(1).split(",")]
        out += _u64(field("subgroup_bits")) + _u64(field("degree")) + _u64(len(w)) + b"".join(_u64(x) for x in w)
    return out


def _fri_config_bytes(fc):
    out = _u64(fc["rate_bits"]) + _u64(fc["cap_height"]) + _u64(fc["num_query_rounds"]) + struct.pack("<I", fc["proof_of_work_bits"])
    rs = fc["reduction_strategy"]
    if "Fixed" in rs:
        out += b"\x00" + _u64(len(rs["Fixed"])) + b"".join(_u64(x) for x in rs["Fixed"])
    elif "ConstantArityBits" in rs:
        out += b"\x01" + _u64(rs["ConstantArityBits"][0]) + _u64(rs["ConstantArityBits"][1])
    else:
        out += b"\x02" + _u64(rs["MinSize"]) if rs.get("MinSize") is not None else b"\x02\x00"
    return out


def common_data_to_bytes(common):
    """`write_common_circuit_data` of plonky2-near (the part of verifier_data.bin after the verifier-only data)"""
    c = common["config"]
    out = b"".join(_u64(c[k]) for k in ("num_wires", "num_routed_wires", "num_constants", "security_bits", "num_challenges",
                                        "max_quotient_degree_factor"))
    out += bytes([1 if c["use_base_arithmetic_gate"] else 0, 1 if c["zero_knowledge"] else 0])
    out += _fri_config_bytes(c["fri_config"])
    fp = common["fri_params"]
    out += _fri_config_bytes(fp["config"])
    out += _u64(len(fp["reduction_arity_bits"])) + b"".join(_u64(x) for x in fp["reduction_arity_bits"])
    out += _u64(fp["degree_bits"]) + bytes([1 if fp["hiding"] else 0])
    si = common["selectors_info"]
    out += _u64(len(si["selector_indices"])) + b"".join(_u64(x) for x in si["selector_indices"])
    out += _u64(len(si["groups"])) + b"".join(_u64(g["start"]) + _u64(g["end"]) for g in si["groups"])
    out += b"".join(_u64(common[k]) for k in ("quotient_degree_factor", "num_gate_constraints", "num_constants", "num_public_inputs"))
    out += _u64(len(common["k_is"])) + b"".join(_u64(x) for x in common["k_is"])
    out += _u64(common["num_partial_products"]) + _u64(common.get("num_lookup_polys", 0)) + _u64(common.get("num_lookup_selectors", 0))
    if common.get("luts"):
        raise ValueError("lookup tables are not used on this path")
    out += _u64(0)
    out += _u64(len(common["gates"])) + b"".join(_gate_bytes(g) for g in common["gates"])
    return out


def verifier_only_to_bytes(verifier_only, hasher):
    """`VerifierOnlyCircuitData::to_bytes`: cap height (u64), the cap's hashes, the circuit digest -- OutputTask.verifier_data"""
    cap = verifier_only["constants_sigmas_cap"]
    h = len(cap).bit_length() - 1
    assert 1 << h == len(cap)
    return _u64(h) + b"".join(S._hash_bytes(x, hasher) for x in cap) + S._hash_bytes(verifier_only["circuit_digest"], hasher)


def verifier_only_from_bytes(data, hasher):
    (h,) = struct.unpack_from("<Q", data, 0)
    if h > 16 or len(data) != 8 + 32 * ((1 << h) + 1):
        raise ValueError("not a VerifierOnlyCircuitData")
    rd = S._Reader(data[8:], hasher)
    return {"constants_sigmas_cap": [rd.hash() for _ in range(1 << h)], "circuit_digest": rd.hash()}


def verifier_data_to_bytes(verifier_only, common, hasher):
    """`VerifierCircuitData::to_bytes(&DefaultGateSerializer)`: verifier_data.bin of bin/prove_block.rs"""
    return verifier_only_to_bytes(verifier_only, hasher) + common_data_to_bytes(common)


# ------------------------------------------------------------------------------------------------- NATS messages
def input_task_to_json(message, approval, validator, signature_index):
    """InputTask as signatures.rs:188-197 builds it: approval = the 64 signature bytes (approval[2..]), validator = the 32
    public-key bytes of the borsh ValidatorStake"""
    if len(approval) != 64 or len(validator) != 32:
        raise ValueError("InputTask carries a 64-byte signature and a 32-byte public key")
    return json.dumps({"message": list(bytes(message)), "approval": list(bytes(approval)), "validator": list(bytes(validator)),
                       "signature_index": int(signature_index)}, separators=(",", ":")).encode()


def input_task_from_json(data):
    j = json.loads(data)
    out = {k: bytes(j[k]) for k in ("message", "approval", "validator")}
    out["signature_index"] = int(j["signature_index"])
    if out["signature_index"] < 0:
        raise ValueError("negative signature_index")
    return out


def output_task_to_json(proof_bytes, verifier_only, signature_index, hasher=S.HASH_GL):
    return json.dumps({"proof": list(bytes(proof_bytes)), "verifier_data": list(verifier_only_to_bytes(verifier_only, hasher)),
                       "signature_index": int(signature_index)}, separators=(",", ":")).encode()


def output_task_from_json(data, hasher=S.HASH_GL):
    j = json.loads(data)
    return {"proof": bytes(j["proof"]), "verifier_only": verifier_only_from_bytes(bytes(j["verifier_data"]), hasher),
            "signature_index": int(j["signature_index"])}


# ---------------------------------------------------------------------------------------------- Groth16 encodings
P_BN254 = 21888242871839275222246405745257275088696311157297823662689037894645226208583
_EXP_SQRT = (P_BN254 + 1) // 4
_F27_82 = 27 * pow(82, -1, P_BN254) % P_BN254      # Verifier.sol FRACTION_27_82_FP: real part of 3 / (9 + i)
_F3_82 = 3 * pow(82, -1, P_BN254) % P_BN254        # FRACTION_3_82_FP
_HALF = pow(2, -1, P_BN254)


class ProofInvalid(ValueError):
    """Verifier.sol `revert ProofInvalid()`"""


def _sqrt_fp(a):
    x = pow(a, _EXP_SQRT, P_BN254)
    if x * x % P_BN254 != a:
        raise ProofInvalid("not a square in Fp")
    return x


def _is_square(a):
    x = pow(a, _EXP_SQRT, P_BN254)
    return x * x % P_BN254 == a


def _sqrt_fp2(a0, a1, hint):
    d = _sqrt_fp((a0 * a0 + a1 * a1) % P_BN254)
    if hint:
        d = (P_BN254 - d) % P_BN254
    x0 = _sqrt_fp((a0 + d) * _HALF % P_BN254)
    inv = pow(2 * x0 % P_BN254, P_BN254 - 2, P_BN254)
    if 2 * x0 * inv % P_BN254 != 1:
        raise ProofInvalid("no inverse")
    x1 = a1 * inv % P_BN254
    if a0 != (x0 * x0 - x1 * x1) % P_BN254 or a1 != 2 * x0 * x1 % P_BN254:
        raise ProofInvalid("not a square in Fp2")
    return x0, x1


def compress_g1(x, y):
    if x >= P_BN254 or y >= P_BN254:
        raise ProofInvalid("G1 coordinate not reduced")
    if x == 0 and y == 0:
        return 0
    y_pos = _sqrt_fp((x * x * x + 3) % P_BN254)
    if y == y_pos:
        return x << 1
    if y == (P_BN254 - y_pos) % P_BN254:
        return (x << 1) | 1
    raise ProofInvalid("G1 point not on the curve")


def decompress_g1(c):
    if c == 0:
        return 0, 0
    x = c >> 1
    if x >= P_BN254:
        raise ProofInvalid("G1 x not reduced")
    y = _sqrt_fp((x * x * x + 3) % P_BN254)
    return x, (P_BN254 - y) % P_BN254 if c & 1 else y


def _g2_rhs(x0, x1):
    n3ab = x0 * x1 % P_BN254 * (P_BN254 - 3) % P_BN254
    a3, b3 = pow(x0, 3, P_BN254), pow(x1, 3, P_BN254)
    y0 = (_F27_82 + a3 + n3ab * x1) % P_BN254
    y1 = (P_BN254 - (_F3_82 + b3 + n3ab * x0) % P_BN254) % P_BN254
    return y0, y1


def compress_g2(x0, x1, y0, y1):
    if max(x0, x1, y0, y1) >= P_BN254:
        raise ProofInvalid("G2 coordinate not reduced")
    if (x0 | x1 | y0 | y1) == 0:
        return 0, 0
    r0, r1 = _g2_rhs(x0, x1)
    d = _sqrt_fp((r0 * r0 + r1 * r1) % P_BN254)
    hint = not _is_square((r0 + d) * _HALF % P_BN254)
    p0, p1 = _sqrt_fp2(r0, r1, hint)
    if (y0, y1) == (p0, p1):
        return (x0 << 2) | (2 if hint else 0), x1
    if (y0, y1) == ((P_BN254 - p0) % P_BN254, (P_BN254 - p1) % P_BN254):
        return (x0 << 2) | (2 if hint else 0) | 1, x1
    raise ProofInvalid("G2 point not on the curve")


def decompress_g2(c0, c1):
    if c0 == 0 and c1 == 0:
        return 0, 0, 0, 0
    x0, x1 = c0 >> 2, c1
    if x0 >= P_BN254 or x1 >= P_BN254:
        raise ProofInvalid("G2 x not reduced")
    y0, y1 = _sqrt_fp2(*_g2_rhs(x0, x1), bool(c0 & 2))
    if c0 & 1:
        y0, y1 = (P_BN254 - y0) % P_BN254, (P_BN254 - y1) % P_BN254
    return x0, x1, y0, y1


def compress_proof(proof8):
    """Verifier.sol:427-434 `compressProof`: uint256[8] (A.x, A.y, B.x1, B.x0, B.y1, B.y0, C.x, C.y) -> uint256[4]"""
    p = [int(x) for x in proof8]
    out = [compress_g1(p[0], p[1]), 0, 0, compress_g1(p[6], p[7])]
    out[2], out[1] = compress_g2(p[3], p[2], p[5], p[4])
    return out


def decompress_proof(c4):
    """the decompression at the head of `verifyCompressedProof` (Verifier.sol:446-449), back to the uint256[8] order"""
    c = [int(x) for x in c4]
    ax, ay = decompress_g1(c[0])
    bx0, bx1, by0, by1 = decompress_g2(c[2], c[1])
    cx, cy = decompress_g1(c[3])
    return [ax, ay, bx1, bx0, by1, by0, cx, cy]


def proof_to_raw_bytes(proof8):
    """gnark `proof.WriteRawTo` (web-api.go:90-98): the eight coordinates as 32-byte big-endian words"""
    return b"".join(int(x).to_bytes(32, "big") for x in proof8)


def proof_from_raw_bytes(raw):
    if len(raw) < 256:
        raise ValueError("a raw Groth16 proof has 256 bytes (+ commitment data, unused here)")
    return [int.from_bytes(raw[32 * i:32 * i + 32], "big") for i in range(8)]


def proof_to_web_api_json(proof8, inputs):
    """the JSON body POST /proof answers with (web-api.go:100-104: decimal strings)"""
    return {"inputs": [str(int(x)) for x in inputs], "proof": [str(int(x)) for x in proof8]}
