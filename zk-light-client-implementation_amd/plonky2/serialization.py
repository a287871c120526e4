"""plonky2 proof <-> bytes / JSON, the formats the reference writes next to every proof
(near_bft_finality/src/bin/prove_block.rs:320-458: proof.bin = `ProofWithPublicInputs::to_bytes()`,
proof.json = serde_json of the same struct) and that gnark-plonky2-verifier reads
(types/deserialize.go, variables/deserialize.go).

Binary layout (plonky2 `util/serialization.rs` `write_proof_with_public_inputs`; no length prefixes except
where noted -- every length follows from common_data):
  wires_cap | zs_partial_products_cap | quotient_polys_cap            2^cap_height hashes each
  openings: constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys
            (+ lookup_zs, lookup_zs_next: empty here)                   extension elements, 2 x u64 LE
  commit_phase_merkle_caps                                              one cap per FRI reduction
  per query round: 4 x (leaf elements u64 LE, u8 sibling count, siblings)
                   per reduction: arity extension elements, u8 sibling count, siblings
  final_poly coefficients | pow_witness u64
  u64 LE count of public inputs | public inputs u64 LE
Hash encodings: Poseidon-Goldilocks digest = 4 x u64 LE; Poseidon-BN128 digest = 32 bytes little-endian
(crypto/plonky2_bn128/src/config.rs:36-70 `to_bytes`).  JSON: Goldilocks digest {"elements": [..]}, BN128 digest a
decimal string.  Pinned by the reference's golden pair proof.bin / proof.json (tests/test_plonky2_serialization.py).
"""
import struct

HASH_GL, HASH_BN128 = 0, 1


def shapes(common):
    cfg, fp = common["config"], common["fri_params"]
    nch = cfg["num_challenges"]
    routed = cfg["num_routed_wires"]
    return {
        "cap": 1 << fp["config"]["cap_height"],
        "openings": [("constants", common["num_constants"]), ("plonk_sigmas", routed), ("wires", cfg["num_wires"]),
                     ("plonk_zs", nch), ("plonk_zs_next", nch), ("partial_products", nch * common["num_partial_products"]),
                     ("quotient_polys", nch * common["quotient_degree_factor"])],
        "leaf_widths": [common["num_constants"] + routed, cfg["num_wires"], nch * (1 + common["num_partial_products"]),
                        nch * common["quotient_degree_factor"]],
        "arity_bits": fp["reduction_arity_bits"],
        "lde_bits": fp["degree_bits"] + fp["config"]["rate_bits"],
        "cap_height": fp["config"]["cap_height"],
        "final_len": 1 << (fp["degree_bits"] - sum(fp["reduction_arity_bits"])),
        "rounds": fp["config"]["num_query_rounds"],
    }


class _Reader:
    """offsets=True: every field element is reported as its byte offset instead of its value (proof_offsets)"""

    def __init__(self, b, hasher, offsets=False):
        self.b, self.o, self.hasher, self.offsets = memoryview(b), 0, hasher, offsets

    def u64s(self, n):
        if self.offsets:
            v = [self.o + 8 * i for i in range(n)]
            if self.o + 8 * n > len(self.b):
                raise ValueError("proof bytes too short")
        else:
            v = struct.unpack_from("<%dQ" % n, self.b, self.o)
        self.o += 8 * n
        return list(v)

    def count(self):
        v = struct.unpack_from("<Q", self.b, self.o)[0]
        self.o += 8
        return v

    def exts(self, n):
        v = self.u64s(2 * n)
        return [[v[2 * i], v[2 * i + 1]] for i in range(n)]

    def hash(self):
        if self.hasher == HASH_GL:
            return {"elements": self.u64s(4)}
        v = int.from_bytes(self.b[self.o:self.o + 32], "little")
        self.o += 32
        return str(v)

    def merkle_proof(self):
        n = self.b[self.o]
        self.o += 1
        return {"siblings": [self.hash() for _ in range(n)]}


def proof_offsets(template, common, hasher):
    """the proof_from_bytes structure of `template` (any proof of the circuit) with byte offsets in place of the field
    elements: proofs of one circuit all have the same layout, so a consumer can gather its inputs straight from the bytes"""
    assert hasher == HASH_GL
    return proof_from_bytes(template, common, hasher, offsets=True)


def proof_from_bytes(data, common, hasher, offsets=False):
    sh = shapes(common)
    r = _Reader(data, hasher, offsets)
    cap = lambda: [r.hash() for _ in range(sh["cap"])]
    proof = {"wires_cap": cap(), "plonk_zs_partial_products_cap": cap(), "quotient_polys_cap": cap()}
    proof["openings"] = {k: r.exts(n) for k, n in sh["openings"]}
    proof["openings"]["lookup_zs"] = []
    proof["openings"]["lookup_zs_next"] = []
    caps = [cap() for _ in sh["arity_bits"]]
    rounds = []
    for _ in range(sh["rounds"]):
        init = [[r.u64s(w), r.merkle_proof()] for w in sh["leaf_widths"]]
        steps = [{"evals": r.exts(1 << a), "merkle_proof": r.merkle_proof()} for a in sh["arity_bits"]]
        rounds.append({"initial_trees_proof": {"evals_proofs": init}, "steps": steps})
    final = r.exts(sh["final_len"])
    pow_witness = r.u64s(1)[0]
    proof["opening_proof"] = {"commit_phase_merkle_caps": caps, "query_round_proofs": rounds, "final_poly": {"coeffs": final},
                              "pow_witness": pow_witness}
    npi = r.count()
    pis = r.u64s(npi)
    if r.o != len(data):
        raise ValueError("trailing bytes in proof: %d of %d consumed" % (r.o, len(data)))
    return {"proof": proof, "public_inputs": pis}


def _hash_bytes(h, hasher):
    if hasher == HASH_GL:
        return struct.pack("<4Q", *[int(x) for x in h["elements"]])
    return int(h).to_bytes(32, "little")


def proof_to_bytes(pj, common, hasher):
    sh = shapes(common)
    out = bytearray()
    p = pj["proof"]
    for key in ("wires_cap", "plonk_zs_partial_products_cap", "quotient_polys_cap"):
        for h in p[key]:
            out += _hash_bytes(h, hasher)
    for k, n in sh["openings"]:
        assert len(p["openings"][k]) == n, k
        for a, b in p["openings"][k]:
            out += struct.pack("<2Q", int(a), int(b))
    op = p["opening_proof"]
    for cap in op["commit_phase_merkle_caps"]:
        for h in cap:
            out += _hash_bytes(h, hasher)

    def mp(m):
        out.append(len(m["siblings"]))
        for h in m["siblings"]:
            out.extend(_hash_bytes(h, hasher))
    for q in op["query_round_proofs"]:
        for leaf, m in q["initial_trees_proof"]["evals_proofs"]:
            out += struct.pack("<%dQ" % len(leaf), *[int(x) for x in leaf])
            mp(m)
        for st in q["steps"]:
            for a, b in st["evals"]:
                out += struct.pack("<2Q", int(a), int(b))
            mp(st["merkle_proof"])
    for a, b in op["final_poly"]["coeffs"]:
        out += struct.pack("<2Q", int(a), int(b))
    out += struct.pack("<Q", int(op["pow_witness"]))
    out += struct.pack("<Q", len(pj["public_inputs"]))
    out += struct.pack("<%dQ" % len(pj["public_inputs"]), *[int(x) for x in pj["public_inputs"]])
    return bytes(out)


def proof_size(common, hasher):
    """exact size in bytes of a serialised proof for this circuit (num_public_inputs from common_data)"""
    sh = shapes(common)
    hb = 32
    n = 3 * sh["cap"] * hb + 16 * sum(k for _, k in sh["openings"]) + len(sh["arity_bits"]) * sh["cap"] * hb
    per = 0
    depth = sh["lde_bits"] - sh["cap_height"]
    for w in sh["leaf_widths"]:
        per += 8 * w + 1 + depth * hb
    bits = sh["lde_bits"]
    for a in sh["arity_bits"]:
        bits -= a
        per += 16 * (1 << a) + 1 + max(bits - sh["cap_height"], 0) * hb
    n += per * sh["rounds"] + 16 * sh["final_len"] + 8 + 8 + 8 * common["num_public_inputs"]
    return n


def write_proof_files(directory, common, verifier_only, proof, hasher, prefix=""):
    """The files near_bft_finality/src/bin/prove_block.rs:320-458 writes next to a proof (`<prefix>proof.bin`, `<prefix>proof.json`,
    `<prefix>common_data.json`, `<prefix>verifier_data.json`; prefix "b0_" / "bn_" for the epoch blocks, "" otherwise) and, when the
    proof has the 33+ public inputs of a block proof, `<prefix>hash.json` = hex of public inputs 1..33 (:305-311,441-443).
    `proof` is the JSON form or the `to_bytes` bytes.  `<prefix>verifier_data.bin` (VerifierCircuitData::to_bytes with plonky2's
    default gate serializer, zklc_amd/formats.py) is written when every gate of the circuit is in that serializer, which is the
    case for the wrap proofs the reference writes."""
    import json
    import os
    os.makedirs(directory, exist_ok=True)
    if isinstance(proof, (bytes, bytearray, memoryview)):
        raw, pj = bytes(proof), proof_from_bytes(bytes(proof), common, hasher)
    else:
        raw, pj = proof_to_bytes(proof, common, hasher), proof
    with open(os.path.join(directory, prefix + "proof.bin"), "wb") as f:
        f.write(raw)
    for name, obj in (("proof.json", pj), ("common_data.json", common), ("verifier_data.json", verifier_only)):
        with open(os.path.join(directory, prefix + name), "w") as f:
            json.dump(obj, f, indent=2)
    from .. import formats
    try:
        vbin = formats.verifier_data_to_bytes(verifier_only, common, hasher)
    except ValueError:
        vbin = None          # a gate outside plonky2's DefaultGateSerializer (the u32 gates of the inner circuits)
    if vbin is not None:
        with open(os.path.join(directory, prefix + "verifier_data.bin"), "wb") as f:
            f.write(vbin)
    pis = pj["public_inputs"]
    if len(pis) >= 33 and all(int(x) < 256 for x in pis[1:33]):
        with open(os.path.join(directory, prefix + "hash.json"), "w") as f:
            f.write(bytes(int(x) for x in pis[1:33]).hex())


def read_proof_files(directory, hasher, prefix=""):
    """(common, verifier_only, proof json) as the reference reads them back (signatures.rs:225-230 from bytes; the gnark service
    from the JSON files): proof.bin wins over proof.json when both exist"""
    import json
    import os
    with open(os.path.join(directory, prefix + "common_data.json")) as f:
        common = json.load(f)
    with open(os.path.join(directory, prefix + "verifier_data.json")) as f:
        vd = json.load(f)
    pbin = os.path.join(directory, prefix + "proof.bin")
    if os.path.exists(pbin):
        with open(pbin, "rb") as f:
            proof = proof_from_bytes(f.read(), common, hasher)
    else:
        with open(os.path.join(directory, prefix + "proof.json")) as f:
            proof = json.load(f)
    return common, vd, proof
