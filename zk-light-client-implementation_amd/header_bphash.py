"""Host-side mirror of the reference's block-hash proofs.

Reference: near_bft_finality/src/prove_crypto/sha256.rs:108-172 (`prove_sub_hashes_u32`) and
near_bft_finality/src/prove_block_data/header_bphash.rs:34-139 (`prove_header_hash`, `prove_bp_hash`): compositions of
`sha256_proof_u32` and `recursive_proof`, both of which run on the GPU here (zklc_amd.plonky2.sha256.Sha256Prover,
zklc_amd.plonky2.recursion.RecursionProver).  NEAR's block hash is
    sha256( sha256( sha256(inner_lite) || sha256(inner_rest) ) || prev_hash )
and every one of the three outer hashes gets its own SHA-256 proof, tied to the proofs of its inputs by a recursion proof
whose public inputs carry the digests as u32 words.  Proofs are (common_data, verifier_only, proof) triples, as in recursion.py.
"""
import hashlib


def vec_u32_to_u8(words):
    """near_bft_finality/src/utils.rs:15-25: big-endian bytes of each u32"""
    return b"".join(int(w).to_bytes(4, "big") for w in words)


class BlockHashProver:
    def __init__(self, ctx, sha=None, recursion=None):
        from .plonky2 import HASH_GL
        from .plonky2.recursion import RecursionProver
        from .plonky2.sha256 import Sha256Prover
        self.sha = sha or Sha256Prover(ctx, HASH_GL)
        self.recursion = recursion or RecursionProver(ctx, HASH_GL)

    def _sha(self, msg, digest):
        (common, vd), proof = self.sha.sha256_proof_u32(msg, digest)
        return (common, vd, proof)

    def _rec(self, first, second, pis):
        rc, proof = self.recursion.recursive_proof(first, second, pis)
        return (rc.common, rc.verifier_only, proof)

    def prove_sub_hashes_u32(self, set_pis_1, set_pis_2, pis_hash_1, pis_hash_2, final_hash, first, second=None):
        """sha256.rs:108-172: verify one or two hash proofs, then prove that `final_hash` is the SHA-256 of the concatenation of
        their digests (of the first digest and the bytes `pis_hash_2` when there is no second proof); the result carries the
        final digest words as public inputs"""
        pis = []
        if set_pis_1:
            pis += [int(x) for x in pis_hash_1]
        if set_pis_2:
            pis += [int(x) for x in pis_hash_2]
        inner = self._rec(first, second, pis if (set_pis_1 or set_pis_2) else None)
        msg = vec_u32_to_u8([int(x) & 0xFFFFFFFF for x in inner[2]["public_inputs"]])
        if second is None:
            msg += bytes(int(x) & 0xFF for x in pis_hash_2)
        final = bytes(final_hash) if final_hash is not None else hashlib.sha256(msg).digest()
        hp = self._sha(msg, final)
        return self._rec(inner, hp, hp[2]["public_inputs"])

    def prove_header_hash(self, header_hash, prev_hash, inner_lite, inner_rest, public_inputs=None):
        """header_bphash.rs:34-111 (`HeaderData` = prev_hash, inner_lite, inner_rest)"""
        p1 = self._sha(inner_lite, hashlib.sha256(inner_lite).digest())
        p2 = self._sha(inner_rest, hashlib.sha256(inner_rest).digest())
        p3 = self.prove_sub_hashes_u32(True, True, p1[2]["public_inputs"], p2[2]["public_inputs"], None, p1, p2)
        p4 = self.prove_sub_hashes_u32(True, False, p3[2]["public_inputs"], list(bytes(prev_hash)), header_hash, p3, None)
        if public_inputs is not None:
            return self._rec(p4, None, [int(x) for x in public_inputs])
        return p4

    def prove_bp_hash(self, bp_hash, validators):
        """header_bphash.rs:121-139: bp_hash = sha256(borsh(Vec<ValidatorStake>)) = sha256(le32(len) || entries)"""
        data = len(validators).to_bytes(4, "little") + b"".join(bytes(v) for v in validators)
        return self._sha(data, bp_hash)

    def close(self):
        self.sha.close()
        self.recursion.close()
