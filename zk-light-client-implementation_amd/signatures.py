"""Host-side mirror of the reference's approval pre-verification.

Reference: near_bft_finality/src/prove_block_data/signatures.rs
  * generate_signed_message            :24-39
  * the loop of prove_approvals        :56-123  (native `sig.verify` :79, the
    `valid_keys` layout :107-112, the stake accounting :59-68,113-118,
    `panic!` on an invalid signature :119-121)
and the byte layout constants of near_bft_finality/src/types.rs:7-17.

What the reference does one signature at a time on the CPU is done here as
ONE batched launch of the gfx950 Ed25519 kernel (Context.ed25519_verify_batch).
Proof generation (ed25519_proof_reuse_circuit / recursive_proof) is not part
of this function; it consumes the same (msg, sig, pk) triples.
"""
import numpy as np

TYPE_BYTE = 1
STAKE_BYTES = 16
PK_HASH_BYTES = 32
SIG_BYTES = 64


class InvalidSignature(Exception):
    """The reference panics ("Invalid signature.") -- signatures.rs:119-121."""

    def __init__(self, positions):
        self.positions = list(positions)
        super().__init__("Invalid signature at approval position(s) %s" % self.positions)


def generate_signed_message(ch_height: int, nb_height: int, nb_prev_hash: bytes) -> bytes:
    """Approval::get_data_for_sig: borsh(ApprovalInner) || le64(nb_height)."""
    if len(nb_prev_hash) != 32:
        raise ValueError("prev hash must be 32 bytes")
    if ch_height + 1 == nb_height:
        inner = b"\x00" + bytes(nb_prev_hash)  # Endorsement(prev_hash)
    else:
        inner = b"\x01" + int(ch_height).to_bytes(8, "little")  # Skip(height)
    return inner + int(nb_height).to_bytes(8, "little")


def slice_approvals(approvals, validators):
    """The borsh slicing of signatures.rs:72-86: returns (positions, pks[n,32], sigs[n,64])."""
    if len(approvals) != len(validators):
        raise ValueError("approvals and validators must have the same length")  # assert_eq! :56
    pos, pks, sigs = [], [], []
    for i, (ap, va) in enumerate(zip(approvals, validators)):
        if len(ap) == SIG_BYTES + 2 * TYPE_BYTE:  # Option tag + key type + 64-byte signature
            vl = len(va)
            if vl < STAKE_BYTES + PK_HASH_BYTES + TYPE_BYTE:
                raise ValueError("validator %d: borsh ValidatorStake too short" % i)
            if ap[0] != 1 or ap[1] != 0 or va[vl - STAKE_BYTES - PK_HASH_BYTES - TYPE_BYTE] != 0:
                raise ValueError("entry %d: only ED25519 keys/signatures are supported" % i)
            pos.append(i)
            sigs.append(bytes(ap[2:]))
            pks.append(bytes(va[vl - STAKE_BYTES - PK_HASH_BYTES:vl - STAKE_BYTES]))
    pk_arr = np.frombuffer(b"".join(pks), dtype=np.uint8).reshape(-1, 32) if pks else np.zeros((0, 32), np.uint8)
    sg_arr = np.frombuffer(b"".join(sigs), dtype=np.uint8).reshape(-1, 64) if sigs else np.zeros((0, 64), np.uint8)
    return pos, pk_arr, sg_arr


def verify_approvals(ctx, msg, approvals, validators, strict=True):
    """Batched form of the pre-check loop of prove_approvals.

    Returns (valid_keys, valid_positions, valid_stake, total_stake) where
    valid_keys = concat(pos as u8 || pk[32]) exactly as signatures.rs:107-112.
    strict=True raises InvalidSignature like the reference's panic; with
    strict=False invalid entries are simply left out.
    """
    pos, pks, sigs = slice_approvals(approvals, validators)
    total_stake = sum(int.from_bytes(bytes(v[len(v) - STAKE_BYTES:]), "little") for v in validators)
    ok = ctx.ed25519_verify_batch(pks, sigs, msg) if pos else np.zeros(0, np.uint8)
    bad = [p for p, o in zip(pos, ok) if not o]
    if bad and strict:
        raise InvalidSignature(bad)
    valid_keys = bytearray()
    valid_pos = []
    valid_stake = 0
    for p, o, pk in zip(pos, ok, pks):
        if o:
            valid_keys.append(p & 0xFF)  # `pos as u8` (signatures.rs:107)
            valid_keys += pk.tobytes()
            valid_pos.append(p)
            v = validators[p]
            valid_stake += int.from_bytes(bytes(v[len(v) - STAKE_BYTES:]), "little")
    return bytes(valid_keys), valid_pos, valid_stake, total_stake


class ApprovalProver:
    """`prove_approvals` (signatures.rs:43-141) on one GPU: batched native pre-check, one proof of the reference's Ed25519
    circuit per present approval (`ed25519_proof_reuse_circuit`, prove_crypto/ed25519.rs:44-64: circuits cached per message
    length), the left fold `agg = recursive_proof(agg, sig_i)` (:97-105) and the closing `recursive_proof(agg, None,
    sha256(valid_keys))` (:125-139).  Everything after the pre-check is plonky2 proving on the GPU
    (zklc_amd.plonky2.Prover); witnesses come from the native interpreter (csrc/plonky2_witness.cpp)."""

    def __init__(self, ctx, witness_threads=None, device_witness=True, witness_batch=8):
        from .plonky2 import HASH_GL
        from .plonky2.recursion import RecursionProver
        self.ctx = ctx
        self.threads = witness_threads
        self.device_witness, self.witness_batch = device_witness, max(1, min(64, witness_batch))
        self._dwit, self._dbuf = {}, {}
        self._ed = {}                       # message length in bits -> (CircuitData, targets, Prover, verifier_only)
        self.recursion = RecursionProver(ctx, HASH_GL, threads=witness_threads)

    def ed25519_circuit(self, msg_len_bytes, example=None):
        """get_ed25519_circuit_targets (ed25519.rs:18-42): build once per message length"""
        from .plonky2 import HASH_GL
        from .plonky2 import ed25519_circuit as E
        ent = self._ed.get(msg_len_bytes)
        if ent is None:
            data, targets, _ = E.build_cached(msg_len_bytes)
            prover = data.prover(self.ctx, HASH_GL)
            ent = self._ed[msg_len_bytes] = (data, targets, prover, prover.verifier_data())
        if example is not None and ent[0]._program is None:
            ent[0].witness_program(example)
        return ent

    def ed25519_proofs(self, msg, sigs, pks):
        """one proof per (signature, public key): (common, verifier_only, proof bytes) triples.  The witnesses are generated on
        the GPU in batches (csrc/plonky2_witness_dev.hip) and proven from HBM; `device_witness=False` uses the host interpreter"""
        from .plonky2 import ed25519_circuit as E
        data, targets, prover, vd = self.ed25519_circuit(len(msg))
        fills = [E.fill_ecdsa_targets(targets, msg, bytes(s), bytes(p)) for s, p in zip(sigs, pks)]
        if fills:
            self.ed25519_circuit(len(msg), example=fills[0])
        common = data.common_data()
        out = []
        if self.device_witness and fills:
            import torch
            dw = self._dwit.get(len(msg))
            if dw is None:
                dw = self._dwit[len(msg)] = data.device_witness(self.ctx)
            chunk = min(len(fills), self.witness_batch)
            # one wire-matrix buffer PER CIRCUIT (= per message length): the device interpreter writes only the cells its program has
            # slots for and relies on the rest being zero, so a buffer must never be shared between two circuits of the same shape.
            # The zero fill runs on torch's stream, the kernels on the context's non-blocking stream: wait for the fill.
            dbuf = self._dbuf.get(len(msg))
            if dbuf is None or dbuf.shape[0] < chunk:
                dbuf = self._dbuf[len(msg)] = torch.zeros((chunk, dw.num_wires, dw.n_rows), dtype=torch.int64,
                                                          device="cuda:%d" % self.ctx.device_id)
                torch.cuda.synchronize(self.ctx.device_id)
            for c0 in range(0, len(fills), chunk):
                pis = dw.run(dbuf.data_ptr(), fills[c0:c0 + chunk], stream=self.ctx.stream_ptr())
                for k in range(len(pis)):
                    out.append((common, vd, prover.prove_dev(dbuf[k].data_ptr(), [int(x) for x in pis[k]], stream=self.ctx.stream_ptr())))
            return out
        chunk = max(1, self.threads or 4)
        for c0 in range(0, len(fills), chunk):
            wires, pis = data.generate_witness_native(fills[c0:c0 + chunk], threads=self.threads)
            for k in range(len(wires)):
                out.append((common, vd, prover.prove_bytes(wires[k], [int(x) for x in pis[k]])))
        return out

    def _precheck(self, msg, approvals, validators):
        """the batched pre-check of one approval set, once: `valid_keys_early` and `prove_approvals` of the same block share it
        (one launch and one slicing instead of two and three)"""
        import hashlib
        h = hashlib.sha256(bytes(msg))          # keyed on CONTENT: a list mutated or re-allocated between the two calls is a different set
        for part in (approvals, validators):
            h.update(len(part).to_bytes(4, "little"))
            for x in part:
                h.update(len(x).to_bytes(4, "little") + bytes(x))
        key = h.digest()
        if getattr(self, "_pre", (None,))[0] != key:
            valid_keys, valid_pos, _, _ = verify_approvals(self.ctx, msg, approvals, validators, strict=True)
            _, pks, sigs = slice_approvals(approvals, validators)
            self._pre = (key, valid_keys, valid_pos, pks, sigs)
        return self._pre[1:]

    def valid_keys_early(self, msg, approvals, validators):
        """the `valid_keys` bytes prove_approvals will return: they only need the batched pre-check"""
        return self._precheck(msg, approvals, validators)[0]

    def prove_approvals(self, msg, approvals, validators, tree=False):
        """-> ((RecursiveCircuit, proof), valid_keys); raises InvalidSignature like the reference's panic (:119-121).
        tree=True (opt-in: a different proof tree than the reference's, same statement and public inputs) aggregates the
        signature proofs pairwise -- depth log2(n) instead of the serial chain of n - 1 fold steps of signatures.rs:97-105, every
        level's recursions independent of each other -- and closes with the same sha256(valid_keys) proof."""
        import hashlib
        valid_keys, valid_pos, pks, sigs = self._precheck(msg, approvals, validators)
        self._pre = (None,)                                   # one block's worth: the lists may be mutated or reused afterwards
        if not valid_pos:
            raise ValueError("no approvals present")         # the reference indexes agg_data_proof[0] (:131) and panics
        proofs = self.ed25519_proofs(msg, [s.tobytes() for s in sigs], [p.tobytes() for p in pks])
        agg = self.tree_fold(proofs) if tree else self.left_fold(proofs)
        rc, proof = self.recursion.recursive_proof(agg, None, list(hashlib.sha256(valid_keys).digest()))
        return (rc, proof), valid_keys

    def left_fold(self, proofs):
        """signatures.rs:97-105: agg = recursive_proof(agg, sig_i); proofs travel as `to_bytes` bytes between the steps (:225-230)"""
        agg = proofs[0]
        for nxt in proofs[1:]:
            rc, proof = self.recursion.recursive_proof(agg, nxt, raw=True)
            agg = (rc.common, rc.verifier_only, proof)
        return agg

    def tree_fold(self, proofs):
        """pairwise aggregation in signature order (an odd node moves up unchanged): ((p0 p1)(p2 p3)) .. -- the circuit shapes settle on
        R(ed, ed), R(R, R) and the few mixed ones an odd count needs, each built once (RecursionProver's cache)"""
        level = list(proofs)
        while len(level) > 1:
            nxt = []
            for i in range(0, len(level) - 1, 2):
                rc, proof = self.recursion.recursive_proof(level[i], level[i + 1], raw=True)
                nxt.append((rc.common, rc.verifier_only, proof))
            if len(level) % 2:
                nxt.append(level[-1])
            level = nxt
        return level[0]

    def close(self):
        for dw in self._dwit.values():
            dw.close()
        self._dwit, self._dbuf = {}, {}
        for _, _, prover, _ in self._ed.values():
            prover.close()
        self._ed = {}
        self.recursion.close()
