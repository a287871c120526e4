"""Host-side mirror of the reference's block-finality DAG (BASELINE configs[2]: the full Block_i proof).

Reference:
  near_bft_finality/src/prove_bft/block_finality.rs   prove_consecutive_heights_proofs :30-96, prove_block_header :98-198,
                                                      prove_block_finality :200-650
  near_bft_finality/src/prove_bft/bft.rs              prove_block_bft :38-500 (5 blocks = a randomly selected block, 6 = epoch blocks)
Every leaf is one of the proofs this package already produces on the GPU -- SHA-256 chains for header hashes and bp_hash
(header_bphash.py), the signature fold (signatures.ApprovalProver), keys / stakes (keys_stakes.py), equalities and
consecutive heights (primitives.py) -- and every inner node is `recursive_proof`.  The order of the recursions, the public
inputs each one exposes and the byte offsets into them are those of the reference; proofs are (common_data, verifier_only,
proof) triples.  Block headers are borsh bytes ([version 1][prev_hash 32][inner_lite 208][inner_rest ..][key type 1][sig 64]).
"""
from . import signatures as SG

TYPE_BYTE, PK_HASH_BYTES, INNER_LITE_BYTES, BLOCK_HEIGHT_BYTES, SIG_BYTES = 1, 32, 208, 8, 64


def pi_bytes(proof, lo, hi=None):
    pis = proof[2]["public_inputs"]
    return bytes(int(x) & 0xFF for x in pis[lo:hi])


class BlockProver:
    def __init__(self, ctx, approval_prover=None, parts=None):
        """parts = (approvals, hashes, keys, primitives) overrides the GPU provers (the CPU test of the DAG wiring passes
        stand-ins that check their inputs and return the public inputs a real proof would carry)"""
        self.counts, self.seconds = {}, {}
        self.ctx, self._pipeline = ctx, None
        if parts is not None:
            self.approvals, self.hashes, self.keys, self.prims = parts
            self.recursion = self.approvals.recursion
            return
        from .header_bphash import BlockHashProver
        from .keys_stakes import KeysStakesProver
        from .primitives import PrimitiveProver
        self.approvals = approval_prover or SG.ApprovalProver(ctx)
        self.recursion = self.approvals.recursion
        self.hashes = BlockHashProver(ctx)
        self.hashes.recursion.close()
        self.hashes.recursion = self.recursion
        self.keys = KeysStakesProver(ctx, sha=self.hashes.sha, recursion=self.recursion)
        self.prims = PrimitiveProver(ctx)

    def _timed(self, what, fn, *a, **kw):
        import time
        t0 = time.perf_counter()
        r = fn(*a, **kw)
        self.counts[what] = self.counts.get(what, 0) + 1
        self.seconds[what] = self.seconds.get(what, 0.0) + time.perf_counter() - t0
        return r

    def _rec(self, first, second=None, pis=None):
        rc, proof = self._timed("recursive_proof", self.recursion.recursive_proof, first, second,
                                None if pis is None else [int(x) for x in pis])
        return (rc.common, rc.verifier_only, proof)

    def _eq(self, a, b):
        return self._timed("prove_eq_array", self.prims.prove_eq_array, a, b)

    # ---- block_finality.rs:30-96
    def prove_consecutive_heights_proofs(self, proofs):
        if len(proofs) < 3:
            raise ValueError("prove_consecutive_heights_proofs needs at least three header proofs")
        h = [pi_bytes(p, 32, 40) for p in proofs]
        p1 = self.prims.prove_consecutive_heights(h[0], h[1])
        p2 = self.prims.prove_consecutive_heights(h[1], h[2])
        agg = self._rec(p1, p2)
        if len(proofs) == 4:
            agg = self._rec(agg, self.prims.prove_consecutive_heights(h[2], h[3]))
        return agg

    # ---- block_finality.rs:98-198
    def prove_block_header(self, hash_bytes, block_bytes, height=None, epoch_id=None, prev_hash=None, last_ds_final_hash=None,
                           last_final_hash=None, bp_hash=None, next_epoch_id=None):
        pis = bytes(hash_bytes)
        if height is not None:
            pis += int(height).to_bytes(8, "little")
        for part in (epoch_id, prev_hash, last_ds_final_hash, last_final_hash):
            if part is not None:
                pis += bytes(part)
        if bp_hash is not None:
            assert len(pis) == 32
            pis += bytes(bp_hash)
            if next_epoch_id is not None:
                pis += bytes(next_epoch_id)
        o = TYPE_BYTE + PK_HASH_BYTES
        return self._timed("prove_header_hash", self.hashes.prove_header_hash, hash_bytes, block_bytes[TYPE_BYTE:o],
                           block_bytes[o:o + INNER_LITE_BYTES],
                           block_bytes[o + INNER_LITE_BYTES:len(block_bytes) - TYPE_BYTE - SIG_BYTES], list(pis))

    # ---- block_finality.rs:200-650
    def prove_block_finality(self, current_block_header_proof, msg_to_sign, next_block_approvals, validators, proofs,
                             consecutive_heights):
        """The nodes are the reference's; the host evaluates the ones that do not depend on the signature aggregate first (the
        keys / stakes proof needs only valid_keys, which the GPU pre-check yields at once, and sha256(valid_keys)), so that after
        `prove_approvals` returns only the joining recursions remain."""
        import hashlib
        cur_hash = pi_bytes(current_block_header_proof, 0, 32)
        cur_epoch_id = pi_bytes(current_block_header_proof, 40, 72)
        if not 3 <= len(proofs) <= 4:
            raise ValueError("prove_block_finality takes 3 or 4 header proofs, got %d" % len(proofs))
        ks = None
        ks_elsewhere = msg_to_sign is not None and hasattr(self.approvals, "keys_stakes_early")   # proven by another thread
        if msg_to_sign is not None and not ks_elsewhere and hasattr(self.approvals, "valid_keys_early"):
            vk = self.approvals.valid_keys_early(msg_to_sign, next_block_approvals, validators)
            ks = self._timed("prove_valid_keys_stakes", self.keys.prove_valid_keys_stakes_in_validators_list, vk,
                             hashlib.sha256(vk).digest(), validators)
        block_n_1 = self._rec(proofs[0], self._eq(cur_epoch_id, pi_bytes(proofs[0], 0, 32)), proofs[0][2]["public_inputs"][0:32])
        if validators is not None:
            bp = self._timed("prove_bp_hash", self.hashes.prove_bp_hash, pi_bytes(proofs[1], 32, 64), validators)
            block_0 = self._rec(proofs[1], bp, proofs[1][2]["public_inputs"][0:32])
        else:
            block_0 = self._rec(proofs[1], None, proofs[1][2]["public_inputs"][0:32])
        agg = self._rec(block_n_1, block_0, block_n_1[2]["public_inputs"] + block_0[2]["public_inputs"])
        prev_hash_p = self._eq(pi_bytes(proofs[2], 72, 104), cur_hash)
        n2 = len(proofs[2][2]["public_inputs"])
        if consecutive_heights is not None:
            ds_p = self._eq(pi_bytes(proofs[2], n2 - 64, n2 - 32), cur_hash)
            inner = self._rec(prev_hash_p, self._rec(ds_p, consecutive_heights))
        else:
            inner = self._rec(prev_hash_p)
        tail = self._rec(proofs[2], inner)                                            # Block_i+1
        if len(proofs) == 4:
            if consecutive_heights is not None:
                n3 = len(proofs[3][2]["public_inputs"])
                block_i_2 = self._rec(proofs[3], self._eq(pi_bytes(proofs[3], n3 - 32), cur_hash))
            else:
                block_i_2 = self._rec(proofs[3])
            tail = self._rec(tail, block_i_2)
        aggregation = agg
        if ks_elsewhere:
            ks = self._timed("wait_keys_stakes", self.approvals.keys_stakes_early, msg_to_sign, next_block_approvals, validators)
        if msg_to_sign is not None:
            (rc, sig_proof), valid_keys = self._timed("prove_approvals", self.approvals.prove_approvals, msg_to_sign,
                                                      next_block_approvals, validators)
            sig = (rc.common, rc.verifier_only, sig_proof)
            if ks is None:
                ks = self._timed("prove_valid_keys_stakes", self.keys.prove_valid_keys_stakes_in_validators_list, valid_keys,
                                 pi_bytes(sig, 0), validators)
            elif pi_bytes(sig, 0) != hashlib.sha256(valid_keys).digest():
                raise ValueError("the signature aggregate does not carry sha256(valid_keys) of the approvals it was given")
            aggregation = self._rec(self._rec(sig, ks, ks[2]["public_inputs"]), agg, agg[2]["public_inputs"])
        aggregation = self._rec(aggregation, tail, aggregation[2]["public_inputs"])
        pis = current_block_header_proof[2]["public_inputs"] + aggregation[2]["public_inputs"]
        return self._rec(aggregation, current_block_header_proof, pis)

    # ---- the block-header proofs of prove_block_bft (bft.rs:64-205): independent leaves of the DAG
    def header_jobs(self, ep2_last_block_bytes, ep2_last_block_hash, ep1_first_block_bytes, ep1_first_block_hash, blocks,
                    ep3_last_block_bytes=None, ep3_last_block_hash=None):
        """{name: (hash_bytes, block_bytes, keyword arguments of prove_block_header)} in the order the reference proves them.
        The sharded driver (zklc_amd.distributed) hands each job to a rank; prove_block_bft takes the results back through
        `header_proofs`."""
        o = TYPE_BYTE + PK_HASH_BYTES + INNER_LITE_BYTES

        def bp_hash_of(b):
            return b[o - 2 * PK_HASH_BYTES:o - PK_HASH_BYTES]
        q = TYPE_BYTE + PK_HASH_BYTES + BLOCK_HEIGHT_BYTES + PK_HASH_BYTES
        jobs = {"ep2_lb": (ep2_last_block_hash, ep2_last_block_bytes, {"bp_hash": bp_hash_of(ep2_last_block_bytes)}),
                "ep1_fb": (ep1_first_block_hash, ep1_first_block_bytes, {"bp_hash": bp_hash_of(ep1_first_block_bytes),
                                                                         "next_epoch_id": ep1_first_block_bytes[q:q + PK_HASH_BYTES]})}

        def header(name, k, *names):
            f, raw = blocks[k]
            jobs[name] = (f["hash"], raw, {nm: f[nm] for nm in names})
        header("b4", 0, "height", "epoch_id", "prev_hash")
        header("b3", 1, "height", "epoch_id", "prev_hash")
        header("b2", 2, "height", "epoch_id", "prev_hash", "last_ds_final_hash", "last_final_hash")
        header("b1", 3, "height", "epoch_id", "prev_hash", "last_ds_final_hash", "last_final_hash")
        if len(blocks) == 5:
            header("bi0", 4, "height", "epoch_id")
        elif len(blocks) == 6:
            header("bi0", 4, "height", "epoch_id", "prev_hash", "last_ds_final_hash")
            header("bn_1", 5, "height", "epoch_id")
            if ep3_last_block_bytes is not None:
                jobs["ep3_lb"] = (ep3_last_block_hash, ep3_last_block_bytes, {})
        else:
            raise ValueError("Invalid blocks.len() %d" % len(blocks))
        return jobs

    def prove_header_job(self, job):
        return self.prove_block_header(job[0], job[1], **job[2])

    # ---- bft.rs:38-500
    def prove_block_bft(self, ep2_last_block_bytes, ep2_last_block_hash, ep1_first_block_bytes, ep1_first_block_hash, blocks,
                        validators, ep3_last_block_bytes=None, ep3_last_block_hash=None, validators_n_1=None, header_proofs=None,
                        pipelined=False):
        """blocks: [(fields, header bytes)] in the order [Block_i+4, .., Block_i] (a randomly selected block) or
        [Block_4, .., Block_0, Block_n-1] (epoch blocks); fields = dict with hash, height, prev_hash, epoch_id,
        last_ds_final_hash, last_final_hash, approvals.  Returns (proof of Block_i / Block_0, proof of Block_n-1 or None).
        header_proofs: {name: proof} or a callable name -> proof for header proofs made elsewhere (header_jobs).
        pipelined=True: the same DAG through zklc_amd.pipeline.BlockPipeline (several proofs in flight on their own HIP streams,
        witnesses on the GPU) -- byte-identical proofs, a fraction of the time; its contexts and circuits are created on first use."""
        if pipelined:
            from .pipeline import BlockPipeline, BlockWindow
            if header_proofs is not None:
                raise ValueError("prove_block_bft(pipelined=True) proves its own header proofs: header_proofs must be None")
            if self.ctx is None:
                raise ValueError("prove_block_bft(pipelined=True) needs a BlockProver constructed over a zklc Context")
            if self._pipeline is None:
                # the pipeline owns its own contexts, resident circuits and a nested BlockProver (beside this object's sequential
                # provers: ~2x the resident set; construct a BlockPipeline directly when only the pipelined form is wanted)
                self._pipeline = BlockPipeline(self.ctx.device_id, wrap=False)
            res = self._pipeline.prove_block_bft(BlockWindow(
                (ep2_last_block_bytes, ep2_last_block_hash), (ep1_first_block_bytes, ep1_first_block_hash), blocks, validators,
                (ep3_last_block_bytes, ep3_last_block_hash) if ep3_last_block_bytes is not None else None, validators_n_1))
            for k, v in res.dag_counts.items():
                self.counts[k] = self.counts.get(k, 0) + v
            for k, v in res.dag_seconds.items():
                self.seconds[k] = self.seconds.get(k, 0.0) + v
            return res.block, res.block_n_1
        jobs = self.header_jobs(ep2_last_block_bytes, ep2_last_block_hash, ep1_first_block_bytes, ep1_first_block_hash, blocks,
                                ep3_last_block_bytes, ep3_last_block_hash)

        def hp(name):
            got = None
            if callable(header_proofs):
                got = header_proofs(name)
            elif header_proofs is not None:
                got = header_proofs.get(name)
            return got if got is not None else self.prove_header_job(jobs[name])
        ep2_lb = hp("ep2_lb")
        ep1_fb = hp("ep1_fb")
        n = len(ep1_fb[2]["public_inputs"])
        neph = self._eq(pi_bytes(ep2_lb, 0, 32), pi_bytes(ep1_fb, n - 32))
        ep1_fb = self._rec(ep1_fb, neph, ep1_fb[2]["public_inputs"])
        b4, b3, b2 = hp("b4"), hp("b3"), hp("b2")
        b2 = self._rec(b2, self.prove_consecutive_heights_proofs([b4, b3, b2]), b2[2]["public_inputs"])
        b1 = hp("b1")
        bi0 = hp("bi0")
        bn_1_header = hp("bn_1") if len(blocks) == 6 else None
        hts = [int.from_bytes(pi_bytes(p, 32, 40), "little") for p in (b2, b1, bi0)]
        chain = [b2, b1, bi0]
        if bn_1_header is not None:
            hts.append(int.from_bytes(pi_bytes(bn_1_header, 32, 40), "little"))
            chain.append(bn_1_header)
        # (the reference's test is h1 + 1 == h2 && ... on heights that come in DEscending order: bft.rs:227-263)
        consecutive = self.prove_consecutive_heights_proofs(chain) if all(hts[k] + 1 == hts[k + 1] for k in range(len(hts) - 1)) else None

        def finality(header_proof, next_block, approvals, vals, proofs):
            msg = SG.generate_signed_message(int.from_bytes(pi_bytes(header_proof, 32, 40), "little"),
                                             int.from_bytes(pi_bytes(next_block, 32, 40), "little"), pi_bytes(next_block, 72, 104))
            return self.prove_block_finality(header_proof, msg, approvals, vals, proofs, consecutive)

        def three_hashes(p, flag):
            n_ = len(p[2]["public_inputs"])
            return self._rec(p, None, [flag] + p[2]["public_inputs"][0:32] + p[2]["public_inputs"][n_ - 64:])
        if len(blocks) == 5:
            bi = finality(bi0, b1, blocks[3][0]["approvals"], validators, [ep2_lb, ep1_fb, b1, b2])
            return three_hashes(bi, 0), None
        b0 = finality(bi0, b1, blocks[3][0]["approvals"], validators, [ep2_lb, ep1_fb, b1, b2])
        ep3_lb = hp("ep3_lb")
        b_n_1 = finality(bn_1_header, b0, blocks[4][0]["approvals"], validators_n_1, [ep3_lb, ep2_lb, b0, b1])
        return three_hashes(b0, 1), three_hashes(b_n_1, 1)

    def close(self):
        self.approvals.close()
        self.hashes.sha.close()
        self.prims.close()
        self.keys.close_circuits()       # the sha / recursion provers it shares with the others are closed above, once
        if self._pipeline is not None:
            self._pipeline.close()
            self._pipeline = None
