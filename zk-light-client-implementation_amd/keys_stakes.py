"""Host-side mirror of `prove_valid_keys_stakes_in_valiators_list`.

Reference: near_bft_finality/src/prove_block_data/keys_stakes.rs:18-266.  The circuit takes the borsh bytes of every validator
and the `valid_keys` list of prove_approvals (signatures.rs:107-112: position byte + 32-byte key per valid approval), ties each
listed key to the key bytes of the validator at that position, adds up the 16-byte little-endian stakes byte by byte (valid
ones and all of them), checks the byte-wise comparison 3 * valid >= 2 * all the way the reference writes it, and exposes
valid_keys and the 17-byte valid stake sum as public inputs; a SHA-256 proof of valid_keys (= the public inputs of the
approvals proof) and one recursion proof tie it together.  The circuit depends on the positions listed in valid_keys, exactly as
in the reference (:73-75), so it is built per call.
"""
from .plonky2.recursion import RecursiveCircuitBuilder
from .primitives import byte_and_carry, select_not_smaller, times_small, upper_bytes_match

STAKE_BYTES, PK_HASH_BYTES = 16, 32
STAKE_SUM_LEN = STAKE_BYTES + 1


def keys_stakes_circuit(valid_keys, validator_lens):
    """keys_stakes.rs:29-243: returns (CircuitData, validator byte targets, valid_keys byte targets)"""
    b = RecursiveCircuitBuilder()
    vt = [b.add_virtual_targets(n) for n in validator_lens]
    kt = b.add_virtual_targets(len(valid_keys))

    def add_stake(acc, val):
        crr = b.zero()
        for j in range(STAKE_BYTES):
            acc[j], crr = byte_and_carry(b, b.add(b.add(acc[j], val[len(val) - STAKE_BYTES + j]), crr))
        acc[STAKE_SUM_LEN - 1] = b.add(acc[STAKE_SUM_LEN - 1], crr)

    valid_sum = [b.zero()] * STAKE_SUM_LEN
    for i in range(0, len(valid_keys), PK_HASH_BYTES + 1):
        pos = valid_keys[i]
        val = vt[pos]
        for j in range(PK_HASH_BYTES):
            b.connect(val[len(val) - STAKE_BYTES - PK_HASH_BYTES + j], kt[i + 1 + j])
        add_stake(valid_sum, val)
    all_sum = [b.zero()] * STAKE_SUM_LEN
    for val in vt:
        add_stake(all_sum, val)
    seven, h = b.constant(7), b.constant(100)
    b.connect(upper_bytes_match(b, b.sub(valid_sum[STAKE_SUM_LEN - 1], h)), seven)
    b.connect(upper_bytes_match(b, b.sub(all_sum[STAKE_SUM_LEN - 1], h)), seven)
    three_valid, two_all = times_small(b, valid_sum, 3), times_small(b, all_sum, 2)
    res = select_not_smaller(b, three_valid, two_all, len(three_valid))
    for x, y in zip(three_valid, res):
        b.connect(x, y)
    for t in kt + valid_sum:
        b.register_public_input(t)
    return b.build(), vt, kt


class KeysStakesProver:
    def __init__(self, ctx, sha=None, recursion=None):
        from .plonky2 import HASH_GL
        from .plonky2.recursion import RecursionProver
        from .plonky2.sha256 import Sha256Prover
        self.ctx = ctx
        self.sha = sha or Sha256Prover(ctx, HASH_GL)
        self.recursion = recursion or RecursionProver(ctx, HASH_GL)
        # the circuit depends on WHICH validators signed (keys_stakes.rs:73-75) and on the lengths of the validator records, not on
        # the key or stake bytes: the last few shapes stay resident (the reference rebuilds the circuit on every call)
        self._cache, self.cache_size, self.cache_hits = [], 4, 0

    def prove_valid_keys_stakes_in_validators_list(self, valid_keys, valid_keys_hash, validators):
        """-> (common, verifier_only, proof) of the aggregated proof (keys_stakes.rs:244-265); public inputs = valid_keys bytes
        then the 17 bytes of the valid stake sum.  Raises AssertionError when the stake condition or the hash does not hold."""
        from .plonky2 import HASH_GL
        valid_keys = bytes(valid_keys)
        shape = (tuple(valid_keys[0::PK_HASH_BYTES + 1]), tuple(len(v) for v in validators))
        ent = next((e for e in self._cache if e[0] == shape), None)
        if ent is None:
            from .plonky2.circuit_cache import load_or_build

            def build():
                data, vt, kt = keys_stakes_circuit(valid_keys, [len(v) for v in validators])
                data.witness_program([t for ts in vt for t in ts] + list(kt))
                return data, {"vt": vt, "kt": kt}
            data, aux, _ = load_or_build("keys_stakes", shape, build)
            vt, kt = aux["vt"], aux["kt"]
            ent = (shape, data, vt, kt, data.prover(self.ctx, HASH_GL))
            self._cache.append(ent)
            if len(self._cache) > self.cache_size:
                self._cache.pop(0)[4].close()
        else:
            self.cache_hits += 1
        _, data, vt, kt, prover = ent
        pw = {t: x for ts, v in zip(vt, validators) for t, x in zip(ts, bytes(v))}
        pw.update(zip(kt, valid_keys))
        if data._program is None:
            data.witness_program(list(pw))
        wires, pis = data.generate_witness_native([pw])
        ks = (data.common_data(), prover.verifier_data(), prover.prove(wires[0], [int(x) for x in pis[0]]))
        keys = bytes(int(x) & 0xFF for x in ks[2]["public_inputs"][:len(valid_keys)])
        (hc, hv), hp = self.sha.sha256_proof_u32(keys, valid_keys_hash)
        rc, proof = self.recursion.recursive_proof(ks, (hc, hv, hp), ks[2]["public_inputs"])
        return rc.common, rc.verifier_only, proof

    def close_circuits(self):
        """the keys / stakes circuits resident on the GPU (not the SHA-256 / recursion provers, which may be shared)"""
        for e in self._cache:
            e[4].close()
        self._cache = []

    def close(self):
        self.close_circuits()
        self.sha.close()
        self.recursion.close()
