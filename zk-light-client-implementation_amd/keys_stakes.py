"""Host-side mirror of `prove_valid_keys_stakes_in_valiators_list`.

Reference: near_bft_finality/src/prove_block_data/keys_stakes.rs:18-266.  The circuit takes the borsh bytes of every validator
and the `valid_keys` list of prove_approvals (signatures.rs:107-112: position byte + 32-byte key per valid approval), ties each
listed key to the key bytes of the validator at that position, adds up the 16-byte little-endian stakes byte by byte (valid
ones and all of them), checks the byte-wise comparison 3 * valid >= 2 * all the way the reference writes it, and exposes
valid_keys and the 17-byte valid stake sum as public inputs; a SHA-256 proof of valid_keys (= the public inputs of the
approvals proof) and one recursion proof tie it together.  The circuit depends on the positions listed in valid_keys, exactly as
in the reference (:73-75), so it is built per call.
"""
from .plonky2.builder import P
from .plonky2.recursion import RecursiveCircuitBuilder

STAKE_BYTES, PK_HASH_BYTES = 16, 32
STAKE_SUM_LEN = STAKE_BYTES + 1
CONSTANT1, CONSTANT2 = 0xFFFFFFFEFFFFFF00, 0xFFFFFFFF00000000


def keys_stakes_circuit(valid_keys, validator_lens):
    """keys_stakes.rs:29-243: returns (CircuitData, validator byte targets, valid_keys byte targets)"""
    b = RecursiveCircuitBuilder()
    zero, neg_one = b.zero(), b.neg_one()
    vt = [b.add_virtual_targets(n) for n in validator_lens]
    kt = b.add_virtual_targets(len(valid_keys))

    def byte_and_carry(t):
        bits = b.split_le_63(t, 64)
        return b.le_sum_small(bits[0:8]), b.le_sum_small(bits[8:16])

    def add_stake(acc, val):
        crr = b.zero()
        for j in range(STAKE_BYTES):
            acc[j], crr = byte_and_carry(b.add(b.add(acc[j], val[len(val) - STAKE_BYTES + j]), crr))
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
    c1, c2, seven, h = b.constant(CONSTANT1), b.constant(CONSTANT2), b.constant(7), b.constant(100)

    def upper_bytes_match(sub):
        """:119-131 -- how many of the bytes 1..7 of `sub` equal those of the pattern chosen by (sub == -1)"""
        chs = b.select(b.is_equal(sub, neg_one), c2, c1)
        cb, sb = b.split_le_63(chs, 64), b.split_le_63(sub, 64)
        s = zero
        for j in range(8, 64, 8):
            s = b.add(s, b.is_equal(b.le_sum_small(cb[j:j + 8]), b.le_sum_small(sb[j:j + 8])))
        return s
    b.connect(upper_bytes_match(b.sub(valid_sum[STAKE_SUM_LEN - 1], h)), seven)
    b.connect(upper_bytes_match(b.sub(all_sum[STAKE_SUM_LEN - 1], h)), seven)
    three, two = b.constant(3), b.two()
    three_valid, crr = [], b.zero()
    for i in range(STAKE_SUM_LEN):
        lo, crr = byte_and_carry(b.mul_add(valid_sum[i], three, crr))
        three_valid.append(lo)
    three_valid.append(crr)
    two_all, crr = [], b.zero()
    for i in range(STAKE_SUM_LEN):
        lo, crr = byte_and_carry(b.mul_add(all_sum[i], two, crr))
        two_all.append(lo)
    two_all.append(crr)
    res = [None] * len(three_valid)
    prev = (zero, zero)
    for i in range(len(three_valid) - 1, -1, -1):
        is_equal = b.is_equal(three_valid[i], two_all[i])
        s = upper_bytes_match(b.sub(three_valid[i], two_all[i]))
        is_negative = b.is_equal(s, seven)
        b.connect(s, b.select(is_negative, seven, zero))
        if i == len(three_valid) - 1:
            res[i] = b.select(is_negative, two_all[i], three_valid[i])
            prev = (is_equal, is_negative)
        else:
            q = b.is_equal(prev[0], prev[1])
            prev = (b.select(q, prev[0], is_equal), b.select(q, prev[1], is_negative))
            res[i] = b.select(prev[1], two_all[i], three_valid[i])
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

    def prove_valid_keys_stakes_in_validators_list(self, valid_keys, valid_keys_hash, validators):
        """-> (common, verifier_only, proof) of the aggregated proof (keys_stakes.rs:244-265); public inputs = valid_keys bytes
        then the 17 bytes of the valid stake sum.  Raises AssertionError when the stake condition or the hash does not hold."""
        from .plonky2 import HASH_GL
        valid_keys = bytes(valid_keys)
        data, vt, kt = keys_stakes_circuit(valid_keys, [len(v) for v in validators])
        pw = {t: x for ts, v in zip(vt, validators) for t, x in zip(ts, bytes(v))}
        pw.update(zip(kt, valid_keys))
        data.witness_program(list(pw))
        wires, pis = data.generate_witness_native([pw])
        prover = data.prover(self.ctx, HASH_GL)
        try:
            ks = (data.common_data(), prover.verifier_data(), prover.prove(wires[0], [int(x) for x in pis[0]]))
        finally:
            prover.close()
        keys = bytes(int(x) & 0xFF for x in ks[2]["public_inputs"][:len(valid_keys)])
        (hc, hv), hp = self.sha.sha256_proof_u32(keys, valid_keys_hash)
        rc, proof = self.recursion.recursive_proof(ks, (hc, hv, hp), ks[2]["public_inputs"])
        return rc.common, rc.verifier_only, proof

    def close(self):
        self.sha.close()
        self.recursion.close()
