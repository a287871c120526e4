"""The wiring of the finality DAG (zklc_amd/prove_bft.py = near_bft_finality/src/prove_bft/{block_finality,bft}.rs) on the CPU:
the GPU provers are replaced by stand-ins that CHECK the statement they are asked to prove (hash chains with hashlib, equalities,
consecutive heights, the stake condition, bp_hash) and return the public inputs a real proof would carry, so every byte offset the
DAG slices out of public inputs and headers is exercised on the reference's mainnet window.  (The real thing runs in
tests/test_gpu_plonky2.py::test_full_block_proof_on_a_mainnet_window.)"""
import hashlib

import pytest

import zklc_amd  # noqa: F401
from zklc_amd.prove_bft import BlockProver
from conftest import load_golden


class _RC:
    def __init__(self, tag):
        self.common, self.verifier_only = {"circuit": tag}, {"vd": tag}


class FakeRecursion:
    def __init__(self):
        self.calls = 0

    def recursive_proof(self, first, second=None, public_inputs=None, raw=False):
        for inner in (first, second):
            if inner is not None:
                assert len(inner) == 3 and "public_inputs" in inner[2]
        self.calls += 1
        return _RC("rec%d" % self.calls), {"public_inputs": [int(x) for x in (public_inputs or [])]}


def _triple(tag, pis):
    return ({"circuit": tag}, {"vd": tag}, {"public_inputs": [int(x) for x in pis]})


class FakePrims:
    def prove_eq_array(self, a, b):
        assert bytes(a) == bytes(b), "prove_eq_array: arrays differ"
        return _triple("eq", bytes(a))

    def prove_consecutive_heights(self, h1, h2):
        assert int.from_bytes(h1, "little") == int.from_bytes(h2, "little") + 1, "heights are not consecutive"
        return _triple("heights", bytes(h1) + bytes(h2))


class FakeHashes:
    def prove_header_hash(self, header_hash, prev_hash, inner_lite, inner_rest, public_inputs=None):
        assert len(inner_lite) == 208 and len(prev_hash) == 32
        inner = hashlib.sha256(hashlib.sha256(inner_lite).digest() + hashlib.sha256(inner_rest).digest()).digest()
        assert hashlib.sha256(inner + bytes(prev_hash)).digest() == bytes(header_hash), "header hash"
        return _triple("header", public_inputs if public_inputs is not None else
                       [int.from_bytes(header_hash[4 * i:4 * i + 4], "big") for i in range(8)])

    def prove_bp_hash(self, bp_hash, validators):
        data = len(validators).to_bytes(4, "little") + b"".join(validators)
        assert hashlib.sha256(data).digest() == bytes(bp_hash), "bp_hash"
        return _triple("bp", [int.from_bytes(bp_hash[4 * i:4 * i + 4], "big") for i in range(8)])


class FakeKeys:
    def prove_valid_keys_stakes_in_validators_list(self, valid_keys, valid_keys_hash, validators):
        assert hashlib.sha256(valid_keys).digest() == bytes(valid_keys_hash)
        stake = lambda v: int.from_bytes(v[-16:], "little")
        vs = 0
        for i in range(0, len(valid_keys), 33):
            v = validators[valid_keys[i]]
            assert v[-48:-16] == valid_keys[i + 1:i + 33]
            vs += stake(v)
        assert 3 * vs >= 2 * sum(stake(v) for v in validators), "less than two thirds of the stake"
        return _triple("keys", bytes(valid_keys) + vs.to_bytes(17, "little"))


class FakeApprovals:
    def __init__(self):
        self.recursion = FakeRecursion()
        self.msgs = []

    def prove_approvals(self, msg, approvals, validators):
        self.msgs.append(msg)
        vk = b"".join(bytes([pos]) + validators[pos][-48:-16] for pos, a in enumerate(approvals) if len(a) == 66)
        return (_RC("sig"), {"public_inputs": list(hashlib.sha256(vk).digest())}), vk


def _window():
    w = load_golden("block_window_HPi5.json")
    hx = bytes.fromhex
    blocks = []
    for blk in w["blocks"]:
        f = {k: hx(blk[k]) for k in ("hash", "prev_hash", "epoch_id", "last_ds_final_hash", "last_final_hash")}
        f["height"] = blk["height"]
        f["approvals"] = [hx(a) for a in blk["approvals"]]
        blocks.append((f, hx(blk["bytes"])))
    return w, blocks, [hx(v) for v in w["validators"]]


def test_dag_wiring_on_the_mainnet_window():
    w, blocks, validators = _window()
    hx = bytes.fromhex
    ap = FakeApprovals()
    bp = BlockProver(None, parts=(ap, FakeHashes(), FakeKeys(), FakePrims()))
    bi, none = bp.prove_block_bft(hx(w["ep2_last_block"]["bytes"]), hx(w["ep2_last_block"]["hash"]), hx(w["ep1_first_block"]["bytes"]),
                                  hx(w["ep1_first_block"]["hash"]), blocks, validators)
    assert none is None
    assert bi[2]["public_inputs"] == [0] + list(hx(w["blocks"][4]["hash"])) + list(hx(w["ep2_last_block"]["hash"])) + \
        list(hx(w["ep1_first_block"]["hash"]))
    # the approvals were checked against Endorsement(hash(Block_i)) || height(Block_i+1)  (signatures.rs:24-39)
    assert ap.msgs == [b"\x00" + hx(w["blocks"][4]["hash"]) + (w["blocks"][3]["height"]).to_bytes(8, "little")]
    assert bp.counts == {"prove_header_hash": 7, "prove_eq_array": 3, "recursive_proof": 15, "prove_bp_hash": 1,
                         "prove_approvals": 1, "prove_valid_keys_stakes": 1}


def test_dag_rejects_inconsistent_inputs():
    w, blocks, validators = _window()
    hx = bytes.fromhex
    args = lambda: (hx(w["ep2_last_block"]["bytes"]), hx(w["ep2_last_block"]["hash"]), hx(w["ep1_first_block"]["bytes"]),
                    hx(w["ep1_first_block"]["hash"]))
    mk = lambda: BlockProver(None, parts=(FakeApprovals(), FakeHashes(), FakeKeys(), FakePrims()))
    # a wrong epoch block (the epoch_id of Block_i is the hash of the last block of epoch i-2)
    with pytest.raises(AssertionError):
        mk().prove_block_bft(hx(w["ep1_first_block"]["bytes"]), hx(w["ep1_first_block"]["hash"]), hx(w["ep1_first_block"]["bytes"]),
                             hx(w["ep1_first_block"]["hash"]), blocks, validators)
    # a validator list that is not the one committed to by next_bp_hash of Block_0(epoch i-1)
    with pytest.raises(AssertionError, match="bp_hash"):
        mk().prove_block_bft(*args(), blocks, validators[:-1] + [validators[0]])
    # a gap in the heights of Block_i+2..i+4
    gap = [(dict(f), raw) for f, raw in blocks]
    gap[0], gap[1] = gap[1], gap[0]
    with pytest.raises(AssertionError):
        mk().prove_block_bft(*args(), gap, validators)
    # too few approvals for two thirds of the stake
    few = [(dict(f), raw) for f, raw in blocks]
    few[3][0]["approvals"] = [a if i % 3 else b"\x00" for i, a in enumerate(few[3][0]["approvals"])]
    with pytest.raises(AssertionError, match="two thirds"):
        mk().prove_block_bft(*args(), few, validators)


def test_header_hash_chain_wiring():
    """header_bphash.rs:34-111 / sha256.rs:108-172 with checking stand-ins for the SHA-256 and recursion provers: the messages of
    the three outer hashes are the big-endian u32 public inputs of the inner proofs (|| the raw prev_hash bytes)"""
    from zklc_amd.header_bphash import BlockHashProver

    class FakeSha:
        def __init__(self):
            self.msgs = []

        def sha256_proof_u32(self, msg, digest=None):
            d = hashlib.sha256(msg).digest()
            assert digest is None or bytes(digest) == d, "sha256_proof_u32: the given hash is not sha256(msg)"
            self.msgs.append(bytes(msg))
            return ({"circuit": "sha%d" % ((8 * len(msg) + 64 + 512) // 512)}, {"vd": 0}), \
                {"public_inputs": [int.from_bytes(d[4 * i:4 * i + 4], "big") for i in range(8)]}
    w, blocks, _ = _window()
    f, raw = blocks[4]
    sha = FakeSha()
    bh = BlockHashProver(None, sha=sha, recursion=FakeRecursion())
    lite, rest = raw[33:241], raw[241:len(raw) - 65]
    common, vd, proof = bh.prove_header_hash(f["hash"], raw[1:33], lite, rest)
    assert proof["public_inputs"] == [int.from_bytes(f["hash"][4 * i:4 * i + 4], "big") for i in range(8)]
    h_lite, h_rest = hashlib.sha256(lite).digest(), hashlib.sha256(rest).digest()
    assert sha.msgs == [lite, rest, h_lite + h_rest, hashlib.sha256(h_lite + h_rest).digest() + raw[1:33]]
    assert bh.prove_header_hash(f["hash"], raw[1:33], lite, rest, public_inputs=[1, 2])[2]["public_inputs"] == [1, 2]
    with pytest.raises(AssertionError):
        bh.prove_header_hash(f["prev_hash"], raw[1:33], lite, rest)


def test_dag_wiring_epoch_blocks():
    """the 6-block branch of prove_block_bft (bft.rs:334-496, the path of bin/prove_epoch.rs:222-232) on the reference's epoch data:
    Block_0 of epoch CRTZ.. and Block_n-1 of epoch HPi5.., each with its own validator set and epoch-ancestor blocks"""
    w = load_golden("block_window_epoch_CRTZ.json")
    hx = bytes.fromhex
    blocks = []
    for blk in w["blocks"]:
        f = {k: hx(blk[k]) for k in ("hash", "prev_hash", "epoch_id", "last_ds_final_hash", "last_final_hash")}
        f["height"] = blk["height"]
        f["approvals"] = [hx(a) for a in blk["approvals"]]
        blocks.append((f, hx(blk["bytes"])))
    ap = FakeApprovals()
    bp = BlockProver(None, parts=(ap, FakeHashes(), FakeKeys(), FakePrims()))
    b0, bn_1 = bp.prove_block_bft(hx(w["ep2_last_block"]["bytes"]), hx(w["ep2_last_block"]["hash"]), hx(w["ep1_first_block"]["bytes"]),
                                  hx(w["ep1_first_block"]["hash"]), blocks, [hx(v) for v in w["validators"]],
                                  ep3_last_block_bytes=hx(w["ep3_last_block"]["bytes"]), ep3_last_block_hash=hx(w["ep3_last_block"]["hash"]),
                                  validators_n_1=[hx(v) for v in w["validators_n_1"]])
    assert b0[2]["public_inputs"] == [1] + list(hx(w["blocks"][4]["hash"])) + list(hx(w["ep2_last_block"]["hash"])) + \
        list(hx(w["ep1_first_block"]["hash"]))
    assert bn_1[2]["public_inputs"] == [1] + list(hx(w["blocks"][5]["hash"])) + list(hx(w["ep3_last_block"]["hash"])) + \
        list(hx(w["ep2_last_block"]["hash"]))
    assert len(ap.msgs) == 2 and bp.counts["prove_header_hash"] == 9 and bp.counts["prove_valid_keys_stakes"] == 2


def test_pipelined_form_refuses_what_it_would_silently_ignore():
    """`prove_block_bft(pipelined=True)` proves its own header proofs on its own contexts -- a caller that passes
    `header_proofs`, or a BlockProver built over stand-ins (no zklc Context), gets an error instead of a proof that ignored them"""
    w, blocks, validators = _window()
    hx = bytes.fromhex
    args = (hx(w["ep2_last_block"]["bytes"]), hx(w["ep2_last_block"]["hash"]), hx(w["ep1_first_block"]["bytes"]),
            hx(w["ep1_first_block"]["hash"]), blocks, validators)
    bp = BlockProver(None, parts=(FakeApprovals(), FakeHashes(), FakeKeys(), FakePrims()))
    with pytest.raises(ValueError, match="header_proofs"):
        bp.prove_block_bft(*args, header_proofs={"b1": None}, pipelined=True)
    with pytest.raises(ValueError, match="Context"):
        bp.prove_block_bft(*args, pipelined=True)
