#!/usr/bin/env python3
"""Builds tests/golden/block_window_HPi5.json from the reference's mainnet data set (run in the build container only):

  data/epochs/HPi5yy.../random-{0..4}.json + validators.json   Block_i .. Block_i+4 of epoch i and its 100 block producers
  data/epochs/3JMehu.../block-0.json                            Block_0 of epoch i-1
  data/epochs/89PT9S.../block-last.json                         Block_n-1 of epoch i-2
-- the window near_bft_finality/src/bin/prove_random.rs:57-62 proves (utils.rs:318-400 `set_blocks`).  The reference turns the
JSON views into borsh bytes with near-primitives; here BlockHeaderV4 / ValidatorStake::V1 are serialised by hand, and every
header is CHECKED: sha256(sha256(sha256(inner_lite) || sha256(inner_rest)) || prev_hash) must equal the block hash in the JSON,
and sha256(borsh(validators)) must equal next_bp_hash of Block_0(epoch i-1)."""
import hashlib
import json
import os
import sys

ALPH = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz"
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
E = os.path.join(REF, "data", "epochs")
EPOCH_I, EPOCH_I1, EPOCH_I2 = ("HPi5yyZHZ91t5S4SPAAfEZwGYEqq5i6QjzXoVMi8ksae", "3JMehuv86nBynJ33VBUGAvfd9Ts8EfvytGJ8i8e45XPi",
                               "89PT9SkLXB1FZHvW7EdQHxiSpm5ybuTCvjrGZWWhXMTz")


def b58d(s, size):
    n = 0
    for c in s:
        n = n * 58 + ALPH.index(c)
    return n.to_bytes(size, "big")


def h32(s):
    return b58d(s, 32)


def sig(s):
    assert s.startswith("ed25519:")
    return b"\0" + b58d(s[8:], 64)


def header_bytes(j):
    """borsh(BlockHeader::BlockHeaderV4) and the block hash"""
    lite = (j["height"].to_bytes(8, "little") + h32(j["epoch_id"]) + h32(j["next_epoch_id"]) + h32(j["prev_state_root"])
            + h32(j["outcome_root"]) + int(j["timestamp_nanosec"]).to_bytes(8, "little") + h32(j["next_bp_hash"])
            + h32(j["block_merkle_root"]))
    assert j["validator_proposals"] == [] and j["challenges_result"] == []
    rest = (h32(j["block_body_hash"]) + h32(j["chunk_receipts_root"]) + h32(j["chunk_headers_root"]) + h32(j["chunk_tx_root"])
            + h32(j["challenges_root"]) + h32(j["random_value"]) + (0).to_bytes(4, "little")
            + len(j["chunk_mask"]).to_bytes(4, "little") + bytes(1 if x else 0 for x in j["chunk_mask"])
            + int(j["gas_price"]).to_bytes(16, "little") + int(j["total_supply"]).to_bytes(16, "little") + (0).to_bytes(4, "little")
            + h32(j["last_final_block"]) + h32(j["last_ds_final_block"]) + j["block_ordinal"].to_bytes(8, "little")
            + j["prev_height"].to_bytes(8, "little")
            + (b"\0" if j["epoch_sync_data_hash"] is None else b"\1" + h32(j["epoch_sync_data_hash"]))
            + len(j["approvals"]).to_bytes(4, "little") + b"".join(b"\0" if a is None else b"\1" + sig(a) for a in j["approvals"])
            + j["latest_protocol_version"].to_bytes(4, "little"))
    prev = h32(j["prev_hash"])
    digest = hashlib.sha256(hashlib.sha256(hashlib.sha256(lite).digest() + hashlib.sha256(rest).digest()).digest() + prev).digest()
    assert digest == h32(j["hash"]), "block hash mismatch: the borsh restatement is wrong for this header"
    return bytes([3]) + prev + lite + rest + sig(j["signature"]), digest


def load(path):
    j = json.load(open(path))
    return j.get("header", j)


def block(path):
    j = load(path)
    raw, digest = header_bytes(j)
    return {"hash": digest.hex(), "bytes": raw.hex(), "height": j["height"], "prev_hash": h32(j["prev_hash"]).hex(),
            "epoch_id": h32(j["epoch_id"]).hex(), "next_epoch_id": h32(j["next_epoch_id"]).hex(),
            "bp_hash": h32(j["next_bp_hash"]).hex(), "last_ds_final_hash": h32(j["last_ds_final_block"]).hex(),
            "last_final_hash": h32(j["last_final_block"]).hex(),
            "approvals": [("00" if a is None else (b"\1" + sig(a)).hex()) for a in j["approvals"]]}


def validators_of(epoch):
    out = []
    for e in json.load(open(os.path.join(E, epoch, "validators.json")))["result"]:
        assert e["validator_stake_struct_version"] == "V1" and e["public_key"].startswith("ed25519:")
        acc = e["account_id"].encode()
        out.append(b"\0" + len(acc).to_bytes(4, "little") + acc + b"\0" + b58d(e["public_key"][8:], 32) + int(e["stake"]).to_bytes(16, "little"))
    return out


def bp_hash(validators):
    return hashlib.sha256(len(validators).to_bytes(4, "little") + b"".join(validators)).hexdigest()


def dump(name, out):
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), name)
    json.dump(out, open(dst, "w"), separators=(",", ":"))
    print("wrote", dst, os.path.getsize(dst), "bytes")


def main():
    # a randomly selected block: bin/prove_random.rs:57-62
    validators = validators_of(EPOCH_I)
    ep1 = block(os.path.join(E, EPOCH_I1, "block-0.json"))
    ep2 = block(os.path.join(E, EPOCH_I2, "block-last.json"))
    assert bp_hash(validators) == ep1["bp_hash"]
    blocks = [block(os.path.join(E, EPOCH_I, "random-%d.json" % k)) for k in (4, 3, 2, 1, 0)]      # set_blocks order
    assert all(blocks[k]["prev_hash"] == blocks[k + 1]["hash"] for k in range(4)) and blocks[4]["epoch_id"] == ep2["hash"]
    dump("block_window_HPi5.json",
         {"source": "data/epochs/{%s,%s,%s} of the reference (NEAR mainnet, heights %d..%d)" % (EPOCH_I[:6], EPOCH_I1[:6], EPOCH_I2[:6],
                                                                                           blocks[4]["height"], blocks[0]["height"]),
          "ep1_first_block": ep1, "ep2_last_block": ep2, "blocks": blocks, "validators": [v.hex() for v in validators]})
    # the epoch blocks Block_0(epoch i) and Block_n-1(epoch i-1): bin/prove_epoch.rs:222-232 (epoch i = CRTZ.., i-1 = HPi5.., ...)
    ei, ei1, ei2, ei3 = "CRTZ7cQd77rvfS57Y7M36P1vLhran9HyQFEpTLxHRf9t", EPOCH_I, EPOCH_I1, EPOCH_I2
    vals, vals_n_1 = validators_of(ei), validators_of(ei1)
    e1, e2, e3 = (block(os.path.join(E, ei1, "block-0.json")), block(os.path.join(E, ei2, "block-last.json")),
                  block(os.path.join(E, ei3, "block-last.json")))
    blks = [block(os.path.join(E, ei, "block-%d.json" % k)) for k in (4, 3, 2, 1, 0)] + [block(os.path.join(E, ei1, "block-last.json"))]
    assert bp_hash(vals) == e1["bp_hash"] and all(blks[k]["prev_hash"] == blks[k + 1]["hash"] for k in range(5))
    dump("block_window_epoch_CRTZ.json",
         {"source": "data/epochs/{%s,%s,%s,%s} of the reference (NEAR mainnet, heights %d..%d)" % (ei[:6], ei1[:6], ei2[:6], ei3[:6],
                                                                                              blks[5]["height"], blks[0]["height"]),
          "ep1_first_block": e1, "ep2_last_block": e2, "ep3_last_block": e3, "blocks": blks,
          "validators": [v.hex() for v in vals], "validators_n_1": [v.hex() for v in vals_n_1]})


if __name__ == "__main__":
    main()
