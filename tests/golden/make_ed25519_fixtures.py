#!/usr/bin/env python3
"""Generate tests/golden/ed25519_*.json from the reference's own fixtures.

Run in the BUILD container only (needs /root/reference):
    python tests/golden/make_ed25519_fixtures.py
Sources (reference root = /root/reference):
  data/validators_ordered.json + data/block_header.json + data/next_block_header.json   (C2, 100 slots)
  data/validators_ordered_small.json + data/block_header_small.json
      + data/next_block_header_small{,_skip}.json                                      (C1, 3 slots)
  data/epochs/<epoch>/{block-i,block-(i+1),validators}.json                             (extra mainnet sets)
  crypto/plonky2_ed25519/src/main.rs:68-80                                              (fixed triple)
Each output entry keeps the raw protocol bytes the reference slices
(signatures.rs:72-86): borsh Option<Signature> = [1][0][sig64] | [0] and the
validator tail [key_type][pk32][stake u128 LE].  `expect` is what the
reference's fixtures pin: every PRESENT approval verifies (the reference
panics otherwise, signatures.rs:119-121).
"""
import json, os, sys

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
B58 = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz"


def b58decode(s):
    n = 0
    for c in s:
        n = n * 58 + B58.index(c)
    pad = len(s) - len(s.lstrip("1"))
    b = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return b"\x00" * pad + b


def load(path):
    j = json.load(open(os.path.join(REF, path)))
    return j["result"] if "result" in j else j


def header(path):
    r = load(path)
    return r["header"] if "header" in r else r


def make_set(name, validators_path, cur_path, next_path):
    vals = load(validators_path)
    cur, nxt = header(cur_path), header(next_path)
    ch, nh = cur["height"], nxt["height"]
    prev = b58decode(nxt["prev_hash"])
    assert len(prev) == 32
    msg = (b"\x00" + prev if ch + 1 == nh else b"\x01" + ch.to_bytes(8, "little")) + nh.to_bytes(8, "little")
    entries = []
    assert len(vals) == len(nxt["approvals"]), (len(vals), len(nxt["approvals"]))
    for v, a in zip(vals, nxt["approvals"]):
        kt, pk = v["public_key"].split(":")
        assert kt == "ed25519"
        pk = b58decode(pk)
        assert len(pk) == 32
        stake = int(v["stake"])
        tail = b"\x00" + pk + stake.to_bytes(16, "little")
        if a is None:
            approval = b"\x00"
        else:
            st, sg = a.split(":")
            assert st == "ed25519"
            sg = b58decode(sg)
            assert len(sg) == 64
            approval = b"\x01\x00" + sg
        entries.append({"account_id": v["account_id"], "validator_tail": tail.hex(),
                        "approval": approval.hex()})
    out = {"source": [validators_path, cur_path, next_path],
           "current_height": ch, "next_height": nh, "next_prev_hash": prev.hex(),
           "msg": msg.hex(), "entries": entries,
           "expect_valid": sum(1 for e in entries if len(e["approval"]) == 132)}
    json.dump(out, open(os.path.join(OUT, f"ed25519_{name}.json"), "w"), indent=0)
    print(name, len(entries), out["expect_valid"], len(msg))


if __name__ == "__main__":
    make_set("near_c2_100", "data/validators_ordered.json", "data/block_header.json", "data/next_block_header.json")
    make_set("near_c1_small", "data/validators_ordered_small.json", "data/block_header_small.json",
             "data/next_block_header_small.json")
    make_set("near_c1_small_skip", "data/validators_ordered_small.json", "data/block_header_small.json",
             "data/next_block_header_small_skip.json")
    ep = "data/epochs/CRTZ7cQd77rvfS57Y7M36P1vLhran9HyQFEpTLxHRf9t"
    make_set("near_epoch_block01", f"{ep}/validators.json", f"{ep}/block-0.json", f"{ep}/block-1.json")
    make_set("near_epoch_random01", f"{ep}/validators.json", f"{ep}/random-0.json", f"{ep}/random-1.json")
    triple = {"source": "crypto/plonky2_ed25519/src/main.rs:68-80", "msg": b"test message".hex(),
              "pk": bytes([59, 106, 39, 188, 206, 182, 164, 45, 98, 163, 168, 208, 42, 111, 13, 115, 101, 50, 21, 119,
                           29, 226, 67, 166, 58, 192, 72, 161, 139, 89, 218, 41]).hex(),
              "sig": bytes([104, 196, 204, 44, 176, 120, 225, 128, 47, 67, 245, 210, 247, 65, 201, 66, 34, 159, 217,
                            32, 175, 224, 14, 12, 31, 231, 83, 160, 214, 122, 250, 68, 250, 203, 33, 143, 184, 13, 247,
                            140, 185, 25, 122, 25, 253, 195, 83, 102, 240, 255, 30, 21, 108, 249, 77, 184, 36, 72, 9,
                            198, 49, 12, 68, 8]).hex(), "expect": True}
    json.dump(triple, open(os.path.join(OUT, "ed25519_fixed_triple.json"), "w"), indent=0)
