#!/usr/bin/env python3
"""Groth16 known-answer vector of the reference (BUILD container only): the verifying key constants of
contracts/hardhat/contracts/Verifier.sol:57-83 and the proof / public inputs of contracts/hardhat/test/proof_with_witness.json,
plus the tampered vectors of contracts/hardhat/test/verify.ts:9-27.  Written to tests/golden/groth16_kat.json."""
import json
import os
import re

REF = "/root/reference/contracts/hardhat"
OUT = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(REF, "contracts/Verifier.sol")).read()
consts = {m[1]: int(m[2]) for m in re.finditer(r"uint256 constant (\w+) = (\d+);", src)}
pw = json.load(open(os.path.join(REF, "test/proof_with_witness.json")))
ts = open(os.path.join(REF, "test/verify.ts")).read()
bad_inputs = re.search(r"const incorrectInputs = \[(.*?)\]", ts, re.S)[1]
bad_proof = re.search(r"const incorrectProof = \[(.*?)\]", ts, re.S)[1]
keep = ["ALPHA_X", "ALPHA_Y", "CONSTANT_X", "CONSTANT_Y"] + ["%s_NEG_%s_%d" % (g, c, i) for g in ("BETA", "GAMMA", "DELTA") for c in "XY" for i in (0, 1)] + \
       ["PUB_%d_%s" % (i, c) for i in range(4) for c in "XY"]
out = {"source": ["contracts/hardhat/contracts/Verifier.sol:57-83", "contracts/hardhat/test/proof_with_witness.json",
                  "contracts/hardhat/test/verify.ts:9-27"],
       "vk": {k: str(consts[k]) for k in keep},
       "inputs": pw["inputs"], "proof": pw["proof"],
       "incorrect_inputs": re.findall(r"\d+", bad_inputs), "incorrect_proof": re.findall(r"\d+", bad_proof)}
json.dump(out, open(os.path.join(OUT, "groth16_kat.json"), "w"), indent=1)
print("ok", len(out["vk"]))
