#!/usr/bin/env python3
"""Fixtures for tests/test_formats.py (BUILD container only; the reference root is /root/reference):
  * the three golden verifier_data.bin files of near_bft_finality/proofs (VerifierCircuitData::to_bytes, written by
    bin/prove_block.rs:320-458) with the common_data.json / verifier_data.json next to them -- they pin the byte layout of
    zklc_amd.formats.{verifier_only,common_data,verifier_data}_to_bytes;
  * the second Groth16 proof of the contracts' tests (16-input variant) is not needed: the 4-input KAT is in groth16_kat.json.
"""
import json
import os

REF = "/root/reference/near_bft_finality/proofs"
OUT = os.path.dirname(os.path.abspath(__file__))
out = {"source": [], "cases": []}
for d in ("random/CGZPhFRkL3NvmGaXWBc6N7qJD519EUe6vyNpaEyDe2Ev", "epoch/CbAHBGJ8VQot2m6KhH9PLasMgcDtkPJBfp9bjAEMJ8UK",
          "epoch/4RjXBrNcu39wutFTuFpnRHgNqgHxLMcGBKNEQdtkSBhy"):
    p = os.path.join(REF, d)
    out["source"].append("near_bft_finality/proofs/%s/{verifier_data.bin,verifier_data.json,common_data.json}" % d)
    out["cases"].append({"name": d, "verifier_data_bin": open(os.path.join(p, "verifier_data.bin"), "rb").read().hex(),
                         "verifier_only": json.load(open(os.path.join(p, "verifier_data.json"))),
                         "common_data": json.load(open(os.path.join(p, "common_data.json")))})
json.dump(out, open(os.path.join(OUT, "plonky2_verifier_data_bins.json"), "w"), separators=(",", ":"))
print(os.path.getsize(os.path.join(OUT, "plonky2_verifier_data_bins.json")))
