#!/usr/bin/env python3
"""Trim the reference's golden plonky2 proofs into small fixtures (BUILD container only).

Sources (reference root /root/reference):
  near_bft_finality/proofs/random/CGZPhFRkL3NvmGaXWBc6N7qJD519EUe6vyNpaEyDe2Ev/{proof,verifier_data,common_data}.json
  gnark-plonky2-verifier/testdata/test_circuit/{proof_with_public_inputs,verifier_only_circuit_data,common_circuit_data}.json
Everything the Fiat-Shamir transcript absorbs is kept verbatim; only the list of FRI query-round
proofs (28 x ~20 KB) is cut to the first KEEP rounds -- the transcript does not absorb them, so
the derived query indices are unchanged and the kept rounds still verify.  Of common_data only
the fields the verifier oracle reads are kept.
"""
import json
import os

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
KEEP = 4


def trim(name, pj, vj, cj):
    p = json.load(open(os.path.join(REF, pj)))
    v = json.load(open(os.path.join(REF, vj)))
    c = json.load(open(os.path.join(REF, cj)))
    p["proof"]["opening_proof"]["query_round_proofs"] = p["proof"]["opening_proof"]["query_round_proofs"][:KEEP]
    common = {k: c[k] for k in ["config", "fri_params", "num_constants", "num_partial_products", "quotient_degree_factor",
                                "num_public_inputs", "num_gate_constraints", "gates", "k_is", "selectors_info"] if k in c}
    out = {"source": [pj, vj, cj], "kept_query_rounds": KEEP, "proof": p, "verifier_data": v, "common_data": common}
    path = os.path.join(OUT, "plonky2_%s.json" % name)
    json.dump(out, open(path, "w"), separators=(",", ":"))
    print(name, os.path.getsize(path))


def copy_bin():
    """proof.bin of the same golden proof (ProofWithPublicInputs::to_bytes, prove_block.rs:320-458): pins the binary format"""
    import shutil
    d = "near_bft_finality/proofs/random/CGZPhFRkL3NvmGaXWBc6N7qJD519EUe6vyNpaEyDe2Ev/"
    shutil.copy(os.path.join(REF, d, "proof.bin"), os.path.join(OUT, "plonky2_near_random_CGZP_proof.bin"))


if __name__ == "__main__":
    copy_bin()
    d = "near_bft_finality/proofs/random/CGZPhFRkL3NvmGaXWBc6N7qJD519EUe6vyNpaEyDe2Ev/"
    trim("near_random_CGZP", d + "proof.json", d + "verifier_data.json", d + "common_data.json")
    t = "gnark-plonky2-verifier/testdata/test_circuit/"
    trim("gnark_test_circuit", t + "proof_with_public_inputs.json", t + "verifier_only_circuit_data.json", t + "common_circuit_data.json")
