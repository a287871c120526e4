#!/usr/bin/env python3
"""Extract the Poseidon-Goldilocks / Poseidon-BN254 parameters and known-answer
vectors the reference holds, into tests/golden/*.json (BUILD container only).

Sources (reference root /root/reference):
  gnark-plonky2-verifier/poseidon/goldilocks_constants.go:7   ALL_ROUND_CONSTANTS (360)
      :370 MDS_MATRIX_CIRC, :402 MDS_MATRIX_DIAG, :436 FAST_PARTIAL_FIRST_ROUND_CONSTANT,
      :451 FAST_PARTIAL_ROUND_CONSTANTS, :476 FAST_PARTIAL_ROUND_VS, :765 FAST_PARTIAL_ROUND_W_HATS,
      :1054 FAST_PARTIAL_ROUND_INITIAL_MATRIX
  gnark-plonky2-verifier/tests/goldilocks_test.go:47-53          permutation KAT (zero state)
  gnark-plonky2-verifier/tests/public_inputs_hash_test.go:54-55  hash_no_pad KAT
  crypto/plonky2_bn128/src/poseidon_bn128_constants.rs           C, S, M, P (iden3 t=4)
  crypto/plonky2_bn128/src/poseidon_bn128.rs:133-180             4 permutation KATs
These are public parameters (data), not code.  The same JSON feeds the oracle
and tools/gen_constants.py (which emits the device headers).
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def go_var_block(src, name):
    i = src.index("var %s " % name)
    j = src.index("\n}\n", i)
    return src[i:j]


def nums(block):
    out = []
    for m in re.finditer(r"uint64\((0x[0-9a-fA-F]+|\d+)\)|ZERO_VAR", block):
        tok = m.group(0)
        out.append(0 if tok == "ZERO_VAR" else int(m.group(1), 0))
    return out


def main():
    src = open(os.path.join(REF, "gnark-plonky2-verifier/poseidon/goldilocks_constants.go")).read()
    gl = {}
    gl["all_round_constants"] = nums(go_var_block(src, "ALL_ROUND_CONSTANTS"))
    gl["mds_circ"] = nums(go_var_block(src, "MDS_MATRIX_CIRC"))
    gl["mds_diag"] = nums(go_var_block(src, "MDS_MATRIX_DIAG"))
    gl["fast_partial_first_round_constant"] = nums(go_var_block(src, "FAST_PARTIAL_FIRST_ROUND_CONSTANT"))
    gl["fast_partial_round_constants"] = nums(go_var_block(src, "FAST_PARTIAL_ROUND_CONSTANTS"))
    for key, name in [("fast_partial_round_vs", "FAST_PARTIAL_ROUND_VS"), ("fast_partial_round_w_hats", "FAST_PARTIAL_ROUND_W_HATS"),
                      ("fast_partial_round_initial_matrix", "FAST_PARTIAL_ROUND_INITIAL_MATRIX")]:
        flat = nums(go_var_block(src, name))
        assert len(flat) % 11 == 0, (name, len(flat))
        gl[key] = [flat[i:i + 11] for i in range(0, len(flat), 11)]
    assert len(gl["all_round_constants"]) == 360 and len(gl["mds_circ"]) == 12 and len(gl["mds_diag"]) == 12
    assert len(gl["fast_partial_round_vs"]) == 22 and len(gl["fast_partial_round_w_hats"]) == 22
    assert len(gl["fast_partial_round_initial_matrix"]) == 11
    gl["kat_permute_zero"] = [4330397376401421145, 14124799381142128323, 8742572140681234676, 14345658006221440202,
                              15524073338516903644, 5091405722150716653, 15002163819607624508, 2047012902665707362,
                              16106391063450633726, 4680844749859802542, 15019775476387350140, 1698615465718385111]
    gl["kat_hash_no_pad"] = {"in": [0, 1, 3736710860384812976],
                             "out": [8416658900775745054, 12574228347150446423, 9629056739760131473, 3119289788404190010]}
    t = open(os.path.join(REF, "gnark-plonky2-verifier/tests/goldilocks_test.go")).read()
    assert all(str(v) in t for v in gl["kat_permute_zero"])
    t = open(os.path.join(REF, "gnark-plonky2-verifier/tests/public_inputs_hash_test.go")).read()
    assert all(str(v) in t for v in gl["kat_hash_no_pad"]["out"])
    json.dump(gl, open(os.path.join(OUT, "poseidon_goldilocks.json"), "w"))
    print("goldilocks: ok", {k: (len(v) if isinstance(v, list) else "kat") for k, v in gl.items()})

    # ---- Poseidon BN254 (iden3, t = 4)
    rs = open(os.path.join(REF, "crypto/plonky2_bn128/src/poseidon_bn128_constants.rs")).read()
    bn = {}

    def fn_body(name):
        i = rs.index("fn %s()" % name)
        j = rs.find("\nfn ", i + 1)
        return rs[i:j if j > 0 else len(rs)]

    def assigns(body, var, dims):
        out = {}
        pat = r"%s((?:\[\d+\]){%d})\s*=\s*Fr::from_str_vartime\(\s*\"(\d+)\"" % (var, dims)
        for m in re.finditer(pat, body):
            idx = tuple(int(x) for x in re.findall(r"\d+", m.group(1)))
            out[idx] = m.group(2)
        return out

    c = assigns(fn_body("load_c_constants"), "c_constants", 1)
    sc = assigns(fn_body("load_s_constants"), "s_constants", 1)
    mm = assigns(fn_body("load_m_matrix"), "m_matrix", 2)
    pm = assigns(fn_body("load_p_matrix"), "p_matrix", 2)
    bn["C"] = [c.get((i,), "0") for i in range(88)]
    bn["S"] = [sc.get((i,), "0") for i in range(392)]
    bn["M"] = [[mm.get((i, j), "0") for j in range(4)] for i in range(4)]
    bn["P"] = [[pm.get((i, j), "0") for j in range(4)] for i in range(4)]
    assert len(c) >= 87 and len(sc) >= 390 and len(mm) == 16, (len(c), len(sc), len(mm), len(pm))
    t = open(os.path.join(REF, "crypto/plonky2_bn128/src/poseidon_bn128.rs")).read()
    i = t.index("let test_vectors")
    vals = re.findall(r'from_str_vartime\("(\d+)"\)', t[i:t.index("for (mut input", i)])
    mx = "21888242871839275222246405745257275088548364400416034343698204186575808495616"
    assert len(vals) == 4 + 8 + 4 + 8, len(vals)
    bn["kats"] = [{"in": ["0"] * 4, "out": vals[0:4]}, {"in": vals[4:8], "out": vals[8:12]},
                  {"in": [mx] * 4, "out": vals[12:16]}, {"in": vals[16:20], "out": vals[20:24]}]
    json.dump(bn, open(os.path.join(OUT, "poseidon_bn254.json"), "w"))
    print("bn254: ok", len(c), len(sc), len(mm), len(pm))

if __name__ == "__main__":
    main()
