"""Circuit-INSTANCE parity of the final wrap circuit with the reference's (judge's finding of round 3, made a committed check).

The reference's last recursion (bin/prove_block.rs:279-287 -> prove_crypto/recursion.rs:36-94) is ONE fixed circuit: its three golden
final proofs (near_bft_finality/proofs/{random/CGZP..,epoch/4RjX..,epoch/CbAH..}) share one verifier_data.json.  Those proofs also
reveal 87 evaluations of each of the circuit's 5 constant / selector and 80 sigma polynomials (tests/golden/
plonky2_wrap_instance_points.json, made by make_wrap_instance_fixture.py) -- so a candidate instance can be compared COLUMN BY COLUMN.

MEASURED STATE: this repo's wrap circuit (host mirror, zklc_amd/plonky2/recursion.py) has the reference's common_data (gate set,
selector groups, degree 2^12, FRI shape, 97 public inputs -- asserted below and in tests/test_gpu_plonky2.py) but is NOT the reference's
instance: 0 of 3 selector, 0 of 2 gate-constant and 0 of 80 sigma columns agree, hence constants_sigmas_cap and circuit_digest differ.
The row -> gate layout is decided by the exact order in which plonky2's Rust `verify_proof` gadgets allocate gate rows and operation
slots (plonky2-near@2244a9d, not vendored, no Rust toolchain here); the mirror follows the Go verifier's order of checks instead.
Consequence: proofs of circuits BUILT by this Python mirror verify against the mirror's own verifier data and
can never be byte-equal to the Rust prover's; the drop-in for the reference is a Rust-built circuit handed to
zklc_plonky2_circuit_create, whose layout the HIP prover takes as given.  The instance check is a STRICT expected failure: the day
the mirror reproduces the reference's layout it turns into a failure that asks for this text to be removed."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden

sys_path_tools = os.path.join(ROOT, "tools")


def _wi():
    import importlib.util
    spec = importlib.util.spec_from_file_location("wrap_instance", os.path.join(sys_path_tools, "wrap_instance.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def wrap():
    wi = _wi()
    return wi, wi.build_wrap(load_golden("block_i_common_2p13.json")), load_golden("plonky2_wrap_instance_points.json")


def test_fixture_is_the_shared_verifier_data_of_the_golden_proofs(wrap):
    _, _, fx = wrap
    assert fx["verifier_data"] == load_golden("plonky2_near_random_CGZP.json")["verifier_data"]
    assert len(fx["points"]) == 3 * 29 and sum(1 for p in fx["points"] if p["kind"] == "zeta") == 3
    assert all(len(p["values"]) == 85 for p in fx["points"])


def test_wrap_common_data_equals_the_reference(wrap):
    _, data, _ = wrap
    golden = load_golden("plonky2_near_random_CGZP.json")["common_data"]
    common = data.common_data()
    assert {k: common[k] for k in golden} == golden
    assert data.degree_bits == 12 and data.num_public_inputs == 97 and data.num_constants == 5 and len(data.groups) == 3


def test_column_check_accepts_the_circuit_it_was_made_from(wrap):
    """positive control of the diagnostic: points computed from THIS circuit's columns by an independent route (coefficients by an
    inverse NTT, Horner evaluation in the extension field) are matched by every column; one changed cell is caught in its column"""
    wi, data, fx = wrap
    from oracle import goldilocks as gl
    cols = np.concatenate([data.constants, data.sigmas])
    xs = [(123456789, 987654321), (gl.GENERATOR * pow(gl.root_of_unity(15), 777, gl.P) % gl.P, 0)]
    pts = []
    for x in xs:
        vals = []
        for col in cols:
            coeffs = gl.ntt([int(v) for v in col], inverse=True)
            acc = (0, 0)
            for c in reversed(coeffs):
                acc = wi.e_mul(acc, x)
                acc = ((acc[0] + c) % gl.P, acc[1])
            vals.append(list(acc))
        pts.append({"kind": "zeta", "x": list(x), "values": vals})
    own = {"num_constants": 5, "num_sigmas": 80, "degree_bits": 12, "points": pts}
    ok, n = wi.match_columns(data.constants, data.sigmas, own)
    assert n == 2 and all(ok)
    sig = data.sigmas.copy()
    sig[17, 1000] ^= np.uint64(1)
    ok, _ = wi.match_columns(data.constants, sig, own)
    assert [k for k, v in enumerate(ok) if not v] == [5 + 17]


def test_matched_columns_today(wrap):
    """the countable form of the finding: how many columns of the mirror's wrap circuit are the reference's (0 / 85 today)"""
    wi, data, fx = wrap
    r = wi.report(data, fx, max_points=3)
    print("wrap instance vs reference: selectors %d/%d, gate constants %d/%d, sigmas %d/%d" % (
        r["selectors_matched"], r["selectors"], r["gate_constants_matched"], r["gate_constants"], r["sigmas_matched"], r["sigmas"]))
    assert (r["selectors"], r["gate_constants"], r["sigmas"]) == (3, 2, 80)


@pytest.mark.xfail(strict=True, reason="measured: the Python mirror's wrap circuit is not the reference's circuit instance (0/85 columns); "
                                       "byte parity with the Rust prover is out of reach for mirror-built circuits -- see the module docstring")
def test_wrap_circuit_is_the_reference_instance(wrap):
    wi, data, fx = wrap
    ok, _ = wi.match_columns(data.constants, data.sigmas, fx, max_points=3)
    assert all(ok)


@pytest.mark.gpu
@pytest.mark.xfail(strict=True, reason="same finding through the GPU preprocessing: constants_sigmas_cap / circuit_digest of the mirror's wrap "
                                       "circuit differ from the reference's verifier_data.json")
def test_gpu_verifier_data_equals_golden(zctx):
    from zklc_amd.plonky2 import HASH_BN128
    from zklc_amd.plonky2.recursion import RecursionProver
    rp = RecursionProver(zctx, HASH_BN128)
    try:
        rc = rp.circuit_for([load_golden("block_i_common_2p13.json")], 97)
        mine = json.loads(json.dumps(rc.verifier_only))
    finally:
        rp.close()
    assert mine == load_golden("plonky2_wrap_instance_points.json")["verifier_data"]


def test_nearest_codeword_decoder_recovers_a_planted_difference():
    """tools/wrap_decode.py (the DISTANCE of round 5's hypothesis log, profiles/r05_wrap_instance_hypotheses.txt): columns that
    differ from the 'reference' in 60 common rows are corrected exactly by the joint decoder (radius 67), a single column with 40
    differing rows by the one-column decoder (radius 44), and 60 rows are beyond the one-column radius"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("wrap_decode", os.path.join(sys_path_tools, "wrap_decode.py"))
    WD = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(WD)
    fx = load_golden("plonky2_wrap_instance_points.json")
    pts = WD.Points(fx)
    neq = sum(2 if x[1] else 1 for x in pts.x)
    assert (neq, WD.radius(neq, 1), WD.radius(neq, 3)) == (90, 44, 67)
    rng = np.random.default_rng(1)
    n = 1 << fx["degree_bits"]
    cand = rng.integers(0, 6, size=(3, n)).astype(np.uint64)
    ref = cand.copy()
    rows = rng.choice(n, 60, replace=False)
    for c in range(3):
        ref[c, rows] = (ref[c, rows] + rng.integers(1, 5, size=60).astype(np.uint64)) % np.uint64(7)
    ref[0, rows[:5]] = 2**32 - 1                      # UNUSED_SELECTOR-sized jumps as well
    vals = [pts.evaluate(ref[c]) for c in range(3)]
    for j in range(len(pts.x)):
        pts.values[j] = [vals[c][j] for c in range(3)]
    res = [pts.residuals(cand[c], c) for c in range(3)]
    got = WD.decode(pts, res, WD.radius(neq, 3))
    assert got is not None and sorted(got[0]) == sorted(rows.tolist())
    assert all(WD.signed(got[1][c][r]) == int(ref[c, r]) - int(cand[c, r]) for c in range(3) for r in got[0])
    assert WD.decode(pts, [res[0]], WD.radius(neq, 1)) is None            # 60 rows: beyond one column's radius
    ref2 = cand[1].copy()
    ref2[rows[:40]] += np.uint64(3)
    v2 = pts.evaluate(ref2)
    for j in range(len(pts.x)):
        pts.values[j] = [v2[j]]
    g1 = WD.decode(pts, [pts.residuals(cand[1], 0)], WD.radius(neq, 1))
    assert g1 is not None and sorted(g1[0]) == sorted(rows[:40].tolist())


def test_layout_variants_and_the_row_budget(wrap):
    """the variants of the hypothesis log build; the unfused evaluators overflow the reference's 2^12 rows (refuted by the row budget)"""
    wi, data, fx = wrap
    inner = load_golden("block_i_common_2p13.json")
    assert set(wi.VARIANTS) >= {"baseline", "literal", "no-memo", "const-first-use", "pi-gate-first"}
    lit = wi.build_wrap(inner, variant="literal")
    assert lit.degree_bits == 13 and data.degree_bits == 12
    assert wi.gate_rows(lit)["PoseidonGate"] == wi.gate_rows(data)["PoseidonGate"] == 2972
    pif = wi.build_wrap(inner, variant="pi-gate-first")
    assert pif.degree_bits == 12 and wi.gate_rows(pif) == wi.gate_rows(data)
