#!/usr/bin/env python3
"""Circuit-INSTANCE parity of the final wrap circuit against the reference's golden verifier data, column by column.

The reference's last recursion (bin/prove_block.rs:279-287 -> prove_crypto/recursion.rs:36-94) is one fixed circuit: its three golden
final proofs share one verifier_data.json.  tests/golden/plonky2_wrap_instance_points.json (made by make_wrap_instance_fixture.py)
holds 87 points per constant / selector / sigma polynomial of THAT circuit.  This tool builds this repo's wrap circuit on the CPU
(host mirror: zklc_amd/plonky2/recursion.py over the Block_i circuit's common data), evaluates its 5 + 80 polynomials at those
points (barycentric formula over the 2^12 subgroup) and reports which columns agree.  Selector columns depend only on the row ->
gate map, the two constant columns on the gate constants per row, sigma column c on the copy classes through wire column c.

    python tools/wrap_instance.py [--inner tests/golden/block_i_common_2p13.json]      # seconds
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = 2**64 - 2**32 + 1
W = 7


def e_mul(a, b):
    return ((a[0] * b[0] + W * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def e_inv(a):
    n = (a[0] * a[0] - W * a[1] * a[1]) % P
    ni = pow(n, P - 2, P)
    return (a[0] * ni % P, (P - a[1]) * ni % P)


def e_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = e_mul(r, a)
        a = e_mul(a, a)
        e >>= 1
    return r


def bary_weights(x, degree_bits):
    """w_r with P(x) = sum_r s_r w_r for the polynomial of degree < n taking the value s_r at w^r (x = (a, b) in the extension)"""
    from oracle import goldilocks as gl
    n = 1 << degree_bits
    om = gl.root_of_unity(degree_bits)
    zh = e_pow(x, n)
    c = e_mul(((zh[0] - 1) % P, zh[1]), (pow(n, P - 2, P), 0))
    out, cur = [], 1
    for _ in range(n):
        d = e_inv(((x[0] - cur) % P, x[1]))
        out.append(e_mul(c, (d[0] * cur % P, d[1] * cur % P)))
        cur = cur * om % P
    return out


def column_values(cols, weights):
    """cols: uint64 array [k, n] (values on the subgroup, natural order) -> k extension-field values"""
    w0 = [w[0] for w in weights]
    w1 = [w[1] for w in weights]
    out = []
    for col in cols:
        s = [int(v) for v in col]
        out.append((sum(a * b for a, b in zip(s, w0)) % P, sum(a * b for a, b in zip(s, w1)) % P))
    return out


def match_columns(constants, sigmas, fixture, max_points=None):
    """-> (bool per column (5 constants then 80 sigmas): agrees at every checked point, number of points checked)"""
    import numpy as np
    cols = np.concatenate([np.asarray(constants, dtype=np.uint64), np.asarray(sigmas, dtype=np.uint64)])
    assert cols.shape == (fixture["num_constants"] + fixture["num_sigmas"], 1 << fixture["degree_bits"]), cols.shape
    pts = fixture["points"]
    # the three zeta points first (extension field: a wrong column agrees by chance with probability 2^-128), then the query points
    pts = [p for p in pts if p["kind"] == "zeta"] + [p for p in pts if p["kind"] != "zeta"]
    pts = pts[:max_points] if max_points else pts
    ok = [True] * len(cols)
    for p in pts:
        got = column_values(cols, bary_weights(tuple(p["x"]), fixture["degree_bits"]))
        for k, (g, w) in enumerate(zip(got, p["values"])):
            ok[k] = ok[k] and g == tuple(w)
    return ok, len(pts)


def build_wrap(inner_common, num_public_inputs=97):
    from zklc_amd.plonky2.recursion import recursive_circuit
    data, _ = recursive_circuit([inner_common], num_public_inputs)
    return data


def report(data, fixture, max_points=4):
    ok, npts = match_columns(data.constants, data.sigmas, fixture, max_points)
    nsel = len(data.groups)
    nc = data.num_constants
    return {"selectors_matched": sum(ok[:nsel]), "selectors": nsel, "gate_constants_matched": sum(ok[nsel:nc]), "gate_constants": nc - nsel,
            "sigmas_matched": sum(ok[nc:]), "sigmas": len(ok) - nc, "points_checked": npts, "columns": ok}


def main():
    inner = os.path.join(ROOT, "tests", "golden", "block_i_common_2p13.json")
    if "--inner" in sys.argv:
        inner = sys.argv[sys.argv.index("--inner") + 1]
    fixture = json.load(open(os.path.join(ROOT, "tests", "golden", "plonky2_wrap_instance_points.json")))
    data = build_wrap(json.load(open(inner)))
    rows = {}
    for g, _ in data.builder.rows:
        rows[g.id().split(" {")[0].split("(")[0]] = rows.get(g.id().split(" {")[0].split("(")[0], 0) + 1
    print("wrap circuit: 2^%d rows, %d used; gate rows: %s" % (data.degree_bits, sum(v for k, v in rows.items() if k != "NoopGate"), rows))
    r = report(data, fixture)
    print("columns equal to the reference's wrap circuit at %d points: selectors %d/%d, gate constants %d/%d, sigmas %d/%d" % (
        r["points_checked"], r["selectors_matched"], r["selectors"], r["gate_constants_matched"], r["gate_constants"],
        r["sigmas_matched"], r["sigmas"]))
    return 0 if all(r["columns"]) else 1


if __name__ == "__main__":
    sys.exit(main())
