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


VARIANTS = {
    "baseline": "the mirror as shipped: plonky2's check order, fused gate evaluators (gate_circuits.CircuitK), operation memo, "
                "constants sorted into ConstantGate rows at build time, PI hash + PublicInputGate at build time",
    "literal": "gate constraints through gate_circuits.LiteralK: one builder operation per field operation, a constant factor as "
               "constant_extension + mul_extension, nothing fused",
    "no-memo": "no cache of identical extension operations (plonky2's arithmetic_results map switched off)",
    "literal+no-memo": "both of the above",
    "const-first-use": "constants handed to the constant generators in first-use order instead of ascending value",
    "pi-gate-first": "public-input hash and PublicInputGate as the FIRST rows (registered before verify_proof) instead of at build time",
}


def build_wrap(inner_common, num_public_inputs=97, variant="baseline"):
    """the wrap circuit (recursion.rs:36-94 over the Block_i circuit's common data) under one of the layout VARIANTS"""
    from zklc_amd.plonky2 import recursion as R
    from zklc_amd.plonky2 import gate_circuits as GC
    from zklc_amd.plonky2 import gates as G
    from zklc_amd.plonky2.builder import Target, standard_recursion_config
    assert variant in VARIANTS, variant

    class B(R.RecursiveCircuitBuilder):
        pass
    b = B(standard_recursion_config())
    if "literal" in variant:
        b.k_adapter = GC.LiteralK
    if "no-memo" in variant:
        class NoMemo(dict):
            def get(self, k, d=None):
                return d

            def __setitem__(self, k, v):
                pass
        b._ext_memo = NoMemo()
    if variant == "const-first-use":
        real_sorted = sorted

        def place(self=b):
            import builtins
            builtins_sorted = builtins.sorted
            try:      # _place_constants sorts the (value, target) items: keep the insertion order of the dict instead
                builtins.sorted = lambda it, *a, **k: list(it)
                R.CircuitBuilder._place_constants(self)
            finally:
                builtins.sorted = builtins_sorted
        b._place_constants = place
        del real_sorted
    if variant == "pi-gate-first":
        pis = b.add_virtual_targets(num_public_inputs)
        for t in pis:
            b.register_public_input(t)
        pi_hash = b.hash_n_to_hash_no_pad(b.public_inputs)
        pi_row = b.add_gate(G.PublicInputGate())
        for i in range(4):
            b.connect(pi_hash[i], Target(pi_row, i))
        keep = list(b.public_inputs)
        pt = R.add_virtual_proof_with_pis(b, inner_common)
        vt = R.add_virtual_verifier_data(b, inner_common)
        R.verify_proof(b, pt, vt, inner_common)
        b.public_inputs = []             # build() would hash them again: the gate is already placed
        b._place_constants()
        while len(b.rows) & (len(b.rows) - 1):
            b.add_gate(G.NoopGate())
        b.public_inputs = keep
        from zklc_amd.plonky2.builder import CircuitData
        return CircuitData(b)
    data, _ = R.recursive_circuit([inner_common], num_public_inputs, builder=b)
    return data


def gate_rows(data):
    rows = {}
    for g, _ in data.builder.rows:
        k = g.id().split(" {")[0].split("(")[0]
        rows[k] = rows.get(k, 0) + 1
    return rows


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
    variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else "baseline"
    data = build_wrap(json.load(open(inner)), variant=variant)
    rows = gate_rows(data)
    print("variant %s: %s" % (variant, VARIANTS[variant]))
    print("wrap circuit: 2^%d rows, %d used; gate rows: %s" % (data.degree_bits, sum(v for k, v in rows.items() if k != "NoopGate"), rows))
    r = report(data, fixture)
    print("columns equal to the reference's wrap circuit at %d points: selectors %d/%d, gate constants %d/%d, sigmas %d/%d" % (
        r["points_checked"], r["selectors_matched"], r["selectors"], r["gate_constants_matched"], r["gate_constants"],
        r["sigmas_matched"], r["sigmas"]))
    return 0 if all(r["columns"]) else 1


if __name__ == "__main__":
    sys.exit(main())
