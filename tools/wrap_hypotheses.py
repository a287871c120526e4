#!/usr/bin/env python3
"""The hypothesis log of the wrap-circuit instance search: every layout variant of tools/wrap_instance.py
built, counted (rows per gate type) and measured against the reference's fixed wrap circuit with the nearest-codeword decoder of
tools/wrap_decode.py (exact recovery when <= 44 rows of a column, or <= 67 rows of the three selector columns jointly, differ).

    python tools/wrap_hypotheses.py [--shifts N] > profiles/r05_wrap_instance_hypotheses.txt        # ~2 min (+ 5 s per shift)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import wrap_decode as WD  # noqa: E402
import wrap_instance as WI  # noqa: E402


def measure(pts, data, neq, shift=0):
    nsel, nc = len(data.groups), data.num_constants
    cols = np.asarray(data.constants, dtype=np.uint64)
    res = [pts.residuals(np.roll(cols[k], shift), k) for k in range(nc)]
    joint = WD.decode(pts, res[:nsel], WD.radius(neq, nsel))
    singles = [WD.decode(pts, [res[k]], WD.radius(neq, 1)) for k in range(nc)]
    return joint, singles


def main():
    shifts = int(sys.argv[sys.argv.index("--shifts") + 1]) if "--shifts" in sys.argv else 0
    fixture = json.load(open(os.path.join(ROOT, "tests", "golden", "plonky2_wrap_instance_points.json")))
    inner = json.load(open(os.path.join(ROOT, "tests", "golden", "block_i_common_2p13.json")))
    pts = WD.Points(fixture)
    neq = sum(2 if x[1] else 1 for x in pts.x)
    r1, r3 = WD.radius(neq, 1), WD.radius(neq, 3)
    print("reference instance: near_bft_finality/proofs/*/verifier_data.json (one file, three copies); %d evaluation points per column" % len(pts.x))
    print("measure: Hamming distance of a candidate column to the reference's, decodable up to %d rows (one column) / %d rows (the 3 "
          "selector columns jointly, one error locator); beyond that the decoder reports '> radius'" % (r1, r3))
    print()
    base = None
    for name, what in WI.VARIANTS.items():
        t0 = time.time()
        data = WI.build_wrap(inner, variant=name)
        rows = WI.gate_rows(data)
        used = sum(v for k, v in rows.items() if k != "NoopGate")
        print("== %s: %s" % (name, what))
        print("   rows used %d -> degree 2^%d; per gate: %s" % (used, data.degree_bits, dict(sorted(rows.items()))))
        if data.degree_bits != fixture["degree_bits"]:
            print("   REFUTED without decoding: the reference's wrap circuit has 2^%d rows (common_data.json), this layout needs 2^%d"
                  % (fixture["degree_bits"], data.degree_bits))
            continue
        joint, singles = measure(pts, data, neq)
        ok, _ = WI.match_columns(data.constants, data.sigmas, fixture, 2)
        nsel, nc = len(data.groups), data.num_constants
        print("   exact columns (2 zeta points): selectors %d/%d, gate constants %d/%d, sigmas %d/%d" % (
            sum(ok[:nsel]), nsel, sum(ok[nsel:nc]), nc - nsel, sum(ok[nc:]), len(ok) - nc))
        print("   decoder: selectors jointly %s; per column %s   (%.0f s)" % (
            "distance > %d rows" % r3 if joint is None else "RECOVERED: %d rows differ %s" % (len(joint[0]), joint[0][:32]),
            ["> %d" % r1 if s is None else "RECOVERED (%d rows)" % len(s[0]) for s in singles], time.time() - t0))
        if name == "baseline":
            base = data
    if shifts and base is not None:
        print()
        print("== baseline under a cyclic row shift d in [-%d, %d] (a different number of rows in an early phase of the builder moves "
              "everything after it): candidate'[r] = candidate[r - d]" % (shifts, shifts))
        hits = []
        for sh in range(-shifts, shifts + 1):
            joint, singles = measure(pts, base, neq, sh)
            if joint is not None or any(s is not None for s in singles):
                hits.append(sh)
                print("   shift %+d: joint %s, singles %s" % (sh, None if joint is None else len(joint[0]), [None if s is None else len(s[0]) for s in singles]))
        print("   %d shifts tested, %d within the radius%s" % (2 * shifts + 1, len(hits), "" if hits else
              " (every shift: selectors jointly > %d rows, every column > %d rows)" % (r3, r1)))


if __name__ == "__main__":
    main()
