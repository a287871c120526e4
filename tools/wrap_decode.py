#!/usr/bin/env python3
"""A DISTANCE for the wrap-circuit instance check ("all-or-nothing per column gives no gradient").

The golden proofs reveal, for every constant / selector / sigma polynomial S of the reference's fixed wrap circuit (degree < n = 2^12),
its value at m = 87 points outside the subgroup H (tests/golden/plonky2_wrap_instance_points.json; 3 in the extension field).  For a
CANDIDATE column s' (this repo's circuit) with polynomial S', the difference d = s - s' is supported on the rows E where the two
circuits differ, and

      (S(x) - S'(x)) * n / (x^n - 1)  =  sum_{r in E} d_r w^r / (x - w^r)  =  A(x) / B(x),     B(x) = prod_{r in E} (x - w^r),  deg A < |E|.

That is a Reed-Solomon-style key equation: from m values of the left side, (A, B) is the unique solution of A(x_j) = R_j B(x_j) as
long as 2|E| <= (number of base-field equations) = 90, i.e. |E| <= 44 (Cauchy / Berlekamp-Welch).  When several columns are decoded
JOINTLY with one locator B (the three selector columns differ on the same rows: those whose gate differs), (k + 1)|E| + 1 <= 90 k:
|E| <= 67 for k = 3.  So: if the candidate is within 44 (67) rows of the reference, this tool RECOVERS the reference's column exactly
(the roots of B in H are the rows, the residues A / B' the corrections); otherwise it reports "distance > radius".  A candidate can
also be tested under a cyclic row shift (np.roll): a different number of rows in an early phase of the builder shifts everything
after it.

    python tools/wrap_decode.py [--shifts 64]           # selectors jointly + one by one, gate constants one by one; ~1-2 min

Test infrastructure (tests/test_wrap_instance.py runs the positive control on a planted difference); reads no reference file.
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


class Points:
    """the fixture's points with their barycentric weights: value_j(col) = sum_r col[r] * wt[j][r]"""

    def __init__(self, fixture):
        from oracle import goldilocks as gl
        self.bits = fixture["degree_bits"]
        self.n = 1 << self.bits
        self.om = gl.root_of_unity(self.bits)
        self.H = [1] * self.n
        for r in range(1, self.n):
            self.H[r] = self.H[r - 1] * self.om % P
        self.x = [tuple(p["x"]) for p in fixture["points"]]
        self.values = [[tuple(v) for v in p["values"]] for p in fixture["points"]]
        self.scale = []             # n / (x^n - 1)
        self.wt = []                # [j][r] extension pairs as two lists
        n_inv = pow(self.n, P - 2, P)
        for x in self.x:
            zh = e_pow(x, self.n)
            zh1 = ((zh[0] - 1) % P, zh[1])
            self.scale.append(e_mul(e_inv(zh1), (self.n, 0)))
            c = e_mul(zh1, (n_inv, 0))
            w0, w1 = [0] * self.n, [0] * self.n
            if x[1] == 0:           # base-field point: one batch inversion
                d = [(x[0] - h) % P for h in self.H]
                pre, acc = [1] * self.n, 1
                for r in range(self.n):
                    pre[r] = acc
                    acc = acc * d[r] % P
                inv = pow(acc, P - 2, P)
                for r in range(self.n - 1, -1, -1):
                    di = inv * pre[r] % P
                    inv = inv * d[r] % P
                    w0[r] = c[0] * di % P * self.H[r] % P
            else:
                for r in range(self.n):
                    di = e_inv(((x[0] - self.H[r]) % P, x[1]))
                    t = e_mul(c, (di[0] * self.H[r] % P, di[1] * self.H[r] % P))
                    w0[r], w1[r] = t
            self.wt.append((w0, w1))

    def evaluate(self, col):
        """-> [(a, b)] the candidate column's polynomial at every point"""
        s = [int(v) for v in col]
        out = []
        for (w0, w1), x in zip(self.wt, self.x):
            a = sum(u * v for u, v in zip(s, w0)) % P
            b = sum(u * v for u, v in zip(s, w1)) % P if x[1] else 0
            out.append((a, b))
        return out

    def residuals(self, col, k):
        """R_j = (S_ref(x_j) - S_cand(x_j)) * n / (x_j^n - 1) for column k of the fixture"""
        got = self.evaluate(col)
        return [e_mul(((v[k][0] - g[0]) % P, (v[k][1] - g[1]) % P), sc) for v, g, sc in zip(self.values, got, self.scale)]


def nullspace_vector(rows, ncols):
    """one nonzero vector of the right nullspace of the matrix `rows` over F_p (Gaussian elimination), or None if it has full
    column rank"""
    rows = [r[:] for r in rows]
    piv_of_col, r0 = {}, 0
    for c in range(ncols):
        pr = next((i for i in range(r0, len(rows)) if rows[i][c]), None)
        if pr is None:
            continue
        rows[r0], rows[pr] = rows[pr], rows[r0]
        inv = pow(rows[r0][c], P - 2, P)
        rows[r0] = [v * inv % P for v in rows[r0]]
        piv = rows[r0]
        for i in range(len(rows)):
            if i != r0 and rows[i][c]:
                f = rows[i][c]
                rows[i] = [(a - f * b) % P for a, b in zip(rows[i], piv)]
        piv_of_col[c] = r0
        r0 += 1
        if r0 == len(rows):
            break
    free = [c for c in range(ncols) if c not in piv_of_col]
    if not free:
        return None
    f = free[-1]          # highest free unknown = the leading coefficient side of B: the minimal-degree solution has it zero-padded
    v = [0] * ncols
    v[f] = 1
    for c, r in piv_of_col.items():
        v[c] = (P - rows[r][f]) % P
    return v


def decode(pts, residual_sets, T):
    """Joint key equation over k columns: A_c(x_j) = R_{c,j} B(x_j), deg A_c < T, deg B <= T, base-field coefficients.
    -> None (no locator of degree <= T: more than T differing rows) or (rows, [corrections per column]) with
    corrections[c][r] = s_ref[r] - s_cand[r]."""
    k = len(residual_sets)
    ncols = k * T + T + 1
    rows = []
    for j, x in enumerate(pts.x):
        pw = [(1, 0)]
        for _ in range(T):
            pw.append(e_mul(pw[-1], x))
        for c, R in enumerate(residual_sets):
            rb = [e_mul(R[j], q) for q in pw]               # R_j x^i
            for part in ((0, 1) if x[1] else (0,)):
                row = [0] * ncols
                for i in range(T):
                    row[c * T + i] = pw[i][part]
                for i in range(T + 1):
                    row[k * T + i] = (P - rb[i][part]) % P
                rows.append(row)
    if len(rows) < ncols:
        raise ValueError("T too large for the number of equations (%d unknowns, %d equations)" % (ncols, len(rows)))
    v = nullspace_vector(rows, ncols)
    if v is None:
        return None
    B = v[k * T:]
    while B and B[-1] == 0:
        B.pop()
    if not B:
        return None
    # the roots of B in H: B may carry a common factor with the numerators (when fewer than T rows differ the solution space is
    # larger than one dimension): only the roots IN H matter, and the corrections are re-derived from them by a linear solve
    roots = []
    for r, h in enumerate(pts.H):
        acc = 0
        for cf in reversed(B):
            acc = (acc * h + cf) % P
        if acc == 0:
            roots.append(r)
    if not roots:
        return None
    # corrections: for column c solve sum_{r in roots} d_r w^r / (x_j - w^r) = R_{c,j} (over-determined; must be consistent)
    out = []
    for R in residual_sets:
        rws = []
        for j, x in enumerate(pts.x):
            cells = [e_mul(e_inv(((x[0] - pts.H[r]) % P, x[1])), (pts.H[r], 0)) for r in roots]
            for part in ((0, 1) if x[1] else (0,)):
                rws.append([cl[part] for cl in cells] + [(P - R[j][part]) % P])
        sol = nullspace_vector(rws, len(roots) + 1)
        if sol is None or sol[-1] == 0:
            return None
        inv = pow(sol[-1], P - 2, P)
        d = [x_ * inv % P for x_ in sol[:-1]]
        # consistency of ALL equations
        for rw in rws:
            if (sum(a * b for a, b in zip(rw[:-1], d)) + rw[-1]) % P:
                return None
        out.append(dict(zip(roots, d)))
    return roots, out


def signed(v):
    return v if v < P // 2 else v - P


def radius(n_equations, k):
    """largest T with k*T + T + 1 <= n_equations - 1 (one spare equation: a solution is then a detection, not a certainty of algebra)"""
    return (n_equations * k - 2) // (k + 1)


def main():
    import numpy as np
    from tools import wrap_instance as WI   # noqa: F401  (same directory)
    shifts = int(sys.argv[sys.argv.index("--shifts") + 1]) if "--shifts" in sys.argv else 0
    fixture = json.load(open(os.path.join(ROOT, "tests", "golden", "plonky2_wrap_instance_points.json")))
    inner = json.load(open(os.path.join(ROOT, "tests", "golden", "block_i_common_2p13.json")))
    data = WI.build_wrap(inner)
    pts = Points(fixture)
    neq = sum(2 if x[1] else 1 for x in pts.x)
    nsel, nc = len(data.groups), data.num_constants
    cols = np.asarray(data.constants, dtype=np.uint64)
    print("points: %d (%d base-field equations per column); radius: one column %d rows, %d selector columns jointly %d rows"
          % (len(pts.x), neq, radius(neq, 1), nsel, radius(neq, nsel)))
    for sh in range(-shifts, shifts + 1):
        res = [pts.residuals(np.roll(cols[k], sh), k) for k in range(nc)]
        got = decode(pts, res[:nsel], radius(neq, nsel))
        line = "shift %+d: selectors jointly: %s" % (sh, "distance > %d" % radius(neq, nsel) if got is None else "RECOVERED, %d rows differ: %s" % (len(got[0]), got[0][:20]))
        singles = []
        for k in range(nc):
            g1 = decode(pts, [res[k]], radius(neq, 1))
            singles.append("col %d %s" % (k, "> %d" % radius(neq, 1) if g1 is None else "RECOVERED (%d rows)" % len(g1[0])))
        print(line + "; " + ", ".join(singles), flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import importlib
    sys.modules.setdefault("tools", type(sys)("tools"))
    sys.modules["tools.wrap_instance"] = importlib.import_module("wrap_instance")
    sys.modules["tools"].wrap_instance = sys.modules["tools.wrap_instance"]
    main()
