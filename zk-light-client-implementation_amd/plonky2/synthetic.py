"""Synthetic circuits of the reference's shapes, with a satisfying witness, built directly as arrays.

The reference's two prover workloads on the signature-aggregation path are
  * the per-signature Ed25519 circuit: 2^17 rows, `wide_ecc_config` (234 wires, 80 routed), dominated by the u32 gates of
    crypto/plonky2_u32 (nonnative arithmetic, SHA-512) -- near_bft_finality/src/prove_crypto/ed25519.rs:26-39
  * the recursion circuit: 2^12 rows, `standard_recursion_config` (135 wires), the 13 gate types listed in
    near_bft_finality/proofs/*/common_data.json -- prove_crypto/recursion.rs:16-97
Their exact gate placement comes from the Rust circuit builders (not rebuilt yet), so benchmarks and
parity tests use circuits with the same dimensions and the same gate TYPES in a stated mix: every row gets a random,
constraint-satisfying assignment for its gate, rows are duplicated in pairs and tied together by copy constraints (so the
permutation argument is non-trivial), and the public inputs are hashed by in-circuit PoseidonGate rows into a
PublicInputGate exactly as `CircuitBuilder::build` does.  The prover's work depends on the shape and the gate types, not
on which values flow where.
"""
import numpy as np

from . import gates as G
from .builder import P, CircuitData, root_of_unity, standard_recursion_config, wide_ecc_config  # noqa: F401
from .prover import poseidon_gate_rows

MASK32 = (1 << 32) - 1


def _limbs2(x, count):
    """[count, R] two-bit limbs of the uint64 array x, little-endian"""
    return np.stack([(x >> np.uint64(2 * j)) & np.uint64(3) for j in range(count)])


def _inv(x):
    return pow(int(x), P - 2, P)


# ---------------------------------------------------------------------------------- vectorised row generators
# each returns (wires [num_wires_used, R] uint64, consts [k, R] uint64)
def rows_arithmetic(gate, R, rng):
    w = np.zeros((4 * gate.num_ops, R), dtype=np.uint64)
    for i in range(gate.num_ops):
        m0 = rng.integers(0, 1 << 31, R, dtype=np.uint64)
        m1 = rng.integers(0, 1 << 31, R, dtype=np.uint64)
        a = rng.integers(0, 1 << 62, R, dtype=np.uint64)
        w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3] = m0, m1, a, m0 * m1 + a
    return w, np.ones((2, R), dtype=np.uint64)


def rows_constant(gate, R, rng):
    c = rng.integers(0, 1 << 63, (gate.num_consts, R), dtype=np.uint64)
    return c.copy(), c


def rows_base_sum(gate, R, rng):
    assert gate.base == 2 and gate.num_limbs <= 63
    x = rng.integers(0, 1 << gate.num_limbs, R, dtype=np.uint64)
    w = np.zeros((1 + gate.num_limbs, R), dtype=np.uint64)
    w[0] = x
    for i in range(gate.num_limbs):
        w[1 + i] = (x >> np.uint64(i)) & np.uint64(1)
    return w, None


def rows_u32_arithmetic(gate, R, rng):
    n = gate.num_ops
    w = np.zeros((38 * n, R), dtype=np.uint64)
    inv_cache = {}
    for i in range(n):
        m0 = rng.integers(0, 1 << 32, R, dtype=np.uint64)
        m1 = rng.integers(0, 1 << 12, R, dtype=np.uint64)     # keeps the number of distinct high halves small (inverse table)
        a = rng.integers(0, 1 << 32, R, dtype=np.uint64)
        out = m0 * m1 + a
        lo, hi = out & np.uint64(MASK32), out >> np.uint64(32)
        for h in np.unique(hi):
            if int(h) not in inv_cache:
                inv_cache[int(h)] = _inv(MASK32 - int(h))
        inv = np.array([inv_cache[int(h)] for h in hi], dtype=np.uint64)
        w[6 * i:6 * i + 6] = np.stack([m0, m1, a, lo, hi, inv])
        w[6 * n + 32 * i:6 * n + 32 * i + 32] = _limbs2(out, 32)
    return w, None


def rows_u32_add_many(gate, R, rng):
    na, n = gate.num_addends, gate.num_ops
    per = na + 3
    w = np.zeros((n * (per + 18), R), dtype=np.uint64)
    for i in range(n):
        adds = rng.integers(0, 1 << 32, (na, R), dtype=np.uint64)
        cin = rng.integers(0, 4, R, dtype=np.uint64)
        s = adds.sum(axis=0) + cin
        res, cout = s & np.uint64(MASK32), s >> np.uint64(32)
        w[per * i:per * i + na] = adds
        w[per * i + na], w[per * i + na + 1], w[per * i + na + 2] = cin, res, cout
        w[per * n + 18 * i:per * n + 18 * i + 16] = _limbs2(res, 16)
        w[per * n + 18 * i + 16:per * n + 18 * i + 18] = _limbs2(cout, 2)
    return w, None


def rows_u32_subtraction(gate, R, rng):
    n = gate.num_ops
    w = np.zeros((21 * n, R), dtype=np.uint64)
    for i in range(n):
        x = rng.integers(0, 1 << 32, R, dtype=np.int64)
        y = rng.integers(0, 1 << 32, R, dtype=np.int64)
        b = rng.integers(0, 2, R, dtype=np.int64)
        d = x - y - b
        bout = (d < 0).astype(np.int64)
        res = (d + (bout << 32)).astype(np.uint64)
        w[5 * i:5 * i + 5] = np.stack([x.astype(np.uint64), y.astype(np.uint64), b.astype(np.uint64), res, bout.astype(np.uint64)])
        w[5 * n + 16 * i:5 * n + 16 * i + 16] = _limbs2(res, 16)
    return w, None


def rows_u32_range_check(gate, R, rng):
    n = gate.num_input_limbs
    w = np.zeros((17 * n, R), dtype=np.uint64)
    for i in range(n):
        x = rng.integers(0, 1 << 32, R, dtype=np.uint64)
        w[i] = x
        w[n + 16 * i:n + 16 * i + 16] = _limbs2(x, 16)
    return w, None


def rows_comparison(gate, R, rng):
    nc, cb = gate.num_chunks, gate.chunk_bits
    assert nc * cb <= 62
    a = rng.integers(0, 1 << gate.num_bits, R, dtype=np.uint64)
    b = rng.integers(0, 1 << gate.num_bits, R, dtype=np.uint64)
    same = rng.integers(0, 4, R) == 0          # a quarter of the rows compare equal values
    b = np.where(same, a, b)
    w = np.zeros((gate.num_wires, R), dtype=np.uint64)
    w[0], w[1] = a, b
    size = 1 << cb
    inv_small = {d: _inv(d % P) for d in range(-size + 1, size) if d}
    msd = np.zeros(R, dtype=np.int64)
    for i in range(nc):
        ca = ((a >> np.uint64(cb * i)) & np.uint64(size - 1)).astype(np.int64)
        cbv = ((b >> np.uint64(cb * i)) & np.uint64(size - 1)).astype(np.int64)
        diff = cbv - ca
        eq = (diff == 0).astype(np.int64)
        w[4 + i], w[4 + nc + i] = ca.astype(np.uint64), cbv.astype(np.uint64)
        w[4 + 2 * nc + i] = np.array([1 if d == 0 else inv_small[int(d)] for d in diff], dtype=np.uint64)
        w[4 + 3 * nc + i] = eq.astype(np.uint64)
        inter = eq * msd
        w[4 + 4 * nc + i] = np.array([int(v) % P for v in inter], dtype=np.uint64)
        msd = inter + (1 - eq) * diff
    w[3] = np.array([int(v) % P for v in msd], dtype=np.uint64)
    top = (size + msd).astype(np.uint64)
    for i in range(cb + 1):
        w[4 + 5 * nc + i] = (top >> np.uint64(i)) & np.uint64(1)
    w[2] = w[4 + 5 * nc + cb]
    return w, None


def rows_poseidon(gate, R, rng):
    ins = rng.integers(0, P, (R, 12), dtype=np.uint64)
    swap = rng.integers(0, 2, R, dtype=np.uint64)
    return np.ascontiguousarray(poseidon_gate_rows(ins, swap).T), None


def rows_noop(gate, R, rng):
    return np.zeros((0, R), dtype=np.uint64), None


# ---------------------------------------------------------------------------------- scalar generators (recursion gates)
def _e_mul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def _e_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def _e_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def _rand_f(rng):
    return int(rng.integers(0, P, dtype=np.uint64))


def _rand_e(rng):
    return (_rand_f(rng), _rand_f(rng))


def _scalar_rows(fill):
    def gen(gate, R, rng):
        cols, consts = None, None
        for r in range(R):
            w, c = fill(gate, rng)
            if cols is None:
                cols = np.zeros((len(w), R), dtype=np.uint64)
                consts = np.zeros((len(c), R), dtype=np.uint64) if c else None
            cols[:, r] = np.array(w, dtype=np.uint64)
            if c:
                consts[:, r] = np.array(c, dtype=np.uint64)
        return cols, consts
    return gen


def _fill_arithmetic_ext(g, rng):
    c0, c1 = _rand_f(rng), _rand_f(rng)
    w = []
    for _ in range(g.num_ops):
        m0, m1, a = _rand_e(rng), _rand_e(rng), _rand_e(rng)
        pr = _e_mul(m0, m1)
        o = ((pr[0] * c0 + a[0] * c1) % P, (pr[1] * c0 + a[1] * c1) % P)
        w += [*m0, *m1, *a, *o]
    return w, [c0, c1]


def _fill_mul_ext(g, rng):
    c0 = _rand_f(rng)
    w = []
    for _ in range(g.num_ops):
        m0, m1 = _rand_e(rng), _rand_e(rng)
        pr = _e_mul(m0, m1)
        w += [*m0, *m1, pr[0] * c0 % P, pr[1] * c0 % P]
    return w, [c0]


def _fill_poseidon_mds(g, rng):
    C = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
    ins = [_rand_e(rng) for _ in range(12)]
    outs = []
    for r in range(12):
        acc = (0, 0)
        for i in range(12):
            v = ins[(i + r) % 12]
            acc = _e_add(acc, (v[0] * C[i] % P, v[1] * C[i] % P))
        if r == 0:
            acc = _e_add(acc, (ins[0][0] * 8 % P, ins[0][1] * 8 % P))
        outs.append(acc)
    return [x for e in ins + outs for x in e], []


def _fill_random_access(g, rng):
    vs = 1 << g.bits
    w = [0] * g.num_wires
    for cp in range(g.num_copies):
        idx = int(rng.integers(0, vs))
        items = [_rand_f(rng) for _ in range(vs)]
        base = (2 + vs) * cp
        w[base], w[base + 1] = idx, items[idx]
        w[base + 2:base + 2 + vs] = items
        for i in range(g.bits):
            w[g.num_routed + cp * g.bits + i] = (idx >> i) & 1
    consts = [_rand_f(rng) for _ in range(g.num_extra_constants)]
    for i, c in enumerate(consts):
        w[(2 + vs) * g.num_copies + i] = c
    return w, consts


def _fill_reducing(g, rng, ext=False):
    n = g.num_coeffs
    alpha, acc = _rand_e(rng), _rand_e(rng)
    w = [0] * g.num_wires
    w[2:4], w[4:6] = alpha, acc
    start_accs = 6 + (2 * n if ext else n)
    for i in range(n):
        coeff = _rand_e(rng) if ext else (_rand_f(rng), 0)
        if ext:
            w[6 + 2 * i:8 + 2 * i] = coeff
        else:
            w[6 + i] = coeff[0]
        acc = _e_add(_e_mul(acc, alpha), coeff)
        if i == n - 1:
            w[0:2] = acc
        else:
            w[start_accs + 2 * i:start_accs + 2 * i + 2] = acc
    return w, []


def _fill_exponentiation(g, rng):
    n = g.num_power_bits
    base = _rand_f(rng)
    bits = [int(rng.integers(0, 2)) for _ in range(n)]
    w = [0] * g.num_wires
    w[0] = base
    w[1:1 + n] = bits
    cur = 1
    for i in range(n):
        prev = 1 if i == 0 else cur * cur % P
        cur = prev * (base if bits[n - 1 - i] else 1) % P
        w[2 + n + i] = cur
    w[1 + n] = cur
    return w, []


def _fill_coset_interpolation(g, rng):
    np_, d = 1 << g.subgroup_bits, g.degree
    ni = g.num_intermediates
    wr = root_of_unity(g.subgroup_bits)
    dom = [pow(wr, i, P) for i in range(np_)]
    shift = (_rand_f(rng) or 1)
    vals = [_rand_e(rng) for _ in range(np_)]
    point = _rand_e(rng)
    si = _inv(shift)
    shifted = (point[0] * si % P, point[1] * si % P)
    start_pt = 1 + 2 * np_
    start_val, start_inter = start_pt + 2, start_pt + 4
    w = [0] * g.num_wires
    w[0] = shift
    for i, v in enumerate(vals):
        w[1 + 2 * i:3 + 2 * i] = v
    w[start_pt:start_pt + 2] = point
    w[start_inter + 4 * ni:start_inter + 4 * ni + 2] = shifted

    def partial(s, e, ev, prod):
        for i in range(s, e):
            term = _e_sub(shifted, (dom[i], 0))
            wv = (vals[i][0] * g.weights[i] % P, vals[i][1] * g.weights[i] % P)
            ev = _e_add(_e_mul(ev, term), _e_mul(wv, prod))
            prod = _e_mul(prod, term)
        return ev, prod
    ev, prod = partial(0, d, (0, 0), (1, 0))
    for i in range(ni):
        w[start_inter + 2 * i:start_inter + 2 * i + 2] = ev
        w[start_inter + 2 * (ni + i):start_inter + 2 * (ni + i) + 2] = prod
        s = 1 + (d - 1) * (i + 1)
        ev, prod = partial(s, min(s + d - 1, np_), ev, prod)
    w[start_val:start_val + 2] = ev
    return w, []


GENERATORS = {
    G.NOOP: rows_noop, G.CONSTANT: rows_constant, G.ARITHMETIC: rows_arithmetic, G.BASE_SUM: rows_base_sum,
    G.POSEIDON: rows_poseidon, G.U32_ARITHMETIC: rows_u32_arithmetic, G.U32_ADD_MANY: rows_u32_add_many,
    G.U32_SUBTRACTION: rows_u32_subtraction, G.U32_RANGE_CHECK: rows_u32_range_check, G.COMPARISON: rows_comparison,
    G.ARITHMETIC_EXT: _scalar_rows(_fill_arithmetic_ext), G.MUL_EXT: _scalar_rows(_fill_mul_ext),
    G.POSEIDON_MDS: _scalar_rows(_fill_poseidon_mds), G.RANDOM_ACCESS: _scalar_rows(_fill_random_access),
    G.REDUCING: _scalar_rows(lambda g, rng: _fill_reducing(g, rng, False)),
    G.REDUCING_EXT: _scalar_rows(lambda g, rng: _fill_reducing(g, rng, True)),
    G.EXPONENTIATION: _scalar_rows(_fill_exponentiation), G.COSET_INTERPOLATION: _scalar_rows(_fill_coset_interpolation),
}


def barycentric_weights(subgroup_bits):
    """weights of the points w^i (i < 2^bits): 1 / prod_{j != i} (x_i - x_j)  (plonky2 interpolation.rs)"""
    w = root_of_unity(subgroup_bits)
    pts = [pow(w, i, P) for i in range(1 << subgroup_bits)]
    out = []
    for i, x in enumerate(pts):
        d = 1
        for j, y in enumerate(pts):
            if i != j:
                d = d * (x - y) % P
        out.append(_inv(d))
    return out


def ed25519_shape_mix(cfg):
    """gate types of the per-signature circuit (u32 gates of crypto/plonky2_u32 + the builder's own), weight = share of rows"""
    return [
        (G.U32ArithmeticGate.new_from_config(cfg), 38), (G.U32AddManyGate.new_from_config(cfg, 3), 10),
        (G.U32SubtractionGate.new_from_config(cfg), 10), (G.U32RangeCheckGate(8), 8), (G.ComparisonGate(32, 16), 3),
        (G.ArithmeticGate.new_from_config(cfg), 18), (G.BaseSumGate(32, 2), 6), (G.ConstantGate(cfg["num_constants"]), 2),
    ]


def recursion_shape_mix(cfg):
    """the 13 gate types of near_bft_finality/proofs/*/common_data.json (PublicInputGate/NoopGate are added by the builder)"""
    return [
        (G.PoseidonGate(), 40), (G.ArithmeticGate.new_from_config(cfg), 14), (G.ArithmeticExtensionGate.new_from_config(cfg), 8),
        (G.MulExtensionGate.new_from_config(cfg), 6), (G.ReducingGate(43), 6), (G.ReducingExtensionGate(32), 6),
        (G.RandomAccessGate.new_from_config(cfg, 4), 6), (G.BaseSumGate(63, 2), 5),
        (G.CosetInterpolationGate(4, 6, barycentric_weights(4)), 3), (G.PoseidonMdsGate(), 2),
        (G.ConstantGate(cfg["num_constants"]), 2),
    ]


def synthetic_circuit(degree_bits, config, mix, num_public_inputs=16, seed=0, fill=0.9):
    """-> (CircuitData, wires uint64 [num_wires, n], public_inputs list)"""
    rng = np.random.default_rng(seed)
    n = 1 << degree_bits
    routed, nw = config["num_routed_wires"], config["num_wires"]
    pi_rows = -(-num_public_inputs // 8)
    budget = int(n * fill) - pi_rows - 1
    assert budget > 2 * len(mix), "circuit too small for the gate mix"
    total_w = sum(w for _, w in mix)
    counts = [max(2, (budget * w // total_w) & ~1) for _, w in mix]
    gate_list = [g for g, _ in mix] + [G.PoseidonGate(), G.PublicInputGate(), G.NoopGate()]
    uniq = sorted(set(gate_list), key=lambda g: (g.degree, g.id()))
    index = {g: i for i, g in enumerate(uniq)}
    row_gate = np.full(n, index[G.NoopGate()], dtype=np.int64)
    ngc = max(g.num_constants for g in uniq)
    row_consts = np.zeros((max(ngc, 1), n), dtype=np.uint64)
    wires = np.zeros((nw, n), dtype=np.uint64)
    sig_col = np.tile(np.arange(routed, dtype=np.int64)[:, None], (1, n))
    sig_row = np.tile(np.arange(n, dtype=np.int64)[None, :], (routed, 1))
    # interleave the gate types over the rows (a real circuit mixes them too)
    order = rng.permutation(np.repeat(np.arange(len(mix)), [c // 2 for c in counts]))
    first_rows = {k: [] for k in range(len(mix))}
    r = 0
    for k in order:
        first_rows[int(k)].append(r)
        r += 2
    body_end = r
    for k, (g, _) in enumerate(mix):
        rows0 = np.array(first_rows[k], dtype=np.int64)
        if len(rows0) == 0:
            continue
        assert g.num_wires <= nw, "%s does not fit %d wires" % (g.id(), nw)
        w, c = GENERATORS[g.code](g, len(rows0), rng)
        for rows in (rows0, rows0 + 1):                     # the row and its duplicate
            row_gate[rows] = index[g]
            wires[:w.shape[0], rows] = w
            if c is not None:
                row_consts[:c.shape[0], rows] = c
        used = min(w.shape[0], routed)                      # tie the duplicate to the original: 2-cycles on every routed column
        for col in range(used):
            sig_row[col, rows0], sig_row[col, rows0 + 1] = rows0 + 1, rows0
    # public inputs: values live in the input wires of the hashing PoseidonGate rows, digest wired into the PublicInputGate
    pis = [int(x) for x in rng.integers(0, P, num_public_inputs, dtype=np.uint64)]
    state = [0] * 12
    r = body_end
    prev_row = None
    for i in range(0, num_public_inputs, 8):
        chunk = pis[i:i + 8]
        state = chunk + state[len(chunk):]
        row = poseidon_gate_rows(np.array([state], dtype=np.uint64))[0]
        row_gate[r] = index[G.PoseidonGate()]
        wires[:135, r] = row
        if prev_row is not None:                            # inputs not overwritten by this chunk = previous outputs
            for j in range(len(chunk), 12):
                sig_col[j, r], sig_row[j, r] = 12 + j, prev_row
                sig_col[12 + j, prev_row], sig_row[12 + j, prev_row] = j, r
        state = [int(x) for x in row[12:24]]
        prev_row = r
        r += 1
    row_gate[r] = index[G.PublicInputGate()]
    wires[:4, r] = np.array(state[:4], dtype=np.uint64)
    if prev_row is not None:
        for j in range(4):
            sig_col[j, r], sig_row[j, r] = 12 + j, prev_row
            # keep an existing 2-cycle partner (none here: outputs 12..15 of the last hash row are otherwise free)
            sig_col[12 + j, prev_row], sig_row[12 + j, prev_row] = j, r
    assert r < n
    data = CircuitData.from_arrays(config, uniq, row_gate, row_consts, sig_col, sig_row, num_public_inputs)
    return data, wires, pis
