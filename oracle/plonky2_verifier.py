"""plonky2 proof verification (BN128-hash outer config) -- transcript, proof-of-work and FRI.
TEST INFRASTRUCTURE.

Restates, line by line, the reference's in-tree Go restatement of the plonky2 verifier:
  gnark-plonky2-verifier/verifier/verifier.go:41-82,143-170   public-input hash, challenges, Verify
  gnark-plonky2-verifier/challenger/challenger.go:42-166      duplex challenger over Poseidon-Goldilocks
  gnark-plonky2-verifier/fri/fri.go:40-73      instance (batches zeta / g*zeta), openings order
  fri/fri.go:75-80       proof of work            fri/fri.go:97-160  Merkle proof to cap (Poseidon-BN254)
  fri/fri.go:187-206     subgroup x               fri/fri.go:208-251 combine initial
  fri/fri.go:314-384     coset interpolation      fri/fri.go:386-497 query round
  fri/fri_utils.go:26-142 oracle / polynomial layout
  poseidon/bn254.go:47-120 hash_no_pad / hash_or_noop / two_to_one / to_vec
  plonk/plonk.go:60-250  vanishing-polynomial identity at zeta (gate evaluators: oracle/plonky2_gates.py)
Both hash configurations are handled: "bn128" (PoseidonBN128GoldilocksConfig, the outer wrap -- all golden
proofs in the tree) and "gl" (PoseidonGoldilocksConfig, the inner proofs; hash = 4 Goldilocks elements).
Everything that involves a hash, a Merkle path, the evaluation-domain order, the FRI folding or a gate
constraint IS checked, which is what pins oracle/poseidon_gl.py, oracle/poseidon_bn254.py,
oracle/plonky2_gates.py and the Merkle conventions against the reference's golden proofs.
"""
from . import goldilocks as gl
from . import poseidon_bn254 as pbn
from . import plonky2_gates as G
from . import poseidon_gl as pgl

P = gl.P


class HasherBN128:
    """PoseidonBN128Hash (crypto/plonky2_bn128/src/config.rs:132-199): digest = one Fr, JSON = decimal string"""
    name = "bn128"
    parse = staticmethod(lambda h: int(h))
    dump = staticmethod(lambda h: str(h))
    hash_or_noop = staticmethod(pbn.hash_or_noop)
    hash_no_pad = staticmethod(pbn.hash_no_pad)
    two_to_one = staticmethod(pbn.two_to_one)
    to_vec = staticmethod(pbn.hash_to_vec)
    pad_to = 9      # config.rs:168-176: pad10*1 to a multiple of RATE * GOLDILOCKS_ELEMENTS


class HasherGL:
    """PoseidonHash (plonky2 [UPSTREAM]; poseidon/goldilocks.go:72-90): digest = 4 Goldilocks elements,
    JSON = {"elements": [..4..]}"""
    name = "gl"
    parse = staticmethod(lambda h: tuple(int(x) for x in h["elements"]))
    dump = staticmethod(lambda h: {"elements": [int(x) for x in h]})
    hash_or_noop = staticmethod(lambda v: tuple(pgl.hash_or_noop(v)))
    hash_no_pad = staticmethod(lambda v: tuple(pgl.hash_no_pad(v)))
    two_to_one = staticmethod(lambda a, b: tuple(pgl.two_to_one(a, b)))
    to_vec = staticmethod(lambda h: list(h))
    pad_to = 8      # plonky2 plonk/config.rs `hash_pad` [UPSTREAM]: pad10*1 to a multiple of the sponge rate


def hash_pad(H, inp):
    v = list(inp) + [1]
    while (len(v) + 1) % H.pad_to:
        v.append(0)
    return H.hash_no_pad(v + [1])


def hasher_of(verifier_json):
    return HasherGL if isinstance(verifier_json["circuit_digest"], dict) else HasherBN128


class Challenger:
    def __init__(self, hasher=HasherBN128):
        self.H = hasher
        self.state = [0] * 12
        self.inp = []
        self.out = []

    def observe(self, e):
        self.out = []
        self.inp.append(e % P)
        if len(self.inp) == 8:
            self._duplex()

    def observe_many(self, es):
        for e in es:
            self.observe(e)

    def observe_hash(self, h):
        self.observe_many(self.H.to_vec(h))

    def observe_cap(self, cap):
        for h in cap:
            self.observe_hash(h)

    def observe_ext(self, x):
        self.observe_many(list(x))

    def challenge(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()

    def challenges(self, n):
        return [self.challenge() for _ in range(n)]

    def ext_challenge(self):
        a, b = self.challenges(2)
        return (a, b)

    def _duplex(self):
        assert len(self.inp) <= 8
        for i, e in enumerate(self.inp):
            self.state[i] = e
        self.inp = []
        self.state = pgl._perm(self.state)
        self.out = list(self.state[:8])


def ext_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = gl.ext_mul(r, a)
        a = gl.ext_mul(a, a)
        e >>= 1
    return r


def reduce_with_powers(terms, alpha):
    s = (0, 0)
    for t in reversed(terms):
        s = gl.ext_add(gl.ext_mul(s, alpha), t)
    return s


def parse_proof(proof_json, verifier_json):
    H = hasher_of(verifier_json)
    hp = H.parse
    pr = proof_json["proof"]
    ext = lambda v: [(int(a) % P, int(b) % P) for a, b in v]
    o = pr["openings"]
    op = pr["opening_proof"]
    rounds = []
    for q in op["query_round_proofs"]:
        init = [([int(x) for x in ep[0]], [hp(s) for s in ep[1]["siblings"]]) for ep in q["initial_trees_proof"]["evals_proofs"]]
        steps = [(ext(st["evals"]), [hp(s) for s in st["merkle_proof"]["siblings"]]) for st in q["steps"]]
        rounds.append((init, steps))
    return {
        "public_inputs": [int(x) for x in proof_json["public_inputs"]],
        "hasher": H,
        "wires_cap": [hp(x) for x in pr["wires_cap"]],
        "zs_pp_cap": [hp(x) for x in pr["plonk_zs_partial_products_cap"]],
        "quotient_cap": [hp(x) for x in pr["quotient_polys_cap"]],
        "openings": {k: ext(o[k]) for k in ["constants", "plonk_sigmas", "wires", "plonk_zs", "plonk_zs_next", "partial_products", "quotient_polys"]},
        "commit_caps": [[hp(x) for x in cap] for cap in op["commit_phase_merkle_caps"]],
        "final_poly": ext(op["final_poly"]["coeffs"]),
        "pow_witness": int(op["pow_witness"]),
        "rounds": rounds,
        "circuit_digest": hp(verifier_json["circuit_digest"]),
        "constants_sigmas_cap": [hp(x) for x in verifier_json["constants_sigmas_cap"]],
    }


def merkle_verify(H, leaf, index, siblings, cap):
    cur = H.hash_or_noop(leaf)
    for s in siblings:
        cur = H.two_to_one(s, cur) if index & 1 else H.two_to_one(cur, s)
        index >>= 1
    return cur == cap[index]


def challenges(pf, common):
    cfg = common["config"]
    nch = cfg["num_challenges"]
    ch = Challenger(pf["hasher"])
    ch.observe_hash(pf["circuit_digest"])
    ch.observe_many(pgl.hash_no_pad(pf["public_inputs"]))
    ch.observe_cap(pf["wires_cap"])
    betas, gammas = ch.challenges(nch), ch.challenges(nch)
    ch.observe_cap(pf["zs_pp_cap"])
    alphas = ch.challenges(nch)
    ch.observe_cap(pf["quotient_cap"])
    zeta = ch.ext_challenge()
    o = pf["openings"]
    batch0 = o["constants"] + o["plonk_sigmas"] + o["wires"] + o["plonk_zs"] + o["partial_products"] + o["quotient_polys"]
    batch1 = o["plonk_zs_next"]
    for x in batch0 + batch1:
        ch.observe_ext(x)
    fri_alpha = ch.ext_challenge()
    fri_betas = []
    for cap in pf["commit_caps"]:
        ch.observe_cap(cap)
        fri_betas.append(ch.ext_challenge())
    for c in pf["final_poly"]:
        ch.observe_ext(c)
    ch.observe(pf["pow_witness"])
    pow_response = ch.challenge()
    indices = ch.challenges(cfg["fri_config"]["num_query_rounds"])
    return {"betas": betas, "gammas": gammas, "alphas": alphas, "zeta": zeta, "fri_alpha": fri_alpha, "fri_betas": fri_betas,
            "pow_response": pow_response, "query_indices": indices, "batches": [batch0, batch1]}


def check_vanishing(pf, common, ch):
    """plonk.go:209-250: vanishing(zeta) == Z_H(zeta) * reduce_with_powers(quotient chunk openings, zeta^n)"""
    K = G.ExtK
    o = pf["openings"]
    degree_bits = common["fri_params"]["degree_bits"]
    n = 1 << degree_bits
    zeta = ch["zeta"]
    zeta_pow_n = ext_pow(zeta, n)
    zh = gl.ext_sub(zeta_pow_n, (1, 0))
    l0 = gl.ext_mul(zh, gl.ext_inv(gl.ext_sub(gl.ext_mul(zeta, (n % P, 0)), (n % P, 0))))
    gates = [G.gate_from_id(g) for g in common["gates"]]
    pih = pgl.hash_no_pad(pf["public_inputs"])
    terms = G.vanishing_terms(K, common, gates, zeta, l0, o["constants"], o["plonk_sigmas"], o["wires"], o["plonk_zs"],
                              o["plonk_zs_next"], o["partial_products"], ch["betas"], ch["gammas"], pih)
    qdf = common["quotient_degree_factor"]
    for i, a in enumerate(ch["alphas"]):
        van = reduce_with_powers(terms, (a, 0))
        t = reduce_with_powers(o["quotient_polys"][i * qdf:(i + 1) * qdf], zeta_pow_n)
        assert van == gl.ext_mul(zh, t), "vanishing polynomial identity, challenge %d" % i


def verify(proof_json, verifier_json, common, max_rounds=None):
    """Returns the derived challenges; raises AssertionError on any failed check."""
    pf = parse_proof(proof_json, verifier_json)
    cfg = common["config"]
    fp = common["fri_params"]
    fc = fp["config"]
    assert not fp["hiding"]
    ch = challenges(pf, common)
    # proof of work (fri.go:75-80): the response must fit in 64 - pow_bits bits
    assert ch["pow_response"] < (1 << (64 - fc["proof_of_work_bits"])), "proof of work"
    degree_bits, rate_bits, cap_h = fp["degree_bits"], fc["rate_bits"], fc["cap_height"]
    n_log = degree_bits + rate_bits
    nch = cfg["num_challenges"]
    if "selectors_info" in common:
        check_vanishing(pf, common, ch)
    # oracle layout (fri_utils.go:60-142)
    n_pre = common["num_constants"] + cfg["num_routed_wires"]
    widths = [n_pre, cfg["num_wires"], nch * (1 + common["num_partial_products"]), nch * common["quotient_degree_factor"]]
    all_polys = [(k, i) for k in range(4) for i in range(widths[k])]
    zs_polys = [(2, i) for i in range(nch)]
    g = gl.root_of_unity(degree_bits)
    zeta = ch["zeta"]
    points = [zeta, gl.ext_mul((g, 0), zeta)]
    polys = [all_polys, zs_polys]
    alpha = ch["fri_alpha"]
    reduced_openings = [reduce_with_powers(b, alpha) for b in ch["batches"]]
    caps = [pf["constants_sigmas_cap"], pf["wires_cap"], pf["zs_pp_cap"], pf["quotient_cap"]]
    assert len(pf["final_poly"]) == 1 << (degree_bits - sum(fp["reduction_arity_bits"]))
    rounds = pf["rounds"] if max_rounds is None else pf["rounds"][:max_rounds]
    for rnd, (init, steps) in enumerate(rounds):
        x_index = ch["query_indices"][rnd] % (1 << n_log)   # low n_log bits of the challenge (fri.go:399-400)
        cap_index = x_index >> (n_log - cap_h)
        for k in range(4):
            leaf, sib = init[k]
            assert len(leaf) == widths[k] and len(sib) == n_log - cap_h
            assert merkle_verify(pf["hasher"], leaf, x_index, sib, caps[k]), "initial tree %d, round %d" % (k, rnd)
        # x = g_mult * w^(bitrev(x_index)) (fri.go:187-206)
        rev = int(format(x_index, "0%db" % n_log)[::-1], 2)
        x = gl.GENERATOR * pow(gl.root_of_unity(n_log), rev, P) % P
        # combine (fri.go:208-251)
        s = (0, 0)
        for b in range(2):
            evals = [(init[k][0][i], 0) for (k, i) in polys[b]]
            red = reduce_with_powers(evals, alpha)
            num = gl.ext_sub(red, reduced_openings[b])
            den = gl.ext_sub((x, 0), points[b])
            s = gl.ext_mul(ext_pow(alpha, len(evals)), s)
            s = gl.ext_add(gl.ext_mul(num, gl.ext_inv(den)), s)
        old = s
        bits = n_log
        idx = x_index
        for i, arity_bits in enumerate(fp["reduction_arity_bits"]):
            evals, sib = steps[i]
            arity = 1 << arity_bits
            within, coset = idx & (arity - 1), idx >> arity_bits
            assert evals[within] == old, "fri consistency, round %d step %d" % (rnd, i)
            # interpolate the coset evaluations at beta (fri.go:314-384)
            gk = gl.root_of_unity(arity_bits)
            rev_within = int(format(within, "0%db" % arity_bits)[::-1], 2)
            start = pow(gl.inv(gk), rev_within, P) * x % P
            xs = [(start * pow(gk, j, P) % P, 0) for j in range(arity)]
            ys = [None] * arity
            for j in range(arity):
                ys[int(format(j, "0%db" % arity_bits)[::-1], 2)] = evals[j]
            beta = ch["fri_betas"][i]
            acc = (0, 0)
            for a in range(arity):
                term = ys[a]
                for b2 in range(arity):
                    if a != b2:
                        term = gl.ext_mul(term, gl.ext_mul(gl.ext_sub(beta, xs[b2]), gl.ext_inv(gl.ext_sub(xs[a], xs[b2]))))
                acc = gl.ext_add(acc, term)
            old = acc
            flat = [c for e in evals for c in e]
            bits -= arity_bits
            assert len(sib) == bits - cap_h
            assert merkle_verify(pf["hasher"], flat, coset, sib, pf["commit_caps"][i]), "commit-phase tree %d, round %d" % (i, rnd)
            assert coset >> (bits - cap_h) == cap_index
            x = pow(x, arity, P)
            idx = coset
        # final polynomial (fri.go:253-259, 493-497)
        ev = (0, 0)
        for c in reversed(pf["final_poly"]):
            ev = gl.ext_add(gl.ext_mul(ev, (x, 0)), c)
        assert ev == old, "final polynomial, round %d" % rnd
    return ch
