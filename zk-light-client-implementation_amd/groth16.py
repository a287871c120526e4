"""`groth16.Prove` of the wrap step on the GPU: the assembly above the MSM / NTT kernels.

Reference: gnark-plonky2-verifier/cmd/web-api.go:77 (`groth16.Prove(r1cs, pk, witness)`), :84 (`groth16.Verify`), :90-98 (the 256
proof bytes); gnark v0.9.1 backend/groth16/bn254/prove.go is un-vendored (go.mod:8).  What it computes, and what runs where:

  host    a, b, c = (A w, B w, C w) per constraint (the solved witness is the caller's: gnark's solver is the Go side)
  GPU     computeH: 3 inverse NTTs, 3 coset NTTs, (a b - c) / (5^n - 1) pointwise, 1 coset inverse NTT  (zklc_bn254_fr_ntt_dev,
          zklc_bn254_fr_mul_sub_scale_dev) -- data stays in HBM between the seven transforms
  GPU     Ar  = alpha + sum_i w_i A_i + r delta                       one G1 MSM (alpha, delta among the bases)        stream 1
          Bs1 = beta1 + sum_i w_i B1_i + s delta1                     one G1 MSM                                        stream 1
          Bs  = beta  + sum_i w_i B_i + s delta                       one G2 MSM                                        stream 2
          Krs = sum_priv w_i K_i + sum_j h_j Z_j + s Ar + r Bs1 - r s delta1
                the K sum: one G1 MSM on stream 1; the Z sum: one G1 MSM after computeH on stream 3; the rest: a five-point MSM at the end
          every base is a fixed point of the key: ZKLC_GROTH16_FIXED=1 runs the four big sums over fixed-base tables (measured: no gain)
  host    the eight coordinates out of Montgomery form -> the uint256[8] / 256-byte / compressed encodings (zklc_amd/formats.py)
The proving key arrives as affine points in gnark-crypto's memory layout (what a cgo shim hands over) and stays
resident on the device.  No CPU fallback: every transform and MSM is a kernel launch through the C ABI.
"""
import os

import numpy as np

R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
_M64 = (1 << 64) - 1


def fr_to_mont_words(x):
    m = (x % R) * (1 << 256) % R
    return [(m >> (64 * i)) & _M64 for i in range(4)]


def fr_to_regular_words(x):
    x %= R
    return [(x >> (64 * i)) & _M64 for i in range(4)]


def fp_from_mont_words(w):
    m = sum(int(w[i]) << (64 * i) for i in range(4))
    return m * pow(1 << 256, P - 2, P) % P


def fp_to_mont_words(x):
    m = (x % P) * (1 << 256) % P
    return [(m >> (64 * i)) & _M64 for i in range(4)]


def g1_words(pt):
    """affine point (x, y) or None (infinity) -> 8 u64 in gnark's layout"""
    return [0] * 8 if pt is None else fp_to_mont_words(pt[0]) + fp_to_mont_words(pt[1])


def g2_words(pt):
    """((x0, x1), (y0, y1)) or None -> 16 u64: X.A0, X.A1, Y.A0, Y.A1"""
    if pt is None:
        return [0] * 16
    (x0, x1), (y0, y1) = pt
    return fp_to_mont_words(x0) + fp_to_mont_words(x1) + fp_to_mont_words(y0) + fp_to_mont_words(y1)


class Groth16Prover:
    """One proving key resident on one GPU.  pk: dict with n (domain size), n_public, point lists A, B1, K, Z (G1), B2 (G2) and
    alpha1, beta1, delta1 (G1), beta2, delta2 (G2) as affine integer tuples (None = infinity) or as uint64 arrays in gnark's
    layout under the same keys with the suffix `_words`."""

    def __init__(self, ctx, pk):
        import torch
        self.ctx, self.torch = ctx, torch
        self.dev = torch.device("cuda", ctx.device_id)
        self.n, self.n_public = int(pk["n"]), int(pk["n_public"])
        self.log_n = self.n.bit_length() - 1
        assert 1 << self.log_n == self.n

        def pts(name, conv, width):
            if name + "_words" in pk:
                a = np.ascontiguousarray(pk[name + "_words"], dtype=np.uint64).reshape(-1, width)
            else:
                v = pk[name]
                a = np.array([conv(p) for p in (v if isinstance(v, list) else [v])], dtype=np.uint64).reshape(-1, width)
            return a
        A, B1, K, Z = pts("A", g1_words, 8), pts("B1", g1_words, 8), pts("K", g1_words, 8), pts("Z", g1_words, 8)
        B2 = pts("B2", g2_words, 16)
        al, be1, de1 = pts("alpha1", g1_words, 8), pts("beta1", g1_words, 8), pts("delta1", g1_words, 8)
        be2, de2 = pts("beta2", g2_words, 16), pts("delta2", g2_words, 16)
        # gnark's ProvingKey stores G1.A, G1.B and G2.B WITHOUT their points at infinity (wires that occur in no A / B column) and
        # keeps the positions as InfinityA / InfinityB (NbInfinityA / NbInfinityB of them); `Prove` filters the wire values the same
        # way before the multi-exponentiations.  pk["infinity_a"] / pk["infinity_b"]: bool [n_wires], True = removed.  Without the
        # masks the arrays are taken as complete (a (0, 0) entry is the point at infinity for the MSM kernel as well).
        inf_a, inf_b = pk.get("infinity_a"), pk.get("infinity_b")
        self.n_wires = len(inf_a) if inf_a is not None else len(inf_b) if inf_b is not None else A.shape[0]
        self.keep_a = None if inf_a is None else ~np.asarray(inf_a, dtype=bool)
        self.keep_b = None if inf_b is None else ~np.asarray(inf_b, dtype=bool)
        n_a = self.n_wires if self.keep_a is None else int(self.keep_a.sum())
        n_b = self.n_wires if self.keep_b is None else int(self.keep_b.sum())
        assert A.shape[0] == n_a and B1.shape[0] == B2.shape[0] == n_b, "key arrays do not match the infinity masks"
        assert K.shape[0] == self.n_wires - 1 - self.n_public and Z.shape[0] == self.n - 1
        up = lambda a: torch.from_numpy(a.view(np.int64)).to(self.dev)
        # MSM operands: every base of the four sums is a FIXED point of the key -- the per-proof points Ar and Bs1 of Krs are folded
        # in by a four-point sum at the end (prove_words).  ZKLC_GROTH16_FIXED=1 puts the operands on fixed-base tables (zklc.h:
        # rows of 2^(c w) P_i, one bucket set, no closing doublings; 16 x the memory, built once here).  Measured at 2^22 on one
        # box (profiles/r06o_groth16_quickbench.txt): 71.2 ms with the tables, 69.6 ms without -- the table form saves the serial
        # tail of a sum, which the three streams below already hide under the other sums' slice kernels, and its 4.3 GB of records
        # per operand no longer sit in the 256 MB last-level cache the way 2^22 plain 64-byte records do.  Default: plain.
        self.fixed = os.environ.get("ZKLC_GROTH16_FIXED", "0") == "1"
        self.n_priv = K.shape[0]
        ops = {"A": (np.concatenate([A, al, de1]), 1), "B1": (np.concatenate([B1, be1, de1]), 1),
               "B2": (np.concatenate([B2, be2, de2]), 2), "K": (K, 1), "Z": (Z, 1)}
        self.delta1_words = de1.reshape(8).copy()
        self.n_op = {k: v[0].shape[0] for k, v in ops.items()}
        self.d_op = {}
        for name, (arr, group) in ops.items():
            if arr.shape[0] == 0:                                 # no private wire: the K sum is skipped
                self.d_op[name] = None
                continue
            d_pts = up(arr)
            if self.fixed:
                self.d_op[name] = ctx.bn254_msm_fixed_table(d_pts, arr.shape[0], group, stream=ctx.stream_ptr())
                ctx.synchronize()                                 # d_pts goes back to torch's allocator below
                del d_pts
            else:
                self.d_op[name] = d_pts
        lib = ctx._lib
        wsb = lambda fn, k: torch.empty(int(fn(k)), dtype=torch.uint8, device=self.dev)
        # one workspace per stream: A, B1 and K run one after the other on ctx, B2 on ctx2, computeH and Z on ctx3
        self.ws1 = wsb(lib.zklc_bn254_g1_msm_workspace_bytes, max(self.n_op["A"], self.n_op["B1"], self.n_op["K"]))
        self.ws2 = wsb(lib.zklc_bn254_g2_msm_workspace_bytes, self.n_op["B2"])
        self.ws3 = wsb(lib.zklc_bn254_g1_msm_workspace_bytes, self.n_op["Z"])
        self.wsn = torch.empty(int(lib.zklc_bn254_fr_ntt_workspace_bytes(self.log_n)), dtype=torch.uint8, device=self.dev)
        self.den = np.array(fr_to_mont_words(pow((pow(5, self.n, R) - 1) % R, R - 2, R)), dtype=np.uint64)
        # resident operands of the Montgomery -> regular conversion of h (a pointwise (a * 1_raw - 0) * 1: see prove_words)
        self.one_raw = torch.zeros((self.n, 4), dtype=torch.int64, device=self.dev)
        self.one_raw[:, 0] = 1
        self.zero = torch.zeros((self.n, 4), dtype=torch.int64, device=self.dev)
        # Three streams per proof (round 6): the sums are independent of each other once Krs is split into sum w_i K_i, sum h_j Z_j
        # and s Ar + r Bs1 - r s delta1, so A, B1 and the K sum (ctx), B2 (ctx2: its slice kernel keeps one wave per SIMD and its
        # serial tail a handful of lanes, both of which the G1 kernels fill) and computeH followed by the Z sum (ctx3) run side by
        # side -- only the Z sum (2^22 of the 2^23 points of gnark's Krs sum) waits for h; the latency-bound tails of one sum
        # (segment / window / final kernels, ~1.6 ms each) lie under the slice kernels of the others.
        from .context import Context
        self.ctx2 = Context(ctx.device_id)
        self.ctx3 = Context(ctx.device_id)           # (the device's high priority for this stream: measured, no effect -- profiles/r06z2_*)
        torch.cuda.synchronize(self.dev)
        self.last_ms = {}

    def close(self):
        for name in ("ctx2", "ctx3"):
            c = getattr(self, name, None)
            if c is not None:
                c.close()
                setattr(self, name, None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ntt(self, d, flags, coset, ctx=None):
        ctx = ctx or self.ctx
        ctx._check(ctx._lib.zklc_bn254_fr_ntt_dev(ctx._h, ctx.stream_ptr(), d.data_ptr(), self.log_n, flags, coset,
                                                  self.wsn.data_ptr(), self.wsn.numel()))

    def _compute_h_enqueue(self, d, ctx):
        """gnark `computeH` on ctx's stream, in place in d[0]; d = the three device tensors [n, 4] (Montgomery) of A w, B w, C w.
        Leaves h in REGULAR form (the MSM takes non-Montgomery scalars): the last step is a multiplication by 1 with the pointwise
        kernel  a <- (a * b - c) * scale,  b = raw 1 (the Montgomery form of 2^-256: strips one factor 2^256), c = 0, scale = 1."""
        from . import _lib
        for x in d:
            self._ntt(x, _lib.NTT_INVERSE, 0, ctx)            # evaluations on the subgroup -> coefficients
            self._ntt(x, 0, 1, ctx)                           # -> evaluations on the coset 5 <w>
        ctx._check(ctx._lib.zklc_bn254_fr_mul_sub_scale_dev(ctx._h, ctx.stream_ptr(), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                                                            self.den.ctypes.data, self.n))
        self._ntt(d[0], _lib.NTT_INVERSE, 1, ctx)             # coset evaluations of (a b - c) / Z -> coefficients of h
        mont_one = np.array(fr_to_mont_words(1), dtype=np.uint64)
        ctx._check(ctx._lib.zklc_bn254_fr_mul_sub_scale_dev(ctx._h, ctx.stream_ptr(), d[0].data_ptr(), self.one_raw.data_ptr(),
                                                            self.zero.data_ptr(), mont_one.ctypes.data, self.n))

    def compute_h(self, a, b, c):
        """gnark `computeH`; a, b, c: uint64 [n, 4] (Montgomery) constraint evaluations -> device tensor of the n coefficients of h
        in regular form"""
        torch = self.torch
        d = [torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint64).view(np.int64)).to(self.dev) for x in (a, b, c)]
        torch.cuda.current_stream(self.dev).synchronize()
        self._compute_h_enqueue(d, self.ctx)
        # b and c go back to torch's allocator when this function returns, and torch hands blocks out on ITS stream: the kernels
        # that still read them run on ctx's stream, so drain it first
        self.ctx.synchronize()
        return d[0]

    def _msm(self, ctx, name, d_sc, d_out, d_inf, ws, group=1):
        n = self.n_op[name]
        if self.fixed:
            ctx.bn254_msm_fixed_dev(self.d_op[name], d_sc, n, d_out, d_inf, ws, ws.numel(), group=group, stream=ctx.stream_ptr())
        elif group == 1:
            ctx.bn254_g1_msm_dev(self.d_op[name], d_sc, n, d_out, d_inf, ws, ws.numel(), stream=ctx.stream_ptr())
        else:
            ctx._check(ctx._lib.zklc_bn254_g2_msm_dev(ctx._h, ctx.stream_ptr(), self.d_op[name].data_ptr(), d_sc.data_ptr(), n,
                                                      d_out.data_ptr(), d_inf.data_ptr(), ws.data_ptr(), ws.numel()))

    def prove(self, witness, abc, r, s):
        """witness: all wire values (ints, witness[0] = 1); abc = (a, b, c): per-constraint values of A w, B w, C w (ints, padded
        to n by this function); r, s: the prover's blinding scalars.  Returns the proof as 8 integers in the order of gnark's
        WriteRawTo / Verifier.sol: A.x, A.y, B.x1, B.x0, B.y1, B.y0, C.x, C.y."""
        assert len(witness) == self.n_wires and witness[0] % R == 1
        mont = lambda v: np.array([fr_to_mont_words(x) for x in list(v) + [0] * (self.n - len(v))], dtype=np.uint64)
        w_reg = np.array([fr_to_regular_words(x) for x in witness], dtype=np.uint64)
        return self.prove_words(w_reg, tuple(mont(v) for v in abc), r, s)

    def prove_words(self, w_reg, abc_mont, r, s):
        """the same with the operands as arrays, the way a cgo shim hands them over: w_reg uint64 [n_wires, 4] = the witness in
        regular (non-Montgomery) form (gnark converts with `fr.Element.BigInt` before MultiExp as well), abc_mont three uint64
        [n, 4] arrays in gnark's Montgomery layout"""
        import time
        torch = self.torch
        t0 = time.perf_counter()
        tsync = lambda: torch.cuda.current_stream(self.dev).synchronize()   # torch builds operands on ITS stream; the kernels run on the contexts'
        w_reg = np.ascontiguousarray(w_reg, dtype=np.uint64).reshape(self.n_wires, 4)
        tail = lambda *xs: np.array([fr_to_regular_words(x) for x in xs], dtype=np.uint64)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(self.dev)
        # the witness crosses PCIe ONCE; the scalar vectors are assembled from it on the device
        d_w = up(w_reg)
        pick = lambda keep: d_w if keep is None else d_w[torch.from_numpy(np.nonzero(keep)[0]).to(self.dev)]
        sc_a = torch.cat([pick(self.keep_a), up(tail(1, r))])
        sc_b = torch.cat([pick(self.keep_b), up(tail(1, s))])
        sc_kp = d_w[1 + self.n_public:]                        # the private wires: a view, no copy
        outs = {k: torch.zeros(16 if k == "B2" else 8, dtype=torch.int64, device=self.dev) for k in ("A", "B1", "B2", "K", "Z")}
        infs = {k: torch.ones(1, dtype=torch.int32, device=self.dev) for k in outs}      # 1 = infinity until a sum has run
        tsync()
        t1 = time.perf_counter()
        # stream 2: Bs = beta + sum_i w_i B_i + s delta (G2);  stream 1: Ar, then Bs1, then the K part of Krs
        self._msm(self.ctx2, "B2", sc_b, outs["B2"], infs["B2"], self.ws2, group=2)
        self._msm(self.ctx, "A", sc_a, outs["A"], infs["A"], self.ws1)
        self._msm(self.ctx, "B1", sc_b, outs["B1"], infs["B1"], self.ws1)
        if self.n_priv:
            self._msm(self.ctx, "K", sc_kp, outs["K"], infs["K"], self.ws1)
        # stream 3: computeH (the three operands cross PCIe while the sums above run), then the Z part of Krs straight from h
        d = [up(x) for x in abc_mont]
        tsync()
        self._compute_h_enqueue(d, self.ctx3)
        t2 = time.perf_counter()
        self._msm(self.ctx3, "Z", d[0], outs["Z"], infs["Z"], self.ws3)        # the first n - 1 coefficients of h
        self.ctx.synchronize()
        self.ctx3.synchronize()
        if int(infs["A"][0]) or int(infs["B1"][0]):
            raise ValueError("groth16: a proof element is the point at infinity (degenerate key or witness)")
        # Krs = sum_priv w_i K_i + sum_j h_j Z_j + s Ar + r Bs1 - r s delta1: five points, through the same kernels
        h = lambda t: t.cpu().numpy().view(np.uint64)
        part = lambda k: np.zeros(8, np.uint64) if int(infs[k][0]) else h(outs[k])
        pts5 = np.stack([part("K"), part("Z"), h(outs["A"]), h(outs["B1"]), self.delta1_words])
        k_w, k_inf = self.ctx.bn254_g1_msm(pts5, tail(1, 1, s, r, (R - r * s % R) % R))
        self.ctx2.synchronize()
        t3 = time.perf_counter()
        if k_inf or int(infs["B2"][0]):
            raise ValueError("groth16: a proof element is the point at infinity (degenerate key or witness)")
        a_w, b_w = h(outs["A"]), h(outs["B2"])
        f = fp_from_mont_words
        proof = [f(a_w[0:4]), f(a_w[4:8]), f(b_w[4:8]), f(b_w[0:4]), f(b_w[12:16]), f(b_w[8:12]), f(k_w[0:4]), f(k_w[4:8])]
        self.last_ms = {"witness_upload_and_scalars": (t1 - t0) * 1e3, "operands_of_compute_h_cross_pcie_beside_A_B1_K_B2": (t2 - t1) * 1e3,
                        "compute_h_z_and_tails": (t3 - t2) * 1e3, "total": (t3 - t0) * 1e3, "fixed_base": self.fixed}
        return proof


def _g1_check(pt, what):
    """affine G1 point (x, y) or None: coordinates reduced and on y^2 = x^3 + 3 (the cofactor of G1 is 1: on the curve = in the group)"""
    from .formats import ProofInvalid
    if pt is None:
        return
    x, y = int(pt[0]), int(pt[1])
    if not (0 <= x < P and 0 <= y < P):
        raise ProofInvalid("%s: coordinate not reduced" % what)
    if (y * y - x * x * x - 3) % P:
        raise ProofInvalid("%s: not on the curve" % what)


def _g2_check_curve(pt, what):
    from .formats import ProofInvalid, _g2_rhs
    if pt is None:
        return
    (x0, x1), (y0, y1) = pt
    if not all(0 <= int(v) < P for v in (x0, x1, y0, y1)):
        raise ProofInvalid("%s: coordinate not reduced" % what)
    r0, r1 = _g2_rhs(int(x0), int(x1))
    if (y0 * y0 - y1 * y1 - r0) % P or (2 * y0 * y1 - r1) % P:
        raise ProofInvalid("%s: not on the twist curve" % what)


def g2_neg(pt):
    return None if pt is None else (pt[0], ((P - pt[1][0]) % P, (P - pt[1][1]) % P))


class Groth16Verifier:
    """`groth16.Verify` (gnark-plonky2-verifier/cmd/web-api.go:84; gnark v0.9.1 backend/groth16/bn254/verify.go, un-vendored) over
    the kernels: the proof's points are validated the way gnark validates them (`Proof.isValid`: Ar, Krs, Bs in their subgroups;
    the decoder has already put them on the curve), the public inputs are folded into one G1 point by the MSM kernel
    (kSum = K[0] + sum_i x_i K[i + 1]), and  e(Ar, Bs) e(Krs, -delta) e(kSum, -gamma) e(alpha, -beta) = 1  is one launch of the
    pairing kernel.  The G2 subgroup test is a multi-exponentiation too: [r - 1] B = -B holds exactly for the points of order
    dividing r (r - 1 < r: the kernel's reduction of scalars modulo r does not touch it, and its addition formulas hold on the whole
    twist curve).

    vk: dict with alpha1 (G1), beta2, gamma2, delta2 (G2) and K (list of n_public + 1 G1 points) as affine integer tuples --
    the layout of oracle.groth16.setup / of `formats.vk_from_gnark_bytes`.  The key's points are validated once, here."""

    def __init__(self, ctx, vk):
        self.ctx = ctx
        self.n_public = len(vk["K"]) - 1
        if self.n_public < 0:
            raise ValueError("verifying key without K[0]")
        _g1_check(vk["alpha1"], "vk.alpha1")
        for i, k in enumerate(vk["K"]):
            _g1_check(k, "vk.K[%d]" % i)
        for name in ("beta2", "gamma2", "delta2"):
            _g2_check_curve(vk[name], "vk." + name)
            self._g2_subgroup(vk[name], "vk." + name)
        self.k_words = np.array([g1_words(p) for p in vk["K"]], dtype=np.uint64).reshape(-1, 8)
        self.alpha_words = g1_words(vk["alpha1"])
        self.neg_g2_words = [g2_words(g2_neg(vk[name])) for name in ("delta2", "gamma2", "beta2")]

    def _g2_subgroup(self, pt, what):
        from .formats import ProofInvalid
        if pt is None:
            return
        out, inf = self.ctx.bn254_g2_msm(np.array([g2_words(pt)], dtype=np.uint64), np.array([fr_to_regular_words(R - 1)], dtype=np.uint64))
        if inf or [int(v) for v in out] != g2_words(g2_neg(pt)):
            raise ProofInvalid("%s: not in the r-torsion subgroup of the twist" % what)

    def public_input_point(self, public_inputs):
        """kSum as 8 words in gnark's layout + the infinity flag"""
        if len(public_inputs) != self.n_public:
            raise ValueError("invalid witness size: %d public inputs, the key has %d" % (len(public_inputs), self.n_public))
        sc = np.array([fr_to_regular_words(1)] + [fr_to_regular_words(int(x)) for x in public_inputs], dtype=np.uint64)
        return self.ctx.bn254_g1_msm(self.k_words, sc)

    def verify(self, proof8, public_inputs):
        """proof8: the eight integers of gnark's WriteRawTo / Verifier.sol order (A.x, A.y, B.x1, B.x0, B.y1, B.y0, C.x, C.y).
        Returns True / False for the pairing equation; raises formats.ProofInvalid for a proof whose points are not valid group
        elements (gnark: an error before any pairing) and ValueError for a wrong number of public inputs."""
        from .formats import ProofInvalid
        p = [int(v) for v in proof8]
        if len(p) != 8:
            raise ValueError("a Groth16 proof has eight words")
        a, c = (p[0], p[1]), (p[6], p[7])
        b = ((p[3], p[2]), (p[5], p[4]))
        if a == (0, 0) or c == (0, 0) or b == ((0, 0), (0, 0)):
            # gnark-crypto encodes infinity as all-zero coordinates; e(O, .) = 1 would drop a factor of the equation
            raise ProofInvalid("a proof element is the point at infinity")
        _g1_check(a, "proof.Ar")
        _g1_check(c, "proof.Krs")
        _g2_check_curve(b, "proof.Bs")
        self._g2_subgroup(b, "proof.Bs")
        l_words, l_inf = self.public_input_point(public_inputs)
        g1 = np.array([[g1_words(a), g1_words(c), [0] * 8 if l_inf else [int(v) for v in l_words], self.alpha_words]], dtype=np.uint64)
        g2 = np.array([[g2_words(b)] + self.neg_g2_words], dtype=np.uint64)
        ok, _ = self.ctx.bn254_pairing_check(g1, g2, 4)
        return bool(int(ok[0]))
