#!/usr/bin/env python3
"""Generates csrc/goldilocks_mul_asm.inc: batches of 2, 3 and 4 independent Goldilocks multiplications (canonical results) as one
hand-scheduled gfx950 inline-asm statement each, built from the instruction lists, scheduler, hazard checker and simulator of
tools/gen_poseidon_asm.py.

Why: hipcc's code for `gl_mul` is 28 instructions (tools' count on gfx950: 6 v_mad_u64_u32, 11 v_mov that build zero-extended 64-bit
addends, compares and selects for the carries); with the carries in SGPR pairs a multiply-reduce with a canonical result is 19.
A single dependent carry chain would need two wait states after every flag write (gfx940+), so a statement interleaves the 2-4
multiplications it is given -- the NTT passes multiply the 15 elements of a butterfly group by their table twiddles in batches
(goldilocks_ntt_group.cuh).  Every list is executed by the simulator against big-integer arithmetic before the file is written;
tests/test_hostsim_goldilocks.py re-runs that and checks the committed file is current.
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_poseidon_asm as G  # noqa: E402

ROOT = G.ROOT
P, M32 = G.P, G.M32
hi = G.hi

# register convention of these statements (all declared as clobbers): temporaries v[VB ..), flag pairs s[FB ..) + vcc
VB = 64
FB, NF = 36, 12
TMP_PER_STREAM = 8
TMP_PER_RANGE_STREAM = 14


def T(n):
    return "v%d" % (VB + n)


def F(n):
    return "F%d" % n


# ---- one more instruction for the lists: ("sor", Fdst, Fa, Fb) = s_or_b64
_flags_read0, _flag_written0, _emit0 = G.flags_read, G.flag_written, G.emit


def flags_read(ins):
    return [ins[2], ins[3]] if ins[0] == "sor" else _flags_read0(ins)


def flag_written(ins):
    return ins[1] if ins[0] == "sor" else _flag_written0(ins)


def fmt_flag(f):
    n = int(f[1:])
    return "vcc" if n == NF else "s[%d:%d]" % (FB + 2 * n, FB + 2 * n + 1)


def emit(ins):
    if ins[0] == "sor":
        return "s_or_b64 %s, %s, %s" % (fmt_flag(ins[1]), fmt_flag(ins[2]), fmt_flag(ins[3]))
    return _emit0(ins, None)


G.flags_read, G.flag_written, G.fmt_flag = flags_read, flag_written, fmt_flag


def simulate(prog, regs):
    """G.simulate + s_or_b64 (the flag dictionary lives inside G.simulate, so the program is run in one piece by a local copy of
    its loop for the new instruction only: flags are modelled as registers "F<k>")"""
    R = dict(regs)
    FL = {}

    def val(x):
        return (x & M32) if isinstance(x, int) else R[x]

    def val64(x):
        return R[x] | (R[hi(x)] << 32)
    for ins in prog:
        op = ins[0]
        if op == "mad":
            _, D, fo, a, b, C = ins
            v = val(a) * val(b) + (val64(C) if C is not None else 0)
            FL[fo] = v >> 64
            assert FL[fo] <= 1
            R[D], R[hi(D)] = v & M32, (v >> 32) & M32
        elif op in ("add_co", "addc"):
            v = val(ins[3]) + val(ins[4]) + (FL[ins[5]] if op == "addc" else 0)
            FL[ins[2]] = v >> 32
            R[ins[1]] = v & M32
        elif op in ("sub_co", "subb"):
            v = val(ins[3]) - val(ins[4]) - (FL[ins[5]] if op == "subb" else 0)
            FL[ins[2]] = 1 if v < 0 else 0
            R[ins[1]] = v & M32
        elif op == "cnd":
            R[ins[1]] = val(ins[3]) if FL[ins[4]] else val(ins[2])
        elif op == "sor":
            FL[ins[1]] = FL[ins[2]] | FL[ins[3]]
        elif op == "sandn2":
            FL[ins[1]] = FL[ins[2]] & (1 - FL[ins[3]])
        elif op == "nop":
            pass
        else:
            raise ValueError(op)
    return R


# ------------------------------------------------------------------------------------------------ instruction lists
def mulmod_canonical(a, b, out, base, fa, fb, fd):
    """out = a * b mod p, CANONICAL (< p), for any 64-bit a, b.  a, b, out: (lo, hi) register names; temporaries T(base .. base+7);
    fa, fb: this stream's flags; fd: the dump flag (never read).
    The 128-bit product is four v_mad_u64_u32 + four carry adds (gen_poseidon_asm.mulmod); the reduction is plonky2's reduce128 with
    the last two steps merged: the value TP = t0 + w2 * eps is 2^64 = eps short when the v_mad carried and p too large when TP >= p,
    i.e. when TP + eps carries; never both (a carried TP is <= 2^64 - 2^33), and both are repaired by the same TP + eps."""
    P0, S, P3, f = T(base), T(base + 2), T(base + 4), T(base + 6)
    LO, w2, w3 = P0, P3, hi(P3)
    TP, TT = S, P3
    return [
        ("mad", P0, fd, a[0], b[0], None),
        ("mad", S, fd, a[0], b[1], None),
        ("mad", S, fa, a[1], b[0], S),                 # S = a0 b1 + a1 b0, 65th bit in fa
        ("mad", P3, fd, a[1], b[1], None),
        ("addc", hi(P3), fd, hi(P3), 0, fa),           # weight 2^96
        ("add_co", hi(P0), fa, hi(P0), S),             # w1
        ("addc", P3, fa, P3, hi(S), fa),               # w2
        ("addc", hi(P3), fd, hi(P3), 0, fa),           # w3
        ("sub_co", LO, fa, LO, w3),                    # 2^96 = -1
        ("subb", hi(LO), fa, hi(LO), 0, fa),
        ("cnd", f, 0, -1, fa),                         # borrow: the wrapped value is 2^64 = eps too large
        ("sub_co", LO, fa, LO, f),
        ("subb", hi(LO), fd, hi(LO), 0, fa),
        ("mad", TP, fa, w2, -1, LO),                   # 2^64 = eps; fa = carry
        ("add_co", TT, fb, TP, -1),                    # TT = TP + eps; carry-out = (TP >= p)
        ("addc", hi(TT), fb, hi(TP), 0, fb),
        ("sor", fa, fa, fb),
        ("cnd", out[0], TP, TT, fa),
        ("cnd", out[1], hi(TP), hi(TT), fa),
    ]


def build_mul(n):
    """n independent products r_k = a_k * b_k; returns (program, outs, ins, nops)"""
    fd = F(NF - 1)
    outs = [("r%dl" % k, "r%dh" % k) for k in range(n)]
    a = [("a%dl" % k, "a%dh" % k) for k in range(n)]
    b = [("b%dl" % k, "b%dh" % k) for k in range(n)]
    streams = [mulmod_canonical(a[k], b[k], outs[k], TMP_PER_STREAM * k, F(2 * k), F(2 * k + 1), fd) for k in range(n)]
    prog, nops = G.schedule(streams)
    return prog, [r for o in outs for r in o], [r for x in a for r in x] + [r for x in b for r in x], nops


def range4_stream(x, out, base, fa, fd):
    """out = x (x - 1)(x - 2)(x - 3) = y (y + 2), y = x (x - 3): the range check of a two-bit limb (plonky2_gates.cuh:
    p2_range_product), loose result.  x canonical.  Temporaries T(base .. base+13)."""
    sc, f = (T(base), T(base + 2), T(base + 4)), T(base + 6)
    D, Y, Y2 = T(base + 8), T(base + 10), T(base + 12)
    return ([
        ("sub_co", D, fa, x[0], 3),
        ("subb", hi(D), fa, x[1], 0, fa),
        ("cnd", f, 0, -1, fa),                         # borrow: x - 3 + 2^64 is eps too large
        ("sub_co", D, fa, D, f),
        ("subb", hi(D), fd, hi(D), 0, fa),
    ] + G.mulmod(x, (D, hi(D)), (Y, hi(Y)), sc, f, fa, fd) + [
        ("add_co", Y2, fa, Y, 2),
        ("addc", hi(Y2), fa, hi(Y), 0, fa),
        ("cnd", f, 0, -1, fa),                         # carry: y + 2 - 2^64 is eps too small
        ("add_co", Y2, fa, Y2, f),
        ("addc", hi(Y2), fd, hi(Y2), 0, fa),
    ] + G.mulmod((Y, hi(Y)), (Y2, hi(Y2)), out, sc, f, fa, fd))


def build_range4(n):
    fd = F(NF - 1)
    outs = [("r%dl" % k, "r%dh" % k) for k in range(n)]
    x = [("x%dl" % k, "x%dh" % k) for k in range(n)]
    streams = [range4_stream(x[k], outs[k], TMP_PER_RANGE_STREAM * k, F(k), fd) for k in range(n)]
    prog, nops = G.schedule(streams, pool=[F(k) for k in range(n, NF - 1)])      # temporaries of the reductions (G.reduce128)
    return prog, [r for o in outs for r in o], [r for v in x for r in v], nops


def check_constant_bus(prog):
    G.check_constant_bus([i for i in prog if i[0] not in ("sor", "sandn2")])


def statement(name, prog, outs, ins, n_tmp, comment):
    names = list(outs) + list(ins)
    assert len(names) <= 30, "an asm statement takes at most 30 operands"
    idx = {nm: "%%%d" % k for k, nm in enumerate(names)}

    def tr(i):
        return tuple(idx.get(x, x) if isinstance(x, str) else x for x in i)
    lines = [emit(tr(i)) for i in prog]
    clob = ["v%d" % (VB + i) for i in range(n_tmp)] + ["s%d" % i for i in range(FB, FB + 2 * NF)] + ["vcc", "scc"]
    c = ["// %s" % comment, "ZKLC_D void %s(%s) {" % (name, ", ".join(["u32 &%s" % x for x in outs] + ["u32 %s" % x for x in ins])),
         "    asm volatile("]
    c += ['        "%s\\n\\t"' % ln for ln in lines]
    c.append("        : " + ", ".join('"=&v"(%s)' % x for x in outs))
    c.append("        : " + ", ".join('"v"(%s)' % x for x in ins))
    c.append("        : " + ", ".join('"%s"' % x for x in clob) + ");")
    c.append("}")
    return "\n".join(c)


def selftest():
    rng = random.Random(77)
    edge = [0, 1, P - 1, P, P + 1, 2**64 - 1, 2**32 - 1, 2**32, 2**63, 0xFFFFFFFF00000000, 0xFFFFFFFF]
    for n in (1, 2, 3, 4):
        prog, outs, ins, _ = build_mul(n)
        G.check_hazards(prog)
        check_constant_bus(prog)
        for it in range(400):
            vals = [rng.choice(edge) if rng.random() < 0.3 else rng.getrandbits(64) for _ in range(2 * n)]
            regs = {}
            for nm, v in zip(["a%d" % k for k in range(n)] + ["b%d" % k for k in range(n)], vals):
                regs[nm + "l"], regs[nm + "h"] = v & M32, v >> 32
            R = simulate(prog, regs)
            for k in range(n):
                got = R["r%dl" % k] | (R["r%dh" % k] << 32)
                assert got == vals[k] * vals[n + k] % P, ("mul%d" % n, k, vals)
    for n in (2, 3, 4):
        prog, outs, ins, _ = build_range4(n)
        G.check_hazards(prog)
        check_constant_bus(prog)
        for it in range(300):
            vals = [rng.choice([0, 1, 2, 3, 4, P - 1, P - 2]) if rng.random() < 0.4 else rng.randrange(P) for _ in range(n)]
            regs = {}
            for k, v in enumerate(vals):
                regs["x%dl" % k], regs["x%dh" % k] = v & M32, v >> 32
            R = simulate(prog, regs)
            for k, v in enumerate(vals):
                got = R["r%dl" % k] | (R["r%dh" % k] << 32)
                assert got % P == v * (v - 1) * (v - 2) * (v - 3) % P, ("range4_%d" % n, k, v)
    return True


def render():
    parts = ["// GENERATED by tools/gen_gl_asm.py -- do not edit.  Batches of independent Goldilocks multiplications (canonical results),",
             "// one hand-scheduled asm statement each; every instruction list was executed against big-integer arithmetic by the generator.",
             "#pragma once", ""]
    for n in (2, 3, 4):
        prog, outs, ins, nops = build_mul(n)
        G.check_hazards(prog)
        check_constant_bus(prog)
        parts.append(statement("gl_mul%d_asm" % n, prog, outs, ins, TMP_PER_STREAM * n,
                               "r_k = a_k * b_k mod p (canonical), k < %d: %d instructions, %d s_nop" % (n, len(prog) - nops, nops)))
        parts.append("")
    for n in (2, 3, 4):
        prog, outs, ins, nops = build_range4(n)
        G.check_hazards(prog)
        check_constant_bus(prog)
        parts.append(statement("p2_range4_%d_asm" % n, prog, outs, ins, TMP_PER_RANGE_STREAM * n,
                               "r_k = x_k (x_k - 1)(x_k - 2)(x_k - 3) mod p (loose; x_k canonical), k < %d: %d instructions, %d s_nop"
                               % (n, len(prog) - nops, nops)))
        parts.append("")
    return "\n".join(parts)


def main():
    assert selftest()
    out = os.path.join(ROOT, "zk-light-client-implementation_amd", "csrc", "goldilocks_mul_asm.inc")
    text = render()
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        assert open(out).read() == text, "csrc/goldilocks_mul_asm.inc is not current: run tools/gen_gl_asm.py"
        print("ok")
        return
    open(out, "w").write(text)
    print("wrote", out, len(text.splitlines()), "lines")


if __name__ == "__main__":
    main()
