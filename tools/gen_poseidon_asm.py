#!/usr/bin/env python3
"""Generates csrc/poseidon_gl_asm.inc: the Poseidon-Goldilocks permutation as hand-scheduled gfx950 inline-asm statements.

Why: the compiler's code for the permutation is ~31 k VALU instructions (profiles/r02_gl_quickbench_pmc_sq_v0.csv), a third of
them v_mov / v_cmp / v_cndmask that emulate carries and build zero-extended 64-bit addends for v_mad_u64_u32.  Here
  * carries live in SGPR pairs (VOP3B carry-out of v_mad_u64_u32 / v_add_co / v_sub_co) and are consumed by v_addc / v_subb;
  * a 64 x 64 product is four v_mad_u64_u32 with natural 64-bit addends (no zero-extension moves);
  * the dense layers of the fast partial rounds (the 11 x 11 initial matrix, the w_hat dot products) accumulate 64-bit COLUMNS
    (x0 k0 | x0 k1 + x1 k0 | x1 k1, one carry counter each) and are reduced once per output;
  * the next round's constants enter the MDS accumulators as the initial addend (no separate constant layer).
gfx940+ needs two wait states between a VALU that writes an SGPR / VCC and a VALU that reads it, and hipcc pads nothing inside
inline asm: every statement therefore interleaves several independent instruction streams (three S-boxes, two MDS rows, the
S-box + dot product + state updates of a partial round) and the scheduler below only places a flag reader >= 3 slots after its
writer (s_nop otherwise).  The same instruction lists are EXECUTED by the simulator in this file against big-integer
arithmetic before anything is written, so a logic error never reaches the GPU; tests/test_hostsim_goldilocks.py re-runs it.

Register convention inside a statement (all declared as clobbers): temporaries v[VB .. VB+NT), constants s[SB .. SB+48),
carry flags s[FB ..) in pairs + vcc.  State words are compiler-allocated 32-bit operands (%k).

Reference for WHAT is computed: gnark-plonky2-verifier/poseidon/goldilocks.go:92-115 (rounds), :138-145 (x^7), :172-216 (MDS),
:231-331 (fast partial rounds); plonky2's poseidon.rs is the un-vendored original.
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2**64 - 2**32 + 1
M32 = 0xFFFFFFFF
MDS_C = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]

VB, NT = 48, 60          # fixed temporary VGPRs v[48:107]
LB = 32                  # s[32:33] table pointer, s34 round counter of the looped statement
SB = 36                  # constants s[36:83]
FB, NF = 84, 9           # flag pairs s[84:85] .. s[100:101]; flag index NF = vcc


def T(n):
    assert 0 <= n < NT
    return "v%d" % (VB + n)


def K(n):
    assert 0 <= n < 48
    return "s%d" % (SB + n)


def F(n):
    return "F%d" % n


def hi(r):
    assert r[0] in "vs", r
    return "%s%d" % (r[0], int(r[1:]) + 1)


# ------------------------------------------------------------------------------------------------ instruction lists
def mulmod(a, b, out, sc, f, fl, fd):
    """out = a * b mod p (loose).  a, b, out: (lo, hi) register names; sc = (P0, S, P3) scratch pair bases; f: scratch VGPR;
    fl: this stream's flag; fd: the dump flag (never read)"""
    P0, S, P3 = sc
    return [
        ("mad", P0, fd, a[0], b[0], None),
        ("mad", S, fd, a[0], b[1], None),
        ("mad", S, fl, a[1], b[0], S),                 # S = a0 b1 + a1 b0, 65th bit in fl
        ("mad", P3, fd, a[1], b[1], None),
        ("addc", hi(P3), fd, hi(P3), 0, fl),           # weight 2^96
        ("add_co", hi(P0), fl, hi(P0), S),             # w1
        ("addc", P3, fl, P3, hi(S), fl),               # w2
        ("addc", hi(P3), fd, hi(P3), 0, fl),           # w3
    ] + reduce128(P0, P3, hi(P3), None, S, out, f, fl, fd)


def reduce128(LO, w2, w3, w4, TP, out, f, fl, fd):
    """out = {LO} + w2 * eps - {w4, w3}  (eps = 2^32 - 1; 2^64 = eps, 2^96 = -1, 2^128 = -2^32 mod p); TP: scratch pair"""
    return [
        ("sub_co", LO, fl, LO, w3),
        ("subb", hi(LO), fl, hi(LO), w4 if w4 is not None else 0, fl),
        ("cnd", f, 0, -1, fl),                         # borrow: the wrapped value is 2^64 = eps too large
        ("sub_co", LO, fl, LO, f),
        ("subb", hi(LO), fd, hi(LO), 0, fl),
        ("mad", TP, fl, w2, -1, LO),
        ("cnd", f, 0, -1, fl),                         # carry: add eps back
        ("add_co", out[0], fl, TP, f),
        ("addc", out[1], fd, hi(TP), 0, fl),
    ]


def sbox_stream(x, base, fl, fd, out=None):
    """x <- x^7 (three multiplications deep: x2, x3 = x2 x, x4 = x2^2, x7 = x3 x4); temporaries T(base .. base+10)"""
    X2, X3 = (T(base), T(base + 1)), (T(base + 2), T(base + 3))
    sc = (T(base + 4), T(base + 6), T(base + 8))
    f = T(base + 10)
    out = out or x
    return (mulmod(x, x, X2, sc, f, fl, fd) + mulmod(X2, x, X3, sc, f, fl, fd) + mulmod(X2, X2, X2, sc, f, fl, fd)
            + mulmod(X3, X2, out, sc, f, fl, fd))


def mds_row_stream(r, lo, hi_, out, base, rc, fl, fd):
    """row r of the MDS layer + the next constant layer: out = sum_i C[i] s[(i + r) % 12] (+ 8 s[0] for r = 0) + rc, loose.
    rc = (SGPR pair with the constant's low word zero-extended, the same for its high word)"""
    Al, Bl, Ah, Bh, f = T(base), T(base + 2), T(base + 4), T(base + 6), T(base + 8)
    terms = [((i + r) % 12, MDS_C[i]) for i in range(12)]
    a_terms, b_terms = terms[:6], terms[6:]
    if r == 0:
        a_terms = a_terms + [(0, 8)]
    ins = []
    for k, (j, c) in enumerate(a_terms):
        ins.append(("mad", Al, fd, lo[j], c, Al if k else None))
        ins.append(("mad", Ah, fd, hi_[j], c, Ah if k else None))
    ins.append(("needconst",))
    for k, (j, c) in enumerate(b_terms):
        ins.append(("mad", Bl, fd, lo[j], c, Bl if k else rc[0]))
        ins.append(("mad", Bh, fd, hi_[j], c, Bh if k else rc[1]))
    ins += [
        ("add64", Al, Al, Bl),                         # sl < 2^42
        ("add64", Ah, Ah, Bh),                         # sh < 2^42; value = sl + sh 2^32
        ("mad", Al, fd, hi(Ah), -1, Al),               # sh.hi 2^64 = sh.hi eps: no carry (both < 2^42)
        ("add_co", hi(Al), fl, hi(Al), Ah),            # + sh.lo 2^32
        ("cnd", f, 0, -1, fl),
        ("add_co", out[0], fl, Al, f),
        ("addc", out[1], fd, hi(Al), 0, fl),
    ]
    return ins


def dot_streams(xs, ks, base, flags, fd, yterm=None, needconst=True):
    """sum_j x_j k_j as four column accumulators (64 bits + a carry counter each).  xs: [(x0, x1)] registers, ks: [(k0, k1)]
    SGPRs.  Returns the four streams; temporaries T(base .. base+11).  yterm = (y, 25, event): adds 25 y once `event` is up"""
    C0, C1a, C1b, C2 = T(base), T(base + 2), T(base + 4), T(base + 6)
    n0, n1a, n1b, n2 = T(base + 8), T(base + 9), T(base + 10), T(base + 11)
    st = [[("mov", n, 0)] + ([("needconst",)] if needconst else []) for n in (n0, n1a, n1b, n2)]
    for j, ((x0, x1), (k0, k1)) in enumerate(zip(xs, ks)):
        for s_, (acc, cnt, xa, kb) in enumerate(((C0, n0, x0, k0), (C1a, n1a, x0, k1), (C1b, n1b, x1, k0), (C2, n2, x1, k1))):
            st[s_].append(("mad", acc, flags[s_], xa, kb, acc if j else None))
            st[s_].append(("addc", cnt, fd, cnt, 0, flags[s_]))
    for s_ in range(4):
        st[s_].append(("signal", "rd%d_%d" % (base, s_)))     # the x_j have been read: they may be overwritten from here on
    if yterm:
        (y0, y1), c, ev = yterm
        st[0] += [("wait", ev), ("mad", C0, flags[0], y0, c, C0), ("addc", n0, fd, n0, 0, flags[0])]
        st[1] += [("wait", ev), ("mad", C1a, flags[1], y1, c, C1a), ("addc", n1a, fd, n1a, 0, flags[1])]
    return st


def fold_stream(base, out, f, fl, fd, waits):
    """the 160-bit value of dot_streams' accumulators -> out (loose)"""
    C0, C1a, C1b, C2 = T(base), T(base + 2), T(base + 4), T(base + 6)
    n0, n1a, n1b, n2 = T(base + 8), T(base + 9), T(base + 10), T(base + 11)
    ins = [("wait", w) for w in waits]
    ins += [
        ("add_co", C1a, fl, C1a, C1b),
        ("addc", hi(C1a), fl, hi(C1a), hi(C1b), fl),
        ("addc", n1a, fd, n1a, n1b, fl),               # column 1 = {n1a, C1a}
        ("add_co", hi(C0), fl, hi(C0), C1a),           # w1
        ("addc", C2, fl, C2, hi(C1a), fl),             # w2 = C2.lo + C1.hi + c
        ("addc", hi(C2), fl, hi(C2), n1a, fl),         # w3 = C2.hi + n1 + c
        ("addc", n2, fd, n2, 0, fl),                   # w4 = n2 + c
        ("add_co", C2, fl, C2, n0),                    # w2 += n0
        ("addc", hi(C2), fl, hi(C2), 0, fl),
        ("addc", n2, fd, n2, 0, fl),
    ]
    return ins + reduce128(C0, C2, hi(C2), n2, C1a, out, f, fl, fd)


def update_stream(y, v, s, sc, f, fl, fd, evs):
    """s <- s + y * v (loose); v = (v0, v1) SGPRs"""
    P0, S, P3 = sc
    return [("wait", e) for e in evs] + [
        ("mad", P0, fd, y[0], v[0], None),
        ("mad", S, fd, y[0], v[1], None),
        ("mad", S, fl, y[1], v[0], S),
        ("mad", P3, fd, y[1], v[1], None),
        ("addc", hi(P3), fd, hi(P3), 0, fl),
        ("add_co", hi(P0), fl, hi(P0), S),
        ("addc", P3, fl, P3, hi(S), fl),
        ("addc", hi(P3), fd, hi(P3), 0, fl),
        ("add_co", P0, fl, P0, s[0]),                  # + s (the sum stays below 2^128)
        ("addc", hi(P0), fl, hi(P0), s[1], fl),
        ("addc", P3, fl, P3, 0, fl),
        ("addc", hi(P3), fd, hi(P3), 0, fl),
    ] + reduce128(P0, P3, hi(P3), None, S, s, f, fl, fd)


# ------------------------------------------------------------------------------------------------------- scheduler
def flags_read(ins):
    op = ins[0]
    if op in ("addc", "subb"):
        return [ins[5]]
    if op == "cnd":
        return [ins[4]]
    return []


def flag_written(ins):
    return ins[2] if ins[0] in ("mad", "add_co", "addc", "sub_co", "subb") else None


def schedule(streams, prologue=()):
    """greedy round-robin merge; a flag reader is placed >= 3 slots after the flag's writer"""
    out = list(prologue)
    pos = len(out)
    lastw, events = {}, set()
    heads = [0] * len(streams)
    waited = False
    rr = nops = 0
    while any(h < len(s) for h, s in zip(heads, streams)):
        chosen = None
        progressed = False
        for k in range(len(streams)):
            i = (rr + k) % len(streams)
            while heads[i] < len(streams[i]) and streams[i][heads[i]][0] in ("needconst", "wait", "signal"):
                m = streams[i][heads[i]]
                if m[0] == "needconst":
                    if not waited:
                        out.append(("waitcnt",))
                        pos += 1
                        waited = True
                elif m[0] == "signal":
                    events.add(m[1])
                elif m[1] not in events:
                    break
                heads[i] += 1
                progressed = True
            if heads[i] >= len(streams[i]):
                continue
            ins = streams[i][heads[i]]
            if ins[0] == "wait":
                continue
            if all(pos - lastw.get(f, -9) >= 3 for f in flags_read(ins)):
                chosen = i
                break
        if chosen is None:
            if progressed:
                continue
            assert any(h < len(s) and s[h][0] != "wait" for h, s in zip(heads, streams)), "deadlock: every stream waits"
            out.append(("nop",))
            nops += 1
            pos += 1
            continue
        ins = streams[chosen][heads[chosen]]
        out.append(ins)
        w = flag_written(ins)
        if w is not None:
            lastw[w] = pos
        heads[chosen] += 1
        pos += 1
        rr = (chosen + 1) % len(streams)
    return out, nops


def check_hazards(prog):
    lastw = {}
    for pos, ins in enumerate(prog):
        for f in flags_read(ins):
            assert pos - lastw.get(f, -9) >= 3, ("flag hazard", pos, ins)
        w = flag_written(ins) if ins[0] not in ("nop", "waitcnt", "sload") else None
        if w is not None:
            lastw[w] = pos


# ------------------------------------------------------------------------------------------------------- simulator
def simulate(prog, regs, consts=None):
    """executes the instruction list on a dict of 32-bit registers (operands "%k", temporaries, SGPR constants)"""
    R = dict(regs)
    FL = {}

    def val(x):
        if isinstance(x, int):
            return x & M32
        return R[x]

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
        elif op == "add64":
            v = (val64(ins[2]) + val64(ins[3])) & (2**64 - 1)
            R[ins[1]], R[hi(ins[1])] = v & M32, v >> 32
        elif op == "mov":
            R[ins[1]] = val(ins[2])
        elif op == "sload":
            _, base, n, off = ins
            for k in range(n):
                R["s%d" % (int(base[1:]) + k)] = consts[off // 4 + k]      # `consts` starts at the statement's table pointer
        elif op in ("nop", "waitcnt"):
            pass
        else:
            raise ValueError(op)
    return R


# --------------------------------------------------------------------------------------------------------- printing
def fmt_flag(f):
    n = int(f[1:])
    return "vcc" if n == NF else "s[%d:%d]" % (FB + 2 * n, FB + 2 * n + 1)


def fmt_pair(r):
    return "%s[%d:%d]" % (r[0], int(r[1:]), int(r[1:]) + 1)


def fmt_src(x):
    return str(x) if isinstance(x, int) else x


def emit(ins, ptr_operand):
    op = ins[0]
    if op == "mad":
        _, D, fo, a, b, C = ins
        return "v_mad_u64_u32 %s, %s, %s, %s, %s" % (fmt_pair(D), fmt_flag(fo), fmt_src(a), fmt_src(b), fmt_pair(C) if C else "0")
    if op == "add_co":
        return "v_add_co_u32_e64 %s, %s, %s, %s" % (ins[1], fmt_flag(ins[2]), fmt_src(ins[3]), fmt_src(ins[4]))
    if op == "addc":
        return "v_addc_co_u32_e64 %s, %s, %s, %s, %s" % (ins[1], fmt_flag(ins[2]), fmt_src(ins[3]), fmt_src(ins[4]), fmt_flag(ins[5]))
    if op == "sub_co":
        return "v_sub_co_u32_e64 %s, %s, %s, %s" % (ins[1], fmt_flag(ins[2]), fmt_src(ins[3]), fmt_src(ins[4]))
    if op == "subb":
        return "v_subb_co_u32_e64 %s, %s, %s, %s, %s" % (ins[1], fmt_flag(ins[2]), fmt_src(ins[3]), fmt_src(ins[4]), fmt_flag(ins[5]))
    if op == "cnd":
        return "v_cndmask_b32_e64 %s, %s, %s, %s" % (ins[1], fmt_src(ins[2]), fmt_src(ins[3]), fmt_flag(ins[4]))
    if op == "add64":
        return "v_lshl_add_u64 %s, %s, 0, %s" % (fmt_pair(ins[1]), fmt_pair(ins[2]), fmt_pair(ins[3]))
    if op == "mov":
        return "v_mov_b32_e32 %s, %s" % (ins[1], fmt_src(ins[2]))
    if op == "sload":
        _, base, n, off = ins
        b = int(base[1:])
        return "s_load_dwordx%d s[%d:%d], %s, 0x%x" % (n, b, b + n - 1, ptr_operand, off)
    if op == "waitcnt":
        return "s_waitcnt lgkmcnt(0)"
    if op == "nop":
        return "s_nop 0"
    raise ValueError(op)


def check_constant_bus(prog):
    """a gfx9 VOP3 instruction may read ONE SGPR (pair) through the constant bus; a carry-in counts"""
    for ins in prog:
        if ins[0] in ("nop", "waitcnt", "sload"):
            continue
        srcs = [x for x in ins[3:] if isinstance(x, str)] if ins[0] != "cnd" else [x for x in ins[2:] if isinstance(x, str)]
        if ins[0] == "add64":
            srcs = [ins[2], ins[3]]
        if ins[0] == "mov":
            srcs = [ins[2]] if isinstance(ins[2], str) else []
        sg = set(x for x in srcs if x[0] in "sF")
        assert len(sg) <= 1, ("constant bus", ins)


CLOBBERS = ["v%d" % (VB + i) for i in range(NT)] + ["s%d" % i for i in range(LB, FB + 2 * NF)] + ["vcc", "scc"]


def statement(name, prog, inouts, ins_ops, outs=(), ptr=False, comment="", loop=None, pre=()):
    """C++ text of one inline function holding one asm statement.  inouts / ins_ops / outs: operand register names "%k" are
    assigned here in this order: outs ("=&v"), inouts ("+v"), inputs ("v"), then the table pointer ("s").
    loop = (rounds, stride_bytes): `prog` is the body of a loop over table rows; the pointer lives in s[LB:LB+1], the
    counter in s(LB+2); `pre`: instructions before the loop (they use the pointer operand's copy as well)."""
    names = list(outs) + list(inouts) + list(ins_ops)
    idx = {n: "%%%d" % k for k, n in enumerate(names)}
    ptr_op = "%%%d" % len(names)

    def tr(ins):
        return tuple(idx.get(x, x) if isinstance(x, str) else x for x in ins)
    if loop:
        lp = "s[%d:%d]" % (LB, LB + 1)
        lines = ["s_mov_b64 %s, %s" % (lp, ptr_op), "s_movk_i32 s%d, %d" % (LB + 2, loop[0])]
        lines += [emit(tr(i), lp) for i in pre]
        lines.append("pgl_loop_%=:")
        lines += [emit(tr(i), lp) for i in prog]
        lines += ["s_add_u32 s%d, s%d, %d" % (LB, LB, loop[1]), "s_addc_u32 s%d, s%d, 0" % (LB + 1, LB + 1),
                  "s_sub_u32 s%d, s%d, 1" % (LB + 2, LB + 2), "s_cmp_lg_u32 s%d, 0" % (LB + 2), "s_cbranch_scc1 pgl_loop_%="]
    else:
        lines = [emit(tr(i), ptr_op) for i in prog]
    params = ["u32 &%s" % n for n in list(outs) + list(inouts)] + ["u32 %s" % n for n in ins_ops] + (["const u32 *tab"] if ptr else [])
    c = ["// %s" % comment if comment else "", "ZKLC_D void %s(%s) {" % (name, ", ".join(params)), "    asm volatile("]
    for ln in lines:
        c.append('        "%s\\n\\t"' % ln)
    c.append("        : " + ", ".join(['"=&v"(%s)' % n for n in outs] + ['"+v"(%s)' % n for n in inouts]))
    c.append("        : " + ", ".join(['"v"(%s)' % n for n in ins_ops] + (['"s"(tab)'] if ptr else [])))
    c.append("        : " + ", ".join('"%s"' % x for x in CLOBBERS) + ");")
    c.append("}")
    return "\n".join(x for x in c if x != "")


# ---------------------------------------------------------------------------------------------- statement builders
def build_sbox3():
    fd = F(0)
    xs = [("x%dl" % i, "x%dh" % i) for i in range(3)]
    streams = [sbox_stream(xs[i], 12 * i, F(1 + i), fd) for i in range(3)]
    prog, nops = schedule(streams)
    return prog, [r for x in xs for r in x], nops


def build_mds2(rows):
    fd = F(0)
    lo = ["l%d" % i for i in range(12)]
    hi_ = ["h%d" % i for i in range(12)]
    outs = []
    streams = []
    for k, r in enumerate(rows):
        o = ("o%dl" % k, "o%dh" % k)
        outs += list(o)
        streams.append(mds_row_stream(r, lo, hi_, o, 10 * k, (K(4 * k), K(4 * k + 2)), F(1 + k), fd))
    prologue = [("sload", K(0), 8, 32 * (rows[0] // 2))]
    prog, nops = schedule(streams, prologue)
    return prog, outs, lo + hi_, nops


def build_partial():
    """one fast partial round: y = s0^7 + rc; s0' = 25 y + sum_j w_j s_j; s_j += y v_j.  Table row (48 dwords):
    w (k0, k1) x 11 | rc lo, hi | v (v0, v1) x 11 | 2 pad"""
    fd = F(0)
    s0 = ("q0l", "q0h")
    sj = [("q%dl" % j, "q%dh" % j) for j in range(1, 12)]
    Y = (T(12), T(13))
    DB = 14                       # dot accumulators T(14..25), fold scratch f = T(26)
    sbox = sbox_stream(s0, 0, F(1), fd, out=Y)
    f = T(10)
    sbox += [("needconst",), ("mov", f, K(23)), ("add_co", Y[0], F(1), Y[0], K(22)), ("addc", Y[1], F(1), Y[1], f, F(1)),
             ("cnd", f, 0, -1, F(1)), ("add_co", Y[0], F(1), Y[0], f), ("addc", Y[1], fd, Y[1], 0, F(1)), ("signal", "y")]
    ks = [(K(2 * j), K(2 * j + 1)) for j in range(11)]
    dots = dot_streams(sj, ks, DB, [F(2), F(3), F(4), F(5)], fd, yterm=(Y, 25, "y"))
    for k, d in enumerate(dots):
        d.append(("signal", "dot%d" % k))
    fold = fold_stream(DB, s0, T(26), F(2), fd, ["dot0", "dot1", "dot2", "dot3"])
    # state updates: three streams; the first reuses the S-box scratch (free once y exists)
    usc = [((T(4), T(6), T(8)), T(10), F(1)), ((T(28), T(30), T(32)), T(34), F(6)), ((T(36), T(38), T(40)), T(42), F(7))]
    ups = [[], [], []]
    for j in range(11):
        sc, fj, fl = usc[j % 3]
        ups[j % 3] += update_stream(Y, (K(24 + 2 * j), K(25 + 2 * j)), sj[j], sc, fj, fl, fd, ["y"] + ["rd%d_%d" % (DB, q) for q in range(4)])
    # the fold rewrites s0 (an S-box input) only after the S-box finished: it waits for the dot products, which wait for y
    prologue = [("sload", K(0), 16, 0), ("sload", K(16), 16, 64), ("sload", K(32), 16, 128)]
    prog, nops = schedule([sbox] + dots + [fold] + ups, prologue)
    return prog, list(s0) + [r for x in sj for r in x], nops


def build_fullround():
    """one full round in place: twelve S-boxes (three streams of four; results in temporaries), then the MDS layer + the next
    constant layer written back to the state operands (three streams of four rows).  The 48 dwords of constants are fetched
    at the top and arrive behind the S-boxes.  Table row: {lo, 0, hi, 0} x 12"""
    fd = F(0)
    xs = [("x%dl" % i, "x%dh" % i) for i in range(12)]
    Y = [(T(2 * i), T(2 * i + 1)) for i in range(12)]
    streams = []
    for k in range(3):
        st = []
        for i in range(k, 12, 3):
            st += sbox_stream(xs[i], 24 + 12 * k, F(1 + k), fd, out=Y[i])
            st.append(("signal", "sb%d" % i))
        streams.append(st)
    lo, hi_ = [y[0] for y in Y], [y[1] for y in Y]
    for m in range(3):
        st = [("wait", "sb%d" % i) for i in range(12)]
        for r in range(m, 12, 3):
            st += mds_row_stream(r, lo, hi_, xs[r], 24 + 12 * m, (K(4 * r), K(4 * r + 2)), F(1 + m), fd)
        streams.append(st)
    prologue = [("sload", K(0), 16, 0), ("sload", K(16), 16, 64), ("sload", K(32), 16, 128)]
    prog, nops = schedule(streams, prologue)
    return prog, [r for x in xs for r in x], nops


def build_partial_loop():
    """the body of the loop over the 22 fast partial rounds (one asm statement, pointer and counter in SGPRs):
    y = s0^7 + rc; s0' = 25 y + sum_j w_j s_j; s_j += y v_j.  Table row (48 dwords): w (k0, k1) x 11 | 2 pad | v (v0, v1) x 11 |
    rc lo, hi.  The w block of the NEXT row is fetched as soon as the dot products have read this row's, the v block at the
    top of the body (first used ~200 slots later), so no load latency is exposed."""
    fd = F(0)
    s0 = ("q0l", "q0h")
    sj = [("q%dl" % j, "q%dh" % j) for j in range(1, 12)]
    Y = (T(12), T(13))
    DB = 14
    sbox = sbox_stream(s0, 0, F(1), fd, out=Y)
    f = T(10)
    sbox += [("needconst",), ("mov", f, K(47)), ("add_co", Y[0], F(1), Y[0], K(46)), ("addc", Y[1], F(1), Y[1], f, F(1)),
             ("cnd", f, 0, -1, F(1)), ("add_co", Y[0], F(1), Y[0], f), ("addc", Y[1], fd, Y[1], 0, F(1)), ("signal", "y")]
    ks = [(K(2 * j), K(2 * j + 1)) for j in range(11)]
    dots = dot_streams(sj, ks, DB, [F(2), F(3), F(4), F(5)], fd, yterm=(Y, 25, "y"), needconst=False)
    for k, d in enumerate(dots):
        d.append(("signal", "dot%d" % k))
    rds = ["rd%d_%d" % (DB, q) for q in range(4)]
    prefetch = [("wait", e) for e in rds] + [("sload", K(0), 8, 192), ("sload", K(8), 16, 192 + 32)]
    fold = fold_stream(DB, s0, T(26), F(2), fd, ["dot0", "dot1", "dot2", "dot3"])
    usc = [((T(4), T(6), T(8)), T(10), F(1)), ((T(28), T(30), T(32)), T(34), F(6)), ((T(36), T(38), T(40)), T(42), F(7))]
    ups = [[], [], []]
    for j in range(11):
        sc, fj, fl = usc[j % 3]
        ups[j % 3] += update_stream(Y, (K(24 + 2 * j), K(25 + 2 * j)), sj[j], sc, fj, fl, fd, ["y"] + rds)
    pre = [("sload", K(0), 8, 0), ("sload", K(8), 16, 32)]
    top = [("waitcnt",), ("sload", K(24), 8, 96), ("sload", K(32), 16, 128)]
    prog, nops = schedule([sbox] + dots + [prefetch, fold] + ups, top)
    return pre, prog, list(s0) + [r for x in sj for r in x], nops


def build_init2(n_out):
    """n_out (1 or 2) outputs of the 11 x 11 initial matrix of the fast partial rounds: t_d = sum_{r=1..11} s_r init[r-1][d-1].
    Table row per output (24 dwords): (k0, k1) x 11 | 2 pad"""
    fd = F(0)
    sj = [("q%dl" % j, "q%dh" % j) for j in range(1, 12)]
    streams, outs = [], []
    for k in range(n_out):
        ks = [(K(24 * k + 2 * j), K(24 * k + 2 * j + 1)) for j in range(11)]
        dots = dot_streams(sj, ks, 14 * k, [F(1 + 4 * k + q) for q in range(4)], fd)
        for q, d in enumerate(dots):
            d.append(("signal", "d%d_%d" % (k, q)))
        o = ("t%dl" % k, "t%dh" % k)
        outs += list(o)
        streams += dots + [fold_stream(14 * k, o, T(14 * k + 12), F(1 + 4 * k), fd, ["d%d_%d" % (k, q) for q in range(4)])]
    prologue = [("sload", K(0), 8, 0), ("sload", K(8), 16, 32)] + ([("sload", K(24), 8, 96), ("sload", K(32), 16, 128)] if n_out == 2 else [])
    prog, nops = schedule(streams, prologue)
    return prog, outs, [r for x in sj for r in x], nops


# ------------------------------------------------------------------------------------------------------ self-test
def rnd64(rng):
    return rng.choice([rng.getrandbits(64), 2**64 - 1, P - 1, P, P + 1, 0, 1, 2**32 - 1, 2**32, 2**64 - 2**32, rng.getrandbits(32) << 32])


def get64(R, pair):
    return R[pair[0]] | (R[pair[1]] << 32)


def selftest(consts, tables):
    rng = random.Random(2024)
    # full rounds (every layer of constants)
    prog, ops, _ = build_fullround()
    check_hazards(prog)
    check_constant_bus(prog)
    for layer in range(8):
        for _ in range(25):
            st = [rnd64(rng) for _ in range(12)]
            regs = {}
            for i, x in enumerate(st):
                regs["x%dl" % i], regs["x%dh" % i] = x & M32, x >> 32
            R = simulate(prog, regs, tables["rc"][48 * layer:48 * layer + 48])
            y = [pow(x, 7, P) for x in st]
            for r in range(12):
                want = sum(MDS_C[i] * y[(i + r) % 12] for i in range(12)) + (8 * y[0] if r == 0 else 0) + consts["next"][layer][r]
                assert get64(R, ("x%dl" % r, "x%dh" % r)) % P == want % P, "full round"
    # the 22 partial rounds as the loop runs them (the w block of row i + 1 is fetched inside iteration i)
    pre, body, ops, _ = build_partial_loop()
    check_hazards(body + body)
    check_constant_bus(body)
    for _ in range(12):
        st = [rnd64(rng) for _ in range(12)]
        regs = {}
        for i, x in enumerate(st):
            regs["q%dl" % i], regs["q%dh" % i] = x & M32, x >> 32
        R = simulate(pre, regs, tables["partial"])
        cur = list(st)
        for rnd in range(22):
            R = simulate(body, R, tables["partial"][48 * rnd:])
            y = (pow(cur[0], 7, P) + consts["fp_rc"][rnd]) % P
            d = (25 * y + sum(consts["w"][rnd][j - 1] * cur[j] for j in range(1, 12))) % P
            cur = [d] + [(cur[j] + y * consts["v"][rnd][j - 1]) % P for j in range(1, 12)]
            for j in range(12):
                assert get64(R, ("q%dl" % j, "q%dh" % j)) % P == cur[j], "partial round %d word %d" % (rnd, j)
    # initial matrix
    for d0 in range(0, 11, 2):
        n_out = min(2, 11 - d0)
        prog, outs, ins, _ = build_init2(n_out)
        check_hazards(prog)
        check_constant_bus(prog)
        for _ in range(20):
            s = [rnd64(rng) for _ in range(12)]
            regs = {}
            for i in range(1, 12):
                regs["q%dl" % i], regs["q%dh" % i] = s[i] & M32, s[i] >> 32
            R = simulate(prog, regs, tables["init"][24 * d0:24 * d0 + 24 * n_out])
            for k in range(n_out):
                want = sum(s[r] * consts["init"][r - 1][d0 + k] for r in range(1, 12)) % P
                assert get64(R, ("t%dl" % k, "t%dh" % k)) % P == want, "init"
    return True


def load_constants():
    j = json.load(open(os.path.join(ROOT, "tests", "golden", "poseidon_goldilocks.json")))
    rc = j["all_round_constants"]
    first = j["fast_partial_first_round_constant"]
    # constant layer added after the MDS of full round k (k = 0..3 first half, 4..7 second half)
    nxt = [rc[12:24], rc[24:36], rc[36:48], first, rc[12 * 27:12 * 28], rc[12 * 28:12 * 29], rc[12 * 29:12 * 30], [0] * 12]
    consts = {"next": nxt, "fp_rc": j["fast_partial_round_constants"], "w": j["fast_partial_round_w_hats"],
              "v": j["fast_partial_round_vs"], "init": j["fast_partial_round_initial_matrix"], "rc0": rc[0:12], "rc26": rc[12 * 26:12 * 27]}
    t_rc = []
    for layer in nxt:
        for c in layer:
            t_rc += [c & M32, 0, c >> 32, 0]
    t_part = []
    for r in range(22):
        row = []
        for x in consts["w"][r]:
            row += [x & M32, x >> 32]
        row += [0, 0]
        for x in consts["v"][r]:
            row += [x & M32, x >> 32]
        row += [consts["fp_rc"][r] & M32, consts["fp_rc"][r] >> 32]
        assert len(row) == 48
        t_part += row
    t_part += [0] * 48            # the loop's last iteration prefetches one row past the end
    t_init = []
    for d in range(11):
        row = []
        for r in range(1, 12):
            x = consts["init"][r - 1][d]
            row += [x & M32, x >> 32]
        row += [0, 0]
        t_init += row
    return consts, {"rc": t_rc, "partial": t_part, "init": t_init}


def c_table(name, vals):
    lines = ["ZKLC_CONST_ARRAY u32 %s[%d] = {" % (name, len(vals))]
    for i in range(0, len(vals), 8):
        lines.append("    " + ", ".join("0x%08xu" % v for v in vals[i:i + 8]) + ",")
    lines.append("};")
    return "\n".join(lines)


def main():
    consts, tables = load_constants()
    selftest(consts, tables)
    if "--check" in sys.argv:
        print("gen_poseidon_asm: simulator self-test OK")
        return
    parts = ["// GENERATED by tools/gen_poseidon_asm.py -- do not edit.  Hand-scheduled gfx950 statements of the Poseidon-Goldilocks",
             "// permutation (see the generator for the derivation, the hazard rule and the simulator that checks every list).",
             "// Tables: PGL_ASM_RC[layer][row] = {lo, 0, hi, 0} of the constant added after the MDS of full round `layer`;",
             "// PGL_ASM_PARTIAL[round] = w_hat pairs | pad | v pairs | rc (48 dwords, + one padding row); PGL_ASM_INIT[d] = column d of the initial",
             "// matrix as (lo, hi) pairs | pad (24 dwords).",
             c_table("PGL_ASM_RC", tables["rc"]), c_table("PGL_ASM_PARTIAL", tables["partial"]), c_table("PGL_ASM_INIT", tables["init"]),
             "#if defined(__HIP_DEVICE_COMPILE__)"]
    stats = []
    prog, ops, nops = build_fullround()
    stats.append(("full round", len(prog), nops))
    parts.append(statement("pgl_asm_full_round", prog, ops, [], ptr=True,
                           comment="one full round in place: x^7 on the twelve words, MDS, next constant layer (tab: 48 dwords of PGL_ASM_RC)"))
    pre, prog, ops, nops = build_partial_loop()
    stats.append(("partial round", len(prog), nops))
    parts.append(statement("pgl_asm_partial_rounds", prog, ops, [], ptr=True, loop=(22, 192), pre=pre,
                           comment="the 22 fast partial rounds (tab: PGL_ASM_PARTIAL)"))
    for n_out in (2, 1):
        prog, outs, ins, nops = build_init2(n_out)
        stats.append(("init x%d" % n_out, len(prog), nops))
        parts.append(statement("pgl_asm_init%d" % n_out, prog, [], ins, outs=outs, ptr=True,
                               comment="%d output(s) of the initial matrix of the fast partial rounds (tab: %d dwords of PGL_ASM_INIT)" % (n_out, 24 * n_out)))
    parts.append("#endif")
    total = 8 * stats[0][1] + 22 * stats[1][1] + 5 * stats[2][1] + stats[3][1]
    parts.insert(5, "// instructions per statement (of which s_nop): " + "; ".join("%s %d (%d)" % s for s in stats)
                 + "; permutation ~%d + constant layers / canonicalisation" % total)
    open(os.path.join(ROOT, "zk-light-client-implementation_amd", "csrc", "poseidon_gl_asm.inc"), "w").write("\n".join(parts) + "\n")
    print("generated poseidon_gl_asm.inc:", stats, "total ~", total)


if __name__ == "__main__":
    main()
