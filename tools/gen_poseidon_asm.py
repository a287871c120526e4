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

VB, NT = 48, 64          # fixed temporary VGPRs v[48:111]
LB = 32                  # s[32:33] table pointer, s34 round counter of the looped statement, s35 = 25 * 2^10
SB = 36                  # constants s[36:83]: two buffers of four multiply-accumulate slots (6 dwords each)
FB, NF = 84, 7           # flag pairs s[84:85] .. s[96:97]; flag index NF = vcc
S25K, S22, S12, S10 = "s35", "s98", "s99", "s100"      # 25 << 10, 1 << 22, 1 << 12, 1 << 10 (set by the statements that use them)
ONE = "v%d" % (VB + NT - 1)                            # a VGPR holding 1


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


_XUID = [0]


def xflag():
    """a fresh TEMPORARY flag: written once, read once by an s_andn2; the scheduler binds it to a free flag pair of its pool at the
    writer and releases the pair at the reader"""
    _XUID[0] += 1
    return "X%d" % _XUID[0]


def reduce128(LO, w2, w3, w4, TP, out, f, fl, fd):
    """out = {LO} + w2 * eps - {w4, w3}  (eps = 2^32 - 1; 2^64 = eps, 2^96 = -1, 2^128 = -2^32 mod p); TP: scratch pair.
    Round 6: the two conditional corrections (a borrow means the wrapped value is eps too large, a carry that it is eps too small)
    no longer build a 0 / -1 mask with v_cndmask: x - eps b = (x.lo + b, x.hi - (b & ~c)) with c the carry of the low add, and the
    AND-NOT of the two flags is ONE s_andn2_b64 on the SCALAR unit -- seven VALU instructions instead of nine (the kernel is VALU
    issue-bound; the scalar unit idles)."""
    x1, x2 = xflag(), xflag()
    return [
        ("sub_co", LO, fl, LO, w3),
        ("subb", hi(LO), fl, hi(LO), w4 if w4 is not None else 0, fl),      # borrow b: wrapped by 2^64 = eps too large
        ("addc", LO, x1, LO, 0, fl),                   # low word + b, carry c
        ("sandn2", fl, fl, x1),                        # b & ~c
        ("subb", hi(LO), fd, hi(LO), 0, fl),           # high word - (b & ~c): together - eps b
        ("mad", TP, fl, w2, -1, LO),                   # carry c1: wrapped, eps too small
        ("subb", out[0], x2, TP, 0, fl),               # low word - c1, borrow b2
        ("sandn2", fl, fl, x2),                        # c1 & ~b2
        ("addc", out[1], fd, hi(TP), 0, fl),           # high word + (c1 & ~b2): together + eps c1
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
    x1 = xflag()
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
        ("add_co", hi(Al), fl, hi(Al), Ah),            # + sh.lo 2^32, carry c: + eps (as in reduce128)
        ("subb", out[0], x1, Al, 0, fl),
        ("sandn2", fl, fl, x1),
        ("addc", out[1], fd, hi(Al), 0, fl),
    ]
    return ins


def limbs22(k):
    return [k & 0x3FFFFF, (k >> 22) & 0x3FFFFF, k >> 44]


class MacStream:
    """A sequence of 64 x 64-bit multiply-accumulates x * k (k a constant of the table) WITHOUT carries: k and k' = 2^32 k mod p
    are split into limbs of 22 bits (ka, kb, kc), so that  x k = x0 k + x1 k'  (mod p)  is six v_mad_u64_u32 into three 64-bit
    column accumulators of weights 1, 2^22, 2^44 -- every product is below 2^54 and a column takes hundreds of them before it
    could overflow.  The constants travel through two SGPR buffers of four slots (6 dwords each): a buffer is fetched while the
    other one is used (`s_waitcnt` + the next `s_load` at the head of every group of four slots).  The table is built here in
    the order the stream consumes it."""

    def __init__(self, fd):
        self.ins, self.table, self.n, self.fd = [], [], 0, fd

    def _slot(self, dwords):
        if self.n % 4 == 0:
            g = self.n // 4
            nb = 24 * ((g + 1) % 2)
            self.ins += [("waitcnt",), ("sload", K(nb), 16, 96 * (g + 1)), ("sload", K(nb + 16), 8, 96 * (g + 1) + 64)]
        base = 24 * ((self.n // 4) % 2) + 6 * (self.n % 4)
        self.table += dwords
        self.n += 1
        return base

    def const(self, acc, k):
        """acc = k (initialises the three columns)"""
        b = self._slot(limbs22(k % P) + [0, 0, 0])
        self.ins += [("mad", acc[c], self.fd, ONE, K(b + c), None) for c in range(3)]

    def mac(self, acc, x, k):
        k %= P
        b = self._slot(limbs22(k) + limbs22((k << 32) % P))
        self.ins += [("mad", acc[c], self.fd, x[h], K(b + 3 * h + c), acc[c]) for h in range(2) for c in range(3)]

    def raw(self, *ins):
        self.ins += list(ins)

    def finish(self, even_groups=True):
        """pads the table to whole groups (an even number of them for a looped body: the fetch at the head of the last group
        then brings the NEXT block's first group into buffer 0) plus the one group the last fetch reads"""
        while self.n % 4 or (even_groups and (self.n // 4) % 2):
            self.table += [0] * 6
            self.n += 1
        return self.n // 4


def prefetch0(off=0):
    return [("sload", K(0), 16, off), ("sload", K(16), 8, off + 64)]


def fold3(acc, TP, f, out, fl, fd):
    """the value C0 + C1 2^22 + C2 2^44 of three column accumulators (each < 2^60) -> out (loose 64 bits).  Scratch: TP (pair), f"""
    C0, C1, C2 = acc
    return [
        ("mad", C0, fd, C1, S22, C0),                  # A = C0 + C1.lo 2^22                      (< 2^61)
        ("mad", TP, fd, hi(C1), S22, None),            # B = C1.hi 2^22            weight 2^32   (< 2^51)
        ("mad", TP, fd, C2, S12, TP),                  # B += C2.lo 2^12
        ("mad", C1, fd, hi(C2), S12, None),            # D = C2.hi 2^12            weight 2^64   (< 2^41)
        ("add_co", hi(C0), fl, hi(C0), TP),            # w1
        ("addc", C1, fl, C1, hi(TP), fl),              # w2
        ("addc", hi(C1), fd, hi(C1), 0, fl),           # w3
    ] + reduce128(C0, C1, hi(C1), None, TP, out, f, fl, fd)


# ------------------------------------------------------------------------------------------------------- scheduler
def flags_read(ins):
    op = ins[0]
    if op in ("addc", "subb"):
        return [ins[5]]
    if op == "cnd":
        return [ins[4]]
    if op == "sandn2":
        return [ins[2], ins[3]]
    return []


def flag_written(ins):
    if ins[0] == "sandn2":
        return ins[1]
    return ins[2] if ins[0] in ("mad", "add_co", "addc", "sub_co", "subb") else None


def bind_flags(ins, binding):
    """the instruction with its temporary flags (xflag) replaced by the flag pairs they are bound to"""
    return tuple(binding.get(x, x) if isinstance(x, str) and x[0] == "X" else x for x in ins)


def schedule(streams, prologue=(), priority=False, pool=()):
    """greedy merge (round-robin, or always the first stream that can issue when priority=True: later streams are filler);
    a flag reader is placed >= 3 slots after the flag's writer.  `pool`: the flag pairs no stream owns; a temporary flag (xflag)
    takes one of them when its writer is placed and gives it back when its reader (the s_andn2) is -- a writer that finds the pool
    empty waits like a hazard."""
    out = list(prologue)
    pos = len(out)
    lastw, events = {}, set()
    free, binding = list(pool), {}
    heads = [0] * len(streams)
    waited = False
    rr = nops = 0
    while any(h < len(s) for h, s in zip(heads, streams)):
        chosen = None
        progressed = False
        for k in range(len(streams)):
            i = (rr + k) % len(streams)
            if priority:
                i = k
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
            w_ = flag_written(ins)
            if w_ is not None and w_[0] == "X" and w_ not in binding and not free:
                assert binding, "schedule(): a temporary flag is needed and the pool is empty (pass pool=[...])"
                continue                        # no flag pair free for its temporary: another stream goes first
            if all(pos - lastw.get(binding.get(f, f), -9) >= 3 for f in flags_read(ins)):
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
        w = flag_written(ins)
        if w is not None and w[0] == "X" and w not in binding:
            binding[w] = free.pop(0)
        released = [f for f in flags_read(ins) if f[0] == "X"] if ins[0] == "sandn2" else []
        ins = bind_flags(ins, binding)
        for f in released:
            free.append(binding.pop(f))
        out.append(ins)
        w = flag_written(ins)
        if w is not None:
            lastw[w] = pos
        heads[chosen] += 1
        pos += 1
        rr = 0 if priority else (chosen + 1) % len(streams)
    return out, nops


def check_hazards(prog):
    lastw = {}
    for pos, ins in enumerate(prog):
        for f in flags_read(ins):
            assert pos - lastw.get(f, -9) >= 3, ("flag hazard", pos, ins)
        assert not any(isinstance(x, str) and x[0] == "X" for x in ins), ("unbound temporary flag", ins)
        w = flag_written(ins) if ins[0] not in ("nop", "waitcnt", "sload", "smov") else None
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
        elif op == "sandn2":
            FL[ins[1]] = FL[ins[2]] & (1 - FL[ins[3]])
        elif op == "add64":
            v = (val64(ins[2]) + val64(ins[3])) & (2**64 - 1)
            R[ins[1]], R[hi(ins[1])] = v & M32, v >> 32
        elif op in ("mov", "smov"):
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
    if op == "sandn2":
        return "s_andn2_b64 %s, %s, %s" % (fmt_flag(ins[1]), fmt_flag(ins[2]), fmt_flag(ins[3]))
    if op == "add64":
        return "v_lshl_add_u64 %s, %s, 0, %s" % (fmt_pair(ins[1]), fmt_pair(ins[2]), fmt_pair(ins[3]))
    if op == "mov":
        return "v_mov_b32_e32 %s, %s" % (ins[1], fmt_src(ins[2]))
    if op == "smov":
        return "s_mov_b32 %s, 0x%x" % (ins[1], ins[2])
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
        if ins[0] in ("nop", "waitcnt", "sload", "smov", "sandn2"):
            continue
        srcs = [x for x in ins[3:] if isinstance(x, str)] if ins[0] != "cnd" else [x for x in ins[2:] if isinstance(x, str)]
        if ins[0] == "add64":
            srcs = [ins[2], ins[3]]
        if ins[0] == "mov":
            srcs = [ins[2]] if isinstance(ins[2], str) else []
        sg = set(x for x in srcs if x[0] in "sF")
        assert len(sg) <= 1, ("constant bus", ins)


CLOBBERS = ["v%d" % (VB + i) for i in range(NT)] + ["s%d" % i for i in range(LB, FB + 2 * NF)] + ["s98", "s99", "s100", "vcc", "scc"]


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
    if any(i[0] == "sload" for i in list(prog) + list(pre)):
        lines.append("s_waitcnt lgkmcnt(0)")       # a prefetch may still be in flight: its SGPRs are ours until it lands
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
    prog, nops = schedule(streams, prologue, pool=[F(4), F(5), F(6), F(7)])
    return prog, [r for x in xs for r in x], nops


STMT_PRE = [("smov", S25K, 25 << 10), ("smov", S22, 1 << 22), ("smov", S12, 1 << 12), ("smov", S10, 1 << 10), ("mov", ONE, 1)]


def build_fullround_init(cs):
    """the LAST full round of the first half merged with the dense 11 x 11 initial matrix of the fast partial rounds: twelve
    S-boxes, then out = G y + g with G = diag(1, Init^T) MDS (64-bit constants, carry-free multiply-accumulates in groups of
    four outputs) and g = diag(1, Init^T) first_round_constants.  Returns (program, operands, nops, table)."""
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
    D = MacStream(fd)
    D.raw(*[("wait", "sb%d" % i) for i in range(12)])
    accs = [(T(24 + 6 * t), T(26 + 6 * t), T(28 + 6 * t)) for t in range(4)]
    folds = [[] for _ in range(4)]
    for g in range(3):
        rows = list(range(4 * g, 4 * g + 4))
        if g:
            D.raw(*[("wait", "f%d_%d" % (g - 1, t)) for t in range(4)])
        for t, r in enumerate(rows):
            D.const(accs[t], cs["g"][r])
        for i in range(12):
            for t, r in enumerate(rows):
                D.mac(accs[t], Y[i], cs["G"][r][i])
        D.raw(("signal", "m%d" % g))
        for t, r in enumerate(rows):
            folds[t] += [("wait", "m%d" % g)] + fold3(accs[t], T(48 + 4 * t), T(50 + 4 * t), xs[r], F(1 + t), fd) + [("signal", "f%d_%d" % (g, t))]
    D.finish(even_groups=False)
    prog, nops = schedule(streams + [D.ins] + folds, STMT_PRE + prefetch0(), pool=[F(5), F(6), F(7)])
    return prog, [r for x in xs for r in x], nops, D.table + [0] * 24


def build_partial_block(cs, b):
    """eleven fast partial rounds (block b = 0, 1) in the LAZY form: with u = the state words 1..11 at the start of the block and
    z_k = s0^7 of round k,   s0 <- 25 z_q + K_q + sum_i u_i w_q[i] + sum_{k<q} z_k c_q[k]   (c_q[k] = <v_k, w_q>, K_q collects
    the round constants), and the words 1..11 are only materialised at the end of the block: u_j += Kv_j + sum_k z_k v_k[j].
    Every product is a carry-free multiply-accumulate (MacStream): the S-box -> fold chain of a round has priority and the
    dot products of the NEXT round (double-buffered accumulators) fill its hazard slots.  Returns (body, operands, nops, table)."""
    fd = F(0)
    s0 = ("q0l", "q0h")
    u = [("q%dl" % j, "q%dh" % j) for j in range(1, 12)]
    Z = [(T(2 * k), T(2 * k + 1)) for k in range(11)]
    ACC = [(T(34), T(36), T(38)), (T(40), T(42), T(44))]
    B = cs["blocks"][b]
    chain = []
    D = MacStream(fd)
    for q in range(11):
        acc = ACC[q % 2]
        chain += sbox_stream(s0, 22, F(1), fd, out=Z[q]) + [("signal", "z%d" % q), ("wait", "dot%d" % q),
                                                            ("mad", acc[0], fd, Z[q][0], 25, acc[0]),
                                                            ("mad", acc[1], fd, Z[q][1], S25K, acc[1])]
        chain += fold3(acc, T(46), T(32), s0, F(1), fd) + [("signal", "fold%d" % q)]
        if q >= 2:
            D.raw(("wait", "fold%d" % (q - 2)))
        D.const(acc, B["K"][q])
        for i in range(11):
            D.mac(acc, u[i], B["w"][q][i])
        for k in range(q):
            D.raw(("wait", "z%d" % k))
            D.mac(acc, Z[k], B["c"][q][k])
        D.raw(("signal", "dot%d" % q))
    accs = [(T(22 + 6 * t), T(24 + 6 * t), T(26 + 6 * t)) for t in range(4)]
    folds = [[] for _ in range(4)]
    groups = [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10]]
    for g, js in enumerate(groups):
        D.raw(("wait", "fold10"), *[("wait", "mf%d_%d" % (g - 1, t)) for t in range(len(groups[g - 1]) if g else 0)])
        for t, j in enumerate(js):
            D.const(accs[t], B["Kv"][j])
            D.raw(("mad", accs[t][0], fd, u[j][0], 1, accs[t][0]), ("mad", accs[t][1], fd, u[j][1], S10, accs[t][1]))
        for k in range(11):
            for t, j in enumerate(js):
                D.mac(accs[t], Z[k], B["v"][k][j])
        D.raw(("signal", "mat%d" % g))
        for t, j in enumerate(js):
            folds[t] += [("wait", "mat%d" % g)] + fold3(accs[t], T(46 + 4 * t), T(48 + 4 * t), u[j], F(2 + t), fd) + [("signal", "mf%d_%d" % (g, t))]
    n_groups = D.finish(even_groups=True)
    prog, nops = schedule([chain, D.ins] + folds, priority=True, pool=[F(6), F(7)])
    return prog, list(s0) + [r for x in u for r in x], nops, D.table, n_groups


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
    # the fourth full round merged with the initial matrix of the partial rounds, against the two separate layers
    prog, ops, _, tab = build_fullround_init(consts)
    assert tab == tables["finit"]
    check_hazards(prog)
    check_constant_bus(prog)
    for _ in range(40):
        st = [rnd64(rng) for _ in range(12)]
        regs = {}
        for i, x in enumerate(st):
            regs["x%dl" % i], regs["x%dh" % i] = x & M32, x >> 32
        R = simulate(prog, regs, tab)
        y = [pow(x, 7, P) for x in st]
        t = [(sum(MDS_C[i] * y[(i + r) % 12] for i in range(12)) + (8 * y[0] if r == 0 else 0) + consts["next"][3][r]) % P for r in range(12)]
        want = [t[0]] + [sum(t[r] * consts["init"][r - 1][d - 1] for r in range(1, 12)) % P for d in range(1, 12)]
        for r in range(12):
            assert get64(R, ("x%dl" % r, "x%dh" % r)) % P == want[r], "full round + initial matrix, word %d" % r
    # the 22 partial rounds as the loop runs them (two blocks of eleven; the constant layer that follows is folded in)
    bodies = [build_partial_block(consts, b) for b in range(2)]
    strip = lambda prog: [tuple(x for x in ins) for ins in prog]
    assert strip(bodies[0][0]) == strip(bodies[1][0]), "the two blocks must share one loop body"
    body, n_groups = bodies[0][0], bodies[0][4]
    assert bodies[0][3] + bodies[1][3] + [0] * 24 == tables["pblocks"] and len(bodies[0][3]) == 24 * n_groups
    check_hazards(body + body)
    check_constant_bus(body)
    pre = STMT_PRE + prefetch0()
    for _ in range(12):
        st = [rnd64(rng) for _ in range(12)]
        regs = {}
        for i, x in enumerate(st):
            regs["q%dl" % i], regs["q%dh" % i] = x & M32, x >> 32
        R = simulate(pre, regs, tables["pblocks"])
        cur = list(st)
        for b in range(2):
            R = simulate(body, R, tables["pblocks"][24 * n_groups * b:])
            for rnd in range(11 * b, 11 * b + 11):
                y = (pow(cur[0], 7, P) + consts["fp_rc"][rnd]) % P
                d = (25 * y + sum(consts["w"][rnd][j - 1] * cur[j] for j in range(1, 12))) % P
                cur = [d] + [(cur[j] + y * consts["v"][rnd][j - 1]) % P for j in range(1, 12)]
            want = cur if b == 0 else [(x + c) % P for x, c in zip(cur, consts["rc26"])]
            for j in range(12):
                assert get64(R, ("q%dl" % j, "q%dh" % j)) % P == want[j], "partial block %d word %d" % (b, j)
    return True


def load_constants():
    j = json.load(open(os.path.join(ROOT, "tests", "golden", "poseidon_goldilocks.json")))
    rc = j["all_round_constants"]
    first = j["fast_partial_first_round_constant"]
    # constant layer added after the MDS of full round k (k = 0..3 first half, 4..7 second half)
    nxt = [rc[12:24], rc[24:36], rc[36:48], first, rc[12 * 27:12 * 28], rc[12 * 28:12 * 29], rc[12 * 29:12 * 30], [0] * 12]
    consts = {"next": nxt, "fp_rc": j["fast_partial_round_constants"], "w": j["fast_partial_round_w_hats"],
              "v": j["fast_partial_round_vs"], "init": j["fast_partial_round_initial_matrix"], "rc0": rc[0:12], "rc26": rc[12 * 26:12 * 27]}
    # full round 3 + initial matrix: out = E (MDS y + first), E = diag(1, Init^T)
    mds = [[0] * 12 for _ in range(12)]
    for r in range(12):
        for i in range(12):
            mds[r][(i + r) % 12] += MDS_C[i]
    mds[0][0] += 8
    E = [[0] * 12 for _ in range(12)]
    E[0][0] = 1
    for d in range(1, 12):
        for r in range(1, 12):
            E[d][r] = consts["init"][r - 1][d - 1]
    consts["G"] = [[sum(E[d][r] * mds[r][i] for r in range(12)) % P for i in range(12)] for d in range(12)]
    consts["g"] = [sum(E[d][r] * first[r] for r in range(12)) % P for d in range(12)]
    # the lazy blocks of eleven partial rounds
    consts["blocks"] = []
    for b in range(2):
        w = [consts["w"][11 * b + q] for q in range(11)]
        v = [consts["v"][11 * b + k] for k in range(11)]
        rcs = [consts["fp_rc"][11 * b + k] for k in range(11)]
        c = [[sum(v[k][j] * w[q][j] for j in range(11)) % P for k in range(q)] for q in range(11)]
        Kq = [(25 * rcs[q] + sum(rcs[k] * c[q][k] for k in range(q))) % P for q in range(11)]
        Kv = [sum(rcs[k] * v[k][j] for k in range(11)) % P for j in range(11)]
        if b == 1:                 # the constant layer in front of the second half of the full rounds
            Kq[10] = (Kq[10] + consts["rc26"][0]) % P
            Kv = [(x + consts["rc26"][j + 1]) % P for j, x in enumerate(Kv)]
        consts["blocks"].append({"w": w, "v": v, "c": c, "K": Kq, "Kv": Kv})
    # the same lazy blocks as plain tables for the C++ evaluator of the Poseidon GATE (plonky2_gates.cuh: the S-box inputs of the
    # partial rounds are wires there, so the rounds of a block are independent): limbs22 slots in natural order
    def slot(k):
        k %= P
        return limbs22(k) + limbs22((k << 32) % P)
    lz = {"w": [], "c": [], "v": [], "k": [], "kv": []}
    for blk in consts["blocks"]:
        for q in range(11):
            for i in range(11):
                lz["w"] += slot(blk["w"][q][i])
            for k in range(11):
                lz["c"] += slot(blk["c"][q][k]) if k < q else [0] * 6
            lz["k"] += limbs22(blk["K"][q] % P)
        for j in range(11):                      # [j][k]: the eleven slots one output word needs are consecutive
            for k in range(11):
                lz["v"] += slot(blk["v"][k][j])
        for j in range(11):
            lz["kv"] += limbs22(blk["Kv"][j] % P)
    consts["lazy_tables"] = lz
    t_rc = []
    for layer in nxt:
        for c in layer:
            t_rc += [c & M32, 0, c >> 32, 0]
    t_finit = build_fullround_init(consts)[3]
    t_blocks = build_partial_block(consts, 0)[3] + build_partial_block(consts, 1)[3] + [0] * 24
    return consts, {"rc": t_rc, "finit": t_finit, "pblocks": t_blocks}


def c_table(name, vals):
    lines = ["ZKLC_CONST_ARRAY u32 %s[%d] = {" % (name, len(vals))]
    for i in range(0, len(vals), 8):
        lines.append("    " + ", ".join("0x%08xu" % v for v in vals[i:i + 8]) + ",")
    lines.append("};")
    return "\n".join(lines)


def render(consts=None, tables=None):
    """-> (text of csrc/poseidon_gl_asm.inc, per-statement statistics, instructions per permutation); writes nothing"""
    if consts is None:
        consts, tables = load_constants()
    parts = ["// GENERATED by tools/gen_poseidon_asm.py -- do not edit.  Hand-scheduled gfx950 statements of the Poseidon-Goldilocks",
             "// permutation (see the generator for the derivation, the hazard rule and the simulator that checks every list).",
             "// Tables: PGL_ASM_RC[layer][row] = {lo, 0, hi, 0} of the constant added after the MDS of full round `layer`;",
             "// PGL_ASM_FINIT / PGL_ASM_PBLOCKS: multiply-accumulate slots {ka, kb, kc, k'a, k'b, k'c} (22-bit limbs of k and of 2^32 k mod p)",
             "// in the order the statements consume them, four slots per fetch (generator: MacStream).",
             c_table("PGL_ASM_RC", tables["rc"]), c_table("PGL_ASM_FINIT", tables["finit"]), c_table("PGL_ASM_PBLOCKS", tables["pblocks"]),
             "// The lazy blocks of the partial rounds once more as plain tables (C++ evaluator of the Poseidon gate): slot = {ka, kb, kc, k'a, k'b, k'c};",
             "// PGL_LAZY_W[b][q][i], PGL_LAZY_C[b][q][k] (k < q, zero otherwise), PGL_LAZY_V[b][j][k]: 11 x 11 slots per block b;",
             "// PGL_LAZY_K[b][q], PGL_LAZY_KV[b][j]: three 22-bit limbs of the collected constants.",
             c_table("PGL_LAZY_W", consts["lazy_tables"]["w"]), c_table("PGL_LAZY_C", consts["lazy_tables"]["c"]),
             c_table("PGL_LAZY_V", consts["lazy_tables"]["v"]), c_table("PGL_LAZY_K", consts["lazy_tables"]["k"]),
             c_table("PGL_LAZY_KV", consts["lazy_tables"]["kv"]),
             "#if defined(__HIP_DEVICE_COMPILE__)"]
    stats = []
    prog, ops, nops = build_fullround()
    stats.append(("full round", len(prog), nops))
    parts.append(statement("pgl_asm_full_round", prog, ops, [], ptr=True,
                           comment="one full round in place: x^7 on the twelve words, MDS, next constant layer (tab: 48 dwords of PGL_ASM_RC)"))
    prog, ops, nops, _ = build_fullround_init(consts)
    stats.append(("full round + initial matrix", len(prog), nops))
    parts.append(statement("pgl_asm_full_round_init", prog, ops, [], ptr=True,
                           comment="the fourth full round merged with the initial matrix of the fast partial rounds (tab: PGL_ASM_FINIT)"))
    prog, ops, nops, tab, n_groups = build_partial_block(consts, 0)
    stats.append(("block of 11 partial rounds", len(prog), nops))
    parts.append(statement("pgl_asm_partial_rounds", prog, ops, [], ptr=True, loop=(2, 96 * n_groups), pre=STMT_PRE + prefetch0(),
                           comment="the 22 fast partial rounds as two lazy blocks + the constant layer that follows (tab: PGL_ASM_PBLOCKS)"))
    parts.append("#endif")
    total = 7 * stats[0][1] + stats[1][1] + 2 * stats[2][1]
    parts.insert(5, "// instructions per statement (of which s_nop): " + "; ".join("%s %d (%d)" % s for s in stats)
                 + "; permutation ~%d + first constant layer / canonicalisation" % total)
    return "\n".join(parts) + "\n", stats, total


INC_PATH = os.path.join(ROOT, "zk-light-client-implementation_amd", "csrc", "poseidon_gl_asm.inc")


def budget(consts):
    """Lane-instructions of ONE permutation by component, against a stated floor.  Counted on the very
    instruction lists the statements are emitted from.  Issue classes from profiles/r05a_valu_ubench_warm_sclk.txt: every
    instruction of these lists except v_mov is of the 4-cycle class (v_mad_u64_u32, carry adds / subtracts, v_cndmask with an SGPR
    mask, v_lshl_add_u64), so the count IS the issue time."""
    fd = F(0)
    x, y = ("xl", "xh"), (T(0), T(1))
    valu = lambda lst: sum(1 for i in lst if i[0] not in ("sandn2", "needconst"))
    n_mul = valu(mulmod(x, x, y, (T(4), T(6), T(8)), T(10), F(1), fd))
    n_red = valu(reduce128(T(4), T(8), T(9), None, T(6), y, T(10), F(1), fd))
    n_sbox = valu(sbox_stream(x, 22, F(1), fd, out=y))
    lo, hi_ = [T(2 * i) for i in range(12)], [T(2 * i + 1) for i in range(12)]
    n_rows = [valu(mds_row_stream(r, lo, hi_, x, 24, (K(0), K(2)), F(1), fd)) for r in range(12)]
    n_fold = valu(fold3((T(34), T(36), T(38)), T(46), T(32), x, F(1), fd))
    prog_f, _, nop_f = build_fullround()
    prog_i, _, nop_i, _ = build_fullround_init(consts)
    prog_p, _, nop_p, _, _ = build_partial_block(consts, 0)
    count = lambda prog, ops: sum(1 for i in prog if i[0] in ops)
    sc = ("sload", "waitcnt", "smov", "nop", "sandn2")
    rows = []
    full_sbox = 8 * 12 * n_sbox
    full_mds = 7 * sum(n_rows)
    init_dense = len(prog_i) - 12 * n_sbox - count(prog_i, sc)
    part_sbox = 22 * n_sbox
    part_fold = 22 * (n_fold + 2) + 22 * n_fold              # per round: 25 z + K (2 mads) and the fold; per block: 11 words folded
    part_scalar = 2 * count(prog_p, sc)
    part_lin = 2 * len(prog_p) - part_sbox - part_fold - part_scalar
    scalar = 7 * count(prog_f, sc) + count(prog_i, sc) + part_scalar
    total = 7 * len(prog_f) + len(prog_i) + 2 * len(prog_p)
    # floor: 4 v_mad_u64_u32 + 2 carry adds per 64 x 64 product and 8 instructions per reduction, a mad per MDS term
    # (12 x 12 x 2 halves) + 4 to fold a row, 6 mads per product with a 64-bit table constant (22-bit limbs, carry-free), 3 per
    # constant, 14 per fold of three columns
    f_mul = 4 + 2 + 8
    f_sbox = 4 * f_mul
    f_row = 24 + 4
    lazy_macs = 2 * (sum(11 + q for q in range(11)) + 11 * 11)
    floor = {"sbox_full": 96 * f_sbox, "mds_full": 7 * 12 * f_row, "init": 144 * 6 + 12 * 3 + 12 * 14, "sbox_part": 22 * f_sbox,
             "lin_part": lazy_macs * 6 + 44 * 3 + 44, "fold_part": 44 * 14 + 44}
    out = ["Poseidon-Goldilocks permutation on gfx950: lane-instructions by component (tools/gen_poseidon_asm.py --budget)",
           "VALU: mulmod %d instructions (4 v_mad_u64_u32 + 4 carry adds + %d reduction), x^7 = 4 mulmod = %d, MDS row %s, 3-column fold %d" %
           (n_mul, n_red, n_sbox, sorted(set(n_rows)), n_fold), "",
           "%-58s %8s %7s %8s" % ("component", "emitted", "share", "floor")]
    n_andn2 = 7 * count(prog_f, ("sandn2",)) + count(prog_i, ("sandn2",)) + 2 * count(prog_p, ("sandn2",))
    comp = [("S-boxes of the 8 full rounds (96 x x^7)", full_sbox, floor["sbox_full"]),
            ("MDS + next constants of 7 full rounds (84 rows)", full_mds, floor["mds_full"]),
            ("4th full round's MDS merged with the 11x11 initial matrix", init_dense, floor["init"]),
            ("S-boxes of the 22 partial rounds", part_sbox, floor["sbox_part"]),
            ("partial rounds, lazy linear layer (dot products + materialisation)", part_lin - 0, floor["lin_part"]),
            ("partial rounds, 25 z + K and the column folds", part_fold, floor["fold_part"]),
            ("scalar unit: constant fetches, waits, s_nop, %d s_andn2" % n_andn2, scalar, 0)]
    for name, n, fl in comp:
        out.append("%-58s %8d %6.1f%% %8s" % (name[:58], n, 100.0 * n / total, fl if fl else "-"))
    fsum = sum(fl for _, _, fl in comp)
    out += ["%-58s %8d %6.1f%% %8d" % ("total (statements; + first constant layer / canonicalisation)", total, 100.0, fsum), "",
            "floor model: 64 x 64 product = 4 mad + 2 carry adds, reduction = 8, MDS term = 1 mad per 32-bit half, fold of an MDS row = 4,",
            "constant product = 6 mads (22-bit limbs), fold of three columns = 14.  VALU emitted / floor = %.3f." % ((total - scalar) / fsum),
            "Round 6: the conditional +-eps corrections of every reduction use the scalar unit (x - eps b = (lo + b, hi - (b & ~c)): one",
            "s_andn2_b64 of two carry flags replaces a v_cndmask and folds an add): reduction 9 -> 7 VALU, mulmod 17 -> %d, MDS row 31 -> %d," % (n_mul, min(n_rows)),
            "fold 16 -> %d; VALU instructions per permutation %d (round 5: 16 223), %d s_nop in hazard slots the scheduler could not fill." %
            (n_fold, total - scalar, 7 * nop_f + nop_i + 2 * nop_p), "",
            "FFT-structured MDS (plonky2 `mds_multiply_freq`), costed on this ISA instead of built:",
            "  per 12-vector of 32-bit halves: 3 x fft4 (18 add/sub) + blocks (9 + 27 + 9 = 45 multiplications, ~57 add/sub) + 3 x ifft4 (~30",
            "  add/sub, shifts) = 45 mult + ~105 add/sub; two halves -> 90 mult + 210 add/sub.  The intermediate values exceed 32 bits (35-42),",
            "  so every add/sub is a 64-bit operation: v_lshl_add_u64 (1 instruction, 4-cycle class) for sums, v_sub_co + v_subb (2) for",
            "  differences -> ~90 + 105 + 2 x 105 = ~405 issue slots + the same 12 x 4 row folds, against 288 mads + 84 today (%d): the" % sum(n_rows),
            "  multiply-accumulate is ONE issue slot here and the MDS entries are 6-bit, so trading multiplications for additions loses.",
            "  (v_add3_u32 / v_mad_u32_u24 / v_alignbit are all in the same 4.3-cycle class as v_mad_u64_u32: no cheaper 3-operand form exists.)",
            "  Halving the circulant over x^12 - 1 = (x^6 - 1)(x^6 + 1) (the half-sums and half-differences of the MDS row are integers)",
            "  needs 33-bit operands for the 32 x 32 multiplier or 22-bit limbs (3 x 72 mads + 36 single-pass adds + recombination ~ 120):",
            "  ~350 against 372.  Not built: within the noise of the scheduler's slack, and it lengthens the dependent chain of a round."]
    return "\n".join(out) + "\n"


def main():
    consts, tables = load_constants()
    selftest(consts, tables)
    if "--check" in sys.argv:
        print("gen_poseidon_asm: simulator self-test OK")
        return
    if "--budget" in sys.argv:
        print(budget(consts), end="")
        return
    text, stats, total = render(consts, tables)
    if os.path.exists(INC_PATH) and open(INC_PATH).read() == text:
        print("poseidon_gl_asm.inc is current:", stats, "total ~", total)       # untouched: its mtime drives the incremental build
        return
    open(INC_PATH, "w").write(text)
    print("generated poseidon_gl_asm.inc:", stats, "total ~", total)


if __name__ == "__main__":
    main()
