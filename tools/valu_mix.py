#!/usr/bin/env python3
"""Static VALU instruction mix of the hot kernels (from the gfx950 code objects in build/*.o), for the issue ceiling the bench's `valu`
blocks are priced against.  The measured issue rates (profiles/r02_valu_ubench.txt) fall in two classes: the multi-pass integer class
(v_mad_u64_u32, carry adds, v_cndmask with an SGPR mask, 64-bit shifts ...: 36-38 T lane-instr/s) and the single-pass class (v_mov, v_add_u32,
v_xor / v_and / v_or, 32-bit shifts: 59-66 T).  A kernel whose instructions are a share f of the second class cannot issue faster than
1 / ((1 - f) / 37.7 + f / 62) T lane-instr/s.

    python tools/valu_mix.py            # prints one line per kernel: VALU count, fast-class share, ceiling
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "zk-light-client-implementation_amd", "build")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FAST = re.compile(r"^v_(mov_b32|add_u32|sub_u32|subrev_u32|xor_b32|and_b32|or_b32|lshlrev_b32|lshrrev_b32|accvgpr_(read|write)_b32|not_b32)")
KERNELS = {"goldilocks_hip.o": ("gl_hash_leaves_kernel", "gl_ntt_pass_g4_kernelILb0ELb0", "gl_merkle_level_kernel"),
           "bn254_msm_hip.o": ("msm_slice_kernel", "msm_convert_kernel", "msm_combine_kernel"),
           "plonky2_prover_hip.o": ("p2_quotient_gate_kernel",)}
INT_T, FAST_T = 37.7, 62.0


def disasm(obj):
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, os.path.basename(obj))
        os.symlink(obj, tmp)
        subprocess.run([OBJDUMP, "--offloading", tmp], capture_output=True, cwd=d, check=True)
        co = [f for f in os.listdir(d) if "gfx950" in f]
        return subprocess.run([OBJDUMP, "-d", os.path.join(d, co[0])], capture_output=True, text=True, check=True).stdout.splitlines()


def main():
    for obj, names in KERNELS.items():
        lines = disasm(os.path.join(BUILD, obj))
        heads = [(i, ln) for i, ln in enumerate(lines) if re.match(r"^[0-9a-f]+ <_Z", ln)] + [(len(lines), "")]
        for (a, head), (b, _) in zip(heads, heads[1:]):
            if not any(n in head for n in names):
                continue
            c = collections.Counter()
            for ln in lines[a + 1:b]:
                m = re.match(r"\s+(\S+)\s", ln)
                if m and m.group(1).startswith("v_"):
                    c[m.group(1)] += 1
            tot = sum(c.values())
            fast = sum(n for k, n in c.items() if FAST.match(k))
            f = fast / max(1, tot)
            print("%-64s valu %6d  fast-class share %.3f  issue ceiling %.1f T lane-instr/s" % (head.split("<")[1][:62], tot, f,
                                                                                              1 / ((1 - f) / INT_T + f / FAST_T)))


if __name__ == "__main__":
    sys.exit(main())
