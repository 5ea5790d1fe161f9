#!/usr/bin/env python3
"""Instruction mix of k_direct_accumulate<16>'s basic blocks (the compiler-scheduled arm, LWKZG_DIRECT_ASM=0; the hand-scheduled
loop has tools/gen_direct_asm.py --mix), from hipcc's own assembly (no GPU needed): the evidence
behind "about 5200 instructions per mixed addition, 3571 of them v_mad_u64_u32" in DESIGN.md section 4a.

    python tools/isa_mix.py > profiles/rNN_accumulate_isa_mix.txt
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "lambdaworks_kzg_amd", "csrc", "direct.hip")
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "direct.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                           "-o", out, src], stderr=subprocess.DEVNULL)
    s = open(out).read()
import re as _re
m = _re.search(r"^_ZN3lwk19k_direct_accumulateILi16EEEvPKNS_7AffineT[^\n]*:", s, _re.M)     # the function's own label, not a mention of it
i = m.end()
j = s.index(".end_amdhsa_kernel", i)
lines = [l.strip() for l in s[i:j].split("\n") if l.strip() and not l.strip().startswith(";") and not l.strip().startswith(".s")
         and not l.strip().startswith(".p")]
blocks, cur = [], ["entry", []]
blocks.append(cur)
for l in lines:
    if re.match(r"^\.?[A-Za-z_0-9]+:", l):
        cur = [l, []]
        blocks.append(cur)
    else:
        cur[1].append(l)
print("k_direct_accumulate<16>: %d instructions in %d basic blocks (whole kernel, all paths)" % (len(lines), len(blocks)))
print("blocks with more than 60 instructions, in layout order:")
keys = ["v_mad_u64_u32", "v_and_b32_e32", "v_lshrrev_b64", "v_lshl_add_u64", "v_mul_lo_u32", "v_mov_b32_e32", "v_mov_b64_e32",
        "v_sub_u32_e32", "v_add_u32_e32", "v_add3_u32", "v_lshrrev_b32_e32", "v_cndmask_b32_e32", "global_load_dwordx4"]
for name, ins in blocks:
    if len(ins) <= 60:
        continue
    c = collections.Counter(x.split()[0] for x in ins)
    print("  %-14s %5d  " % (name[:14], len(ins)) + "  ".join("%s=%d" % (k.replace("_e32", ""), c[k]) for k in keys if c[k]))
print()
print("main path of one mixed addition = the gather block (global_load_dwordx4=7), the two products that use the row, the zero")
print("test, and the block holding the remaining products; the block with 2275 multiply-adds is the doubling branch (P + P).")
