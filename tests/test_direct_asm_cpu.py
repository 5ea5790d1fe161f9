"""CPU: the hand-scheduled loop of the direct-table MSM (lambdaworks_kzg_amd/csrc/direct_asm.inc) is what its generator
writes, and the generator's own lane-level simulator runs that instruction stream -- every 64-bit column checked against
overflow, every 32-bit add against wrap-around, no register read before its load was waited for -- to the same point as
affine big-int arithmetic: random scalars at three window widths, sparse scalars (windows skipped under EXEC), and a row
that equals the accumulator (the `redo` flag that sends the blob to the C++ kernel). Host logic only; the GPU parity tests
run the assembled kernel."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_direct_asm_inc_is_current():
    import gen_direct_asm as G
    assert open(G.OUT).read() == G.render(G.build())
    assert open(G.OUT.replace(".inc", "_clobbers.inc")).read().split("\n", 1)[1].strip() == G.clobbers()
    assert G.NUM_VGPRS <= 232 and G.NUM_SGPRS <= 102      # two waves per SIMD; the SGPR file


def test_direct_asm_stream_on_a_simulated_lane():
    import gen_direct_asm as G
    per_row = G.selftest(11, 16, verbose=False) / (2 * 16)
    assert per_row < 4300                                   # VALU instructions per mixed addition (the compiler's schedule: 4814)
    G.selftest(12, 13, verbose=False)
    G.selftest(13, 10, spl=1, verbose=False)
    G.selftest(14, 16, verbose=False, sparse=True)
    redo, _ = G.selftest(15, 16, verbose=False, force_equal=True)
    assert redo == 1


def test_single_collision_after_the_accumulator_has_widened_raises_redo():
    """exactly ONE addition of the lane meets a row equal to +-(the sum of the rows before it), at the 3rd .. 15th
    addition -- by then -X of the accumulator sits at its loop bound and P = U2 - X1 is k p with k up to 41, where the second-limb test
    once dropped the carry of k MOD0 (a 34-bit product taken with v_mul_lo_u32) for k >= 17 and left the flag down (26 of 120 such
    lanes before the fix). The flag must come up every time: it is the only thing between the incomplete formulas and a wrong
    commitment."""
    import gen_direct_asm as G
    for t in range(3, 16):
        for neg in (False, True):
            for seed in (100 + t, 200 + 3 * t):
                redo, _ = G.selftest(seed, 16, spl=1, verbose=False, collide_at=t, collide_neg=neg)
                assert redo == 1, (t, neg, seed)
    for t, c in ((5, 13), (19, 13), (9, 10), (25, 10)):      # the generic-plan widths (20 / 26 additions per scalar)
        redo, _ = G.selftest(300 + t, c, spl=1, verbose=False, collide_at=t)
        assert redo == 1, (t, c)


def test_bucket_asm_inc_is_current_and_accumulates_a_bucket_on_a_simulated_lane():
    """the hand-scheduled light-bucket accumulation of the bucket engine (tools/gen_bucket_asm.py, msm.hip: k_bucket_accumulate_asm): the
    committed .inc files are what the generator writes; a simulated lane walks its entry list (0, 1, 2, 19, 40 entries, either sign,
    weakly reduced rows) to the affine big-int sum in the reduction's format (X, Y, ZZ, ZZZ below 2p, limbs below 2^28, infinity as
    literal zeros); a lane of a heavy bucket stores nothing; one row equal / opposite to the accumulator raises the redo flag"""
    import gen_bucket_asm as Bk
    assert open(Bk.OUT).read() == Bk.render(Bk.build())
    assert open(Bk.OUT.replace(".inc", "_clobbers.inc")).read().split("\n", 1)[1].strip() == Bk.clobbers()
    assert Bk.NUM_SGPRS <= 100
    per_row = Bk.selftest(31, 19, verbose=False) / 19
    assert per_row < 4350                                   # the direct stream's body + the entry walk + the closing products
    for seed, n in ((32, 0), (33, 1), (34, 2), (35, 40)):
        Bk.selftest(seed, n, verbose=False)
    Bk.selftest(36, 9, heavy=True, verbose=False)
    for t, neg in ((2, False), (5, True), (11, False), (17, True)):
        assert Bk.selftest(40 + t, 18, collide_at=t, collide_neg=neg, verbose=False) == 1, (t, neg)


def test_fold_asm_inc_is_current_and_adds_on_a_simulated_lane():
    """the hand-scheduled lane fold (tools/gen_fold_asm.py): committed .inc files are what the generator writes; one simulated
    lane adds 4 / 7 stored lane sums through the memory path (bounds as the accumulation leaves them) to the affine big-int
    sum, skips lane sums at infinity, brings the result under 2p, and raises the redo flag on equal points"""
    import gen_fold_asm as F
    assert open(F.OUT).read() == F.render(F.build())
    assert open(F.OUT.replace(".inc", "_clobbers.inc")).read().split("\n", 1)[1].strip() == F.clobbers()
    assert open(F.OUT.replace(".inc", "_clobbers_pad.inc")).read().split("\n", 1)[1].strip() == F.clobbers(True)
    F.selftest(21, 4, verbose=False)
    F.selftest(22, 7, verbose=False)
    F.selftest(23, 5, verbose=False, with_inf=True)
    assert F.selftest(24, 3, verbose=False, equal=True) == 1


def test_hand_scheduled_streams_are_8_byte_aligned_in_the_built_library(tmp_path):
    """Every 8-byte vector instruction of the two hand-scheduled kernels sits at 0 mod 8 in the code object that ships (so none straddles a
    64-byte fetch line): at 4 mod 8 a multiply-add stream loses 9 % at two waves per SIMD and the accumulation lost 3.3 % (profiles/r03_experiments.md
    sections 8-9). Checked on the disassembly, because the encodings are the assembler's choice, not the generator's."""
    import re
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = os.path.join(ROOT, "lambdaworks_kzg_amd", "lib", "liblambdaworks_kzg.so")
    if not os.path.exists(objdump) or not os.path.exists(lib):
        pytest.skip("needs llvm-objdump and the built library")
    work = str(tmp_path)
    shutil.copy(lib, os.path.join(work, "lib.so"))
    subprocess.run([objdump, "-d", "--offloading", "lib.so"], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    seen = {}
    for f in sorted(os.listdir(work)):
        if "gfx950" not in f:
            continue
        dis = subprocess.run([objdump, "-d", f], cwd=work, capture_output=True, text=True).stdout
        cur = None
        for line in dis.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                if m.group(1).startswith("_Z"):
                    cur = m.group(1) if ("k_direct_accumulate_asm" in m.group(1) or "k_direct_fold_lanes_asm" in m.group(1)) else None
                    if cur:
                        seen[cur] = [0, 0]
                continue
            if cur is None:
                continue
            m = re.match(r"\s*(v_\S+)\s.*//\s*([0-9A-Fa-f]+):\s*((?:[0-9A-Fa-f]{8}\s?)+)", line)
            if m and len(m.group(3).split()) == 2:
                seen[cur][0] += 1
                seen[cur][1] += int(m.group(2), 16) % 8 == 4
    assert len(seen) >= 3, seen          # the accumulation and the two builds of the lane fold
    for name, (n, misaligned) in seen.items():
        assert n > 3000 and misaligned <= 4, (name, n, misaligned)     # (the compiler's own few instructions around the statement are its business)
