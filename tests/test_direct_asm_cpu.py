"""CPU: the hand-scheduled loop of the direct-table MSM (lambdaworks_kzg_amd/csrc/direct_asm.inc) is what its generator
writes, and the generator's own lane-level simulator runs that instruction stream -- every 64-bit column checked against
overflow, every 32-bit add against wrap-around, no register read before its load was waited for -- to the same point as
affine big-int arithmetic: random scalars at three window widths, sparse scalars (windows skipped under EXEC), and a row
that equals the accumulator (the `redo` flag that sends the blob to the C++ kernel). Host logic only; the GPU parity tests
run the assembled kernel."""
import os
import sys

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
