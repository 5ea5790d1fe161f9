#!/usr/bin/env python3
"""Generator (and wave-level simulator driver) of the cooperative G1 subgroup test: lambdaworks_kzg_amd/csrc/subgroup_asm.inc, the body of
k_subgroup_coop_asm (sha256.hip). Shares the instruction IR, the product chains, the quad ADDITION and the wave simulator with
tools/gen_coop_asm.py.

    python tools/gen_subgroup_asm.py            # writes csrc/subgroup_asm.inc (+ _clobbers.inc)
    python tools/gen_subgroup_asm.py --check
    python tools/gen_subgroup_asm.py --selftest

check_point_is_in_subgroup (/root/reference/src/compression.rs:22-27) is [r]P == O; the library tests the equivalent [z^2]P == -phi(P) with
phi(x, y) = (beta x, y) (g1.cuh: g1_in_subgroup_endo): two multiplications by |z| = 0xd201000000010000, 126 doublings and 10 additions in all,
a dependent chain of ~1270 field products on ONE lane per point in k_validate_commitments -- 2 ms whatever the batch. Here a quad of lanes owns
the point (lane c = coordinate c of X, Y, ZZ, ZZZ, as in the cooperative MSM kernel) and a doubling (dbl-2008-s-1, a = 0) is three rounds of one
product per lane:

    round 1   XX = X^2          V = (2Y)^2        -                 -
    round 2   S = X V           W = 2Y V          ZZ3 = ZZ V        MM = (3 XX)^2
    (lane 0)  X3 = MM - 2S, carried
    round 3   M (S - X3)        W Y               -                 ZZZ3 = ZZZ W           then Y3 = M (S - X3) - W Y on lane 1

about 1780 instructions per doubling (one lane: 9 products, ~4300), the additions are gen_coop_asm.add_body. The public bits of |z| drive a scalar
loop (one copy of each body). A point whose order is small can make an addition meet P = +-Q or the accumulator reach infinity: the first
raises the quad's `undetermined` verdict (2: the caller runs the complete-branches test on that point), the second shows as ZZ = 0 at the end, which
is "not in the subgroup" (a point of order r never gets there: every partial multiple is a prefix of z's bits, far below r).

Verdicts (one word per point): 0 = not in G1, 1 = in G1, 2 = undetermined. Points whose kind is not 0 (infinity, invalid) are skipped.
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_direct_asm as G  # noqa: E402
import gen_coop_asm as C  # noqa: E402
from gen_direct_asm import EXEC, MASK, P, VCC, W, Val, borrowed, chain_mul, lit, opnd, s, sp, v, vp  # noqa: E402
from gen_coop_asm import (ACC1, ADDR, HA, HB, LANE, M1, QC, QUAD, R1, R2, R3, RT, T1, T2, T3, CProg, WaveSim, dpp_move, emit_chain, set_exec_and,  # noqa: E402
                          sADDM, sINV, sINVP, sMASK, sMOD, sQ0, sQ1, sQ2, sQ3, sTMP, sTMPB, sTROUBLE, sSTMP)

OUT = os.path.join(G.ROOT, "lambdaworks_kzg_amd", "csrc", "subgroup_asm.inc")

# ---- registers beyond the cooperative kernel's -------------------------------------------------------------------------
_n = [C.NUM_VGPRS]


def vregs(n, align=1):
    while _n[0] % align:
        _n[0] += 1
    r = list(range(_n[0], _n[0] + n))
    _n[0] += n
    return r


HT = vregs(14, 2)           # the point being multiplied (this lane's coordinate): P, then [z]P
HX = vregs(14, 2)           # the affine input: x on lane 0, y on lane 1 (for the final comparison)
KIND = vregs(1)[0]
NUM_VGPRS = _n[0]
assert NUM_VGPRS <= 256, NUM_VGPRS

# scalar registers: the cooperative kernel's constants and masks where its addition expects them; the loop's own after them
_s = [C.NUM_SGPRS]


def sregs(n=1, align=1):
    while _s[0] % align:
        _s[0] += 1
    r = _s[0] if n == 1 else list(range(_s[0], _s[0] + n))
    _s[0] += n
    return r


# (the cooperative kernel's scalars that this one has no use for are reused under new names)
sPTS, sKIND, sOUT = C.sTABLE, C.sPART, C.sCTR           # pairs
sACT, sZBITS = C.sAINF, C.sBINF                       # pairs: quads with a point to test; the bits of |z|
sI, sPASS, sN, sFIRST = C.sJ, C.sJEND, C.sN, C.sUNIT
NUM_SGPRS = C.NUM_SGPRS
SBASE = C.SBASE

OPERANDS = ["points (G1Affine29: x, y of 14 words each)", "kinds (0 = a point to test)", "verdicts out (one word per point)", "number of points",
            "first point of this wave", "lane"]
KP4_1, KP8_4, KP16_1 = borrowed(4, 1), borrowed(8, 4), borrowed(16, 1)
Z_ABS = 0xd201000000010000
BETA = 0x5f19672fdf76ce51ba69c6076a0f77eaddb3a93be6f89688de17d813620a00022e01fffffffefffe
PT_X, PT_Y, PT_Z = C.PT_X, C.PT_Y, C.PT_Z


def dbl_body(p):
    """A <- 2A on the quads of sACT (EXEC = sACT on entry and on exit); A in the point format of gen_coop_asm (X < 10p carried, Y < 6p lazy)"""
    e = p.emit
    # lane 1: Y carried, U = 2Y -> HB; lane 0: X -> HB
    set_exec_and(p, sACT, sQ1)
    for i in range(13):
        e("v_lshrrev_b32", v(T1), lit(W), v(HA[i]))
        e("v_and_b32", v(HA[i]), s(sMASK), v(HA[i]))
        e("v_add_u32", v(HA[i + 1]), v(HA[i + 1]), v(T1))
    for i in range(14):
        e("v_lshlrev_b32", v(HB[i]), lit(1), v(HA[i]))
    set_exec_and(p, sACT, sQ0)
    for i in range(14):
        e("v_mov_b32", v(HB[i]), v(HA[i]))
    e("s_mov_b64", EXEC, sp(sACT))
    op1 = Val(HB, 12, 2)                                    # X (10, 1) or 2Y (12, 2); lanes 2, 3: whatever HB holds (their result is not used)
    # ---- round 1: R1 = HB HB = (XX, V, -, -)
    emit_chain(p, chain_mul(op1, op1, R1, M1, ACC1, T1))
    # ---- round 2: RT <- V; HB = (X, U, ZZ, M), M = 3 XX; lane 3's second factor is M too; R2 = HB RT = (S, W, ZZ3, MM)
    dpp_move(p, RT, R1, (1, 1, 1, 1))                       # V to every lane
    dpp_move(p, R3, R1, (0, 0, 0, 0))                       # XX to every lane (R3 is free)
    set_exec_and(p, sACT, sQ2)
    for i in range(14):
        e("v_mov_b32", v(HB[i]), v(HA[i]))                  # ZZ
    set_exec_and(p, sACT, sQ3)
    for i in range(14):
        e("v_lshl_add_u32", v(HB[i]), v(R3[i]), lit(1), v(R3[i]))     # M = 3 XX      (6, 3)
        e("v_mov_b32", v(RT[i]), v(HB[i]))
    e("s_mov_b64", EXEC, sp(sACT))
    emit_chain(p, chain_mul(Val(HB, 12, 3), Val(RT, 6, 3), R2, M1, ACC1, T1))
    # ---- lane 0: M -> HB (for round 3), X3 = MM - 2S + 8p carried -> R1 (over XX), then HA <- S - X3 + 16p
    dpp_move(p, RT, R2, (3, 3, 3, 3))                       # MM to every lane
    dpp_move(p, M1, R2, (1, 1, 1, 1))                       # W to every lane (M1 is free between products)
    set_exec_and(p, sACT, sQ0)
    for i in range(14):
        e("v_lshl_add_u32", v(HB[i]), v(R1[i]), lit(1), v(R1[i]))     # M = 3 XX
    for i in range(14):
        e("v_lshlrev_b32", v(T1), lit(1), v(R2[i]))                   # 2S                 < 2 2^28
        e("v_sub_u32", v(T1), lit(KP8_4[i]), v(T1))
        if i == 0:
            e("v_add_u32", v(R1[i]), v(RT[i]), v(T1))
        else:
            e("v_add3_u32", v(R1[i]), v(RT[i]), v(T1), v(T2))
        if i < 13:
            e("v_lshrrev_b32", v(T2), lit(W), v(R1[i]))
            e("v_and_b32", v(R1[i]), s(sMASK), v(R1[i]))
    assert 2 + 8 <= PT_X[0]
    for i in range(14):
        e("v_add_u32", v(HA[i]), lit(KP16_1[i]), v(R2[i]))
        e("v_sub_u32", v(HA[i]), v(HA[i]), v(R1[i]))                  # S - X3 + 16p       (18, 3)
    # ---- round 3 operands: lane 1: W x Y; lane 3: W x ZZZ (W from lane 1)
    set_exec_and(p, sACT, sQ1)
    for i in range(14):
        e("v_mov_b32", v(HB[i]), v(R2[i]))                  # W
    set_exec_and(p, sACT, sQ3)
    for i in range(14):
        e("v_mov_b32", v(HB[i]), v(M1[i]))                  # W
    e("s_mov_b64", EXEC, sp(sACT))
    emit_chain(p, chain_mul(Val(HB, 6, 3), Val(HA, 18, 3), R3, M1, ACC1, T1))      # R3 = (M (S - X3), W Y, -, ZZZ3)
    # ---- results home
    dpp_move(p, RT, R3, (0, 0, 0, 0))                       # M (S - X3) to every lane
    set_exec_and(p, sACT, sQ1)
    for i in range(14):                                     # Y3 = M (S - X3) - W Y + 4p
        e("v_add_u32", v(HA[i]), lit(KP4_1[i]), v(RT[i]))
        e("v_sub_u32", v(HA[i]), v(HA[i]), v(R3[i]))
    assert (2 + 4, 1 + 2) == PT_Y
    set_exec_and(p, sACT, sQ0)
    for i in range(14):
        e("v_mov_b32", v(HA[i]), v(R1[i]))                  # X3
    set_exec_and(p, sACT, sQ2)
    for i in range(14):
        e("v_mov_b32", v(HA[i]), v(R2[i]))                  # ZZ3
    set_exec_and(p, sACT, sQ3)
    for i in range(14):
        e("v_mov_b32", v(HA[i]), v(R3[i]))                  # ZZZ3
    e("s_mov_b64", EXEC, sp(sACT))


def build():
    p = CProg()
    e = p.emit
    e("s_mov_b64", sp(sPTS), opnd(0))
    e("s_mov_b64", sp(sKIND), opnd(1))
    e("s_mov_b64", sp(sOUT), opnd(2))
    e("s_mov_b32", s(sN), opnd(3))
    e("s_mov_b32", s(sFIRST), opnd(4))
    e("v_mov_b32", v(LANE), opnd(5))
    for i in range(14):
        e("s_mov_b32", s(sMOD[i]), lit(G.MOD[i]))
    e("s_mov_b32", s(sINV), lit(G.INV))
    e("s_mov_b32", s(sMASK), lit(MASK))
    e("s_mov_b32", s(sINVP), lit(G.INVP))
    for k, q in enumerate((sQ0, sQ1, sQ2, sQ3)):
        e("s_mov_b32", s(q[0]), lit(0x11111111 << k))
        e("s_mov_b32", s(q[1]), lit(0x11111111 << k))
    e("s_mov_b32", s(sZBITS[0]), lit(Z_ABS & 0xFFFFFFFF))
    e("s_mov_b32", s(sZBITS[1]), lit(Z_ABS >> 32))
    e("s_mov_b64", sp(sTROUBLE), lit(0))
    e("v_and_b32", v(QC), lit(3), v(LANE))
    e("v_lshrrev_b32", v(QUAD), lit(2), v(LANE))
    e("v_add_u32", v(T3), s(sFIRST), v(QUAD))               # this quad's point
    e("v_cmp_gt_u32", sp(sACT), s(sN), v(T3))
    e("s_mov_b64", EXEC, sp(sACT))
    e("s_cbranch_execz", ("label", "S_end%="))
    e("v_mov_b32", v(KIND), lit(1))
    e("v_mov_b32", v(T1), lit(4))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(T3), v(T1), sp(sKIND))
    e("global_load_dword", v(KIND), vp(ADDR[0]), ("off",))
    e("v_mov_b32", v(T1), lit(2 * C.LANE_BYTES))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(T3), v(T1), sp(sPTS))
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    e("v_and_b32", v(KIND), lit(0xff), v(KIND))
    e("v_cmp_eq_u32", VCC, lit(0), v(KIND))
    e("s_and_b64", sp(sACT), sp(sACT), VCC)                 # a point (kind 0) inside the range
    e("s_mov_b64", EXEC, sp(sACT))
    e("s_cbranch_execz", ("label", "S_end%="))
    # x -> lane 0, y -> lane 1; ZZ = ZZZ = 1 on lanes 2, 3
    e("s_and_b64", sp(sTMP), sp(sACT), sp(sQ0))
    e("s_and_b64", sp(sTMPB), sp(sACT), sp(sQ1))
    C.lane_loads(p, HA, False)
    e("s_or_b64", sp(sTMP), sp(sQ2), sp(sQ3))
    set_exec_and(p, sACT, sTMP)
    for i in range(14):
        e("v_mov_b32", v(HA[i]), lit(G.R1[i]))
    e("s_mov_b64", EXEC, sp(sACT))
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    for i in range(14):
        e("v_mov_b32", v(HX[i]), v(HA[i]))
        e("v_mov_b32", v(HT[i]), v(HA[i]))
    e("s_mov_b32", s(sPASS), lit(0))
    # ---------------- [|z|] twice: acc = t (bit 63); for i = 62 .. 0: acc = 2 acc; if bit i: acc = acc + t
    p.label("S_pass%=")
    e("s_mov_b32", s(sI), lit(62))
    p.label("S_bit%=")
    dbl_body(p)
    e("s_bitcmp1_b64", sp(sZBITS), s(sI))
    e("s_cbranch_scc0", ("label", "S_no_add%="))
    for i in range(14):
        e("v_mov_b32", v(HB[i]), v(HT[i]))
    e("s_mov_b64", sp(sADDM), sp(sACT))
    e("s_or_b64", sp(sTMP), sp(sQ0), sp(sQ1))
    C.add_body(p)
    p.label("S_no_add%=")
    e("s_mov_b64", EXEC, sp(sACT))
    e("s_sub_u32", s(sI), s(sI), lit(1))
    e("s_cbranch_scc0", ("label", "S_bit%="))               # (the borrow out of 0 - 1 ends the loop)
    e("s_add_u32", s(sPASS), s(sPASS), lit(1))
    e("s_cmp_lt_u32", s(sPASS), lit(2))
    e("s_cbranch_scc0", ("label", "S_compare%="))
    for i in range(14):
        e("v_mov_b32", v(HT[i]), v(HA[i]))                  # t <- [z]P
    e("s_branch", ("label", "S_pass%="))
    # ---------------- q = [z^2]P in HA. In G1  <=>  ZZ_q != 0, X_q == (beta x) ZZ_q, Y_q == -y ZZZ_q
    p.label("S_compare%=")
    # round A: lane 0: beta x; lane 1: y ZZZ_q
    dpp_move(p, RT, HA, (3, 3, 3, 3))                       # ZZZ_q to every lane
    set_exec_and(p, sACT, sQ0)
    beta_m = G.limbs(G.to_mont(BETA))
    for i in range(14):
        e("v_mov_b32", v(RT[i]), lit(beta_m[i]))
    e("s_mov_b64", EXEC, sp(sACT))
    emit_chain(p, chain_mul(Val(HX, 2, 1), Val(RT, 2, 1), R1, M1, ACC1, T1))       # R1 = (beta x, y ZZZ_q, -, -)
    # round B: lane 0: (beta x) ZZ_q
    dpp_move(p, RT, HA, (2, 2, 2, 2))                       # ZZ_q to every lane
    emit_chain(p, chain_mul(Val(R1, 2, 1), Val(RT, 2, 1), R2, M1, ACC1, T1))       # R2 = ((beta x) ZZ_q, -, -, -)
    # the three values that must (not) vanish, one per lane, in R3: lane 0: X_q - (beta x) ZZ_q + 4p; lane 1: Y_q + y ZZZ_q; lane 2: ZZ_q
    set_exec_and(p, sACT, sQ0)
    for i in range(14):
        e("v_add_u32", v(R3[i]), lit(KP4_1[i]), v(HA[i]))
        e("v_sub_u32", v(R3[i]), v(R3[i]), v(R2[i]))        # (10 + 4, 3)
    set_exec_and(p, sACT, sQ1)
    for i in range(14):
        e("v_add_u32", v(R3[i]), v(HA[i]), v(R1[i]))        # (6 + 2, 4)
    e("s_or_b64", sp(sTMP), sp(sQ2), sp(sQ3))
    set_exec_and(p, sACT, sTMP)
    for i in range(14):
        e("v_mov_b32", v(R3[i]), v(HA[i]))
    e("s_mov_b64", EXEC, sp(sACT))
    # one product by the Montgomery form of one brings each under 2p with carried limbs; zero mod p is then 0 or p, limb by limb
    for i in range(14):
        e("v_mov_b32", v(RT[i]), lit(G.R1[i]))
    emit_chain(p, chain_mul(Val(R3, 14, 4), Val(RT, 1, 1), R1, M1, ACC1, T1))
    e("v_or_b32", v(T1), v(R1[0]), v(R1[1]))
    e("v_xor_b32", v(T2), s(sMOD[0]), v(R1[0]))
    for i in range(1, 14):
        if i > 1:
            e("v_or_b32", v(T1), v(T1), v(R1[i]))
        e("v_xor_b32", v(T3), s(sMOD[i]), v(R1[i]))
        e("v_or_b32", v(T2), v(T2), v(T3))
    e("v_cmp_eq_u32", sp(sTMP), lit(0), v(T1))
    e("v_cmp_eq_u32", sp(sTMPB), lit(0), v(T2))
    e("s_or_b64", sp(sTMP), sp(sTMP), sp(sTMPB))            # lanes whose value is zero mod p
    # quad verdict on lane 0: zero(lane 0) & zero(lane 1) & !zero(lane 2)
    e("s_lshr_b64", sp(sTMPB), sp(sTMP), lit(1))
    e("s_and_b64", sp(sADDM), sp(sTMP), sp(sTMPB))          # bit 4q: lanes 0 and 1 vanish
    e("s_lshr_b64", sp(sTMPB), sp(sTMP), lit(2))
    e("s_andn2_b64", sp(sADDM), sp(sADDM), sp(sTMPB))       # ... and ZZ_q does not
    e("s_and_b64", sp(sADDM), sp(sADDM), sp(sQ0))
    e("v_cndmask_b32", v(T1), lit(0), lit(1), sp(sADDM))
    e("v_cndmask_b32", v(T1), v(T1), lit(2), sp(sTROUBLE))  # an addition met P = +-Q (low 56 bits): undetermined
    set_exec_and(p, sACT, sQ0)
    e("v_add_u32", v(T3), s(sFIRST), v(QUAD))
    e("v_mov_b32", v(T2), lit(4))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(T3), v(T2), sp(sOUT))
    e("global_store_dword", vp(ADDR[0]), v(T1), ("off",))
    e("s_nop", ("raw", "1"))
    p.label("S_end%=")
    e("s_mov_b64", EXEC, lit(-1))
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    return p


def clobbers():
    return ", ".join(['"v%d"' % i for i in range(NUM_VGPRS)] + ['"s%d"' % i for i in range(SBASE, NUM_SGPRS) if i not in (32, 33, 34)] +
                     ['"vcc"', '"scc"', '"memory"'])


def render(p):
    lines = ["// generated by tools/gen_subgroup_asm.py -- do not edit (python tools/gen_subgroup_asm.py)",
             "// %d instructions, %d of them VALU; VGPRs v0..v%d, SGPRs s%d..s%d" % (
                 sum(1 for i in p.ins if i[0] not in ("label", "comment")), p.count_valu(), NUM_VGPRS - 1, SBASE, NUM_SGPRS - 1)]
    for t in p.text():
        lines.append('"%s\\n"' % t.replace("\\", "\\\\").replace('"', '\\"'))
    return "\n".join(lines) + "\n"


# ---- self-test -----------------------------------------------------------------------------------------------------------
R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
COFACTOR = 0x396c8c005555e1568c00aaab0000aaab


def curve_point_outside_g1(rnd):
    """a point of E(Fp) that is NOT in G1 (order not dividing r), and one of small order obtained by clearing r"""
    while True:
        x = rnd.randrange(P)
        y2 = (x * x * x + 4) % P
        y = pow(y2, (P + 1) // 4, P)
        if y * y % P == y2:
            pt = (x, y)
            if G.ec_mul(R_ORDER, pt) is not None:
                return pt


def run_points(points, kinds=None, prog=None):
    """one wave over up to 16 affine points (None = slot beyond n); returns the verdict words"""
    prog = prog or build()
    rnd = random.Random(99)
    PTS, KIND, OUTP = 0x100000000000, 0x200000000000, 0x300000000000
    n = len(points)
    kinds = kinds or [0] * n
    mem = {}
    for i, pt in enumerate(points):
        words = (G.limbs(G.to_mont(pt[0]) + P * rnd.randrange(0, 2)) + G.limbs(G.to_mont(pt[1]) + P * rnd.randrange(0, 2))) if pt else [0] * 28
        for k, w in enumerate(words):
            mem[PTS + 112 * i + 4 * k] = w
        mem[KIND + 4 * i] = kinds[i]
        mem[OUTP + 4 * i] = 0xEE
    sim = WaveSim(prog, [PTS, KIND, OUTP, n, 0, np.arange(64, dtype=np.uint64)], mem, lambda a, k: None)
    sim.run(max_steps=20_000_000)
    return [mem[OUTP + 4 * i] for i in range(n)], sim


def selftest(verbose=True):
    rnd = random.Random(2025)
    prog = build()
    pts, want = [], []
    for k in range(6):
        pts.append(G.ec_mul(rnd.randrange(1, R_ORDER), G.G1))
        want.append(1)
    for k in range(4):
        pts.append(curve_point_outside_g1(rnd))
        want.append(0)
    small = G.ec_mul(R_ORDER, curve_point_outside_g1(rnd))       # order divides the cofactor
    pts.append(small)
    want.append(None)                                             # 0 or 2 (never 1)
    got, sim = run_points(pts, prog=prog)
    ok = all((g == w) if w is not None else g in (0, 2) for g, w in zip(got, want))
    if verbose:
        print("subgroup selftest: %s, verdicts %s, %d instructions (%d VALU)" % ("ok" if ok else "MISMATCH", got, sim.executed, sim.valu_executed))
    assert ok, (got, want)
    return sim.valu_executed


def main():
    if "--selftest" in sys.argv:
        selftest()
        return
    text = render(build())
    if "--check" in sys.argv:
        assert open(OUT).read() == text, "csrc/subgroup_asm.inc is stale: run python tools/gen_subgroup_asm.py"
        assert open(OUT.replace(".inc", "_clobbers.inc")).read().split("\n", 1)[1].strip() == clobbers()
        print("subgroup_asm.inc matches its generator")
        return
    with open(OUT, "w") as f:
        f.write(text)
    with open(OUT.replace(".inc", "_clobbers.inc"), "w") as f:
        f.write("// generated by tools/gen_subgroup_asm.py -- do not edit\n" + clobbers() + "\n")
    print("wrote %s: %d VALU instructions in the stream, v0..v%d, s%d..s%d" % (OUT, build().count_valu(), NUM_VGPRS - 1, SBASE, NUM_SGPRS - 1))


if __name__ == "__main__":
    main()
