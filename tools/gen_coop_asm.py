#!/usr/bin/env python3
"""Generator (and WAVE-level simulator) of the cooperative MSM kernel for a handful of blobs: lambdaworks_kzg_amd/csrc/coop_asm.inc,
the body of k_coop_msm_asm (direct.hip).

    python tools/gen_coop_asm.py            # writes csrc/coop_asm.inc (+ _clobbers.inc)
    python tools/gen_coop_asm.py --check    # the committed .inc is what this script writes
    python tools/gen_coop_asm.py --selftest # runs the instruction stream on simulated waves against big-int arithmetic

Why a second kernel family. One blob is 65,536 table rows (16-bit windows) to add up; on a chip of 1024 SIMDs that is a reduction
tree of 16 levels whatever the lane count, and a lone wave issues ONE vector instruction per 4.05 cycles whether the instruction depends on
the previous one or not (tools/ubench_latency.hip, profiles/r05_ubench_latency.txt). With one lane per group addition a level
costs the ~6000 instructions of an add-2008-s; the only way down is to spend more LANES on each addition. Here a QUAD (four adjacent
lanes) owns one point, lane c holding coordinate c of (X, Y, ZZ, ZZZ), and the 14 field products of the addition run as four
rounds of one product per lane, operands exchanged inside the quad by DPP quad_perm moves:

    round 1   U1 = X1 ZZ2        S1 = Y1 ZZZ2       U2 = ZZ1 X2        S2 = ZZZ1 Y2
    round 2   PP = (U2 - U1)^2   RR = (S2 - S1)^2   ZZp = ZZ1 ZZ2      ZZZp = ZZZ1 ZZZ2
    round 3   PPP = P PP         Q = U1 PP          ZZ3 = ZZp PP       -
    round 4   S1 PPP             R (Q - X3)         -                  ZZZ3 = ZZZp PPP          (X3 = RR - PPP - 2Q between 3 and 4)

about 2300 instructions per addition instead of 6000. A wave is 16 quads; a workgroup is ONE wave (no LDS, no barrier):

    rows      every quad turns RPQ consecutive windows of one scalar into table rows (signed digits by the add-a-constant recoding:
              digit_j = window_j(s + K) - (H - 1), K = (H - 1) sum_j 2^(C j) over the signed windows) and adds them up;
    tree      four levels inside the wave (the other quad's point comes by ds_bpermute);
    hand-off  quad 0 stores the wave's sum (agent-scope stores), one lane bumps the group's counter; the wave that finds it at
              group size - 1 loads the group's <= 16 sums (agent-scope loads), one per quad, and goes back to `tree`; everyone else
              ends. The last wave standing stores the blob's sum in the library's XYZZ layout.

P = +-Q inside the formulas is detected as in the other streams (P = U2 - U1 vanishes mod 2^56) and raises the blob's redo flag: the
complete-branches C++ path then recomputes that blob. Infinity (a zero digit, an empty group slot) is a mask.

The simulator below executes the SAME instruction list on 64 lanes (numpy) with exact 64-bit columns, asserts what the algorithm relies
on (no column overflow, no 32-bit wrap where none is allowed, no DPP read of a disabled lane, no register read before its s_waitcnt)
and models the hand-off memory; tests/test_coop_asm_cpu.py runs small problems through several waves against affine big-int arithmetic.
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_direct_asm as G  # noqa: E402
from gen_direct_asm import EXEC, MASK, P, VCC, W, Prog, Val, borrowed, chain_mul, lit, opnd, s, sp, v, vp  # noqa: E402

OUT = os.path.join(G.ROOT, "lambdaworks_kzg_amd", "csrc", "coop_asm.inc")

# ---- registers -------------------------------------------------------------------------------------------------------
_n = [0]


def vregs(n, align=1):
    while _n[0] % align:
        _n[0] += 1
    r = list(range(_n[0], _n[0] + n))
    _n[0] += n
    return r


HA = vregs(14, 2)           # this lane's coordinate of the quad's point A (lane & 3: X, Y, ZZ, ZZZ)
HB = vregs(14, 2)           # ... of the other operand B
RT = vregs(14, 2)           # what a DPP move brought from another lane of the quad
R1, R2, R3 = vregs(14, 2), vregs(14, 2), vregs(14, 2)   # the rounds' products
M1 = vregs(14)
ACC1 = vregs(2, 2)
T1, T2, T3 = vregs(1)[0], vregs(1)[0], vregs(1)[0]
SC = vregs(8, 2)            # the quad's scalar + K, shifted down window by window
ADDR = vregs(2, 2)
LANE, QC, QUAD, POINT, MAG, BPADDR, VRB, SIGN = (vregs(1)[0] for _ in range(8))
HN = vregs(14, 2)           # the NEXT table row of this quad, gathered while the current addition runs
NUM_VGPRS = _n[0]
assert NUM_VGPRS <= 168, NUM_VGPRS       # three waves per SIMD

sINV, sMASK, sINVP, sMOD = G.sINV, G.sMASK, G.sINVP, G.sMOD
SBASE = G.SBASE
_s = [max(sMOD) + 1]


def sregs(n=1, align=1):
    while _s[0] % align:
        _s[0] += 1
    r = _s[0] if n == 1 else list(range(_s[0], _s[0] + n))
    _s[0] += n
    return r


sSTMP = G.sMORE                                        # (s31: a scratch scalar)
sTABLE, sPART, sCTR, sOUT, sREDO = (sregs(2, 2) for _ in range(5))
sC, sNW, sH, sHTOP, sRB, sMASKC = (sregs() for _ in range(6))
sJ, sJEND, sLVL, sN, sUNIT, sSOFF, sCOFF, sPHASE = (sregs() for _ in range(8))
sQ0, sQ1, sQ2, sQ3 = (sregs(2, 2) for _ in range(4))
sAINF, sBINF, sTAKE, sADDM, sTROUBLE, sTMP, sTMPB = (sregs(2, 2) for _ in range(7))
sSUBJ, sMASKJ, sHJ = (sregs() for _ in range(3))
sGRP, sGSZ = sSUBJ, sMASKJ                             # (the hand-off's scalars: the row fetch is over by then)
NUM_SGPRS = _s[0]
assert NUM_SGPRS <= 100, NUM_SGPRS                     # (s100, s101 are XNACK_MASK on gfx950)

OPERANDS = ["window base addresses (device array)", "this blob's scalars", "this blob's partial sums", "this blob's counters", "this blob's sum (out)",
            "this blob's redo flag", "K word 0", "K word 1", "K word 2", "K word 3", "K word 4", "K word 5", "K word 6", "K word 7",
            "c | nw << 8 | rpq << 16 | log2(points) << 24", "wtop", "row_bytes", "waves of this blob at stage 0", "this wave's unit (blockIdx.x)", "lane"]
(O_TABLE, O_SCALARS, O_PART, O_CTR, O_OUT, O_REDO, O_K0) = range(7)
O_PACK, O_WTOP, O_RB, O_N0, O_UNIT, O_LANE = range(14, 20)
LANE_BYTES = 56
UNIT_BYTES = 224

KP4_1, KP8_4, KP16_1 = borrowed(4, 1), borrowed(8, 4), borrowed(16, 1)
# a point as it stands between additions: X < 10p carried, Y < 6p with limbs < 3 2^28, ZZ, ZZZ < 2p carried
PT_X, PT_Y, PT_Z = (10, 1), (6, 3), (2, 1)
PT_ANY = (10, 3)            # what ONE register set may hold across the four lanes of a quad


class CProg(Prog):
    """Prog whose instructions may carry a modifier string (DPP controls, cache-scope bits) behind their operands"""

    def text(self):
        out = []
        prev_vector = False
        for op, args, kw in self.ins:
            if G.KNOB_E64 and op.startswith("v_") and not prev_vector and out:
                out.append("  .p2align 3")
            if op not in ("label", "comment"):
                prev_vector = op.startswith("v_")
            if op == "label":
                out.append("%s:" % args[0])
                continue
            if op == "comment":
                out.append("; " + args[0])
                continue
            line = op
            if G.KNOB_E64 and op in G.E64_OPS and not any(a[0] == "op" or (a[0] == "lit" and 64 < a[1] < 0xFFFFFFF0) for a in args):
                line += "_e64"
            if args:
                line += " " + ", ".join(G.fmt(a) for a in args)
            if kw.get("offset"):
                line += " offset:%d" % kw["offset"]
            if kw.get("mod"):
                line += " " + kw["mod"]
            out.append("  " + line)
        return out


def qperm(a, b, c, d):
    return "quad_perm:[%d,%d,%d,%d] row_mask:0xf bank_mask:0xf" % (a, b, c, d)


def dpp_move(p, dst, src, perm):
    """dst <- src of lane perm[lane & 3] of the same quad, all 14 limbs (EXEC must cover whole quads)"""
    p.emit("s_nop", ("raw", "1"))            # (a VALU result needs two wait states before a DPP instruction reads it)
    for i in range(14):
        p.emit("v_mov_b32_dpp", v(dst[i]), v(src[i]), mod=qperm(*perm), perm=perm)


def set_exec_and(p, a, b):
    p.emit("s_and_b64", EXEC, sp(a), sp(b))


def emit_chain(p, chain):
    p.emit(".p2align", ("raw", "3"))
    for op, args, kw in chain:
        p.emit(op, *args, **kw)


def trouble_check(p, preg):
    """lanes of EXEC whose value in preg (< 8p) is 0 mod p -> VCC (low 56 bits tested, 28 at a time: a false alarm costs time only)"""
    e = p.emit
    e("v_mul_lo_u32", v(T1), v(preg[0]), s(sINVP))
    e("v_and_b32", v(T1), s(sMASK), v(T1))                 # k with k p = value mod 2^28
    e("v_cmp_gt_u32", VCC, lit(8), v(T1))
    e("s_cbranch_vccz", ("label", "C_no_cand%="))
    e("s_mov_b64", sp(sTMPB), VCC)
    e("v_mul_lo_u32", v(T2), v(T1), s(sMOD[0]))            # (k < 8: k MOD0 < 2^31, exact in 32 bits)
    e("v_lshrrev_b32", v(T2), lit(W), v(T2))
    e("v_mul_lo_u32", v(T3), v(T1), s(sMOD[1]))
    e("v_add_u32", v(T2), v(T2), v(T3), wrap=True)
    e("v_lshrrev_b32", v(T3), lit(W), v(preg[0]))
    e("v_add_u32", v(T3), v(T3), v(preg[1]))
    e("v_xor_b32", v(T2), v(T2), v(T3))
    e("v_and_b32", v(T2), s(sMASK), v(T2))
    e("v_cmp_eq_u32", VCC, lit(0), v(T2))
    e("s_and_b64", VCC, sp(sTMPB), VCC)
    p.label("C_no_cand%=")


def add_body(p):
    """A <- A + B on the quads of sADDM (neither at infinity); the quads' four lanes are all enabled"""
    e = p.emit
    any_ = lambda r: Val(r, *PT_ANY)                        # noqa: E731
    e("s_mov_b64", EXEC, sp(sADDM))
    # ---- round 1: RT <- B of lane c ^ 2; R1 = HA RT = (U1, S1, U2, S2)
    dpp_move(p, RT, HB, (2, 3, 0, 1))
    emit_chain(p, chain_mul(any_(HA), any_(RT), R1, M1, ACC1, T1))
    # ---- round 2: lanes 0, 1: HA = HB = (U2 - U1, S2 - S1) + 4p; R2 = HA HB = (PP, RR, ZZp, ZZZp)
    dpp_move(p, RT, R1, (2, 3, 0, 1))
    set_exec_and(p, sADDM, sTMP)                            # sTMP = Q0 | Q1 (set by the caller of add_body)
    for i in range(14):
        e("v_add_u32", v(HA[i]), lit(KP4_1[i]), v(RT[i]))
        e("v_sub_u32", v(HA[i]), v(HA[i]), v(R1[i]))
        e("v_mov_b32", v(HB[i]), v(HA[i]))
    diff = (2 + 4, 1 + 2)
    trouble_check(p, HA)                                    # only lane 0's verdict counts (P); lane 1 tests R, which may vanish
    e("s_and_b64", VCC, VCC, sp(sQ0))
    e("s_or_b64", sp(sTROUBLE), sp(sTROUBLE), VCC)
    e("s_mov_b64", EXEC, sp(sADDM))
    emit_chain(p, chain_mul(Val(HA, *diff), Val(HB, *diff), R2, M1, ACC1, T1))
    # ---- round 3: RT <- PP; HB = (P, U1, ZZp, -); R3 = HB RT = (PPP, Q, ZZ3, -)
    dpp_move(p, RT, R2, (0, 0, 0, 0))
    dpp_move(p, R3, R1, (0, 0, 0, 0))                       # U1 to every lane (R3 is free), then to lane 1's HB
    set_exec_and(p, sADDM, sQ1)
    for i in range(14):
        e("v_mov_b32", v(HB[i]), v(R3[i]))
    set_exec_and(p, sADDM, sQ2)
    for i in range(14):
        e("v_mov_b32", v(HB[i]), v(R2[i]))
    e("s_mov_b64", EXEC, sp(sADDM))
    emit_chain(p, chain_mul(Val(HB, *diff), Val(RT, 2, 1), R3, M1, ACC1, T1))
    # ---- X3 = RR - PPP - 2Q + 8p on lane 1, carried, over RR
    dpp_move(p, RT, R3, (0, 0, 0, 0))                       # PPP to every lane
    set_exec_and(p, sADDM, sQ1)
    for i in range(14):
        e("v_lshl_add_u32", v(T1), v(R3[i]), lit(1), v(RT[i]))           # 2Q + PPP           < 3 2^28
        e("v_sub_u32", v(T1), lit(KP8_4[i]), v(T1))
        if i == 0:
            e("v_add_u32", v(R2[i]), v(R2[i]), v(T1))
        else:
            e("v_add3_u32", v(R2[i]), v(R2[i]), v(T1), v(T2))
        if i < 13:
            e("v_lshrrev_b32", v(T2), lit(W), v(R2[i]))
            e("v_and_b32", v(R2[i]), s(sMASK), v(R2[i]))
    x3 = (2 + 8, 1)
    assert x3[0] <= PT_X[0]
    # ---- round 4 operands: lane 0: S1 x PPP; lane 1: R x (Q - X3 + 16p); lane 3: ZZZp x PPP
    for i in range(14):                                     # (still lane 1) RT <- Q - X3 + 16p, HB <- R
        e("v_add_u32", v(RT[i]), lit(KP16_1[i]), v(R3[i]))
        e("v_sub_u32", v(RT[i]), v(RT[i]), v(R2[i]))
        e("v_mov_b32", v(HB[i]), v(HA[i]))
    t1 = (2 + 16, 1 + 2)
    e("s_mov_b64", EXEC, sp(sADDM))
    dpp_move(p, M1, R1, (1, 1, 1, 1))                       # S1 to every lane (M1 is free between products), then to lane 0's HB
    set_exec_and(p, sADDM, sQ0)
    for i in range(14):
        e("v_mov_b32", v(HB[i]), v(M1[i]))
    set_exec_and(p, sADDM, sQ3)
    for i in range(14):
        e("v_mov_b32", v(HB[i]), v(R2[i]))
    e("s_mov_b64", EXEC, sp(sADDM))
    emit_chain(p, chain_mul(Val(HB, *diff), Val(RT, *t1), R1, M1, ACC1, T1))      # R1 = (S1 PPP, R (Q - X3), -, ZZZ3)
    # ---- results home: HA = (X3, Y3, ZZ3, ZZZ3)
    dpp_move(p, RT, R1, (0, 0, 0, 0))                       # S1 PPP to every lane
    dpp_move(p, M1, R2, (1, 1, 1, 1))                       # X3 to every lane
    set_exec_and(p, sADDM, sQ1)
    for i in range(14):                                     # Y3 = R (Q - X3) - S1 PPP + 4p
        e("v_add_u32", v(HA[i]), lit(KP4_1[i]), v(R1[i]))
        e("v_sub_u32", v(HA[i]), v(HA[i]), v(RT[i]))
    assert (2 + 4, 1 + 2) == PT_Y
    set_exec_and(p, sADDM, sQ0)
    for i in range(14):
        e("v_mov_b32", v(HA[i]), v(M1[i]))
    set_exec_and(p, sADDM, sQ2)
    for i in range(14):
        e("v_mov_b32", v(HA[i]), v(R3[i]))
    set_exec_and(p, sADDM, sQ3)
    for i in range(14):
        e("v_mov_b32", v(HA[i]), v(R1[i]))


def lane_loads(p, dst, agent_scope):
    """dst[0:14] <- the 56 bytes of this lane's coordinate at ADDR (+56 on odd lanes): naturally aligned pieces, masks sTMP (even lanes
    of the loading quads) and sTMPB (odd ones) prepared by the caller"""
    e = p.emit
    mod = "sc1" if agent_scope else None
    e("s_mov_b64", EXEC, sp(sTMP))
    for k in range(3):
        e("global_load_dwordx4", ("v4", dst[4 * k]), vp(ADDR[0]), ("off",), offset=16 * k, mod=mod)
    e("global_load_dwordx2", vp(dst[12]), vp(ADDR[0]), ("off",), offset=48, mod=mod)
    e("s_mov_b64", EXEC, sp(sTMPB))
    e("global_load_dwordx2", vp(dst[0]), vp(ADDR[0]), ("off",), offset=56, mod=mod)
    for k in range(3):
        e("global_load_dwordx4", ("v4", dst[2 + 4 * k]), vp(ADDR[0]), ("off",), offset=64 + 16 * k, mod=mod)


def lane_stores(p, src):
    """the same pieces, stored with agent scope"""
    e = p.emit
    e("s_mov_b64", EXEC, sp(sTMP))
    for k in range(3):
        e("global_store_dwordx4", vp(ADDR[0]), ("v4", src[4 * k]), ("off",), offset=16 * k, mod="sc1")
    e("global_store_dwordx2", vp(ADDR[0]), vp(src[12]), ("off",), offset=48, mod="sc1")
    e("s_mov_b64", EXEC, sp(sTMPB))
    e("global_store_dwordx2", vp(ADDR[0]), vp(src[0]), ("off",), offset=56, mod="sc1")
    for k in range(3):
        e("global_store_dwordx4", vp(ADDR[0]), ("v4", src[2 + 4 * k]), ("off",), offset=64 + 16 * k, mod="sc1")


def row_issue(p):
    """window sJ of the scalar in SC (all lanes): MAG, SIGN, the row's address, its gather into HN (lane 0: x, lane 1: y); SC moves on"""
    e = p.emit
    e("s_add_u32", s(sSUBJ), s(sH), lit(-1))
    e("s_add_u32", s(sSTMP), s(sNW), lit(-1))
    e("s_cmp_eq_u32", s(sJ), s(sSTMP))                      # the top window is unsigned and takes what is left of the scalar
    e("s_cselect_b32", s(sMASKJ), lit(-1), s(sMASKC))
    e("s_cselect_b32", s(sHJ), s(sHTOP), s(sH))
    e("s_cselect_b32", s(sSUBJ), lit(0), s(sSUBJ))
    e("s_lshl_b32", s(sSTMP), s(sJ), lit(3))
    e("s_load_dwordx2", sp(sTMPB), sp(sTABLE), s(sSTMP))
    e("v_and_b32", v(T1), s(sMASKJ), v(SC[0]))
    e("v_cmp_lt_u32", sp(sTMP), v(T1), s(sSUBJ))            # negative digit
    e("v_sub_u32", v(T2), v(T1), s(sSUBJ), wrap=True)
    e("v_sub_u32", v(T3), s(sSUBJ), v(T1), wrap=True)
    e("v_cndmask_b32", v(MAG), v(T2), v(T3), sp(sTMP))
    e("v_cndmask_b32", v(SIGN), lit(0), lit(1), sp(sTMP))
    for i in range(7):
        e("v_alignbit_b32", v(SC[i]), v(SC[i + 1]), v(SC[i]), s(sC))
    e("v_lshrrev_b32", v(SC[7]), s(sC), v(SC[7]))
    e("v_cmp_ne_u32", sp(sTAKE), lit(0), v(MAG))            # quads that have a row to gather
    e("v_mad_u32_u24", v(T3), v(POINT), s(sHJ), v(MAG))
    e("s_waitcnt", ("raw", "lgkmcnt(0)"))
    e("s_sub_u32", s(sTMPB[0]), s(sTMPB[0]), s(sRB))        # minus one row: mag counts from 1
    e("s_subb_u32", s(sTMPB[1]), s(sTMPB[1]), lit(0))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(T3), v(VRB), sp(sTMPB))
    e("s_and_b64", sp(sTMP), sp(sQ0), sp(sTAKE))
    e("s_and_b64", sp(sTMPB), sp(sQ1), sp(sTAKE))
    lane_loads(p, HN, False)
    e("s_mov_b64", EXEC, lit(-1))


def build():
    p = CProg()
    e = p.emit
    # ---------------- prologue
    e("comment", "operands -> fixed registers")
    e("s_mov_b64", sp(sTABLE), opnd(O_TABLE))
    e("s_mov_b64", sp(sPART), opnd(O_PART))
    e("s_mov_b64", sp(sCTR), opnd(O_CTR))
    e("s_mov_b64", sp(sOUT), opnd(O_OUT))
    e("s_mov_b64", sp(sREDO), opnd(O_REDO))
    e("s_and_b32", s(sC), opnd(O_PACK), lit(0xff))
    e("s_bfe_u32", s(sNW), opnd(O_PACK), lit(0x80008))      # 8 bits from bit 8
    e("s_bfe_u32", s(sJEND), opnd(O_PACK), lit(0x80010))    # rpq (for now)
    e("s_bfe_u32", s(sLVL), opnd(O_PACK), lit(0x80018))     # log2(points) (for now)
    e("s_mov_b32", s(sRB), opnd(O_RB))
    e("s_mov_b32", s(sN), opnd(O_N0))
    e("s_mov_b32", s(sUNIT), opnd(O_UNIT))
    e("v_mov_b32", v(LANE), opnd(O_LANE))
    e("v_mov_b32", v(VRB), s(sRB))
    for i in range(14):
        e("s_mov_b32", s(sMOD[i]), lit(G.MOD[i]))
    e("s_mov_b32", s(sINV), lit(G.INV))
    e("s_mov_b32", s(sMASK), lit(MASK))
    e("s_mov_b32", s(sINVP), lit(G.INVP))
    for k, q in enumerate((sQ0, sQ1, sQ2, sQ3)):
        e("s_mov_b32", s(q[0]), lit(0x11111111 << k))
        e("s_mov_b32", s(q[1]), lit(0x11111111 << k))
    e("s_lshl_b32", s(sSTMP), lit(1), s(sC))
    e("s_add_u32", s(sMASKC), s(sSTMP), lit(-1))
    e("s_lshr_b32", s(sH), s(sSTMP), lit(1))
    e("s_lshl_b32", s(sHTOP), lit(1), opnd(O_WTOP))
    e("s_mov_b64", sp(sTROUBLE), lit(0))
    e("s_mov_b64", sp(sAINF), lit(-1))
    e("s_mov_b32", s(sSOFF), lit(0))
    e("s_mov_b32", s(sCOFF), lit(0))
    e("v_and_b32", v(QC), lit(3), v(LANE))
    e("v_lshrrev_b32", v(QUAD), lit(2), v(LANE))
    # the quad's unit u = 16 unit + quad: point = u & (points - 1), window group g = u >> log2(points) (wave-uniform: 16 | points)
    e("s_lshl_b32", s(sSTMP), s(sUNIT), lit(4))
    e("v_add_u32", v(POINT), s(sSTMP), v(QUAD))
    e("s_lshr_b32", s(sJ), s(sSTMP), s(sLVL))               # g
    e("s_lshl_b32", s(sSTMP), lit(1), s(sLVL))
    e("s_add_u32", s(sSTMP), s(sSTMP), lit(-1))
    e("v_and_b32", v(POINT), s(sSTMP), v(POINT))
    e("s_mul_i32", s(sJ), s(sJ), s(sJEND))                  # first window = g rpq
    e("s_add_u32", s(sJEND), s(sJ), s(sJEND))
    e("s_min_u32", s(sJEND), s(sJEND), s(sNW))
    # the scalar (32 bytes, little-endian words) + K
    e("v_mov_b32", v(T1), lit(32))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(POINT), v(T1), opnd(O_SCALARS))
    e("global_load_dwordx4", ("v4", SC[0]), vp(ADDR[0]), ("off",))
    e("global_load_dwordx4", ("v4", SC[4]), vp(ADDR[0]), ("off",), offset=16)
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    e("v_add_co_u32", v(SC[0]), VCC, opnd(O_K0), v(SC[0]))
    for i in range(1, 8):                                   # (VCC and a scalar source in one instruction would be two constant-bus reads)
        e("v_mov_b32", v(T1), opnd(O_K0 + i))
        e("s_nop", ("raw", "1"))
        e("v_addc_co_u32", v(SC[i]), VCC, v(T1), v(SC[i]), VCC)
    # shift down to the first window: whole words first, then the bits
    e("s_mul_i32", s(sSTMP), s(sJ), s(sC))
    e("s_lshr_b32", s(sLVL), s(sSTMP), lit(5))
    e("s_and_b32", s(sSTMP), s(sSTMP), lit(31))
    p.label("C_words%=")
    e("s_cmp_eq_u32", s(sLVL), lit(0))
    e("s_cbranch_scc1", ("label", "C_words_done%="))
    for i in range(7):
        e("v_mov_b32", v(SC[i]), v(SC[i + 1]))
    e("v_mov_b32", v(SC[7]), lit(0))
    e("s_add_u32", s(sLVL), s(sLVL), lit(-1))
    e("s_branch", ("label", "C_words%="))
    p.label("C_words_done%=")
    for i in range(7):
        e("v_alignbit_b32", v(SC[i]), v(SC[i + 1]), v(SC[i]), s(sSTMP))
    e("v_lshrrev_b32", v(SC[7]), s(sSTMP), v(SC[7]))
    e("s_mov_b32", s(sPHASE), lit(0))                       # 0: rows, 1: tree
    row_issue(p)

    # ---------------- one step: fetch B (a table row, or another quad's point), then the common addition
    p.label("C_step%=")
    e("s_cmp_eq_u32", s(sPHASE), lit(0))
    e("s_cbranch_scc0", ("label", "C_fetch_tree%="))
    # ---- a row: the gather issued one addition ago has landed in HN
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    e("v_cmp_eq_u32", sp(sBINF), lit(0), v(MAG))            # no row (zero digit): B at infinity
    e("v_cmp_ne_u32", sp(sTMP), lit(0), v(SIGN))
    e("s_andn2_b64", sp(sTMP), sp(sTMP), sp(sBINF))
    e("s_or_b64", EXEC, sp(sQ0), sp(sQ1))
    for i in range(14):
        e("v_mov_b32", v(HB[i]), v(HN[i]))
    e("s_or_b64", EXEC, sp(sQ2), sp(sQ3))                   # ZZ = ZZZ = 1
    for i in range(14):
        e("v_mov_b32", v(HB[i]), lit(G.R1[i]))
    set_exec_and(p, sTMP, sQ1)                              # negative digit: y <- 4p - y
    for i in range(14):
        e("v_sub_u32", v(HB[i]), lit(KP4_1[i]), v(HB[i]))
    e("s_mov_b64", EXEC, lit(-1))
    e("s_add_u32", s(sJ), s(sJ), lit(1))
    e("s_cmp_lt_u32", s(sJ), s(sJEND))
    e("s_cbranch_scc0", ("label", "C_no_next_row%="))
    row_issue(p)
    p.label("C_no_next_row%=")
    e("s_mov_b64", sp(sTAKE), lit(-1))
    e("s_branch", ("label", "C_have_b%="))
    # ---- the tree: B <- A of quad + (1 << level)
    p.label("C_fetch_tree%=")
    e("s_mov_b64", EXEC, lit(-1))
    e("s_lshl_b32", s(sSTMP), lit(4), s(sLVL))              # lanes between the two quads
    e("v_add_u32", v(BPADDR), s(sSTMP), v(LANE))
    e("v_and_b32", v(BPADDR), lit(63), v(BPADDR))
    e("v_lshlrev_b32", v(BPADDR), lit(2), v(BPADDR))
    for a, b in zip(HA, HB):
        e("ds_bpermute_b32", v(b), v(BPADDR), v(a))
    e("s_lshr_b64", sp(sBINF), sp(sAINF), s(sSTMP))
    # the quads that take part: quad & ((2 << level) - 1) == 0
    e("s_lshl_b32", s(sSTMP), lit(2), s(sLVL))
    e("s_add_u32", s(sSTMP), s(sSTMP), lit(-1))
    e("v_and_b32", v(T1), s(sSTMP), v(QUAD))
    e("v_cmp_eq_u32", sp(sTAKE), lit(0), v(T1))
    e("s_waitcnt", ("raw", "lgkmcnt(0)"))
    p.label("C_have_b%=")
    # ---------------- the addition under the masks: copy (A at infinity, B not), add (neither), nothing (B at infinity)
    e("s_andn2_b64", sp(sTMP), sp(sAINF), sp(sBINF))
    e("s_and_b64", sp(sTMP), sp(sTMP), sp(sTAKE))
    e("s_mov_b64", EXEC, sp(sTMP))
    e("s_cbranch_execz", ("label", "C_no_copy%="))
    for a, b in zip(HA, HB):
        e("v_mov_b32", v(a), v(b))
    p.label("C_no_copy%=")
    e("s_or_b64", sp(sADDM), sp(sAINF), sp(sBINF))
    e("s_andn2_b64", sp(sADDM), sp(sTAKE), sp(sADDM))
    e("s_andn2_b64", sp(sTMP), sp(sTAKE), sp(sBINF))        # taking quads whose B is a point: A is a point afterwards
    e("s_andn2_b64", sp(sAINF), sp(sAINF), sp(sTMP))
    e("s_or_b64", sp(sTMP), sp(sQ0), sp(sQ1))
    e("s_mov_b64", EXEC, sp(sADDM))
    e("s_cbranch_execz", ("label", "C_no_add%="))
    add_body(p)
    p.label("C_no_add%=")
    e("s_mov_b64", EXEC, lit(-1))
    # ---------------- what comes next
    e("s_cmp_eq_u32", s(sPHASE), lit(0))
    e("s_cbranch_scc0", ("label", "C_next_level%="))
    e("s_cmp_lt_u32", s(sJ), s(sJEND))                      # (sJ moved on when the row was taken)
    e("s_cbranch_scc1", ("label", "C_step%="))
    e("s_mov_b32", s(sPHASE), lit(1))
    e("s_mov_b32", s(sLVL), lit(0))
    e("s_branch", ("label", "C_step%="))
    p.label("C_next_level%=")
    e("s_add_u32", s(sLVL), s(sLVL), lit(1))
    e("s_cmp_lt_u32", s(sLVL), lit(4))
    e("s_cbranch_scc1", ("label", "C_step%="))

    # ---------------- quad 0 holds the wave's sum: carry Y, zeros for infinity, store; raise the redo flag if an addition met P = 0
    e("s_mov_b64", EXEC, lit(15))
    for i in range(13):
        e("v_lshrrev_b32", v(T1), lit(W), v(HA[i]))
        e("v_and_b32", v(HA[i]), s(sMASK), v(HA[i]))
        e("v_add_u32", v(HA[i + 1]), v(HA[i + 1]), v(T1))
    e("s_and_b64", EXEC, sp(sAINF), lit(15))
    for i in range(14):
        e("v_mov_b32", v(HA[i]), lit(0))
    e("s_cmp_eq_u64", sp(sTROUBLE), lit(0))
    e("s_cbranch_scc1", ("label", "C_no_trouble%="))
    e("s_mov_b64", EXEC, lit(1))
    e("v_mov_b32", v(T1), lit(1))
    e("v_mov_b32", v(ADDR[0]), s(sREDO[0]))
    e("v_mov_b32", v(ADDR[1]), s(sREDO[1]))
    e("global_store_dword", vp(ADDR[0]), v(T1), ("off",), mod="sc1")
    e("s_mov_b64", sp(sTROUBLE), lit(0))
    p.label("C_no_trouble%=")
    e("s_mov_b64", EXEC, lit(15))
    e("s_cmp_eq_u32", s(sN), lit(1))
    e("s_cbranch_scc0", ("label", "C_publish%="))
    # the last wave standing: the blob's sum, library layout (X, Y, ZZ, ZZZ; zeros = infinity)
    e("v_lshrrev_b32", v(T1), lit(1), v(QC))
    e("v_mov_b32", v(T2), lit(2 * LANE_BYTES))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(T1), v(T2), sp(sOUT))
    e("s_mov_b64", sp(sTMP), lit(5))
    e("s_mov_b64", sp(sTMPB), lit(10))
    lane_stores(p, HA)
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    e("s_endpgm")
    p.label("C_publish%=")
    # partial sum number sSOFF + sUNIT; lane pairs (0, 1) and (2, 3) share an address, the odd lane's piece 56 bytes further on
    e("s_add_u32", s(sSTMP), s(sSOFF), s(sUNIT))
    e("s_lshl_b32", s(sSTMP), s(sSTMP), lit(1))
    e("v_lshrrev_b32", v(T1), lit(1), v(QC))
    e("v_add_u32", v(T1), s(sSTMP), v(T1))
    e("v_mov_b32", v(T2), lit(2 * LANE_BYTES))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(T1), v(T2), sp(sPART))
    e("s_mov_b64", sp(sTMP), lit(5))
    e("s_mov_b64", sp(sTMPB), lit(10))
    lane_stores(p, HA)
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    # the group's counter: the wave that finds group size - 1 carries on
    e("s_lshr_b32", s(sGRP), s(sUNIT), lit(4))
    e("s_lshl_b32", s(sSTMP), s(sGRP), lit(4))
    e("s_sub_u32", s(sGSZ), s(sN), s(sSTMP))
    e("s_min_u32", s(sGSZ), s(sGSZ), lit(16))
    e("s_add_u32", s(sSTMP), s(sCOFF), s(sGRP))
    e("s_lshl_b32", s(sSTMP), s(sSTMP), lit(2))
    e("s_add_u32", s(sTMP[0]), s(sCTR[0]), s(sSTMP))
    e("s_addc_u32", s(sTMP[1]), s(sCTR[1]), lit(0))
    e("s_mov_b64", EXEC, lit(1))
    e("v_mov_b32", v(ADDR[0]), s(sTMP[0]))
    e("v_mov_b32", v(ADDR[1]), s(sTMP[1]))
    e("v_mov_b32", v(T1), lit(1))
    e("global_atomic_add", v(T2), vp(ADDR[0]), v(T1), ("off",), mod="sc0")
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    e("s_nop", ("raw", "1"))
    e("v_readfirstlane_b32", s(sSTMP), v(T2))
    e("s_add_u32", s(sSTMP), s(sSTMP), lit(1))
    e("s_cmp_eq_u32", s(sSTMP), s(sGSZ))
    e("s_cbranch_scc1", ("label", "C_elected%="))
    e("s_endpgm")
    p.label("C_elected%=")
    e("v_mov_b32", v(T1), lit(0))                           # the counter is left at zero for the next call
    e("global_store_dword", vp(ADDR[0]), v(T1), ("off",), mod="sc1")
    # the group's partial sums, one per quad: unit sSOFF + 16 grp + quad for quad < gsize
    e("s_mov_b64", EXEC, lit(-1))
    e("s_lshl_b32", s(sSTMP), s(sGRP), lit(4))
    e("s_add_u32", s(sSTMP), s(sSTMP), s(sSOFF))
    e("v_add_u32", v(T1), s(sSTMP), v(QUAD))
    e("v_lshlrev_b32", v(T1), lit(1), v(T1))
    e("v_lshrrev_b32", v(T2), lit(1), v(QC))
    e("v_add_u32", v(T1), v(T1), v(T2))
    e("v_mov_b32", v(T2), lit(2 * LANE_BYTES))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(T1), v(T2), sp(sPART))
    e("v_cmp_gt_u32", sp(sTAKE), s(sGSZ), v(QUAD))          # quads that have a partial sum to load
    e("s_or_b64", sp(sTMP), sp(sQ0), sp(sQ2))
    e("s_and_b64", sp(sTMP), sp(sTMP), sp(sTAKE))
    e("s_or_b64", sp(sTMPB), sp(sQ1), sp(sQ3))
    e("s_and_b64", sp(sTMPB), sp(sTMPB), sp(sTAKE))
    lane_loads(p, HA, True)
    e("s_mov_b64", EXEC, lit(-1))
    # next stage's bookkeeping while the loads fly
    e("s_add_u32", s(sSOFF), s(sSOFF), s(sN))
    e("s_add_u32", s(sN), s(sN), lit(15))
    e("s_lshr_b32", s(sN), s(sN), lit(4))                   # units of the next stage = groups of this one
    e("s_add_u32", s(sCOFF), s(sCOFF), s(sN))
    e("s_mov_b32", s(sUNIT), s(sGRP))
    e("s_mov_b32", s(sLVL), lit(0))
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    # infinity: ZZ (lane 2 of the quad) is literally zero, or the quad has nothing to load
    e("v_or_b32", v(T1), v(HA[0]), v(HA[1]))
    for i in range(2, 14):
        e("v_or_b32", v(T1), v(T1), v(HA[i]))
    e("v_cmp_eq_u32", VCC, lit(0), v(T1))
    e("s_and_b64", sp(sTMP), VCC, sp(sQ2))
    e("s_lshr_b64", sp(sAINF), sp(sTMP), lit(2))            # -> lane 0 of the quad, then spread over its four lanes
    e("s_lshl_b64", sp(sTMP), sp(sAINF), lit(1))
    e("s_or_b64", sp(sAINF), sp(sAINF), sp(sTMP))
    e("s_lshl_b64", sp(sTMP), sp(sAINF), lit(2))
    e("s_or_b64", sp(sAINF), sp(sAINF), sp(sTMP))
    e("s_orn2_b64", sp(sAINF), sp(sAINF), sp(sTAKE))
    e("s_branch", ("label", "C_step%="))
    return p


def clobbers():
    return ", ".join(['"v%d"' % i for i in range(NUM_VGPRS)] + ['"s%d"' % i for i in range(SBASE, NUM_SGPRS) if i not in (32, 33, 34)] +
                     ['"vcc"', '"scc"', '"memory"'])


def render(p):
    lines = ["// generated by tools/gen_coop_asm.py -- do not edit (python tools/gen_coop_asm.py)",
             "// %d instructions, %d of them VALU; VGPRs v0..v%d, SGPRs s%d..s%d" % (
                 sum(1 for i in p.ins if i[0] not in ("label", "comment")), p.count_valu(), NUM_VGPRS - 1, SBASE, NUM_SGPRS - 1)]
    for t in p.text():
        lines.append('"%s\\n"' % t.replace("\\", "\\\\").replace('"', '\\"'))
    return "\n".join(lines) + "\n"


# ---- simulator: one wave of 64 lanes -----------------------------------------------------------------------------------
M64 = (1 << 64) - 1
U64 = np.uint64


class EndProgram(Exception):
    pass


class WaveSim:
    """Executes a Prog for one wave: 64 lanes as numpy vectors, EXEC / VCC and the mask pairs as 64-bit integers. Memory is the
    dictionary `mem` (word address -> 32-bit word) shared between the waves of a run; addresses inside the synthetic table are
    served by `table_read`."""

    def __init__(self, prog, operands, mem, table_read, strict=True):
        self.p = prog
        self.vr = np.zeros((256, 64), dtype=U64)
        self.sr = [0] * 128
        self.vcc = 0
        self.exec = M64
        self.scc = 0
        self.ops = operands
        self.mem, self.table_read = mem, table_read
        self.labels = {a[0]: i for i, (op, a, _) in enumerate(prog.ins) if op == "label"}
        self.valu_executed = 0
        self.executed = 0
        self.pending = {}           # register -> (lane mask array, values): loads not yet waited for
        self.strict = strict
        self.lanes = np.arange(64, dtype=np.int64)
        self.on_label = None        # debugging hook: called with (label, simulator) whenever a label is reached

    # ---- helpers
    def em(self):
        return np.array([(self.exec >> i) & 1 for i in range(64)], dtype=bool)

    @staticmethod
    def to_mask(b):
        return int(sum(1 << i for i in range(64) if b[i]))

    def rv(self, o):
        """a 32-bit source operand as a vector of 64"""
        k = o[0]
        if k == "v":
            assert o[1] not in self.pending, "read of v%d before s_waitcnt" % o[1]
            return self.vr[o[1]]
        if k == "s":
            return np.full(64, self.sr[o[1]], dtype=U64)
        if k == "lit":
            return np.full(64, o[1], dtype=U64)
        if k == "op":
            x = self.ops[o[1]]
            if isinstance(x, np.ndarray):
                return x.astype(U64)
            return np.full(64, x & 0xFFFFFFFF, dtype=U64)
        raise ValueError(o)

    def rv64(self, o):
        k = o[0]
        if k == "vp":
            return self.rv(v(o[1])) | (self.rv(v(o[1] + 1)) << U64(32))
        if k == "lit":
            x = o[1]
            return np.full(64, x if x < 0x80000000 else (x | 0xFFFFFFFF00000000), dtype=U64)
        return np.full(64, self.g64(o), dtype=U64)

    def g32(self, o):
        k = o[0]
        if k == "s":
            return self.sr[o[1]]
        if k == "lit":
            return o[1]
        if k == "op":
            return self.ops[o[1]] & 0xFFFFFFFF
        raise ValueError(o)

    def g64(self, o):
        k = o[0]
        if k == "sp":
            return self.sr[o[1]] | (self.sr[o[1] + 1] << 32)
        if k == "lit":
            x = o[1]
            return x if x < 0x80000000 else (x | 0xFFFFFFFF00000000)
        if k == "vcc":
            return self.vcc
        if k == "exec":
            return self.exec
        if k == "op":
            return self.ops[o[1]] & M64
        raise ValueError(o)

    def pv(self, o, x, wrap=False):
        assert o[0] == "v"
        if not wrap:
            bad = (x >> U64(32)) != 0
            assert not (bad & self.em()).any(), "32-bit result out of range"
        m = self.em()
        self.vr[o[1]] = np.where(m, x & U64(0xFFFFFFFF), self.vr[o[1]])

    def pv64(self, o, x):
        assert o[0] == "vp"
        m = self.em()
        self.vr[o[1]] = np.where(m, x & U64(0xFFFFFFFF), self.vr[o[1]])
        self.vr[o[1] + 1] = np.where(m, x >> U64(32), self.vr[o[1] + 1])

    def p32(self, o, x):
        assert o[0] == "s" and 0 <= x < (1 << 32), (o, x)
        self.sr[o[1]] = x

    def p64(self, o, x):
        x &= M64
        if o[0] == "sp":
            self.sr[o[1]], self.sr[o[1] + 1] = x & 0xFFFFFFFF, x >> 32
        elif o[0] == "vcc":
            self.vcc = x
        elif o[0] == "exec":
            self.exec = x
        else:
            raise ValueError(o)

    def defer(self, reg, mask, vals):
        """a load's result, visible after the next s_waitcnt (several loads may fill different lanes of one register)"""
        if reg in self.pending:
            m0, v0 = self.pending[reg]
            mask, vals = m0 | mask, np.where(mask, vals, v0)
        self.pending[reg] = (mask, vals)

    def rd_words(self, addr, n):
        t = self.table_read(addr, n)
        if t is not None:
            return t
        assert addr % 4 == 0
        return [self.mem.get(addr + 4 * k, 0xDEADBEEF) for k in range(n)]

    def run(self, max_steps=5_000_000):
        try:
            self._run(max_steps)
        except EndProgram:
            pass
        return self.executed

    def _run(self, max_steps):
        pc = 0
        ins = self.p.ins
        while pc < len(ins):
            op, a, kw = ins[pc]
            pc += 1
            if op == "label" and self.on_label:
                self.on_label(a[0], self)
            if op in ("label", "comment", ".p2align"):
                continue
            self.executed += 1
            assert self.executed < max_steps
            if op.startswith("v_"):
                self.valu_executed += 1
            # ---- vector ALU
            if op == "v_mad_u64_u32":
                x, y, c = self.rv(a[2]), self.rv(a[3]), self.rv64(a[4])
                prod = x * y
                r = prod + c
                assert not ((r < prod) & self.em()).any(), "64-bit column overflow"
                self.pv64(a[0], r)
            elif op == "v_mul_lo_u32":
                self.pv(a[0], (self.rv(a[1]) * self.rv(a[2])) & U64(0xFFFFFFFF))
            elif op == "v_and_b32":
                self.pv(a[0], self.rv(a[1]) & self.rv(a[2]))
            elif op == "v_or_b32":
                self.pv(a[0], self.rv(a[1]) | self.rv(a[2]))
            elif op == "v_xor_b32":
                self.pv(a[0], self.rv(a[1]) ^ self.rv(a[2]))
            elif op == "v_lshrrev_b64":
                self.pv64(a[0], self.rv64(a[2]) >> (self.rv(a[1]) & U64(63)))
            elif op == "v_lshrrev_b32":
                self.pv(a[0], self.rv(a[2]) >> (self.rv(a[1]) & U64(31)))
            elif op == "v_lshlrev_b32":
                self.pv(a[0], self.rv(a[2]) << (self.rv(a[1]) & U64(31)), wrap=kw.get("wrap", False))
            elif op == "v_alignbit_b32":
                x = (self.rv(a[1]) << U64(32)) | self.rv(a[2])
                self.pv(a[0], (x >> (self.rv(a[3]) & U64(31))) & U64(0xFFFFFFFF))
            elif op == "v_add_u32":
                self.pv(a[0], self.rv(a[1]) + self.rv(a[2]), wrap=kw.get("wrap", False))
            elif op == "v_add3_u32":
                self.pv(a[0], self.rv(a[1]) + self.rv(a[2]) + self.rv(a[3]))
            elif op == "v_lshl_add_u32":
                self.pv(a[0], (self.rv(a[1]) << self.rv(a[2])) + self.rv(a[3]))
            elif op == "v_sub_u32":
                x, y = self.rv(a[1]), self.rv(a[2])
                if not kw.get("wrap"):
                    assert not ((x < y) & self.em()).any(), "32-bit subtraction went negative"
                self.pv(a[0], (x - y) & U64(0xFFFFFFFF))
            elif op == "v_add_co_u32":
                r = self.rv(a[2]) + self.rv(a[3])
                self.pv(a[0], r, wrap=True)
                self.p64(a[1], self.to_mask(((r >> U64(32)) != 0) & self.em()))
            elif op == "v_addc_co_u32":
                cin = np.array([(self.g64(a[4]) >> i) & 1 for i in range(64)], dtype=U64)
                r = self.rv(a[2]) + self.rv(a[3]) + cin
                self.pv(a[0], r, wrap=True)
                self.p64(a[1], self.to_mask(((r >> U64(32)) != 0) & self.em()))
            elif op == "v_mov_b32":
                self.pv(a[0], self.rv(a[1]))
            elif op == "v_mov_b32_dpp":
                perm = kw["perm"]
                src_lane = (self.lanes & ~3) | np.array([perm[i & 3] for i in range(64)])
                m = self.em()
                assert m[src_lane][m].all(), "DPP read of a lane that EXEC disables"
                self.pv(a[0], self.rv(a[1])[src_lane])
            elif op == "v_cndmask_b32":
                sel = np.array([(self.g64(a[3]) >> i) & 1 for i in range(64)], dtype=bool)
                self.pv(a[0], np.where(sel, self.rv(a[2]), self.rv(a[1])))
            elif op == "v_mad_u32_u24":
                self.pv(a[0], (self.rv(a[1]) & U64(0xFFFFFF)) * (self.rv(a[2]) & U64(0xFFFFFF)) + self.rv(a[3]))
            elif op in ("v_cmp_lt_u32", "v_cmp_gt_u32", "v_cmp_ne_u32", "v_cmp_eq_u32"):
                x, y = self.rv(a[1]), self.rv(a[2])
                r = {"lt": x < y, "gt": x > y, "ne": x != y, "eq": x == y}[op[6:8]]
                self.p64(a[0], self.to_mask(r & self.em()))
            elif op == "v_readfirstlane_b32":
                m = self.em()
                lane = int(np.argmax(m)) if m.any() else 0
                self.p32(a[0], int(self.rv(a[1])[lane]))
            elif op == "ds_bpermute_b32":
                assert self.exec == M64, "ds_bpermute_b32 with lanes disabled"
                src = (self.rv(a[1]) >> U64(2)).astype(np.int64) & 63
                self.defer(a[0][1], self.em(), self.rv(a[2])[src])
            # ---- scalar ALU
            elif op == "s_mov_b32":
                self.p32(a[0], self.g32(a[1]))
            elif op == "s_mov_b64":
                self.p64(a[0], self.g64(a[1]))
            elif op == "s_and_b32":
                r = self.g32(a[1]) & self.g32(a[2])
                self.p32(a[0], r)
                self.scc = int(r != 0)
            elif op == "s_and_b64":
                r = self.g64(a[1]) & self.g64(a[2])
                self.p64(a[0], r)
                self.scc = int(r != 0)
            elif op == "s_or_b64":
                r = self.g64(a[1]) | self.g64(a[2])
                self.p64(a[0], r)
                self.scc = int(r != 0)
            elif op == "s_andn2_b64":
                r = self.g64(a[1]) & ~self.g64(a[2]) & M64
                self.p64(a[0], r)
                self.scc = int(r != 0)
            elif op == "s_orn2_b64":
                r = (self.g64(a[1]) | (~self.g64(a[2]) & M64)) & M64
                self.p64(a[0], r)
                self.scc = int(r != 0)
            elif op == "s_bfe_u32":
                ctl = self.g32(a[2])
                off, width = ctl & 31, (ctl >> 16) & 0x7f
                r = (self.g32(a[1]) >> off) & ((1 << width) - 1)
                self.p32(a[0], r)
                self.scc = int(r != 0)
            elif op == "s_add_u32":
                x = self.g32(a[1]) + self.g32(a[2])
                self.scc = x >> 32
                self.p32(a[0], x & 0xFFFFFFFF)
            elif op == "s_addc_u32":
                x = self.g32(a[1]) + self.g32(a[2]) + self.scc
                self.scc = x >> 32
                self.p32(a[0], x & 0xFFFFFFFF)
            elif op == "s_sub_u32":
                x = self.g32(a[1]) - self.g32(a[2])
                self.scc = 1 if x < 0 else 0
                self.p32(a[0], x & 0xFFFFFFFF)
            elif op == "s_subb_u32":
                x = self.g32(a[1]) - self.g32(a[2]) - self.scc
                self.scc = 1 if x < 0 else 0
                self.p32(a[0], x & 0xFFFFFFFF)
            elif op == "s_min_u32":
                x, y = self.g32(a[1]), self.g32(a[2])
                self.scc = int(x < y)
                self.p32(a[0], min(x, y))
            elif op == "s_mul_i32":
                self.p32(a[0], (self.g32(a[1]) * self.g32(a[2])) & 0xFFFFFFFF)
            elif op == "s_lshl_b32":
                r = (self.g32(a[1]) << (self.g32(a[2]) & 31)) & 0xFFFFFFFF
                self.p32(a[0], r)
                self.scc = int(r != 0)
            elif op == "s_lshr_b32":
                r = self.g32(a[1]) >> (self.g32(a[2]) & 31)
                self.p32(a[0], r)
                self.scc = int(r != 0)
            elif op == "s_lshr_b64":
                r = self.g64(a[1]) >> (self.g32(a[2]) & 63)
                self.p64(a[0], r)
                self.scc = int(r != 0)
            elif op == "s_lshl_b64":
                r = (self.g64(a[1]) << (self.g32(a[2]) & 63)) & M64
                self.p64(a[0], r)
                self.scc = int(r != 0)
            elif op in ("s_cmp_eq_u32", "s_cmp_lt_u32", "s_cmp_gt_u32"):
                x, y = self.g32(a[0]), self.g32(a[1])
                self.scc = int({"eq": x == y, "lt": x < y, "gt": x > y}[op[6:8]])
            elif op == "s_cmp_eq_u64":
                self.scc = int(self.g64(a[0]) == self.g64(a[1]))
            elif op == "s_bitcmp1_b64":
                self.scc = (self.g64(a[0]) >> (self.g32(a[1]) & 63)) & 1
            elif op == "s_cselect_b32":
                self.p32(a[0], self.g32(a[1]) if self.scc else self.g32(a[2]))
            elif op == "s_cbranch_scc0":
                if not self.scc:
                    pc = self.labels[a[0][1]]
            elif op == "s_cbranch_scc1":
                if self.scc:
                    pc = self.labels[a[0][1]]
            elif op == "s_cbranch_execz":
                if not self.exec:
                    pc = self.labels[a[0][1]]
            elif op == "s_cbranch_vccz":
                if not self.vcc:
                    pc = self.labels[a[0][1]]
            elif op == "s_branch":
                pc = self.labels[a[0][1]]
            elif op == "s_endpgm":
                raise EndProgram()
            elif op == "s_waitcnt":
                for r, (m, val) in self.pending.items():
                    self.vr[r] = np.where(m, val, self.vr[r])
                self.pending = {}
            elif op in ("s_nop", "s_sleep"):
                pass
            # ---- memory
            elif op == "s_load_dwordx2":
                w = self.rd_words(self.g64(a[1]) + self.g32(a[2]), 2)
                self.p64(a[0], w[0] | (w[1] << 32))
            elif op in ("global_load_dwordx4", "global_load_dwordx2", "global_load_dword"):
                n = {"x4": 4, "x2": 2}.get(op[-2:], 1)
                m = self.em()
                addr = self.rv64(a[1])
                vals = np.zeros((n, 64), dtype=U64)
                for lane in range(64):
                    if m[lane]:
                        ad = int(addr[lane]) + kw.get("offset", 0)
                        assert ad % (4 * n) == 0 or not self.strict, "load of %d bytes at 0x%x is not naturally aligned" % (4 * n, ad)
                        w = self.rd_words(ad, n)
                        for k in range(n):
                            vals[k, lane] = w[k]
                for k in range(n):
                    self.defer(a[0][1] + k, m, vals[k])
            elif op in ("global_store_dwordx4", "global_store_dwordx2", "global_store_dword"):
                n = {"x4": 4, "x2": 2}.get(op[-2:], 1)
                m = self.em()
                addr = self.rv64(a[0])
                for lane in range(64):
                    if m[lane]:
                        ad = int(addr[lane]) + kw.get("offset", 0)
                        assert ad % (4 * n) == 0 or not self.strict
                        for k in range(n):
                            self.mem[ad + 4 * k] = int(self.rv(v(a[1][1] + k))[lane])
            elif op == "global_atomic_add":
                m = self.em()
                addr = self.rv64(a[1])
                old = np.zeros(64, dtype=U64)
                for lane in range(64):
                    if m[lane]:
                        ad = int(addr[lane])
                        old[lane] = self.mem.get(ad, 0)
                        self.mem[ad] = (int(old[lane]) + int(self.rv(a[2])[lane])) & 0xFFFFFFFF
                self.defer(a[0][1], m, old)
            else:
                raise ValueError("simulator: unknown instruction " + op)


# ---- self-test ---------------------------------------------------------------------------------------------------------
def recode_constant(c, nw):
    """K = (H - 1) sum over the signed windows (all but the top one) of 2^(c j), as eight 32-bit words"""
    k = sum(((1 << (c - 1)) - 1) << (c * j) for j in range(nw - 1))
    return [(k >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


class Problem:
    """a small MSM: `points` points k_i G, scalars of c (nw - 1) + wtop bits, a synthetic table served on demand"""
    TABLE, SCAL, PART, CTR, OUTP, REDO = 0x100000000000, 0x200000000000, 0x300000000000, 0x400000000000, 0x500000000000, 0x600000000000
    WIN_STRIDE = 1 << 36

    def __init__(self, seed, c, nw, wtop, log_points, rpq, row_bytes=128, scalars=None, point_ks=None):
        rnd = random.Random(seed)
        self.c, self.nw, self.wtop, self.lp, self.rpq, self.rb = c, nw, wtop, log_points, rpq, row_bytes
        self.np = 1 << log_points
        self.h, self.htop = 1 << (c - 1), 1 << wtop
        bits = c * (nw - 1) + wtop
        self.scalars = scalars if scalars is not None else [rnd.randrange(0, 1 << bits) for _ in range(self.np)]
        self.ks = point_ks if point_ks is not None else [rnd.randrange(1, 1 << 60) for _ in range(self.np)]
        self.pts = [G.ec_mul(k, G.G1) for k in self.ks]
        self.rnd = rnd
        self.mem = {}
        self.rows = {}
        g = (nw + rpq - 1) // rpq
        self.n0 = self.np * g // 16
        self.row_reads = 0

    def table_read(self, addr, n):
        if self.SCAL <= addr < self.SCAL + 32 * self.np:
            off = addr - self.SCAL
            i, w = off // 32, (off % 32) // 4
            words = [(self.scalars[i] >> (32 * k)) & 0xFFFFFFFF for k in range(8)]
            return words[w:w + n]
        if self.TABLE <= addr < self.TABLE + 8 * 64:
            j = (addr - self.TABLE) // 8
            base = self.TABLE + (j + 1) * self.WIN_STRIDE
            return [base & 0xFFFFFFFF, base >> 32][:n]
        if addr >= self.TABLE + self.WIN_STRIDE and addr < self.SCAL:
            j = (addr - self.TABLE) // self.WIN_STRIDE - 1
            off = addr - (self.TABLE + (j + 1) * self.WIN_STRIDE)
            idx, w = off // self.rb, (off % self.rb) // 4
            hj = self.htop if j == self.nw - 1 else self.h
            i, mag = idx // hj, idx % hj + 1
            assert 0 <= j < self.nw and i < self.np and w + n <= 28, (j, i, w, n)
            key = (j, i, mag)
            if key not in self.rows:
                pt = G.ec_mul(mag << (self.c * j), self.pts[i])
                self.rows[key] = G.limbs(G.to_mont(pt[0]) + P * self.rnd.randrange(0, 2)) + G.limbs(G.to_mont(pt[1]) + P * self.rnd.randrange(0, 2))
            self.row_reads += 1
            return self.rows[key][w:w + n]
        return None

    def want(self):
        acc = None
        for sc, pt in zip(self.scalars, self.pts):
            if sc:
                acc = G.ec_add(acc, G.ec_mul(sc, pt))
        return acc

    def operands(self, unit):
        pack = self.c | (self.nw << 8) | (self.rpq << 16) | (self.lp << 24)
        return [self.TABLE, self.SCAL, self.PART, self.CTR, self.OUTP, self.REDO] + recode_constant(self.c, self.nw) + \
               [pack, self.wtop, self.rb, self.n0, unit, np.arange(64, dtype=U64)]

    def result(self):
        got = [self.mem.get(self.OUTP + 4 * k) for k in range(56)]
        if any(x is None for x in got):
            return "missing"
        if all(x == 0 for x in got):
            return None
        x, y, zz, zzz = (G.from_mont_limbs(got[14 * t:14 * t + 14]) for t in range(4))
        assert (zz ** 3 - zzz ** 2) % P == 0, "ZZ^3 != ZZZ^2"
        # bounds the library's layout promises: X < 14p, Y < 6p, ZZ, ZZZ < 2p, limbs carried
        for t, b in zip(range(4), (14, 6, 2, 2)):
            l = got[14 * t:14 * t + 14]
            assert all(c < (1 << W) for c in l[:13]) and sum(c << (W * i) for i, c in enumerate(l)) < b * P
        return x * pow(zz, -1, P) % P, y * pow(zzz, -1, P) % P


def run_problem(prob, prog=None, order=None, verbose=False):
    """every wave of the launch, one after the other (any order: the hand-off elects whoever arrives last)"""
    prog = prog or build()
    units = list(range(prob.n0))
    if order == "reverse":
        units.reverse()
    elif order == "shuffle":
        random.Random(7).shuffle(units)
    stats = {"waves": 0, "instructions": 0, "valu": 0, "max_wave_valu": 0}
    for u in units:
        sim = WaveSim(prog, prob.operands(u), prob.mem, prob.table_read)
        sim.run()
        stats["waves"] += 1
        stats["instructions"] += sim.executed
        stats["valu"] += sim.valu_executed
        stats["max_wave_valu"] = max(stats["max_wave_valu"], sim.valu_executed)
    assert all(prob.mem.get(prob.CTR + 4 * k, 0) == 0 for k in range(4096)), "a counter was not left at zero"
    return stats


def selftest(verbose=True):
    prog = build()
    cases = [dict(seed=1, c=4, nw=4, wtop=3, log_points=4, rpq=2),        # 2 waves, one hand-off
             dict(seed=2, c=5, nw=3, wtop=4, log_points=5, rpq=2),        # odd window count: a quad with one row
             dict(seed=3, c=4, nw=4, wtop=3, log_points=6, rpq=4, row_bytes=112)]
    for kw in cases:
        prob = Problem(**kw)
        st = run_problem(prob, prog, order="shuffle")
        got, want = prob.result(), prob.want()
        flagged = prob.mem.get(prob.REDO, 0)
        if verbose:
            print("coop selftest %s: %s, %d waves, %d instructions (%d VALU), redo=%d" % (kw, "ok" if got == want else "MISMATCH", st["waves"],
                                                                                        st["instructions"], st["valu"], flagged))
        assert got == want and not flagged


def main():
    if "--selftest" in sys.argv:
        selftest()
        return
    text = render(build())
    if "--check" in sys.argv:
        assert open(OUT).read() == text, "csrc/coop_asm.inc is stale: run python tools/gen_coop_asm.py"
        assert open(OUT.replace(".inc", "_clobbers.inc")).read().split("\n", 1)[1].strip() == clobbers()
        print("coop_asm.inc matches its generator")
        return
    with open(OUT, "w") as f:
        f.write(text)
    with open(OUT.replace(".inc", "_clobbers.inc"), "w") as f:
        f.write("// generated by tools/gen_coop_asm.py -- do not edit\n" + clobbers() + "\n")
    print("wrote %s: %d VALU instructions in the stream, v0..v%d, s%d..s%d" % (OUT, build().count_valu(), NUM_VGPRS - 1, SBASE, NUM_SGPRS - 1))


if __name__ == "__main__":
    main()
