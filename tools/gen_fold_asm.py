#!/usr/bin/env python3
"""Generator of the hand-scheduled lane fold of the direct-table MSM: lambdaworks_kzg_amd/csrc/direct_fold_asm.inc, the body of
k_direct_fold_lanes_asm (direct.hip). Shares the instruction IR, the product chains, the bound bookkeeping and the lane
simulator with tools/gen_direct_asm.py.

    python tools/gen_fold_asm.py            # writes csrc/direct_fold_asm.inc (+ _clobbers.inc)
    python tools/gen_fold_asm.py --check
    python tools/gen_fold_asm.py --selftest

One wave per unit (a unit = the 64 k lane sums one workgroup of k_direct_accumulate_asm stored): every lane first adds its
own k lane sums (loaded from memory), then six shuffle levels (ds_bpermute) fold the 64 lanes; lane 0 stores the sum in the
library's XYZZ layout. The compiler's schedule of the same tree (k_direct_fold_lanes) issues one instruction per ~9 cycles
on its lone wave -- 0.30 ms per 1024 blobs for nine additions; the accumulation stream shows that a lone wave can issue one
per 4.2 (profiles/r03_experiments.md section 6). The addition is add-2008-s on (X, Y, ZZ, ZZZ):

    U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1, ZZp = ZZ1 ZZ2, ZZZp = ZZZ1 ZZZ2      (six independent products)
    P = U2 - U1, R = S2 - S1, PP = P^2, PPP = P PP, Q = U1 PP
    X3 = R^2 - PPP - 2Q,  Y3 = R (Q - X3) - S1 PPP,  ZZ3 = ZZp PP,  ZZZ3 = ZZZp PPP

with every product writing over an operand that is dead by then, three chains interleaved at a time. Infinity is a lane
mask in a scalar pair (a lane sum that never saw a row is stored as zeros): "A at infinity, B not" copies B under EXEC,
"B at infinity" leaves A. P = 0 (the two sums are equal or opposite) raises the blob's redo flag, as in the accumulation.
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_direct_asm as G  # noqa: E402
from gen_direct_asm import (EXEC, MASK, P, VCC, W, Prog, Sim, Val, borrowed, chain_mul, chain_mul_add, chain_sqr, interleave, lit, opnd,  # noqa: E402
                            s, sp, v, vp)

OUT = os.path.join(G.ROOT, "lambdaworks_kzg_amd", "csrc", "direct_fold_asm.inc")

# ---- registers -------------------------------------------------------------------------------------------------------
_n = [0]


def vregs(n, align=1):
    while _n[0] % align:
        _n[0] += 1
    r = list(range(_n[0], _n[0] + n))
    _n[0] += n
    return r


AX, AY, AZZ, AZZZ = vregs(14), vregs(14), vregs(14), vregs(14)      # the running sum (v0 .. v55: stored as it stands)
BX, BY, BZZ, BZZZ = vregs(14, 2), vregs(14), vregs(14), vregs(14)   # the other operand (v56 .. v111: loaded as it is stored)
D1, D2 = vregs(14), vregs(14)
PP, RR2 = vregs(14), vregs(14)
M1, M2, M3 = vregs(14), vregs(14), vregs(14)
ACC1, ACC2, ACC3 = vregs(2, 2), vregs(2, 2), vregs(2, 2)
T1, T2, T3 = vregs(1)[0], vregs(1)[0], vregs(1)[0]
ADDR = vregs(2, 2)
LANE, BPADDR = vregs(1)[0], vregs(1)[0]
NUM_VGPRS = _n[0]
assert NUM_VGPRS <= 240, NUM_VGPRS
A_ALL, B_ALL = AX + AY + AZZ + AZZZ, BX + BY + BZZ + BZZZ

# scalar registers: the constants sit where the product chains of gen_direct_asm expect them
sINV, sMASK, sINVP, sMOD = G.sINV, G.sMASK, G.sINVP, G.sMOD
_s = [max(sMOD) + 1]


def sregs(n=1, align=1):
    while _s[0] % align:
        _s[0] += 1
    r = _s[0] if n == 1 else list(range(_s[0], _s[0] + n))
    _s[0] += n
    return r


sSRC, sOUT, sREDO = sregs(2, 2), sregs(2, 2), sregs(2, 2)
sPER, sLVL, sEND, sD, sTMP1 = sregs(), sregs(), sregs(), sregs(), sregs()
sAINF, sBINF, sTAKE, sTROUBLE, sT2, sT3 = sregs(2, 2), sregs(2, 2), sregs(2, 2), sregs(2, 2), sregs(2, 2), sregs(2, 2)
NUM_SGPRS = _s[0]
SBASE = G.SBASE
assert NUM_SGPRS <= 100

KP4_1, KP8_4, KP16_1 = borrowed(4, 1), borrowed(8, 4), borrowed(16, 1)
KP32_1, KP32_4 = borrowed(32, 1), borrowed(32, 4)
LANE_WORDS = 56
OPERANDS = ["src (this unit's lane sums)", "out (this unit's partial sum)", "redo flag of the blob", "lane sums per thread", "lane",
            "shuffle levels (6: 64 lanes; 0 in the one-lane self-test)"]

# bounds of the stored lane sums (tools/gen_direct_asm.py): -X < 25p carried, -Y < 17p (limbs < 2 2^28), ZZ, ZZZ < 17p
IN_NX, IN_NY, IN_Z = (25, 1), (17, 2), (17, 1)
# an operand of the addition: a converted lane sum or an earlier sum
OP_X = Val(None, 32, 2)      # 32p - (-X) with one unit borrowed per limb; a sum's X3 is < 10p carried
OP_Y = Val(None, 32, 5)      # 32p - (-Y) with four borrowed; a sum's Y3 is < 2p
OP_Z = Val(None, 17, 1)


def load_and_convert(p, dst_x, dst_y, dst_zz, dst_zzz, inf_mask, index_sreg):
    """dst <- lane sum number `index_sreg` of this lane (64 lanes apart in memory), as (X, Y, ZZ, ZZZ); inf_mask <- ZZ == 0"""
    e = p.emit
    # address = src + ((index * 64 + lane) * 224)
    e("s_lshl_b32", s(sTMP1), s(index_sreg), lit(6))
    e("v_add_u32", v(T1), s(sTMP1), v(LANE))
    e("v_mov_b32", v(T2), lit(4 * LANE_WORDS))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(T1), v(T2), sp(sSRC))
    regs = dst_x + dst_y + dst_zz + dst_zzz
    assert regs == list(range(regs[0], regs[0] + 56)) and regs[0] % 2 == 0
    for k in range(14):
        e("global_load_dwordx4", ("v4", regs[0] + 4 * k), vp(ADDR[0]), ("off",), offset=16 * k)
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    e("v_or_b32", v(T1), v(dst_zz[0]), v(dst_zz[1]))
    for i in range(2, 14):
        e("v_or_b32", v(T1), v(T1), v(dst_zz[i]))
    e("v_cmp_eq_u32", sp(inf_mask), lit(0), v(T1))
    for i in range(14):
        e("v_sub_u32", v(dst_x[i]), lit(KP32_1[i]), v(dst_x[i]))
        e("v_sub_u32", v(dst_y[i]), lit(KP32_4[i]), v(dst_y[i]))


def add_body(p):
    """A <- A + B for the lanes of EXEC (neither at infinity)"""
    e = p.emit
    ax, ay = Val(AX, OP_X.B, OP_X.L), Val(AY, OP_Y.B, OP_Y.L)
    bx, by = Val(BX, OP_X.B, OP_X.L), Val(BY, OP_Y.B, OP_Y.L)
    azz, azzz, bzz, bzzz = (Val(r, OP_Z.B, OP_Z.L) for r in (AZZ, AZZZ, BZZ, BZZZ))
    # group 1: U1 -> AX, U2 -> BX, S1 -> AY        (X1, X2, Y1 are not needed again)
    interleave(p, chain_mul(ax, bzz, AX, M1, ACC1, T1), chain_mul(bx, azz, BX, M2, ACC2, T2), chain_mul(ay, bzzz, AY, M3, ACC3, T3))
    # group 2: S2 -> BY, ZZp -> AZZ, ZZZp -> BZZZ   (ZZZp may not overwrite AZZZ: S2 reads it in this group)
    interleave(p, chain_mul(by, azzz, BY, M1, ACC1, T1), chain_mul(azz, bzz, AZZ, M2, ACC2, T2), chain_mul(azzz, bzzz, BZZZ, M3, ACC3, T3))
    u1, s1, zzp, zzzp = Val(AX, 2, 1), Val(AY, 2, 1), Val(AZZ, 2, 1), Val(BZZZ, 2, 1)
    # P = U2 - U1 + 4p -> BX, R = S2 - S1 + 4p -> BY
    for i in range(14):
        e("v_add_u32", v(BX[i]), lit(KP4_1[i]), v(BX[i]))
        e("v_sub_u32", v(BX[i]), v(BX[i]), v(AX[i]))
        e("v_add_u32", v(BY[i]), lit(KP4_1[i]), v(BY[i]))
        e("v_sub_u32", v(BY[i]), v(BY[i]), v(AY[i]))
    pv, rv = Val(BX, 2 + 4, 1 + 2), Val(BY, 2 + 4, 1 + 2)
    # P == 0 mod p (equal or opposite points)? value = k p, k < 8: low 56 bits, 28 at a time (as in the accumulation)
    e("v_mul_lo_u32", v(T1), v(BX[0]), s(sINVP))
    e("v_and_b32", v(T1), s(sMASK), v(T1))
    e("v_cmp_gt_u32", VCC, lit(8), v(T1))
    e("s_cbranch_vccz", ("label", "F_no_cand%="))
    e("s_mov_b64", sp(sT2), VCC)
    e("v_mul_lo_u32", v(T2), v(T1), s(sMOD[0]))
    e("v_lshrrev_b32", v(T2), lit(W), v(T2))
    e("v_mul_lo_u32", v(T3), v(T1), s(sMOD[1]))
    e("v_add_u32", v(T2), v(T2), v(T3))
    e("v_lshrrev_b32", v(T3), lit(W), v(BX[0]))
    e("v_add_u32", v(T3), v(T3), v(BX[1]))
    e("v_xor_b32", v(T2), v(T2), v(T3))
    e("v_and_b32", v(T2), s(sMASK), v(T2))
    e("v_cmp_eq_u32", VCC, lit(0), v(T2))
    e("s_and_b64", sp(sT2), sp(sT2), VCC)
    e("s_or_b64", sp(sTROUBLE), sp(sTROUBLE), sp(sT2))
    p.label("F_no_cand%=")
    # group 3: PP = P^2 -> PP, R^2 -> RR2
    for i in range(14):
        e("v_lshlrev_b32", v(D1[i]), lit(1), v(BX[i]))
        e("v_lshlrev_b32", v(D2[i]), lit(1), v(BY[i]))
    interleave(p, chain_sqr(pv, D1, PP, M1, ACC1, T1), chain_sqr(rv, D2, RR2, M2, ACC2, T2))
    pp, rr2 = Val(PP, 2, 1), Val(RR2, 2, 1)
    # group 4: PPP = P PP -> BX, Q = U1 PP -> AX, ZZ3 = ZZp PP -> AZZ
    interleave(p, chain_mul(pv, pp, BX, M1, ACC1, T1), chain_mul(u1, pp, AX, M2, ACC2, T2), chain_mul(zzp, pp, AZZ, M3, ACC3, T3))
    ppp, q = Val(BX, 2, 1), Val(AX, 2, 1)
    # X3 = R^2 - PPP - 2Q + 8p, carried -> RR2 (R^2 is dead once its limb is read)
    for i in range(14):
        e("v_lshl_add_u32", v(T1), v(AX[i]), lit(1), v(BX[i]))           # 2Q + PPP           < 3 2^28
        e("v_sub_u32", v(T1), lit(KP8_4[i]), v(T1))
        if i == 0:
            e("v_add_u32", v(RR2[i]), v(RR2[i]), v(T1))
        else:
            e("v_add3_u32", v(RR2[i]), v(RR2[i]), v(T1), v(T2))
        if i < 13:
            e("v_lshrrev_b32", v(T2), lit(W), v(RR2[i]))
            e("v_and_b32", v(RR2[i]), s(sMASK), v(RR2[i]))
    x3 = Val(RR2, 2 + 8, 1)
    # t1 = Q - X3 + 16p -> AX,  t2 = 4p - PPP -> D1  (PPP itself is still needed for ZZZ3)
    for i in range(14):
        e("v_add_u32", v(AX[i]), lit(KP16_1[i]), v(AX[i]))
        e("v_sub_u32", v(AX[i]), v(AX[i]), v(RR2[i]))
        e("v_sub_u32", v(D1[i]), lit(KP4_1[i]), v(BX[i]))
    t1, t2 = Val(AX, 2 + 16, 1 + 2), Val(D1, 4, 2)
    # group 5: ZZZ3 = ZZZp PPP -> AZZZ, Y3 = R t1 + S1 t2 -> AY (over S1, limb by limb)
    interleave(p, chain_mul(zzzp, ppp, AZZZ, M1, ACC1, T1), chain_mul_add(rv, t1, s1, t2, AY, M2, ACC2, T2))
    for i in range(14):
        e("v_mov_b32", v(AX[i]), v(RR2[i]))
    assert x3.B <= OP_X.B and x3.L <= OP_X.L


def build():
    p = Prog()
    e = p.emit
    e("s_mov_b64", sp(sSRC), opnd(0))
    e("s_mov_b64", sp(sOUT), opnd(1))
    e("s_mov_b64", sp(sREDO), opnd(2))
    e("s_mov_b32", s(sPER), opnd(3))
    e("v_mov_b32", v(LANE), opnd(4))
    for i in range(14):
        e("s_mov_b32", s(sMOD[i]), lit(G.MOD[i]))
    e("s_mov_b32", s(sINV), lit(G.INV))
    e("s_mov_b32", s(sMASK), lit(MASK))
    e("s_mov_b32", s(sINVP), lit(G.INVP))
    e("s_mov_b64", sp(sTROUBLE), lit(0))
    e("s_mov_b32", s(sEND), opnd(5))
    e("s_add_u32", s(sEND), s(sPER), s(sEND))          # levels: per-thread sums 1 .. PER-1, then six shuffle levels
    e("s_mov_b32", s(sLVL), lit(0))
    load_and_convert(p, AX, AY, AZZ, AZZZ, sAINF, sLVL)
    e("s_mov_b32", s(sLVL), lit(1))
    p.label("F_loop%=")
    e("s_cmp_lt_u32", s(sLVL), s(sPER))
    e("s_cbranch_scc0", ("label", "F_shuffle%="))
    load_and_convert(p, BX, BY, BZZ, BZZZ, sBINF, sLVL)
    e("s_mov_b64", sp(sTAKE), lit(-1))
    e("s_branch", ("label", "F_have_b%="))
    p.label("F_shuffle%=")
    # B <- A of lane + d, d = 32 >> (level - PER); lanes below d take part
    e("s_sub_u32", s(sTMP1), s(sLVL), s(sPER))
    e("s_lshr_b32", s(sD), lit(32), s(sTMP1))
    e("v_add_u32", v(BPADDR), s(sD), v(LANE))
    e("v_and_b32", v(BPADDR), lit(63), v(BPADDR))
    e("v_lshlrev_b32", v(BPADDR), lit(2), v(BPADDR))
    for a, b in zip(A_ALL, B_ALL):
        e("ds_bpermute_b32", v(b), v(BPADDR), v(a))
    e("s_lshr_b64", sp(sBINF), sp(sAINF), s(sD))
    e("s_lshl_b64", sp(sTAKE), lit(1), s(sD))
    e("s_sub_u32", s(sTAKE[0]), s(sTAKE[0]), lit(1))
    e("s_subb_u32", s(sTAKE[1]), s(sTAKE[1]), lit(0))
    e("s_waitcnt", ("raw", "lgkmcnt(0)"))
    p.label("F_have_b%=")
    # A at infinity, B not: A <- B
    e("s_andn2_b64", sp(sT2), sp(sAINF), sp(sBINF))
    e("s_and_b64", sp(sT2), sp(sT2), sp(sTAKE))
    e("s_mov_b64", EXEC, sp(sT2))
    e("s_cbranch_execz", ("label", "F_no_copy%="))
    for a, b in zip(A_ALL, B_ALL):
        e("v_mov_b32", v(a), v(b))
    p.label("F_no_copy%=")
    # the lanes that add: neither at infinity
    e("s_or_b64", sp(sT3), sp(sAINF), sp(sBINF))
    e("s_andn2_b64", sp(sT3), sp(sTAKE), sp(sT3))
    # new infinity flags of the taking lanes: A and B at infinity
    e("s_andn2_b64", sp(sT2), sp(sTAKE), sp(sBINF))        # taking lanes whose B is a point: A is a point afterwards
    e("s_andn2_b64", sp(sAINF), sp(sAINF), sp(sT2))
    e("s_mov_b64", EXEC, sp(sT3))
    e("s_cbranch_execz", ("label", "F_no_add%="))
    add_body(p)
    p.label("F_no_add%=")
    e("s_mov_b64", EXEC, lit(-1))
    e("s_add_u32", s(sLVL), s(sLVL), lit(1))
    e("s_cmp_lt_u32", s(sLVL), s(sEND))
    e("s_cbranch_scc1", ("label", "F_loop%="))
    # ---- lane 0 holds the unit's sum: bring every coordinate under 2p (a lone lane sum is only < 32p) and store
    # (one product by the Montgomery form of one, R mod p: x (R mod p) / R = x mod p, weakly reduced)
    one = Val(D2, 1, 1)
    for i in range(14):
        e("v_mov_b32", v(D2[i]), lit(G.R1[i]))
    interleave(p, chain_mul(Val(AX, OP_X.B, OP_X.L), one, AX, M1, ACC1, T1), chain_mul(Val(AY, OP_Y.B, OP_Y.L), one, AY, M2, ACC2, T2),
               chain_mul(Val(AZZ, OP_Z.B, OP_Z.L), one, AZZ, M3, ACC3, T3))
    for it in chain_mul(Val(AZZZ, OP_Z.B, OP_Z.L), one, AZZZ, M1, ACC1, T1):
        e(it[0], *it[1], **it[2])
    e("s_and_b64", sp(sT2), sp(sAINF), lit(1))
    e("s_mov_b64", EXEC, sp(sT2))
    e("s_cbranch_execz", ("label", "F_not_inf%="))
    for r in A_ALL:
        e("v_mov_b32", v(r), lit(0))
    p.label("F_not_inf%=")
    e("s_mov_b64", EXEC, lit(1))
    e("v_mov_b32", v(ADDR[0]), s(sOUT[0]))
    e("v_mov_b32", v(ADDR[1]), s(sOUT[1]))
    for k in range(14):
        e("global_store_dwordx4", vp(ADDR[0]), ("v4", 4 * k), ("off",), offset=16 * k)
    e("s_nop", ("raw", "1"))
    e("s_cmp_eq_u64", sp(sTROUBLE), lit(0))
    e("s_cbranch_scc1", ("label", "F_done%="))
    e("v_mov_b32", v(T1), lit(1))
    e("v_mov_b32", v(ADDR[0]), s(sREDO[0]))
    e("v_mov_b32", v(ADDR[1]), s(sREDO[1]))
    e("global_store_dword", vp(ADDR[0]), v(T1), ("off",))
    e("s_nop", ("raw", "1"))
    p.label("F_done%=")
    e("s_mov_b64", EXEC, lit(-1))
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    return p


# Accumulator registers nobody touches, listed as clobbers so that the kernel descriptor allocates them: with more than 256
# registers per lane a SIMD holds ONE wave of this kernel. The fold is one dependent chain per unit and wants every issue slot
# of its SIMD; left to itself the dispatcher puts two such waves on some SIMDs and none on others, and the pairs set the kernel's
# duration (time stamps: waves 139 us on average, 238 us the slowest; profiles/r03_experiments.md section 6).
OCCUPANCY_PAD_AGPRS = 64


def clobbers(pad=False):
    return ", ".join(['"v%d"' % i for i in range(NUM_VGPRS)] + (['"a%d"' % i for i in range(OCCUPANCY_PAD_AGPRS)] if pad else []) +
                     ['"s%d"' % i for i in range(SBASE, NUM_SGPRS) if i not in (32, 33, 34)] + ['"vcc"', '"memory"'])


def render(p):
    lines = ["// generated by tools/gen_fold_asm.py -- do not edit (python tools/gen_fold_asm.py)",
             "// %d instructions, %d of them VALU; VGPRs v0..v%d, SGPRs s%d..s%d" % (
                 sum(1 for i in p.ins if i[0] not in ("label", "comment")), p.count_valu(), NUM_VGPRS - 1, SBASE, NUM_SGPRS - 1)]
    for t in p.text():
        lines.append('"%s\\n"' % t.replace("\\", "\\\\").replace('"', '\\"'))
    return "\n".join(lines) + "\n"


# ---- self-test: one lane adds `per` stored lane sums through the memory path -------------------------------------------
def selftest(seed=1, per=5, verbose=True, with_inf=False, equal=False):
    """(the shuffle levels need the other lanes: on the simulated lane they run with B = A of a lane at infinity, which the
    simulator models as `take` = 0 by giving ds_bpermute no data; the GPU parity tests cover them)"""
    rnd = random.Random(seed)
    prog = build()
    pts, words = [], []
    for k in range(per):
        if with_inf and k in (0, 2):
            pts.append(None)
            words.append([0] * 56)
            continue
        pt = G.ec_mul(3 if equal else rnd.randrange(2, 1 << 40), G.G1)
        t = rnd.randrange(1, P)                                    # ZZ = t^2, ZZZ = t^3
        zz, zzz = t * t % P, t * t * t % P
        X, Y = pt[0] * zz % P, pt[1] * zzz % P
        # as the accumulation stores them: -X < 25p, -Y, ZZ, ZZZ < 17p, Montgomery form, carried limbs
        words.append(G.limbs(G.to_mont((-X) % P) + P * rnd.randrange(0, 20)) + G.limbs(G.to_mont((-Y) % P) + P * rnd.randrange(0, 12)) +
                     G.limbs(G.to_mont(zz) + P * rnd.randrange(0, 12)) + G.limbs(G.to_mont(zzz) + P * rnd.randrange(0, 12)))
        pts.append(pt)
    src, out, redo = 0x100000000000, 0x200000000000, 0x300000000000
    lane = 0
    stored = {}

    def rd(addr, n):
        off = addr - src
        idx, w = off // (4 * 56), (off % (4 * 56)) // 4
        k, ln = idx // 64, idx % 64
        assert ln == lane
        return words[k][w:w + n]

    def wr(addr, ws):
        for k, x in enumerate(ws):
            stored[addr + 4 * k] = x

    class FoldSim(Sim):
        pass

    sim = FoldSim(prog, [src, out, redo, per, lane, 0], rd, wr)
    steps = sim.run()
    got = [stored[out + 4 * k] for k in range(56)]
    want = None
    for pt in pts:
        want = G.ec_add(want, pt)
    flagged = stored.get(redo, 0)
    if equal:
        return flagged
    if want is None:
        ok = all(x == 0 for x in got)
    else:
        x, y, zz, zzz = (G.from_mont_limbs(got[14 * t:14 * t + 14]) for t in range(4))
        ok = (x * pow(zz, -1, P) % P, y * pow(zzz, -1, P) % P) == want and (zz ** 3 - zzz ** 2) % P == 0
        ok = ok and all(sum(c << (W * i) for i, c in enumerate(got[14 * t:14 * t + 14])) < 2 * P for t in range(4))
    if verbose:
        print("fold selftest seed=%d per=%d inf=%s: %s, %d instructions executed, %d VALU, redo=%d" % (seed, per, with_inf, "ok" if ok else "MISMATCH",
                                                                                                    steps, sim.valu_executed, flagged))
    assert ok and not flagged
    return sim.valu_executed


def main():
    if "--selftest" in sys.argv:
        selftest(1, 4)
        selftest(2, 7)
        selftest(3, 5, with_inf=True)
        assert selftest(4, 3, verbose=False, equal=True) == 1
        print("equal points raise the redo flag")
        return
    text = render(build())
    if "--check" in sys.argv:
        assert open(OUT).read() == text, "csrc/direct_fold_asm.inc is stale: run python tools/gen_fold_asm.py"
        assert open(OUT.replace(".inc", "_clobbers.inc")).read().split("\n", 1)[1].strip() == clobbers()
        assert open(OUT.replace(".inc", "_clobbers_pad.inc")).read().split("\n", 1)[1].strip() == clobbers(True)
        print("direct_fold_asm.inc matches its generator")
        return
    with open(OUT, "w") as f:
        f.write(text)
    with open(OUT.replace(".inc", "_clobbers.inc"), "w") as f:
        f.write("// generated by tools/gen_fold_asm.py -- do not edit\n" + clobbers() + "\n")
    with open(OUT.replace(".inc", "_clobbers_pad.inc"), "w") as f:
        f.write("// generated by tools/gen_fold_asm.py -- do not edit (one wave per SIMD: see OCCUPANCY_PAD_AGPRS)\n" + clobbers(True) + "\n")
    print("wrote %s: %d VALU instructions in the stream, v0..v%d, s%d..s%d" % (OUT, build().count_valu(), NUM_VGPRS - 1, SBASE, NUM_SGPRS - 1))


if __name__ == "__main__":
    main()
