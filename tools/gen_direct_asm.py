#!/usr/bin/env python3
"""Generator (and lane-level simulator) of the hand-scheduled gfx950 loop of the direct-table MSM:
lambdaworks_kzg_amd/csrc/direct_asm.inc, the body of k_direct_accumulate_asm (direct.hip).

    python tools/gen_direct_asm.py            # writes csrc/direct_asm.inc
    python tools/gen_direct_asm.py --check    # the committed .inc is what this script writes
    python tools/gen_direct_asm.py --selftest # runs the instruction stream on a simulated lane against big-int arithmetic

What the hand-written loop does differently from the compiler's schedule of field29.cuh / g1.cuh:
  * a fixed register map: the accumulator, the row buffer and every temporary live where this file puts them, so the
    loop has no moves at its merge points and the next row's seven global_load_dwordx4 land in the registers the
    current row vacates after its two products;
  * every Montgomery product is ONE dependent chain of v_mad_u64_u32 per column (no second chain, no 64-bit merge add);
    the instruction-level parallelism comes from interleaving two or three INDEPENDENT products instruction by
    instruction: (X2 ZZ1, Y2 ZZZ1), (P^2, R^2), (P PP, X1 PP), (ZZ1 PP, ZZZ1 PPP, the fused pair of Y3);
  * the accumulator keeps -X and -Y, so P = U2 - X1 and R = S2 - Y1 are plain limb-wise additions, and Y3's fused
    product pair takes -Y1 as it is;
  * all lanes of a wave walk (scalar, window) in lockstep: window shifts, masks and table bases are scalar registers,
    the lane's scalar lives in eight VGPRs that are shifted down by the window width after every digit (no LDS);
  * P = +-Q and an accumulator at infinity never reach the addition formulas: the first row of a lane is copied in
    under an EXEC mask, and a lane whose P = U2 - X1 vanishes mod 2^56 (probability 2^-52 per addition on honest
    data) raises the blob's `redo` flag -- the C++ kernel then recomputes that blob with its complete branches.
    A false alarm costs time, never correctness.

Bounds (B: value < B p, L: limbs 0..12 < L 2^28), the same bookkeeping field29.cuh does in its types; `Val` below carries
them through the generator and asserts what operator* static_asserts: A B <= 2520 and column sums < 2^64.

The simulator executes the SAME instruction list the emitter prints (one lane, 64-bit registers modelled exactly,
every 32-bit add checked against wrap-around where the algorithm relies on none) -- tests/test_direct_asm_cpu.py.
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "lambdaworks_kzg_amd", "csrc", "direct_asm.inc")

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
W, N = 28, 14
MASK = (1 << W) - 1
RMONT = 1 << (W * N)
INV = (-pow(P, -1, 1 << W)) % (1 << W)        # -p^-1 mod 2^28
INVP = pow(P, -1, 1 << W)                     # p^-1 mod 2^28


def limbs(x):
    out = []
    for _ in range(N - 1):
        out.append(x & MASK)
        x >>= W
    out.append(x)
    assert x < (1 << 32)
    return out


def borrowed(k, br):
    """k p with every limb but the top one raised by br 2^28 and the next one lowered by br (tools/gen_field_consts.py)."""
    g = limbs(k * P)
    for i in range(N - 1):
        g[i] += br << W
        g[i + 1] -= br
    assert sum(x << (W * i) for i, x in enumerate(g)) == k * P and all(0 <= x < (1 << 32) for x in g)
    return g


MOD = limbs(P)
R1 = limbs(RMONT % P)

# experiment knobs (A/B builds only; the committed .inc is generated with none of them set)
KNOB_SDST = int(os.environ.get("LWK_ASM_SDST", "0"))      # 1: every chain writes its multiply-add carry-out to a scalar pair of its own
KNOB_BLOCK = int(os.environ.get("LWK_ASM_BLOCK", "1"))    # instructions a chain issues before the next chain takes its turn
KNOB_CONFINE = int(os.environ.get("LWK_ASM_CONFINE", "0"))   # 1: WRONG RESULTS -- every gather lands in a 1 MB part of the table (8 points x 64 rows x the windows): what does the memory side cost?
KNOB_ROTWIN = int(os.environ.get("LWK_ASM_ROTWIN", "0"))     # 1: WRONG RESULTS -- workgroup g reads window (j + g) mod (nw - 1)'s allocation for its window j (all but the top one):
                                                              # the same instructions and gathers, spread over the windows' allocations instead of all in one at a time
KNOB_E64 = int(os.environ.get("LWK_ASM_E64", "1"))           # 1 (shipped): every vector instruction of the loop in an 8-byte encoding (4-byte VOP1 / VOP2 / VOPC forms take their _e64 one) and the
                                                              # vector runs start 8-byte aligned, so no multiply-add sits at 4 mod 8 or straddles a fetch line: -3.3 % time, same box
                                                              # (0: 2340 of the 3548 multiply-adds at 4 mod 8, 340 instructions across a 64-byte line; profiles/r03_experiments.md section 9)
KNOB_PAUSE = os.environ.get("LWK_ASM_PAUSE", "")          # "nop_top" / "sleep_top": s_nop 15 / s_sleep 1 at the head of the loop; "nop_groups": s_nop 15 in front of every product group

# ---- register map ------------------------------------------------------------------------------------------------
V = {}
_next = [0]


def vregs(name, n, align=1):
    while _next[0] % align:
        _next[0] += 1
    V[name] = list(range(_next[0], _next[0] + n))
    _next[0] += n
    return V[name]


NX = vregs("NX", 14)        # -X of the accumulator   (B <= 25, L = 1)
NY = vregs("NY", 14)        # -Y                      (B <= 17, L = 1)
ZZ = vregs("ZZ", 14)
ZZZ = vregs("ZZZ", 14)
ROW = vregs("ROW", 28, 2)   # the gathered table row: QX = ROW[0:14], QY = ROW[14:28]
QX, QY = ROW[:14], ROW[14:]
M1 = vregs("M1", 14)
M2 = vregs("M2", 14)
U = vregs("U", 14)          # U2 -> P -> (K - PPP)
S = vregs("S", 14)          # S2 -> R
D1 = vregs("D1", 14)        # 2P -> PPP
D2 = vregs("D2", 14)        # 2R -> -Q -> (-Q - X3' + K)
PP = vregs("PP", 14)
RR2 = vregs("RR2", 14)      # R^2 -> m digits of the third chain
ACC1 = vregs("ACC1", 2, 2)
ACC2 = vregs("ACC2", 2, 2)
ACC3 = vregs("ACC3", 2, 2)
T1 = vregs("T1", 1)[0]
T2r = vregs("T2", 1)[0]
T3 = vregs("T3", 1)[0]
SC = vregs("SC", 8, 2)      # the lane's current scalar, shifted down by C after every digit
SCN = vregs("SCN", 8, 2)    # the next scalar (prefetched)
POINT = vregs("POINT", 1)[0]
CARRY = vregs("CARRY", 1)[0]
ADDR = vregs("ADDR", 2, 2)
MAG = vregs("MAG", 1)[0]
RAW = vregs("RAW", 1)[0]
VRB = vregs("VRB", 1)[0]    # row_bytes
VTID = vregs("VTID", 1)[0]
NUM_VGPRS = _next[0]
assert NUM_VGPRS <= 232, NUM_VGPRS

SBASE = 28                   # s0 .. s27 stay with the compiler (the 18 scalar registers of the statement's operands live there)
S_ = {}
_snext = [SBASE]


def sregs(name, n=1, align=1):
    while _snext[0] % align or any(r in (32, 33, 34) for r in range(_snext[0], _snext[0] + n)):   # s32..s34: SP / FP / BP of the ABI
        _snext[0] += 1
    S_[name] = _snext[0] if n == 1 else list(range(_snext[0], _snext[0] + n))
    _snext[0] += n
    return S_[name]


sINV, sMASK, sINVP, sMORE = sregs("INV"), sregs("MASK"), sregs("INVP"), sregs("MORE")     # (these four fill s28..s31)
sMOD = sregs("MOD", 14)
sTABLE = sregs("TABLE", 2, 2)
sSC = sregs("SCPTR", 2, 2)
sOUT = sregs("OUT", 2, 2)
sREDO = sregs("REDO", 2, 2)
sSPL, sLPB, sC, sNW, sH, sHTOP, s2C, sMASKC, sMASKTOP = [sregs(n) for n in
                                                           ("SPL", "LPB", "C", "NW", "H", "HTOP", "TWOC", "MASKC", "MASKTOP")]
sRB = sregs("RB")
sQ, sJ = sregs("Q"), sregs("J")
sVALID = sregs("VALID", 2, 2)
sNEG = sregs("NEG", 2, 2)
sVALIDN = sregs("VALIDN", 2, 2)
sNEGN = sregs("NEGN", 2, 2)
sINF = sregs("INF", 2, 2)
sTROUBLE = sregs("TROUBLE", 2, 2)
sACT = sregs("ACT", 2, 2)
sTMP = sregs("TMP", 2, 2)
sTMPB = sregs("TMPB", 2, 2)
sHCMP, sHJ, sMASKJ, sSTMP = sregs("HCMP"), sregs("HJ"), sregs("MASKJ"), sregs("STMP")
sCARRY2 = sregs("CARRY2", 2, 2) if KNOB_SDST else None
sROT = sregs("ROT") if KNOB_ROTWIN else None
NUM_SGPRS = _snext[0]
assert NUM_SGPRS <= 102, NUM_SGPRS

# operands of the asm statement, in this order (direct.hip): all "s" except the last two ("v")
OPERANDS = ["window base addresses (device array)", "scalars", "out", "redo", "spl", "lpb", "c", "nw", "wtop", "h", "htop", "row_bytes",
            "first", "tid"]


# ---- IR ----------------------------------------------------------------------------------------------------------
def v(n):
    return ("v", n)


def vp(n):
    assert n % 2 == 0
    return ("vp", n)


def s(n):
    return ("s", n)


def sp(n):
    assert isinstance(n, list) or n % 2 == 0
    return ("sp", n[0] if isinstance(n, list) else n)


def lit(x):
    return ("lit", x & 0xFFFFFFFF)


def opnd(i):
    return ("op", i)


VCC, EXEC = ("vcc",), ("exec",)


def fmt(o):
    k = o[0]
    if k == "raw":      # s_waitcnt / s_nop / .p2align arguments, printed as they are
        return o[1]
    if k == "v":
        return "v%d" % o[1]
    if k == "vp":
        return "v[%d:%d]" % (o[1], o[1] + 1)
    if k == "v4":
        return "v[%d:%d]" % (o[1], o[1] + 3)
    if k == "s":
        return "s%d" % o[1]
    if k == "sp":
        return "s[%d:%d]" % (o[1], o[1] + 1)
    if k == "lit":
        x = o[1]
        return str(x) if x <= 64 else ("-1" if x == 0xFFFFFFFF else "0x%x" % x)
    if k == "op":
        return "%%%d" % o[1]
    if k == "vcc":
        return "vcc"
    if k == "exec":
        return "exec"
    if k == "off":
        return "off"
    if k == "label":
        return o[1]
    raise ValueError(o)


# vector instructions with a 4-byte encoding (VOP1 / VOP2 / VOPC) that the generator uses; a 32-bit literal makes them 8 bytes anyway (and VOP3 takes none)
E64_OPS = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_mov_b32", "v_cndmask_b32",
           "v_cmp_lt_u32", "v_cmp_gt_u32", "v_cmp_ne_u32", "v_cmp_eq_u32"}


class Prog:
    def __init__(self):
        self.ins = []

    def emit(self, op, *args, **kw):
        self.ins.append((op, args, kw))

    def label(self, name):
        self.ins.append(("label", (name,), {}))

    def text(self):
        out = []
        prev_vector = False
        for op, args, kw in self.ins:
            if KNOB_E64 and op.startswith("v_") and not prev_vector and out:
                out.append("  .p2align 3")        # a run of vector instructions starts 8-byte aligned (the scalar ones in front of it are 4 bytes each)
            if op not in ("label", "comment"):
                prev_vector = op.startswith("v_")
            if op == "label":
                out.append("%s:" % args[0])
                continue
            if op == "comment":
                out.append("; " + args[0])
                continue
            line = op
            if kw.get("e64") or (KNOB_E64 and op in E64_OPS and not any(a[0] == "op" or (a[0] == "lit" and 64 < a[1] < 0xFFFFFFF0) for a in args)):
                line += "_e64"
            if args:
                line += " " + ", ".join(fmt(a) for a in args)
            if "offset" in kw and kw["offset"]:
                line += " offset:%d" % kw["offset"]
            out.append("  " + line)
        return out

    def count_valu(self):
        return sum(1 for op, _, _ in self.ins if op.startswith("v_"))


# ---- value bookkeeping (what field29.cuh keeps in its types) -------------------------------------------------------
class Val:
    def __init__(self, regs, B, L):
        self.r, self.B, self.L = regs, B, L


# A product whose reduction digits m_k are NOT masked to 28 bits ("wide"): m_k = (column * -p^-1) mod 2^32 clears the
# column's low 28 bits just the same, and its upper four bits add a multiple of p one column up -- one instruction less
# per column, a result < 17p instead of < 2p (the callers' bounds say where that is affordable).
WIDE_B, NARROW_B = 17, 2


def check_columns(terms_ll, wide):
    """exact bound of every 64-bit column: terms_ll = [La Lb per product pair], all limbs at their bound"""
    mmax = (1 << 32) - 1 if wide else MASK
    carry = 0
    for k in range(27):
        lo, hi = max(0, k - 13), min(k, 13)
        n = hi - lo + 1
        tot = carry + sum(n * ll * (1 << W) * (1 << W) for ll in terms_ll)
        tot += sum(mmax * MOD[k - i] for i in range(lo, hi + 1) if i <= 13 and (k < 14 or i >= k - 13))
        assert tot < (1 << 64), ("64-bit column", k, terms_ll, wide)
        carry = tot >> W


def chain_mul(a, b, out, m, acc, tmp, wide=False):
    """instruction list of one Montgomery product out = a b / R (product scanning, one dependent chain)."""
    assert a.B * b.B <= 2520, (a.B, b.B)
    check_columns([a.L * b.L], wide)
    return _chain([(a.r, b.r)], None, out, m, acc, tmp, wide)


def chain_sqr(a, dbl, out, m, acc, tmp, wide=False):
    """out = a^2 / R with the cross products taken once against dbl = 2a (dbl registers must already hold 2a)."""
    assert a.B * a.B <= 2520 and 2 * a.L <= 15, (a.B, a.L)     # (the doubled limbs must fit 32 bits)
    check_columns([a.L * a.L], wide)
    return _chain([], (a.r, dbl), out, m, acc, tmp, wide)


def chain_mul_add(a, b, c, d, out, m, acc, tmp, wide=False):
    assert a.B * b.B + c.B * d.B <= 2520, (a.B * b.B + c.B * d.B)
    check_columns([a.L * b.L, c.L * d.L], wide)
    return _chain([(a.r, b.r), (c.r, d.r)], None, out, m, acc, tmp, wide)


def _chain(pairs, square, out, m, acc, tmp, wide=False):
    ins = []
    A = vp(acc[0])
    first = [True]
    sink = VCC
    if KNOB_SDST and acc[0] == ACC2[0]:
        sink = sp(sCARRY2)
    if KNOB_SDST and acc[0] == ACC3[0]:
        sink = sp(sTMP)

    # (the modulus limb goes FIRST: a scalar register as the first source of v_mad_u64_u32 runs 3 % faster than as the second, tools/ubench_sustain.hip)
    def mad(x, y):
        ins.append(("v_mad_u64_u32", (A, sink, x, y, lit(0) if first[0] else A), {}))
        first[0] = False

    def column(k):
        lo, hi = max(0, k - 13), min(k, 13)
        for a, b in pairs:
            for i in range(lo, hi + 1):
                mad(v(a[i]), v(b[k - i]))
        if square:
            a, d = square
            for i in range(lo, hi + 1):
                if 2 * i < k:
                    mad(v(d[i]), v(a[k - i]))
            if k % 2 == 0 and lo <= k // 2 <= hi:
                mad(v(a[k // 2]), v(a[k // 2]))

    for k in range(14):
        column(k)
        for i in range(k):
            mad(s(sMOD[k - i]), v(m[i]))
        if wide:
            ins.append(("v_mul_lo_u32", (v(m[k]), v(acc[0]), s(sINV)), {}))
        else:
            ins.append(("v_mul_lo_u32", (v(tmp), v(acc[0]), s(sINV)), {}))
            ins.append(("v_and_b32", (v(m[k]), s(sMASK), v(tmp)), {}))
        mad(s(sMOD[0]), v(m[k]))
        ins.append(("v_lshrrev_b64", (A, lit(W), A), {}))
    for k in range(14, 27):
        column(k)
        for i in range(k - 13, 14):
            mad(s(sMOD[k - i]), v(m[i]))
        ins.append(("v_and_b32", (v(out[k - 14]), s(sMASK), v(acc[0])), {}))
        if k < 26:
            ins.append(("v_lshrrev_b64", (A, lit(W), A), {}))
        else:
            ins.append(("v_alignbit_b32", (v(out[13]), v(acc[1]), v(acc[0]), lit(W)), {}))
    return ins


def interleave(prog, *chains):
    """merge instruction lists proportionally (each list keeps its order)."""
    if KNOB_E64:
        prog.emit(".p2align", ("raw", "3"))      # (scalar instructions in front of a run of products are 4 bytes each)
    n = [len(c) for c in chains]
    pos = [0] * len(chains)
    total = sum(n)
    done = 0
    while done < total:
        # the chain that is furthest behind its share goes next
        k = min((j for j in range(len(chains)) if pos[j] < n[j]), key=lambda j: (pos[j] + 0.5) / n[j])
        for _ in range(KNOB_BLOCK):
            if pos[k] >= n[k]:
                break
            op, args, kw = chains[k][pos[k]]
            prog.emit(op, *args, **kw)
            pos[k] += 1
            done += 1


# ---- the kernel ----------------------------------------------------------------------------------------------------
KP4_1 = borrowed(4, 1)
KP8_4 = borrowed(8, 4)
KP32_1 = borrowed(32, 1)
NX_B = WIDE_B + 8            # -X3 = PPP (wide) - R^2 - 2(-Q) + 8p


def emit_first_row_and_negation(p):
    """lanes still at infinity copy the row in (sVALID & sINF), the others get sACT; rows of negative digits are negated (sACT & sNEG)"""
    e = p.emit
    # lanes whose accumulator is still at infinity take the row as it is: -X = 4p - QX, -Y = QY (negative digit) or
    # 4p - QY, ZZ = ZZZ = 1 (Montgomery form)
    e("s_and_b64", sp(sTMP), sp(sVALID), sp(sINF))
    e("s_andn2_b64", sp(sACT), sp(sVALID), sp(sINF))
    e("s_andn2_b64", sp(sINF), sp(sINF), sp(sTMP))
    e("s_mov_b64", EXEC, sp(sTMP))
    e("s_cbranch_execz", ("label", "L_no_init%="))
    for i in range(14):
        e("v_sub_u32", v(NX[i]), lit(KP4_1[i]), v(QX[i]))
        if i > 0:
            e("v_add_u32", v(NX[i]), v(NX[i]), v(T1))
        if i < 13:                                         # carried: -X keeps its limbs under 2^28
            e("v_lshrrev_b32", v(T1), lit(W), v(NX[i]))
            e("v_and_b32", v(NX[i]), s(sMASK), v(NX[i]))
    for i in range(14):
        e("v_sub_u32", v(NY[i]), lit(KP4_1[i]), v(QY[i]))
        if i > 0:
            e("v_add_u32", v(NY[i]), v(NY[i]), v(T1))
        if i < 13:
            e("v_lshrrev_b32", v(T1), lit(W), v(NY[i]))
            e("v_and_b32", v(NY[i]), s(sMASK), v(NY[i]))
    for i in range(14):
        e("v_mov_b32", v(ZZ[i]), lit(R1[i]))
    for i in range(14):
        e("v_mov_b32", v(ZZZ[i]), lit(R1[i]))
    e("s_and_b64", EXEC, sp(sTMP), sp(sNEG))
    for i in range(14):
        e("v_mov_b32", v(NY[i]), v(QY[i]))
    p.label("L_no_init%=")
    # negative digit: QY <- 4p - QY on the lanes that add
    e("s_and_b64", EXEC, sp(sACT), sp(sNEG))
    e("s_cbranch_execz", ("label", "L_no_neg%="))
    for i in range(14):
        e("v_sub_u32", v(QY[i]), lit(KP4_1[i]), v(QY[i]))
    p.label("L_no_neg%=")
    e("s_mov_b64", EXEC, lit(-1))


def emit_madd_part_a(p):
    """U2, S2, P, R and the P == 0 test, under sACT; returns the values part B needs"""
    e = p.emit
    # ---- the mixed addition, first part: U2 = QX ZZ1, S2 = QY ZZZ1; P = U2 + (-X1), R = S2 + (-Y1)
    qx, qy = Val(QX, 2, 1), Val(QY, 4, 2)
    nx, ny = Val(NX, NX_B, 1), Val(NY, WIDE_B, 1)
    zz, zzz = Val(ZZ, WIDE_B, 1), Val(ZZZ, WIDE_B, 1)
    e("s_mov_b64", EXEC, sp(sACT))
    e("s_cbranch_execz", ("label", "L_skip_a%="))
    interleave(p, chain_mul(qx, zz, U, M1, ACC1, T1, wide=True), chain_mul(qy, zzz, S, M2, ACC2, T2r, wide=True))
    for i in range(14):
        e("v_add_u32", v(U[i]), v(U[i]), v(NX[i]))
        e("v_add_u32", v(S[i]), v(S[i]), v(NY[i]))
    pp_ = Val(U, WIDE_B + nx.B, 1 + nx.L)
    rr = Val(S, WIDE_B + ny.B, 1 + ny.L)
    # P == 0 mod p? value = k p with k < 64 <=> (low 56 bits) p^-1 < 64 mod 2^56, tested 28 bits at a time
    assert pp_.B <= 64
    e("v_mul_lo_u32", v(T1), v(U[0]), s(sINVP))
    e("v_and_b32", v(T1), s(sMASK), v(T1))
    e("v_cmp_gt_u32", VCC, lit(64), v(T1))
    e("s_cbranch_vccz", ("label", "L_no_cand%="))
    # second limb of k p: ((k MOD0) >> 28) + k MOD1, against limb 1 of P plus limb 0's carry
    e("s_mov_b64", sp(sTMP), VCC)
    # (k MOD0 needs up to 34 bits -- k < 64, MOD0 < 2^28 --, so the carry is taken from the 64-bit product: a v_mul_lo_u32
    # here lost it for k >= 17 and the flag stayed down; VCC, the sink of the multiply-add, was saved above)
    assert 63 * MOD[0] >= 1 << 32 and 63 * MOD[0] < 1 << 36
    e("v_mad_u64_u32", vp(ACC1[0]), VCC, s(sMOD[0]), v(T1), lit(0))
    e("v_lshrrev_b64", vp(ACC1[0]), lit(W), vp(ACC1[0]))
    e("v_mul_lo_u32", v(T3), v(T1), s(sMOD[1]))              # (only its low 28 bits are compared: exact in 32)
    e("v_add_u32", v(T2r), v(ACC1[0]), v(T3))
    e("v_lshrrev_b32", v(T3), lit(W), v(U[0]))
    e("v_add_u32", v(T3), v(T3), v(U[1]))
    e("v_xor_b32", v(T2r), v(T2r), v(T3))
    e("v_and_b32", v(T2r), s(sMASK), v(T2r))
    e("v_cmp_eq_u32", VCC, lit(0), v(T2r))
    e("s_and_b64", sp(sTMP), sp(sTMP), VCC)
    e("s_or_b64", sp(sTROUBLE), sp(sTROUBLE), sp(sTMP))
    p.label("L_no_cand%=")
    p.label("L_skip_a%=")
    return pp_, rr, nx, ny, zz, zzz


def emit_madd_part_b(p, pp_, rr, nx, ny, zz, zzz):
    """the rest of the mixed addition under sACT (the row registers are dead: the next row is already on its way into them)"""
    e = p.emit
    e("s_mov_b64", EXEC, sp(sACT))
    e("s_cbranch_execz", ("label", "L_skip_b%="))
    # ---- second part
    for i in range(14):
        e("v_lshlrev_b32", v(D1[i]), lit(1), v(U[i]))
        e("v_lshlrev_b32", v(D2[i]), lit(1), v(S[i]))
    if KNOB_PAUSE in ("nop_groups", "nop_mid"):
        e("s_nop", ("raw", "15"))
    interleave(p, chain_sqr(pp_, D1, PP, M1, ACC1, T1, wide=True), chain_sqr(rr, D2, RR2, M2, ACC2, T2r))
    pp, rr2 = Val(PP, WIDE_B, 1), Val(RR2, NARROW_B, 1)
    if KNOB_PAUSE == "nop_groups":
        e("s_nop", ("raw", "15"))
    interleave(p, chain_mul(pp_, pp, D1, M1, ACC1, T1, wide=True), chain_mul(nx, pp, D2, M2, ACC2, T2r))
    ppp, nq = Val(D1, WIDE_B, 1), Val(D2, NARROW_B, 1)          # nq = (-X1) PP = -Q
    assert rr2.B + 2 * nq.B <= 8
    # X3 = R^2 - PPP - 2Q  ->  -X3 = PPP - R^2 - 2(-Q) + 8p, carried: limbs < 2^28
    for i in range(14):
        e("v_lshl_add_u32", v(T1), v(D2[i]), lit(1), v(RR2[i]))          # 2(-Q) + R^2        < 3 2^28
        e("v_sub_u32", v(T1), lit(KP8_4[i]), v(T1))                       # 8p (borrowed 4) - that
        if i == 0:
            e("v_add_u32", v(NX[i]), v(D1[i]), v(T1))
        else:
            e("v_add3_u32", v(NX[i]), v(D1[i]), v(T1), v(T2r))
        if i < 13:
            e("v_lshrrev_b32", v(T2r), lit(W), v(NX[i]))
            e("v_and_b32", v(NX[i]), s(sMASK), v(NX[i]))
    nx3 = Val(NX, ppp.B + 8, 1)
    # t1 = (-Q) - (-X3) + 32p  [= -(Q - X3)],  t2 = 32p - PPP
    assert nx3.B <= 32 and ppp.B <= 32
    for i in range(14):
        e("v_add_u32", v(D2[i]), lit(KP32_1[i]), v(D2[i]))
        e("v_sub_u32", v(D2[i]), v(D2[i]), v(NX[i]))
        e("v_sub_u32", v(U[i]), lit(KP32_1[i]), v(D1[i]))
    t1 = Val(D2, nq.B + 32, 1 + 2)
    t2 = Val(U, 32, 2)
    # ZZ3 = ZZ1 PP, ZZZ3 = ZZZ1 PPP, -Y3 = R t1 + (-Y1) t2   (three chains; each output overwrites an input limb by limb:
    # output limb k - 14 is written at column k, the input limb k - 14 was last read at column k - 1)
    if KNOB_PAUSE == "nop_groups":
        e("s_nop", ("raw", "15"))
    interleave(p, chain_mul(zz, pp, ZZ, M1, ACC1, T1, wide=True), chain_mul(zzz, ppp, ZZZ, M2, ACC2, T2r, wide=True),
               chain_mul_add(rr, t1, ny, t2, NY, RR2, ACC3, T3, wide=True))
    assert nx3.B <= nx.B and nx3.L <= nx.L
    p.label("L_skip_b%=")


def build():
    p = Prog()
    e = p.emit
    # ---------------- prologue: operands into fixed registers, constants
    e("comment", "operands -> fixed registers")
    e("s_mov_b64", sp(sTABLE), opnd(0))
    e("s_mov_b64", sp(sSC), opnd(1))
    e("s_mov_b64", sp(sOUT), opnd(2))
    e("s_mov_b64", sp(sREDO), opnd(3))
    e("s_mov_b32", s(sSPL), opnd(4))
    e("s_mov_b32", s(sLPB), opnd(5))
    e("s_mov_b32", s(sC), opnd(6))
    e("s_mov_b32", s(sNW), opnd(7))
    e("s_mov_b32", s(sSTMP), opnd(8))              # wtop
    e("s_mov_b32", s(sH), opnd(9))
    e("s_mov_b32", s(sHTOP), opnd(10))
    e("s_mov_b32", s(sRB), opnd(11))               # row_bytes
    e("v_mov_b32", v(VRB), opnd(11))
    e("v_mov_b32", v(POINT), opnd(12))
    e("v_mov_b32", v(VTID), opnd(13))
    if KNOB_ROTWIN:
        e("v_lshrrev_b32", v(T1), lit(8), v(VTID))
        e("s_nop", ("raw", "0"))
        e("v_readfirstlane_b32", s(sROT), v(T1))
    for i in range(14):
        e("s_mov_b32", s(sMOD[i]), lit(MOD[i]))
    e("s_mov_b32", s(sINV), lit(INV))
    e("s_mov_b32", s(sMASK), lit(MASK))
    e("s_mov_b32", s(sINVP), lit(INVP))
    # masks, 2^C
    e("s_lshl_b32", s(s2C), lit(1), s(sC))
    e("s_add_u32", s(sMASKC), s(s2C), lit(-1))
    e("s_lshl_b32", s(sMASKTOP), lit(1), s(sSTMP))
    e("s_add_u32", s(sMASKTOP), s(sMASKTOP), lit(-1))
    # row addressing: the table is one allocation per window (kernels.h: DirectTable); the row for (window j, point, mag) is at
    # base[j] + (point * hj + mag - 1) * row_bytes, base[j] read from the device array of window addresses when the window comes up
    # state
    e("s_mov_b64", sp(sINF), lit(-1))
    e("s_mov_b64", sp(sTROUBLE), lit(0))
    e("s_mov_b32", s(sQ), lit(0))
    e("s_mov_b32", s(sJ), lit(0))
    e("v_mov_b32", v(CARRY), lit(0))
    # the first scalar, and the second one into SCN
    scalar_load(p, SC, POINT)
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    e("s_cmp_gt_u32", s(sSPL), lit(1))
    e("s_cbranch_scc0", ("label", "L_no_second%="))
    e("v_add_u32", v(T3), s(sLPB), v(POINT))
    scalar_load(p, SCN, T3)
    p.label("L_no_second%=")
    digit_and_address(p)
    e("s_mov_b64", EXEC, sp(sVALIDN))       # (a zero digit has no row: its address would be the row in front of the window's)
    row_loads(p)
    e("s_mov_b64", EXEC, lit(-1))
    # ---------------- the loop
    e(".p2align", ("raw", "6" if KNOB_PAUSE in ("align64", "align64_nop0") else "3"))
    p.label("L_loop%=")
    if KNOB_PAUSE in ("nop_top", "nop_groups", "nop_top2"):
        e("s_nop", ("raw", "15"))
    if KNOB_PAUSE == "nop_top2":
        e("s_nop", ("raw", "15"))
    if KNOB_PAUSE == "nop_top_half":
        e("s_nop", ("raw", "7"))
    if KNOB_PAUSE in ("nop0", "align64_nop0"):
        e("s_nop", ("raw", "0"))
    if KNOB_PAUSE == "sleep_top":
        e("s_sleep", ("raw", "1"))
    e("s_mov_b64", sp(sVALID), sp(sVALIDN))
    e("s_mov_b64", sp(sNEG), sp(sNEGN))
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    # (q, j) of the row after this one
    e("s_add_u32", s(sJ), s(sJ), lit(1))
    e("s_cmp_eq_u32", s(sJ), s(sNW))
    e("s_cbranch_scc0", ("label", "L_same_scalar%="))
    e("s_mov_b32", s(sJ), lit(0))
    e("s_add_u32", s(sQ), s(sQ), lit(1))
    p.label("L_same_scalar%=")
    e("s_cmp_lt_u32", s(sQ), s(sSPL))
    e("s_cselect_b32", s(sMORE), lit(1), lit(0))
    emit_first_row_and_negation(p)
    # the digit after this one (all lanes), its masks and its row address
    e("s_mov_b64", sp(sVALIDN), lit(0))
    e("s_cmp_eq_u32", s(sMORE), lit(0))
    e("s_cbranch_scc1", ("label", "L_no_next%="))
    e("s_cmp_eq_u32", s(sJ), lit(0))
    e("s_cbranch_scc0", ("label", "L_no_new_scalar%="))
    for i in range(8):
        e("v_mov_b32", v(SC[i]), v(SCN[i]))
    e("v_mov_b32", v(CARRY), lit(0))
    e("v_add_u32", v(POINT), s(sLPB), v(POINT))
    e("s_add_u32", s(sSTMP), s(sQ), lit(1))
    e("s_cmp_lt_u32", s(sSTMP), s(sSPL))
    e("s_cbranch_scc0", ("label", "L_no_new_scalar%="))
    e("v_add_u32", v(T3), s(sLPB), v(POINT))
    scalar_load(p, SCN, T3)
    p.label("L_no_new_scalar%=")
    digit_and_address(p)
    p.label("L_no_next%=")
    pp_, rr, nx, ny, zz, zzz = emit_madd_part_a(p)
    # ---- the row is dead: the next one is gathered into its registers
    e("s_mov_b64", EXEC, sp(sVALIDN))
    e("s_cbranch_execz", ("label", "L_no_loads%="))
    row_loads(p)
    p.label("L_no_loads%=")
    emit_madd_part_b(p, pp_, rr, nx, ny, zz, zzz)
    e("s_mov_b64", EXEC, lit(-1))
    e("s_cmp_eq_u32", s(sMORE), lit(0))
    e("s_cbranch_scc0", ("label", "L_loop%="))
    # ---------------- epilogue: lanes still at infinity store literal zeros (the library's marker is ZZ == 0)
    e("s_mov_b64", EXEC, sp(sINF))
    e("s_cbranch_execz", ("label", "L_no_inf%="))
    for r in NX + NY + ZZ + ZZZ:
        e("v_mov_b32", v(r), lit(0))
    p.label("L_no_inf%=")
    e("s_mov_b64", EXEC, lit(-1))
    e("v_mov_b32", v(T1), lit(224))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(VTID), v(T1), sp(sOUT))
    for k in range(14):
        e("global_store_dwordx4", vp(ADDR[0]), ("v4", 4 * k), ("off",), offset=16 * k)
    e("s_nop", ("raw", "1"))
    e("s_cmp_eq_u64", sp(sTROUBLE), lit(0))
    e("s_cbranch_scc1", ("label", "L_done%="))
    e("v_mov_b32", v(T1), lit(1))
    e("v_mov_b32", v(ADDR[0]), s(sREDO[0]))
    e("v_mov_b32", v(ADDR[1]), s(sREDO[1]))
    e("s_mov_b64", EXEC, lit(1))
    e("global_store_dword", vp(ADDR[0]), v(T1), ("off",))
    e("s_nop", ("raw", "1"))
    e("s_mov_b64", EXEC, lit(-1))
    p.label("L_done%=")
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    return p


def scalar_load(p, dst, point_reg):
    """dst[0:8] <- the 32-byte canonical scalar of point `point_reg` (little-endian words)"""
    p.emit("v_mov_b32", v(T1), lit(32))
    p.emit("v_mad_u64_u32", vp(ADDR[0]), VCC, v(point_reg), v(T1), sp(sSC))
    p.emit("global_load_dwordx4", ("v4", dst[0]), vp(ADDR[0]), ("off",))
    p.emit("global_load_dwordx4", ("v4", dst[4]), vp(ADDR[0]), ("off",), offset=16)


def digit_and_address(p):
    """the signed digit of window sJ of the scalar in SC (all lanes): sVALIDN, sNEGN, ADDR; SC shifted down by C"""
    e = p.emit
    e("s_sub_u32", s(sSTMP), s(sNW), lit(1))
    e("s_cmp_eq_u32", s(sJ), s(sSTMP))
    e("s_cselect_b32", s(sMASKJ), s(sMASKTOP), s(sMASKC))
    e("s_cselect_b32", s(sHCMP), lit(-1), s(sH))
    e("s_cselect_b32", s(sHJ), s(sHTOP), s(sH))
    # base of window sJ (asked for now, needed by the last instruction below)
    if KNOB_ROTWIN:
        # index = j == nw - 1 ? j : (j + rot) mod (nw - 1)      (sSTMP = nw - 1 here, scc = (j == nw - 1))
        e("s_add_u32", s(sTMPB[0]), s(sJ), s(sROT))
        e("s_and_b32", s(sTMPB[0]), s(sTMPB[0]), lit(0xffff))
        e("v_cvt_f32_u32", v(T3), s(sTMPB[0]))                   # (a cheap exact mod for small numbers: t - floor(t / m) m)
        e("v_cvt_f32_u32", v(RAW), s(sSTMP))
        e("v_rcp_iflag_f32", v(RAW), v(RAW))
        e("v_mul_f32", v(T3), v(T3), v(RAW))
        e("v_cvt_u32_f32", v(T3), v(T3))
        e("s_nop", ("raw", "0"))
        e("v_readfirstlane_b32", s(sTMPB[1]), v(T3))
        e("s_mul_i32", s(sTMPB[1]), s(sTMPB[1]), s(sSTMP))
        e("s_sub_u32", s(sTMPB[0]), s(sTMPB[0]), s(sTMPB[1]))
        e("s_cmp_ge_u32", s(sTMPB[0]), s(sSTMP))                  # (the reciprocal may be one short)
        e("s_cselect_b32", s(sTMPB[1]), s(sSTMP), lit(0))
        e("s_sub_u32", s(sTMPB[0]), s(sTMPB[0]), s(sTMPB[1]))
        e("s_cmp_eq_u32", s(sJ), s(sSTMP))
        e("s_cselect_b32", s(sSTMP), s(sJ), s(sTMPB[0]))
        e("s_lshl_b32", s(sSTMP), s(sSTMP), lit(3))
    else:
        e("s_lshl_b32", s(sSTMP), s(sJ), lit(3))
    e("s_load_dwordx2", sp(sTMPB), sp(sTABLE), s(sSTMP))
    e("v_and_b32", v(RAW), s(sMASKJ), v(SC[0]))
    e("v_add_u32", v(RAW), v(RAW), v(CARRY))
    e("v_cmp_lt_u32", VCC, s(sHCMP), v(RAW))
    e("v_sub_u32", v(T3), s(s2C), v(RAW))
    for i in range(7):
        e("v_alignbit_b32", v(SC[i]), v(SC[i + 1]), v(SC[i]), s(sC))
    e("v_lshrrev_b32", v(SC[7]), s(sC), v(SC[7]))
    e("v_cndmask_b32", v(MAG), v(RAW), v(T3), VCC)
    e("v_cndmask_b32", v(CARRY), lit(0), lit(1), VCC)
    e("s_mov_b64", sp(sNEGN), VCC)
    e("v_cmp_ne_u32", sp(sVALIDN), lit(0), v(MAG))
    if KNOB_CONFINE:
        # 1: 8 points x 64 rows per window (1 MB: cache)   2: 64 points x all rows (4 GB in 1024 runs of 4 MB: HBM, few pages)
        # 3: all points x 16 rows (134 MB in 65,536 runs of 2 KB: every page of the table, little of its data)
        pmask, mmask = {1: (7, 0x3f), 2: (63, 0x7fff), 3: (4095, 0xf)}[KNOB_CONFINE]
        e("v_and_b32", v(T3), lit(pmask), v(POINT))
        e("v_mad_u32_u24", v(T3), v(T3), s(sHJ), v(MAG))
        e("v_and_b32", v(T3), lit(mmask | (pmask << 15)), v(T3))       # (hj = 2^15 on every window but the top one)
        e("v_or_b32", v(T3), lit(1), v(T3))
    else:
        e("v_mad_u32_u24", v(T3), v(POINT), s(sHJ), v(MAG))
    e("s_waitcnt", ("raw", "lgkmcnt(0)"))
    e("s_sub_u32", s(sTMPB[0]), s(sTMPB[0]), s(sRB))          # minus one row: mag counts from 1
    e("s_subb_u32", s(sTMPB[1]), s(sTMPB[1]), lit(0))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(T3), v(VRB), sp(sTMPB))


def row_loads(p):
    for k in range(7):
        p.emit("global_load_dwordx4", ("v4", ROW[0] + 4 * k), vp(ADDR[0]), ("off",), offset=16 * k)


# ---- emit ------------------------------------------------------------------------------------------------------------
def render(p):
    lines = ["// generated by tools/gen_direct_asm.py -- do not edit (python tools/gen_direct_asm.py)",
             "// %d instructions, %d of them VALU; VGPRs v0..v%d, SGPRs s%d..s%d" % (
                 sum(1 for i in p.ins if i[0] not in ("label", "comment")), p.count_valu(), NUM_VGPRS - 1, SBASE, NUM_SGPRS - 1)]
    for t in p.text():
        t = t.replace("\\", "\\\\").replace('"', '\\"')
        lines.append('"%s\\n"' % t)
    return "\n".join(lines) + "\n"


def clobbers():
    return ", ".join(['"v%d"' % i for i in range(NUM_VGPRS)] + ['"s%d"' % i for i in range(SBASE, NUM_SGPRS) if i not in (32, 33, 34)] +
                     ['"vcc"', '"memory"'])


# ---- simulator: one lane ---------------------------------------------------------------------------------------------
class Sim:
    """Executes a Prog for ONE lane. SGPR pairs that hold lane masks are modelled as this lane's bit only (0 / 1 in
    the low word): a wave-uniform branch on EXEC or VCC then goes the way a wave consisting of this lane would."""

    def __init__(self, prog, operands, mem_read, mem_write):
        self.p = prog
        self.vr = [0] * 256
        self.sr = [0] * 128
        self.vcc = 0
        self.exec = 1
        self.scc = 0
        self.ops = operands
        self.rd, self.wr = mem_read, mem_write
        self.labels = {a[0]: i for i, (op, a, _) in enumerate(prog.ins) if op == "label"}
        self.valu_executed = 0
        self.pending = []        # (register, value) of loads not yet waited for: reading them early is a bug

    # operand access
    def g32(self, o):
        k = o[0]
        if k == "v":
            assert all(o[1] != r for r, _ in self.pending), "read of v%d before s_waitcnt" % o[1]
            return self.vr[o[1]]
        if k == "s":
            return self.sr[o[1]]
        if k == "lit":
            return o[1]
        if k == "op":
            x = self.ops[o[1]]
            return x & 0xFFFFFFFF
        raise ValueError(o)

    def g64(self, o):
        k = o[0]
        if k == "vp":
            return self.g32(v(o[1])) | (self.g32(v(o[1] + 1)) << 32)
        if k == "sp":
            return self.sr[o[1]] | (self.sr[o[1] + 1] << 32)
        if k == "lit":
            x = o[1]
            return x if x < 0x80000000 else (x | 0xFFFFFFFF00000000)     # inline constants sign-extend
        if k == "vcc":
            return self.vcc
        if k == "exec":
            return self.exec
        if k == "op":
            return self.ops[o[1]] & 0xFFFFFFFFFFFFFFFF
        raise ValueError(o)

    def p32(self, o, x):
        assert 0 <= x < (1 << 32), x
        if o[0] == "v":
            if self.exec & 1:
                self.vr[o[1]] = x
        elif o[0] == "s":
            self.sr[o[1]] = x
        elif o[0] == "op":                   # an output operand of the statement (a VGPR: written under EXEC)
            if self.exec & 1:
                self.ops[o[1]] = x
        else:
            raise ValueError(o)

    def p64(self, o, x):
        x &= 0xFFFFFFFFFFFFFFFF
        if o[0] == "vp":
            if self.exec & 1:
                self.vr[o[1]], self.vr[o[1] + 1] = x & 0xFFFFFFFF, x >> 32
        elif o[0] == "sp":
            self.sr[o[1]], self.sr[o[1] + 1] = x & 0xFFFFFFFF, x >> 32
        elif o[0] == "vcc":
            self.vcc = x
        elif o[0] == "exec":
            self.exec = x
        else:
            raise ValueError(o)

    def mask_bit(self, x):   # lane masks: only this lane's bit (bit 0) is meaningful
        return x & 1

    def run(self, max_steps=50_000_000):
        pc, steps = 0, 0
        ins = self.p.ins
        while pc < len(ins):
            op, a, kw = ins[pc]
            pc += 1
            steps += 1
            assert steps < max_steps
            if op in ("label", "comment", ".p2align"):
                continue
            if op.startswith("v_"):
                self.valu_executed += 1
            if op == "v_mad_u64_u32":
                r = self.g32(a[2]) * self.g32(a[3]) + self.g64(a[4])
                assert r < (1 << 64), "64-bit column overflow"
                self.p64(a[0], r)
            elif op == "v_mul_lo_u32":
                self.p32(a[0], (self.g32(a[1]) * self.g32(a[2])) & 0xFFFFFFFF)
            elif op == "v_and_b32":
                self.p32(a[0], self.g32(a[1]) & self.g32(a[2]))
            elif op == "v_xor_b32":
                self.p32(a[0], self.g32(a[1]) ^ self.g32(a[2]))
            elif op == "v_or_b32":
                self.p32(a[0], self.g32(a[1]) | self.g32(a[2]))
            elif op == "ds_bpermute_b32":
                raise ValueError("simulator: cross-lane instruction on a one-lane simulation")
            elif op == "v_lshrrev_b64":
                self.p64(a[0], self.g64(a[2]) >> (self.g32(a[1]) & 63))
            elif op == "v_lshrrev_b32":
                self.p32(a[0], self.g32(a[2]) >> (self.g32(a[1]) & 31))
            elif op == "v_lshlrev_b32":
                x = self.g32(a[2]) << (self.g32(a[1]) & 31)
                assert x < (1 << 32), "shift lost a bit"
                self.p32(a[0], x)
            elif op == "v_alignbit_b32":
                x = (self.g32(a[1]) << 32) | self.g32(a[2])
                self.p32(a[0], (x >> (self.g32(a[3]) & 31)) & 0xFFFFFFFF)
            elif op == "v_add_u32":
                x = self.g32(a[1]) + self.g32(a[2])
                assert x < (1 << 32), "32-bit add wrapped"
                self.p32(a[0], x)
            elif op == "v_add3_u32":
                x = self.g32(a[1]) + self.g32(a[2]) + self.g32(a[3])
                assert x < (1 << 32), "32-bit add wrapped"
                self.p32(a[0], x)
            elif op == "v_lshl_add_u32":
                x = (self.g32(a[1]) << self.g32(a[2])) + self.g32(a[3])
                assert x < (1 << 32), "32-bit add wrapped"
                self.p32(a[0], x)
            elif op == "v_sub_u32":
                x = self.g32(a[1]) - self.g32(a[2])
                assert x >= 0 or not (self.exec & 1), "32-bit subtraction went negative"
                self.p32(a[0], x & 0xFFFFFFFF)
            elif op == "v_mov_b32":
                self.p32(a[0], self.g32(a[1]))
            elif op == "v_cndmask_b32":
                self.p32(a[0], self.g32(a[2]) if self.mask_bit(self.g64(a[3])) else self.g32(a[1]))
            elif op == "v_mad_u32_u24":
                x = (self.g32(a[1]) & 0xFFFFFF) * (self.g32(a[2]) & 0xFFFFFF) + self.g32(a[3])
                assert x < (1 << 32)
                self.p32(a[0], x)
            elif op in ("v_cmp_lt_u32", "v_cmp_gt_u32", "v_cmp_ne_u32", "v_cmp_eq_u32"):
                x, y = self.g32(a[1]), self.g32(a[2])
                r = {"lt": x < y, "gt": x > y, "ne": x != y, "eq": x == y}[op[6:8]]
                self.p64(a[0], 1 if (r and (self.exec & 1)) else 0)
            elif op == "s_mov_b32":
                self.p32(a[0], self.g32(a[1]))
            elif op == "s_mov_b64":
                self.p64(a[0], self.g64(a[1]))
            elif op == "s_and_b64":
                self.p64(a[0], self.g64(a[1]) & self.g64(a[2]))
                self.scc = 1 if self.mask_bit(self.g64(a[0])) else 0
            elif op == "s_or_b64":
                self.p64(a[0], self.g64(a[1]) | self.g64(a[2]))
            elif op == "s_andn2_b64":
                self.p64(a[0], self.g64(a[1]) & ~self.g64(a[2]))
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
            elif op == "s_mul_i32":
                self.p32(a[0], (self.g32(a[1]) * self.g32(a[2])) & 0xFFFFFFFF)
            elif op == "s_mul_hi_u32":
                self.p32(a[0], (self.g32(a[1]) * self.g32(a[2])) >> 32)
            elif op == "s_lshl_b32":
                self.p32(a[0], (self.g32(a[1]) << (self.g32(a[2]) & 31)) & 0xFFFFFFFF)
            elif op == "s_lshr_b32":
                self.p32(a[0], self.g32(a[1]) >> (self.g32(a[2]) & 31))
            elif op == "s_lshr_b64":
                self.p64(a[0], self.g64(a[1]) >> (self.g32(a[2]) & 63))
            elif op == "s_lshl_b64":
                self.p64(a[0], self.g64(a[1]) << (self.g32(a[2]) & 63))
            elif op in ("s_cmp_eq_u32", "s_cmp_lt_u32", "s_cmp_gt_u32"):
                x, y = self.g32(a[0]), self.g32(a[1])
                self.scc = int({"eq": x == y, "lt": x < y, "gt": x > y}[op[6:8]])
            elif op == "s_cmp_eq_u64":
                self.scc = int(self.mask_bit(self.g64(a[0])) == self.mask_bit(self.g64(a[1])))
            elif op == "s_cselect_b32":
                self.p32(a[0], self.g32(a[1]) if self.scc else self.g32(a[2]))
            elif op == "s_cselect_b64":
                self.p64(a[0], self.g64(a[1]) if self.scc else self.g64(a[2]))
            elif op == "s_cbranch_scc0":
                if not self.scc:
                    pc = self.labels[a[0][1]]
            elif op == "s_cbranch_scc1":
                if self.scc:
                    pc = self.labels[a[0][1]]
            elif op == "s_cbranch_execz":
                if not (self.exec & 1):
                    pc = self.labels[a[0][1]]
            elif op == "s_cbranch_vccz":
                if not (self.vcc & 1):
                    pc = self.labels[a[0][1]]
            elif op == "s_waitcnt":
                if "vmcnt" in a[0][1]:       # (an lgkmcnt wait does not make vector loads land)
                    for r, val in self.pending:
                        self.vr[r] = val
                    self.pending = []
            elif op in ("s_nop", "s_sleep"):
                pass
            elif op == "s_load_dwordx2":
                w = self.rd(self.g64(a[1]) + self.g32(a[2]), 2)
                self.p64(a[0], w[0] | (w[1] << 32))
            elif op == "s_branch":
                pc = self.labels[a[0][1]]
            elif op == "global_load_dwordx4":
                if self.exec & 1:
                    addr = self.g64(a[1]) + kw.get("offset", 0)
                    words = self.rd(addr, 4)
                    for k in range(4):
                        self.pending.append((a[0][1] + k, words[k]))
            elif op in ("global_load_dword", "global_load_dwordx2"):      # (the entry walk of tools/gen_bucket_asm.py)
                if self.exec & 1:
                    n = 1 if op.endswith("dword") else 2
                    words = self.rd(self.g64(a[1]) + kw.get("offset", 0), n)
                    for k in range(n):
                        self.pending.append((a[0][1] + k, words[k]))
            elif op == "global_store_dwordx4":
                if self.exec & 1:
                    self.wr(self.g64(a[0]) + kw.get("offset", 0), [self.g32(v(a[1][1] + k)) for k in range(4)])
            elif op == "global_store_dword":
                if self.exec & 1:
                    self.wr(self.g64(a[0]), [self.g32(a[1])])
            else:
                raise ValueError("simulator: unknown instruction " + op)
        return steps


# ---- reference arithmetic for the self-test ----------------------------------------------------------------------------
def to_mont(x):
    return x * RMONT % P


def from_mont_limbs(l):
    return sum(x << (W * i) for i, x in enumerate(l)) * pow(RMONT, -1, P) % P


def ec_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    (x1, y1), (x2, y2) = a, b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return x3, (lam * (x1 - x3) - y1) % P


def ec_mul(k, pt):
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, pt)
        pt = ec_add(pt, pt)
        k >>= 1
    return acc


G1 = (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
      0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)


def selftest(seed=1, c=16, spl=2, verbose=True, sparse=False, force_equal=False, collide_at=0, collide_neg=False):
    """One lane through the whole instruction stream: `spl` scalars of nw windows, rows served from a synthetic table
    (row index -> an honest multiple of the generator, weakly reduced as the table stores it), result against affine
    big-int arithmetic.
    collide_at = t > 1: every digit of every scalar is positive and non-zero, and the t-th row the lane gathers IS the sum of
    the t - 1 rows before it (negated with collide_neg), i.e. exactly ONE addition of the lane meets P = +-Q, after the
    accumulator's -X has widened to its loop bound: returns the redo flag (the flag must be raised every time)."""
    rnd = random.Random(seed)
    prog = build()
    nw = (255 + c - 1) // c
    wtop = 255 - c * (nw - 1)
    h, htop = 1 << (c - 1), 1 << wtop
    top_base = (nw - 1) * 4096 * h
    row_bytes = 128
    lpb = 256
    first = rnd.randrange(256)
    table_addr, sc_addr, out_addr, redo_addr = 0x100000000000, 0x200000000000, 0x300000000000, 0x400000000000
    win_tab_addr = 0x500000000000                     # the device array of window base addresses
    win_stride = 1 << 38                              # window j lives at table_addr + j * win_stride (separate allocations; <= 26 windows)
    R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    scalars = {}
    for q in range(spl):
        k = rnd.randrange(R_ORDER)
        if collide_at:      # windows in [1, h - 1] (no carries, no negative digits, nothing skipped); the top window small enough for k < r
            k = sum(rnd.randrange(1, h) << (c * j) for j in range(nw - 1)) + (rnd.randrange(1, min(htop, 1 << (wtop - 2))) << (c * (nw - 1)))
        if sparse:
            k &= ((1 << c) - 1) << (c * rnd.randrange(nw - 1))      # a single non-zero window
        scalars[first + q * lpb] = k
    # point i of the setup = [i + 2] G; window j base = 2^(c j) of it
    pts = {i: ec_mul(i + 2, G1) for i in scalars}
    cache = {}
    served = []            # the points of the distinct rows, in the order the lane first asked for them

    def row(idx):
        if idx not in cache:
            if idx >= top_base:
                point, d, j = (idx - top_base) // htop, (idx - top_base) % htop + 1, nw - 1
            else:
                j, rem = idx // (4096 * h), idx % (4096 * h)
                point, d = rem // h, rem % h + 1
            pt = ec_mul(d << (c * j), pts[point])
            if force_equal:
                pt = pts[min(pts)]                                  # every row the same point: P + P on the second addition
            if collide_at and len(served) == collide_at - 1:
                pt = None
                for q_ in served:
                    pt = ec_add(pt, q_)
                if collide_neg:
                    pt = (pt[0], (P - pt[1]) % P)
            served.append(pt)
            # weakly reduced Montgomery limbs (< 2p), as a table row holds them
            xs = to_mont(pt[0]) + (P if rnd.random() < 0.5 else 0)
            ys = to_mont(pt[1]) + (P if rnd.random() < 0.5 else 0)
            cache[idx] = (limbs(xs) + limbs(ys), pt)
        return cache[idx]

    stored = {}

    def rd(addr, n):
        if addr >= win_tab_addr:
            j = (addr - win_tab_addr) // 8
            a = table_addr + j * win_stride
            return [a & 0xFFFFFFFF, a >> 32][:n]
        if addr >= table_addr and addr < sc_addr:
            j, off = (addr - table_addr) // win_stride, (addr - table_addr) % win_stride
            base_idx = top_base if j == nw - 1 else j * 4096 * h          # first row of window j in the flat numbering row() decodes
            words = row(base_idx + off // row_bytes)[0]
            k = (off % row_bytes) // 4
            return words[k:k + n]
        off = addr - sc_addr
        point, k = off // 32, (off % 32) // 4
        sc = scalars[point]
        return [(sc >> (32 * (k + t))) & 0xFFFFFFFF for t in range(n)]

    def wr(addr, words):
        for k, wv in enumerate(words):
            stored[addr + 4 * k] = wv

    ops = [win_tab_addr, sc_addr, out_addr, redo_addr, spl, lpb, c, nw, wtop, h, htop, row_bytes, first, 0]
    sim = Sim(prog, ops, rd, wr)
    steps = sim.run()
    got = [stored[out_addr + 4 * k] for k in range(56)]
    redo = stored.get(redo_addr, 0)
    want = None
    for i, k in scalars.items():
        want = ec_add(want, ec_mul(k, pts[i]) if not force_equal else None)
    if force_equal or collide_at:
        return redo, sim.valu_executed
    nxv, nyv, zzv, zzzv = (from_mont_limbs(got[14 * t:14 * t + 14]) for t in range(4))
    if want is None:
        ok = all(x == 0 for x in got[28:42])
    else:
        zz_inv, zzz_inv = pow(zzv, -1, P), pow(zzzv, -1, P)
        ok = ((-nxv) * zz_inv % P, (-nyv) * zzz_inv % P) == want and (zzv ** 3 - zzzv ** 2) % P == 0
    if verbose:
        adds = spl * nw
        print("selftest seed=%d c=%d spl=%d: %s, %d instructions executed, %d VALU (%.0f per row), redo=%d" % (
            seed, c, spl, "ok" if ok else "MISMATCH", steps, sim.valu_executed, sim.valu_executed / adds, redo))
    assert ok and redo == 0
    return sim.valu_executed


def mix():
    """opcode mix of one pass of the loop body (one mixed addition incl. the next digit, the gather and the loop control)"""
    import collections
    prog = build()
    lo = next(k for k, i in enumerate(prog.ins) if i[0] == "label" and i[1][0].startswith("L_loop"))
    hi = max(k for k, i in enumerate(prog.ins) if i[0] == "s_cbranch_scc0" and i[1][0][1].startswith("L_loop"))
    body = [i for i in prog.ins[lo:hi + 1] if i[0] not in ("label", "comment", ".p2align")]
    # the blocks that run once per lane, once per scalar or never on honest data are counted apart
    c = collections.Counter(i[0] for i in body)
    init_lo = next(k for k, i in enumerate(prog.ins) if i[0] == "s_cbranch_execz" and i[1][0][1].startswith("L_no_init"))
    init_hi = next(k for k, i in enumerate(prog.ins) if i[0] == "label" and i[1][0].startswith("L_no_init"))
    cand_lo = next(k for k, i in enumerate(prog.ins) if i[0] == "s_cbranch_vccz")
    cand_hi = next(k for k, i in enumerate(prog.ins) if i[0] == "label" and i[1][0].startswith("L_no_cand"))
    ns_lo = next(k for k, i in enumerate(prog.ins) if i[0] == "s_cbranch_scc0" and i[1][0][1].startswith("L_no_new_scalar"))
    ns_hi = next(k for k, i in enumerate(prog.ins) if i[0] == "label" and i[1][0].startswith("L_no_new_scalar"))
    rare = collections.Counter()
    for lo2, hi2 in ((init_lo, init_hi), (cand_lo, cand_hi), (ns_lo, ns_hi)):
        rare.update(i[0] for i in prog.ins[lo2 + 1:hi2] if i[0] not in ("label", "comment"))
    hot = c - rare
    valu = sum(n for op, n in hot.items() if op.startswith("v_"))
    print("k_direct_accumulate_asm: one pass of the loop body = one mixed addition (tools/gen_direct_asm.py --mix)")
    print("  VALU instructions on the path every row takes: %d   (the compiler's schedule of the same addition: 4814, SQ_INSTS_VALU)" % valu)
    for op, n in sorted(hot.items(), key=lambda kv: -kv[1]):
        print("    %-22s %5d" % (op, n))
    print("  blocks that run once per lane, once per scalar or never on honest data (first-row copy, second limb of the P = 0 test,")
    print("  scalar switch): %d instructions, %d of them VALU" % (sum(rare.values()), sum(n for op, n in rare.items() if op.startswith("v_"))))
    print("  registers: v0..v%d, s%d..s%d; whole stream: %d instructions" % (NUM_VGPRS - 1, SBASE, NUM_SGPRS - 1,
                                                                          sum(1 for i in prog.ins if i[0] not in ("label", "comment"))))


def main():
    if "--mix" in sys.argv:
        return mix()
    if "--selftest" in sys.argv:
        for seed, c in ((1, 16), (2, 13), (3, 10), (4, 16)):
            selftest(seed, c)
        selftest(5, 16, sparse=True)
        redo, _ = selftest(6, 16, verbose=False, force_equal=True)
        assert redo == 1, "P + P must raise the redo flag"
        print("P + P raises the redo flag")
        return
    text = render(build())
    if "--check" in sys.argv:
        assert open(OUT).read() == text, "csrc/direct_asm.inc is stale: run python tools/gen_direct_asm.py"
        print("direct_asm.inc matches its generator")
        return
    with open(OUT, "w") as f:
        f.write(text)
    with open(OUT.replace(".inc", "_clobbers.inc"), "w") as f:
        f.write("// generated by tools/gen_direct_asm.py -- do not edit\n" + clobbers() + "\n")
    print("wrote %s: %d VALU instructions in the stream" % (OUT, build().count_valu()))


if __name__ == "__main__":
    main()
