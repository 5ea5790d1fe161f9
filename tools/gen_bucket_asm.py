#!/usr/bin/env python3
"""Generator of the hand-scheduled light-bucket accumulation of the bucket engine (msm.hip: k_bucket_accumulate_asm,
csrc/bucket_asm.inc) -- Pippenger's bucket accumulation (the reference's msm::pippenger::msm, call sites
/root/reference/src/lib.rs:234-243, 270) with the mixed-addition stream of tools/gen_direct_asm.py.

One lane owns one bucket, as in the compiler-scheduled kernel: lanes are ranked by descending bucket population (k_digit_sort's
`perm`), so the 64 trip counts of a wave are near-equal; the XYZZ accumulator never leaves the registers. What changes against the
direct-table kernel is only WHERE the next row comes from:

    direct: (scalar, window) in lockstep, signed digit of the lane's scalar -> row of the window's table
    bucket: the lane's own entry list  sorted[begin .. end)  of (window * 4096 + point | sign << 31) -> row of the 9 MB fixed-base table

The entry of the row after next is prefetched one addition ahead (entry -> row is a dependent gather), the next row's seven
loads are issued inside the addition into the registers the current row has vacated, exactly as in the direct kernel. The loop ends
when no lane of the wave has an entry left. At the end every lane brings its sum to the form the bucket reduction reads
(G1Xyzz29: X, Y, ZZ, ZZZ below 2p) -- X = 32p - (-X), then one product by 1 per coordinate -- and stores it to buckets[bucket].
Lanes whose rank is below n_heavy (buckets of more than 64 entries: the C++ wave-per-bucket path of the same launch owns them) do
nothing. A lane that meets P = +-Q reports it through the statement's output operand, and the C++ code behind the statement -- the
complete-by-branches formulas -- recomputes that lane's bucket on the spot (no second launch, no flag array).

    python tools/gen_bucket_asm.py             write csrc/bucket_asm.inc (+ clobbers)
    python tools/gen_bucket_asm.py --selftest  run the stream on a simulated lane against affine big-int arithmetic
    python tools/gen_bucket_asm.py --check     the committed .inc is what this file writes
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_direct_asm as G
from gen_direct_asm import (ACC1, ACC2, ADDR, EXEC, M1, M2, NX, NY, T1, T2r, T3, U, VCC, ZZ, ZZZ, KP32_1, R1, MOD, INV, INVP, MASK, P, W,
                            Prog, Sim, Val, chain_mul, interleave, lit, opnd, s, sp, v, vp)

OUT = os.path.join(G.ROOT, "lambdaworks_kzg_amd", "csrc", "bucket_asm.inc")

# registers the direct kernel uses for its scalars and digits serve the entry walk here
KCUR, KEND = G.SC[0], G.SC[1]          # cursor into the lane's entry list, its end (an aligned pair: one dwordx2 load)
ENT, ENT1, BKT = G.SC[2], G.SC[3], G.SC[4]
RANK, V112 = G.POINT, G.VRB
sENT = G.sSC
sBS = G.sregs("BS", 2, 2)
sPERM = G.sregs("PERM", 2, 2)
sLIGHT = G.sregs("LIGHT", 2, 2)
sNHEAVY = G.sregs("NHEAVY")
sM31 = G.sregs("M31")
NUM_SGPRS = G._snext[0]
assert NUM_SGPRS <= 100, NUM_SGPRS        # (s100 and up are reserved on gfx950)

ROW_BYTES = 112                           # G1Affine29: two coordinates of 14 x 28-bit limbs, rows packed (the 9 MB table is cache resident)
BUCKET_BYTES = 224                        # G1Xyzz29
# operands of the asm statement (msm.hip): %0 is its one OUTPUT (a VGPR: 1 on a lane that met P = +-Q and must be redone by the C++ formulas,
# else 0), the inputs follow -- "s" but the last ("v")
OPERANDS = ["(out) trouble", "table", "entries of this blob", "bucket_start of this blob", "perm of this blob", "buckets of this blob", "n_heavy", "rank"]


def row_address_from_entry(p):
    """sNEGN, ADDR <- sign and table row of the entry in ENT (all lanes; lanes without an entry compute garbage and load nothing)"""
    e = p.emit
    e("v_cmp_gt_u32", sp(G.sNEGN), v(ENT), s(sM31))
    e("v_and_b32", v(T3), s(sM31), v(ENT))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(T3), v(V112), sp(G.sTABLE))


def build():
    p = Prog()
    e = p.emit
    e("comment", "operands -> fixed registers")
    e("s_mov_b64", sp(G.sTABLE), opnd(1))
    e("s_mov_b64", sp(sENT), opnd(2))
    e("s_mov_b64", sp(sBS), opnd(3))
    e("s_mov_b64", sp(sPERM), opnd(4))
    e("s_mov_b64", sp(G.sOUT), opnd(5))
    e("s_mov_b32", s(sNHEAVY), opnd(6))
    e("v_mov_b32", v(RANK), opnd(7))
    for i in range(14):
        e("s_mov_b32", s(G.sMOD[i]), lit(MOD[i]))
    e("s_mov_b32", s(G.sINV), lit(INV))
    e("s_mov_b32", s(G.sMASK), lit(MASK))
    e("s_mov_b32", s(G.sINVP), lit(INVP))
    e("s_mov_b32", s(sM31), lit(0x7fffffff))
    e("v_mov_b32", v(V112), lit(ROW_BYTES))
    e("s_mov_b64", sp(G.sINF), lit(-1))
    e("s_mov_b64", sp(G.sTROUBLE), lit(0))
    # light lanes: rank >= n_heavy (the others belong to the wave-per-bucket path and end up with an empty list here)
    e("v_cmp_gt_u32", sp(G.sTMP), s(sNHEAVY), v(RANK))
    e("s_andn2_b64", sp(sLIGHT), lit(-1), sp(G.sTMP))
    # bucket = perm[rank]; begin, end = bucket_start[bucket], bucket_start[bucket + 1]
    e("v_mov_b32", v(T1), lit(4))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(RANK), v(T1), sp(sPERM))
    e("global_load_dword", v(BKT), vp(ADDR[0]), ("off",))
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(BKT), v(T1), sp(sBS))
    e("global_load_dwordx2", vp(KCUR), vp(ADDR[0]), ("off",))
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    e("s_mov_b64", EXEC, sp(G.sTMP))
    e("v_mov_b32", v(KEND), v(KCUR))
    e("s_mov_b64", EXEC, lit(-1))
    e("v_cmp_lt_u32", sp(G.sVALIDN), v(KCUR), v(KEND))
    # the first entry, and the second one a step ahead
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(KCUR), v(T1), sp(sENT))
    e("v_add_u32", v(T3), lit(1), v(KCUR))
    e("v_cmp_lt_u32", VCC, v(T3), v(KEND))
    e("s_mov_b64", EXEC, sp(G.sVALIDN))
    e("global_load_dword", v(ENT), vp(ADDR[0]), ("off",))
    e("s_mov_b64", EXEC, VCC)
    e("global_load_dword", v(ENT1), vp(ADDR[0]), ("off",), offset=4)
    e("s_mov_b64", EXEC, lit(-1))
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    row_address_from_entry(p)
    e("s_mov_b64", EXEC, sp(G.sVALIDN))       # (a lane without an entry has no row)
    G.row_loads(p)
    e("s_mov_b64", EXEC, lit(-1))
    # ---------------- the loop: one mixed addition per pass on every lane that still has an entry
    e(".p2align", ("raw", "3"))
    p.label("L_loop%=")
    e("s_mov_b64", sp(G.sVALID), sp(G.sVALIDN))
    e("s_mov_b64", sp(G.sNEG), sp(G.sNEGN))
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    G.emit_first_row_and_negation(p)
    # the entry after this one (all lanes): cursor, validity, the prefetched entry becomes current, the one after it is asked for
    e("v_add_u32", v(KCUR), lit(1), v(KCUR))
    e("v_cmp_lt_u32", sp(G.sVALIDN), v(KCUR), v(KEND))
    e("v_mov_b32", v(ENT), v(ENT1))
    e("v_add_u32", v(T3), lit(1), v(KCUR))
    e("v_mov_b32", v(T1), lit(4))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(KCUR), v(T1), sp(sENT))
    e("v_cmp_lt_u32", VCC, v(T3), v(KEND))
    e("s_mov_b64", EXEC, VCC)
    e("global_load_dword", v(ENT1), vp(ADDR[0]), ("off",), offset=4)
    e("s_mov_b64", EXEC, lit(-1))
    row_address_from_entry(p)
    e("s_cmp_eq_u64", sp(G.sVALIDN), lit(0))
    e("s_cselect_b32", s(G.sMORE), lit(0), lit(1))
    pp_, rr, nx, ny, zz, zzz = G.emit_madd_part_a(p)
    # ---- the row is dead: the next one is gathered into its registers
    e("s_mov_b64", EXEC, sp(G.sVALIDN))
    e("s_cbranch_execz", ("label", "L_no_loads%="))
    G.row_loads(p)
    p.label("L_no_loads%=")
    G.emit_madd_part_b(p, pp_, rr, nx, ny, zz, zzz)
    e("s_mov_b64", EXEC, lit(-1))
    e("s_cmp_eq_u32", s(G.sMORE), lit(0))
    e("s_cbranch_scc0", ("label", "L_loop%="))
    # ---------------- epilogue: the form the bucket reduction reads (G1Xyzz29: every coordinate below 2p, infinity = literal zeros)
    e("s_mov_b64", EXEC, sp(G.sINF))
    e("s_cbranch_execz", ("label", "L_no_inf%="))
    for r in NX + NY + ZZ + ZZZ:
        e("v_mov_b32", v(r), lit(0))
    p.label("L_no_inf%=")
    e("s_andn2_b64", sp(G.sTMP), lit(-1), sp(G.sINF))
    e("s_mov_b64", EXEC, sp(G.sTMP))
    e("s_cbranch_execz", ("label", "L_no_sum%="))
    assert nx.B <= 32 and ny.B <= 32 and nx.L == 1 and ny.L == 1
    for i in range(14):                                  # X = 32p - (-X), Y = 32p - (-Y): limb-wise, borrowed form (no limb goes negative)
        e("v_sub_u32", v(NX[i]), lit(KP32_1[i]), v(NX[i]))
        e("v_sub_u32", v(NY[i]), lit(KP32_1[i]), v(NY[i]))
    for i in range(14):
        e("v_mov_b32", v(U[i]), lit(R1[i]))              # 1 in Montgomery form
    one = Val(U, 1, 1)
    # (each product writes its result over its own first operand, limb by limb)
    interleave(p, chain_mul(Val(NX, 32, 2), one, NX, M1, ACC1, T1), chain_mul(Val(NY, 32, 2), one, NY, M2, ACC2, T2r))
    interleave(p, chain_mul(Val(ZZ, zz.B, 1), one, ZZ, M1, ACC1, T1), chain_mul(Val(ZZZ, zzz.B, 1), one, ZZZ, M2, ACC2, T2r))
    p.label("L_no_sum%=")
    e("s_mov_b64", EXEC, sp(sLIGHT))
    e("s_cbranch_execz", ("label", "L_no_store%="))
    e("v_mov_b32", v(T1), lit(BUCKET_BYTES))
    e("v_mad_u64_u32", vp(ADDR[0]), VCC, v(BKT), v(T1), sp(G.sOUT))
    for k in range(14):
        e("global_store_dwordx4", vp(ADDR[0]), ("v4", 4 * k), ("off",), offset=16 * k)
    p.label("L_no_store%=")
    e("s_mov_b64", EXEC, lit(-1))
    # the statement's output: 1 on the lanes that met P = +-Q -- the C++ code behind the statement recomputes exactly their buckets
    e("v_cndmask_b32", opnd(0), lit(0), lit(1), sp(G.sTROUBLE), e64=True)
    e("s_waitcnt", ("raw", "vmcnt(0)"))
    return p


def render(p):
    lines = ["// generated by tools/gen_bucket_asm.py -- do not edit (python tools/gen_bucket_asm.py)",
             "// %d instructions, %d of them VALU; VGPRs v0..v%d, SGPRs s%d..s%d" % (
                 sum(1 for i in p.ins if i[0] not in ("label", "comment")), p.count_valu(), G.NUM_VGPRS - 1, G.SBASE, NUM_SGPRS - 1)]
    for t in p.text():
        t = t.replace("\\", "\\\\").replace('"', '\\"')
        lines.append('"%s\\n"' % t)
    return "\n".join(lines) + "\n"


def clobbers():
    return ", ".join(['"v%d"' % i for i in range(G.NUM_VGPRS)] + ['"s%d"' % i for i in range(G.SBASE, NUM_SGPRS) if i not in (32, 33, 34)] +
                     ['"vcc"', '"memory"'])


# ---- self-test: one lane = one bucket ----------------------------------------------------------------------------------
def selftest(seed=1, n_entries=20, heavy=False, collide_at=0, collide_neg=False, verbose=True):
    """One lane through the whole stream: a bucket of `n_entries` entries over a synthetic 9 MB-style table (entry index -> an honest
    multiple of the generator), result against affine big-int arithmetic in the reduction's format (X, Y, ZZ, ZZZ below 2p).
    heavy: the lane's rank is below n_heavy -- it must store nothing. collide_at = t: the t-th row IS the sum of the rows before it
    (negated with collide_neg): the redo flag must come up."""
    rnd = random.Random(seed)
    prog = build()
    table_addr, ent_addr, bs_addr, perm_addr, out_addr, redo_addr = (0x100000000000, 0x200000000000, 0x300000000000, 0x400000000000,
                                                                      0x500000000000, 0x600000000000)
    rank, bucket = rnd.randrange(4096), rnd.randrange(4096)
    begin = rnd.randrange(60000)
    entries = []
    pts = {}
    served = []
    for k in range(n_entries):
        idx = rnd.randrange(20 * 4096)
        while idx in pts:
            idx = rnd.randrange(20 * 4096)
        pt = G.ec_mul(rnd.randrange(1, 1 << 64), G.G1)
        neg = rnd.random() < 0.5
        if collide_at and k == collide_at - 1:
            acc = None
            for q in served:
                acc = G.ec_add(acc, q)
            pt = acc if not collide_neg else (acc[0], (P - acc[1]) % P)
            neg = False
        pts[idx] = pt
        served.append((pt[0], (P - pt[1]) % P) if neg else pt)
        entries.append(idx | (0x80000000 if neg else 0))

    def row_words(idx):
        pt = pts[idx]
        xs = G.to_mont(pt[0]) + (P if rnd.random() < 0.5 else 0)       # weakly reduced, as a table row may be
        ys = G.to_mont(pt[1]) + (P if rnd.random() < 0.5 else 0)
        return G.limbs(xs) + G.limbs(ys)
    cache = {}
    stored = {}

    def rd(addr, n):
        if addr >= redo_addr:
            raise AssertionError("read of the redo flag")
        if addr >= perm_addr and addr < out_addr:
            assert addr == perm_addr + 4 * rank and n == 1
            return [bucket]
        if addr >= bs_addr and addr < perm_addr:
            assert addr == bs_addr + 4 * bucket and n == 2
            return [begin, begin + n_entries]
        if addr >= ent_addr and addr < bs_addr:
            k = (addr - ent_addr) // 4 - begin
            assert 0 <= k < n_entries and n == 1, "entry read outside the lane's list"
            return [entries[k]]
        off = addr - table_addr
        idx, w0 = off // ROW_BYTES, (off % ROW_BYTES) // 4
        if idx not in cache:
            cache[idx] = row_words(idx)
        return cache[idx][w0:w0 + n]

    def wr(addr, words):
        for k, wv in enumerate(words):
            stored[addr + 4 * k] = wv

    n_heavy = rank + 1 if heavy else rnd.randrange(0, rank + 1)
    ops = [0xdead, table_addr, ent_addr, bs_addr, perm_addr, out_addr, n_heavy, rank]
    sim = Sim(prog, ops, rd, wr)
    steps = sim.run()
    redo = sim.ops[0]
    assert redo in (0, 1)
    if heavy:
        assert not stored, "a lane of a heavy bucket stored something"
        return 0
    if collide_at:
        return redo
    base = out_addr + BUCKET_BYTES * bucket
    got = [stored[base + 4 * k] for k in range(56)]
    assert set(stored) == set(base + 4 * k for k in range(56))
    want = None
    for q in served:
        want = G.ec_add(want, q)
    xv, yv, zzv, zzzv = (sum(x << (W * i) for i, x in enumerate(got[14 * t:14 * t + 14])) for t in range(4))
    if want is None:
        ok = all(x == 0 for x in got)
    else:
        ok = all(val < 2 * P for val in (xv, yv, zzv, zzzv)) and all(x <= MASK for x in got)
        rinv = pow(G.RMONT, -1, P)
        x_, y_, zz_, zzz_ = (val * rinv % P for val in (xv, yv, zzv, zzzv))
        ok = ok and (x_ * pow(zz_, -1, P) % P, y_ * pow(zzz_, -1, P) % P) == want and (zz_ ** 3 - zzz_ ** 2) % P == 0
    if verbose:
        print("bucket selftest seed=%d entries=%d: %s, %d instructions executed, %d VALU (%.0f per row), redo=%d" % (
            seed, n_entries, "ok" if ok else "MISMATCH", steps, sim.valu_executed, sim.valu_executed / max(1, n_entries), redo))
    assert ok and redo == 0
    return sim.valu_executed


def main():
    if "--selftest" in sys.argv:
        for seed, n in ((1, 20), (2, 1), (3, 33), (4, 0), (5, 2)):
            selftest(seed, n)
        selftest(6, 12, heavy=True)
        print("a lane of a heavy bucket stores nothing")
        for t, neg in ((2, False), (7, True), (15, False)):
            assert selftest(10 + t, 18, collide_at=t, collide_neg=neg) == 1
        print("a row equal / opposite to the accumulator raises the redo flag")
        return
    text = render(build())
    if "--check" in sys.argv:
        assert open(OUT).read() == text, "csrc/bucket_asm.inc is stale: run python tools/gen_bucket_asm.py"
        print("bucket_asm.inc matches its generator")
        return
    with open(OUT, "w") as f:
        f.write(text)
    with open(OUT.replace(".inc", "_clobbers.inc"), "w") as f:
        f.write("// generated by tools/gen_bucket_asm.py -- do not edit\n" + clobbers() + "\n")
    print("wrote %s: %d VALU instructions in the stream" % (OUT, build().count_valu()))


if __name__ == "__main__":
    main()
