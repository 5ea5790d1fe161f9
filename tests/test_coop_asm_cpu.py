"""CPU: the cooperative MSM kernel (lambdaworks_kzg_amd/csrc/coop_asm.inc: four lanes per group addition, waves handing their sums on
through memory) is what tools/gen_coop_asm.py writes, and the generator's WAVE-level simulator -- 64 lanes, exact 64-bit columns,
no 32-bit wrap where the algorithm relies on none, no DPP read of a lane EXEC disables, no register read before its s_waitcnt, the
hand-off memory shared between the simulated waves -- runs that instruction stream to the same point as affine big-int arithmetic:
the quad addition with P, Q, infinity in every lane position, small multi-wave problems in any arrival order, zero digits, empty
scalars, and equal / opposite partial sums (the redo flag that sends the blob to the complete-branches kernel). Host logic only; the GPU
parity tests (tests/test_gpu_coop.py) run the assembled kernel."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def test_coop_asm_inc_is_current():
    import gen_coop_asm as C
    assert open(C.OUT).read() == C.render(C.build())
    assert open(C.OUT.replace(".inc", "_clobbers.inc")).read().split("\n", 1)[1].strip() == C.clobbers()
    assert C.NUM_VGPRS <= 168 and C.NUM_SGPRS <= 100       # three waves per SIMD; s100 / s101 are XNACK_MASK on gfx950


def _add_prog(C, G):
    p = C.CProg()
    e = p.emit
    for i in range(14):
        e("s_mov_b32", G.s(C.sMOD[i]), G.lit(G.MOD[i]))
    e("s_mov_b32", G.s(C.sINV), G.lit(G.INV))
    e("s_mov_b32", G.s(C.sMASK), G.lit(G.MASK))
    e("s_mov_b32", G.s(C.sINVP), G.lit(G.INVP))
    for k, q in enumerate((C.sQ0, C.sQ1, C.sQ2, C.sQ3)):
        e("s_mov_b32", G.s(q[0]), G.lit(0x11111111 << k))
        e("s_mov_b32", G.s(q[1]), G.lit(0x11111111 << k))
    e("s_mov_b64", G.sp(C.sADDM), G.opnd(0))
    e("s_mov_b64", G.sp(C.sTROUBLE), G.lit(0))
    e("s_or_b64", G.sp(C.sTMP), G.sp(C.sQ0), G.sp(C.sQ1))
    C.add_body(p)
    return p


def _coords(G, rnd, pt, top):
    """a point as the stream keeps it between additions: X < 10p carried, Y < 6p with limbs < 3 2^28, ZZ, ZZZ < 2p carried"""
    P = G.P
    t = rnd.randrange(1, P)
    zz, zzz = t * t % P, t * t * t % P
    X, Y = pt[0] * zz % P, pt[1] * zzz % P
    ylimbs = [a + b + c for a, b, c in zip(G.limbs(G.to_mont(Y) + P * rnd.randrange(0, 2 if top else 1)), G.limbs(P * (2 if top else 0)),
                                            G.limbs(P * (2 if top else 0)))]
    return [G.limbs(G.to_mont(X) + P * (9 if top else rnd.randrange(0, 9))), ylimbs,
            G.limbs(G.to_mont(zz) + P * (1 if top else rnd.randrange(0, 2))), G.limbs(G.to_mont(zzz) + P * (1 if top else rnd.randrange(0, 2)))]


def test_quad_addition_on_a_simulated_wave():
    """A <- A + B on sixteen quads at once, operands at the top of their declared bounds on some quads; the quads the mask leaves out
    keep their point; about 2250 vector instructions per addition (one lane per addition: 6000)"""
    import gen_coop_asm as C
    import gen_direct_asm as G
    P = G.P
    rnd = random.Random(5)
    prog = _add_prog(C, G)
    for addm in ((1 << 64) - 1, 0x0F0F00FF0000FFF0):
        sim = C.WaveSim(prog, [addm], {}, lambda a, n: None)
        As, Bs = [], []
        for q in range(16):
            a, b = G.ec_mul(rnd.randrange(2, R), G.G1), G.ec_mul(rnd.randrange(2, R), G.G1)
            As.append(a)
            Bs.append(b)
            ca, cb = _coords(G, rnd, a, q % 3 == 0), _coords(G, rnd, b, q % 3 == 1)
            for c in range(4):
                for i in range(14):
                    sim.vr[C.HA[i]][4 * q + c] = ca[c][i]
                    sim.vr[C.HB[i]][4 * q + c] = cb[c][i]
        sim.run()
        for q in range(16):
            co = [[int(sim.vr[C.HA[i]][4 * q + c]) for i in range(14)] for c in range(4)]
            x, y, zz, zzz = (G.from_mont_limbs(l) for l in co)
            got = (x * pow(zz, -1, P) % P, y * pow(zzz, -1, P) % P)
            taking = (addm >> (4 * q)) & 1
            assert got == (G.ec_add(As[q], Bs[q]) if taking else As[q]), (hex(addm), q)
            if taking:    # what the next addition is told about its operands
                vals = [sum(c << (G.W * i) for i, c in enumerate(l)) for l in co]
                assert vals[0] < 10 * P and vals[1] < 6 * P and vals[2] < 2 * P and vals[3] < 2 * P
                assert all(c < (1 << 28) for c in co[0][:13] + co[2][:13] + co[3][:13]) and all(c < 3 << 28 for c in co[1][:13])
        assert sim.sr[C.sTROUBLE[0]] == 0 and sim.sr[C.sTROUBLE[1]] == 0
        assert sim.valu_executed < 2300


@pytest.mark.parametrize("which", ["equal", "opposite"])
def test_equal_or_opposite_operands_raise_the_trouble_mask(which):
    """P = +-Q on quads 3 and 12 only: exactly their lane-0 bits come up in the trouble mask (the kernel turns a non-zero mask into the
    blob's redo flag); every other quad's sum is right"""
    import gen_coop_asm as C
    import gen_direct_asm as G
    P = G.P
    rnd = random.Random(8)
    prog = _add_prog(C, G)
    sim = C.WaveSim(prog, [(1 << 64) - 1], {}, lambda a, n: None)
    As, Bs = [], []
    for q in range(16):
        a = G.ec_mul(rnd.randrange(2, R), G.G1)
        b = G.ec_mul(rnd.randrange(2, R), G.G1)
        if q in (3, 12):
            b = a if which == "equal" else (a[0], (-a[1]) % P)
        As.append(a)
        Bs.append(b)
        ca, cb = _coords(G, rnd, a, q == 3), _coords(G, rnd, b, q == 12)
        for c in range(4):
            for i in range(14):
                sim.vr[C.HA[i]][4 * q + c] = ca[c][i]
                sim.vr[C.HB[i]][4 * q + c] = cb[c][i]
    sim.run()
    trouble = sim.sr[C.sTROUBLE[0]] | (sim.sr[C.sTROUBLE[1]] << 32)
    assert trouble == (1 << 12) | (1 << 48)
    for q in range(16):
        if q in (3, 12):
            continue
        co = [[int(sim.vr[C.HA[i]][4 * q + c]) for i in range(14)] for c in range(4)]
        x, y, zz, zzz = (G.from_mont_limbs(l) for l in co)
        assert (x * pow(zz, -1, P) % P, y * pow(zzz, -1, P) % P) == G.ec_add(As[q], Bs[q])


def test_trouble_in_every_quad_position():
    """P = +-Q in EVERY lane position. Sixteen runs: P = Q on quad q, P = -Q on quad q + 5, quad q + 9 left out of the
    addition's mask (it keeps its point and can raise nothing, although its operands are equal): the trouble mask is exactly the two
    lane-0 bits, every other quad's sum is right"""
    import gen_coop_asm as C
    import gen_direct_asm as G
    P = G.P
    rnd = random.Random(88)
    prog = _add_prog(C, G)
    for q0 in range(16):
        qe, qo, qm = q0, (q0 + 5) % 16, (q0 + 9) % 16
        addm = ((1 << 64) - 1) & ~(0xF << (4 * qm))
        sim = C.WaveSim(prog, [addm], {}, lambda a, n: None)
        As, Bs = [], []
        for q in range(16):
            a = G.ec_mul(rnd.randrange(2, R), G.G1)
            b = G.ec_mul(rnd.randrange(2, R), G.G1)
            if q in (qe, qm):
                b = a
            if q == qo:
                b = (a[0], (-a[1]) % P)
            As.append(a)
            Bs.append(b)
            ca, cb = _coords(G, rnd, a, q == qe), _coords(G, rnd, b, q == qo)
            for c in range(4):
                for i in range(14):
                    sim.vr[C.HA[i]][4 * q + c] = ca[c][i]
                    sim.vr[C.HB[i]][4 * q + c] = cb[c][i]
        sim.run()
        trouble = sim.sr[C.sTROUBLE[0]] | (sim.sr[C.sTROUBLE[1]] << 32)
        assert trouble == (1 << (4 * qe)) | (1 << (4 * qo)), (q0, hex(trouble))
        for q in range(16):
            if q in (qe, qo):
                continue
            co = [[int(sim.vr[C.HA[i]][4 * q + c]) for i in range(14)] for c in range(4)]
            x, y, zz, zzz = (G.from_mont_limbs(l) for l in co)
            want = As[q] if q == qm else G.ec_add(As[q], Bs[q])
            assert (x * pow(zz, -1, P) % P, y * pow(zzz, -1, P) % P) == want, (q0, q)


@pytest.mark.parametrize("kw,order", [
    (dict(seed=1, c=4, nw=4, wtop=3, log_points=4, rpq=2), "shuffle"),                    # 2 waves, one hand-off
    (dict(seed=2, c=5, nw=3, wtop=4, log_points=5, rpq=2), "reverse"),                    # an odd window count: quads with one row
    (dict(seed=3, c=4, nw=4, wtop=3, log_points=6, rpq=4, row_bytes=112), None),          # packed rows, one window group
    (dict(seed=4, c=3, nw=5, wtop=2, log_points=8, rpq=1), "shuffle"),                    # 80 waves: 80 -> 5 -> 1, a last group of 5
    (dict(seed=5, c=4, nw=2, wtop=4, log_points=4, rpq=2), None),                         # one wave: no hand-off at all
])
def test_small_problems_through_every_wave(kw, order):
    import gen_coop_asm as C
    prob = C.Problem(**kw)
    C.run_problem(prob, order=order)
    assert prob.result() == prob.want()
    assert prob.mem.get(prob.REDO, 0) == 0


def test_zero_digits_empty_scalars_and_the_empty_sum():
    """digits that vanish (no row: B at infinity), scalars that are zero (a quad with nothing to add), a blob of zeros (the sum is the
    point at infinity: literal zeros in the library's layout), one non-zero scalar, scalars at the top of their range"""
    import gen_coop_asm as C
    geo = dict(c=4, nw=4, wtop=3, log_points=4, rpq=2)
    top = (1 << 15) - 1
    rnd = random.Random(3)
    prob = C.Problem(seed=6, scalars=[0] * 16, **geo)
    C.run_problem(prob)
    assert prob.result() is None and prob.want() is None
    for scalars in ([0] * 9 + [0x0101] + [0] * 6, [rnd.choice((0, 0x0010, 0x7000, 0x0f0f, top)) for _ in range(16)], [top] * 16,
                    [0x0888] * 16):    # (0x888: every signed digit at its extreme, H = 8)
        prob = C.Problem(seed=7, scalars=scalars, **geo)
        C.run_problem(prob, order="reverse")
        assert prob.result() == prob.want(), scalars
        assert prob.mem.get(prob.REDO, 0) == 0


@pytest.mark.parametrize("which", ["equal", "opposite"])
def test_equal_partial_sums_raise_the_redo_flag(which):
    """two quads of a wave carry the same point and the same scalar (or the opposite point): the first tree level adds P to +-P, which the
    formulas cannot do -- the blob's redo flag must come up (the complete-branches kernel recomputes the blob)"""
    import gen_coop_asm as C
    rnd = random.Random(11)
    ks = [rnd.randrange(1, 1 << 60) for _ in range(16)]
    ks[5] = ks[4] if which == "equal" else R - ks[4]
    scalars = [rnd.randrange(1, 1 << 15) for _ in range(16)]
    scalars[5] = scalars[4]
    prob = C.Problem(seed=12, c=4, nw=4, wtop=3, log_points=4, rpq=2, scalars=scalars, point_ks=ks)
    C.run_problem(prob)
    assert prob.mem.get(prob.REDO, 0) == 1


def test_subgroup_asm_inc_is_current_and_judges_points_on_a_simulated_wave():
    """the cooperative subgroup test (csrc/subgroup_asm.inc, tools/gen_subgroup_asm.py: doublings in three rounds of one product per lane, the
    cooperative kernel's addition, the bits of |z| as a scalar loop): the committed files are what the generator writes; a simulated wave judges
    points of G1 (verdict 1), points of E(Fp) outside G1 (0), a point of small order (0 or `undetermined`, never 1), skips slots whose kind is not 0
    and slots beyond n; about 1780 vector instructions per doubling (one lane per point: nine products, ~4300)"""
    import gen_subgroup_asm as S
    import gen_direct_asm as G
    assert open(S.OUT).read() == S.render(S.build())
    assert open(S.OUT.replace(".inc", "_clobbers.inc")).read().split("\n", 1)[1].strip() == S.clobbers()
    assert S.NUM_VGPRS <= 168 and S.NUM_SGPRS <= 100
    rnd = random.Random(77)
    prog = S.build()
    pts, kinds, want = [], [], []
    for k in range(5):
        pts.append(G.ec_mul(rnd.randrange(1, R), G.G1)); kinds.append(0); want.append(1)
    for k in range(3):
        pts.append(S.curve_point_outside_g1(rnd)); kinds.append(0); want.append(0)
    pts.append(G.ec_mul(R, S.curve_point_outside_g1(rnd))); kinds.append(0); want.append(None)    # small order: 0 or 2
    pts.append(G.G1); kinds.append(0); want.append(1)                                                # the generator itself
    pts.append(None); kinds.append(1); want.append(0xEE)                                             # infinity: skipped, its word untouched
    pts.append(S.curve_point_outside_g1(rnd)); kinds.append(2); want.append(0xEE)                    # invalid encoding upstream: skipped
    pts.append(G.ec_mul(rnd.randrange(1, R), G.G1)); kinds.append(0x100); want.append(1)             # the sign flag in bit 8 is not a kind
    got, sim = S.run_points(pts, kinds, prog)
    for g, w in zip(got, want):
        assert (g in (0, 2)) if w is None else g == w, (got, want)
    assert sim.valu_executed < 126 * 1850 + 10 * 2300 + 4000
