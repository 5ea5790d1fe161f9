// g1.cuh -- BLS12-381 G1 (y^2 = x^3 + 4) group law for gfx950, host+device.
//
// Replaces on the hot path what the reference gets from lambdaworks-math's
// ShortWeierstrassProjectivePoint<BLS12381Curve>::{operate_with, operate_with_self, neg,
// to_affine} (un-vendored; call sites /root/reference/src/lib.rs:664-688,
// src/compression.rs:25,42,98) and restates the reference's own
// compress_g1_point / decompress_g1_point / check_point_is_in_subgroup
// (/root/reference/src/compression.rs:22-103) for device use.
//
// NOT a translation of the reference's homogeneous-projective formulas: bucket accumulators use
// extended-Jacobian XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; EFD "xyzz", a = 0)
// because the mixed addition with an affine SRS point is 8M+2S against 11M+ for projective, and
// SRS points stay affine (96 B) in HBM. All formulas are complete by explicit branches
// (P+P, P+(-P), P+O, O+P): formula-generated blobs do hit them (hard part c).
#pragma once
#include "field.cuh"
#include "field29.cuh"

namespace lwk {

// ---- representation-generic point types -----------------------------------------------------------
// Two instantiations: saturated Fp (host, setup kernels) and the lazy 28-bit-limb F29<B> of
// field29.cuh (MSM hot loop), whose value bounds (multiples of p) are part of the member types below
// and are re-derived by the compiler through every formula (`auto` temporaries).

LWK_HD bool literal_zero(const Fp &a) { return a.is_zero(); }
template <int B, bool I>
LWK_HD bool literal_zero(const F29<B, I> &a) { return a.is_literal_zero(); }

template <class FX, class FY>
struct alignas(16) AffineT {
    FX x;
    FY y;  // the point at infinity is not representable (never needed in tables)
};

template <class FX, class FY, class FZ>
struct alignas(16) XyzzT {
    FX x;
    FY y;
    FZ zz, zzz;  // infinity <=> zz is literally zero

    LWK_HD static XyzzT infinity() {
        XyzzT r;
        r.x = FX::zero();
        r.y = FY::zero();
        r.zz = FZ::zero();
        r.zzz = FZ::zero();
        return r;
    }
    LWK_HD bool is_inf() const { return literal_zero(zz); }
    template <class QX, class QY>
    LWK_HD static XyzzT from_affine(const QX &qx, const QY &qy) {
        XyzzT r;
        r.x = qx;
        r.y = qy;
        r.zz = FZ::one();
        r.zzz = FZ::one();
        return r;
    }
};

typedef AffineT<Fp, Fp> G1Affine;    // 96 B, Montgomery 12x32
typedef XyzzT<Fp, Fp, Fp> G1Xyzz;    // 192 B
// hot-loop forms. Bounds: table coordinates < 2p; an accumulator leaves every formula below with
// X < 14p, Y < 6p, ZZ, ZZZ < 2p (derivation in DESIGN.md section 4a; enforced by static_asserts).
typedef AffineT<F29<2>, F29<2>> G1Affine29;          // 112 B
typedef XyzzT<F29<14>, F29<6>, F29<2>> G1Xyzz29;     // 224 B
// same layouts, field products inlined (see F29's INL): the accumulate loop's view of the same memory
typedef AffineT<F29<2, true>, F29<2, true>> G1Affine29i;
typedef XyzzT<F29<14, true>, F29<6, true>, F29<2, true>> G1Xyzz29i;
static_assert(sizeof(G1Affine29i) == sizeof(G1Affine29) && sizeof(G1Xyzz29i) == sizeof(G1Xyzz29), "layout");

LWK_HD Fp fp_from_u32(uint32_t v) {
    uint32_t raw[12];
#pragma unroll
    for (int i = 0; i < 12; i++) raw[i] = (i == 0) ? v : 0u;
    return fe_from_raw<FpParams>(raw);
}

LWK_HD bool g1_on_curve(const G1Affine &p) {
    Fp l = sqr(p.y);
    Fp r = sqr(p.x) * p.x + fp_from_u32(4);
    return l == r;
}

// 2Q for affine Q (mdbl-2008-s-1, a = 0)
template <class X, class QX, class QY>
LWK_HD X xyzz_dbl_affine(const QX &qx, const QY &qy) {
    if (qy.is_zero()) return X::infinity();  // order-2 point: cannot occur in G1, kept for completeness
    auto u = dbl(qy);
    auto v = sqr(u);
    auto w = u * v;
    auto s = qx * v;
    auto xx = sqr(qx);
    auto m = dbl(xx) + xx;
    X r;
    auto x3 = normed(sqr(m) - dbl(s));  // a long lazy chain ends here (a no-op with 29-bit limbs)
    r.x = x3;
    r.y = mul_sub(m, s - x3, w, qy);
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2P (dbl-2008-s-1, a = 0)
template <class X>
LWK_HD X xyzz_dbl(const X &p) {
    if (p.is_inf() || p.y.is_zero()) return X::infinity();
    auto u = dbl(p.y);
    auto v = sqr(u);
    auto w = u * v;
    auto s = p.x * v;
    auto xx = sqr(p.x);
    auto m = dbl(xx) + xx;
    X r;
    auto x3 = normed(sqr(m) - dbl(s));  // a long lazy chain ends here (a no-op with 29-bit limbs)
    r.x = x3;
    r.y = mul_sub(m, s - x3, w, p.y);
    r.zz = v * p.zz;
    r.zzz = w * p.zzz;
    return r;
}

// acc + Q for affine Q = (qx, qy) (madd-2008-s), complete
template <class X, class QX, class QY>
LWK_HD X xyzz_madd(const X &acc, const QX &qx, const QY &qy) {
    if (acc.is_inf()) return X::from_affine(qx, qy);
    auto u2 = qx * acc.zz;
    auto s2 = qy * acc.zzz;
    auto pp_ = u2 - acc.x;
    auto rr = s2 - acc.y;
    if (pp_.is_zero()) {
        if (rr.is_zero()) return xyzz_dbl_affine<X>(qx, qy);
        return X::infinity();
    }
    auto pp = sqr(pp_);
    auto ppp = pp_ * pp;
    auto qq = acc.x * pp;
    X r;
    auto x3 = normed(sqr(rr) - ppp - dbl(qq));  // a long lazy chain ends here (a no-op with 29-bit limbs)
    r.x = x3;
    r.y = mul_sub(rr, qq - x3, acc.y, ppp);
    r.zz = acc.zz * pp;
    r.zzz = acc.zzz * ppp;
    return r;
}

LWK_HD G1Xyzz xyzz_madd(const G1Xyzz &acc, const G1Affine &q) { return xyzz_madd(acc, q.x, q.y); }

// The same addition, split around its last use of the affine operand: (1) the two products that bring Q to the
// accumulator's scale plus the degenerate cases (acc = O, P + P, P - P), (2) everything else. `row_is_dead()` runs
// between the two -- the hot loops issue the gather of their NEXT table row there, into the registers Q just
// vacated: no second row buffer, no copy, and eight products of time for the gather to land.
template <class X, class QX, class QY, class F>
LWK_HD void xyzz_madd_split(X &acc, const QX &qx, const QY &qy, F &&row_is_dead) {
    bool done = false;
    decltype(qx * acc.zz) u2;
    decltype(qy * acc.zzz) s2;
    if (acc.is_inf()) {
        acc = X::from_affine(qx, qy);
        done = true;
    } else {
        u2 = qx * acc.zz;
        s2 = qy * acc.zzz;
        if ((u2 - acc.x).is_zero()) {
            if ((s2 - acc.y).is_zero()) acc = xyzz_dbl_affine<X>(qx, qy);
            else acc = X::infinity();
            done = true;
        }
    }
    row_is_dead();
    if (done) return;
    auto pp_ = u2 - acc.x;
    auto rr = s2 - acc.y;
    auto pp = sqr(pp_);
    auto ppp = pp_ * pp;
    auto qq = acc.x * pp;
    auto x3 = normed(sqr(rr) - ppp - dbl(qq));
    X r;
    r.x = x3;
    r.y = mul_sub(rr, qq - x3, acc.y, ppp);
    r.zz = acc.zz * pp;
    r.zzz = acc.zzz * ppp;
    acc = r;
}

// a + b (add-2008-s), complete
template <class X>
LWK_HD X xyzz_add(const X &a, const X &b) {
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    auto u1 = a.x * b.zz;
    auto u2 = b.x * a.zz;
    auto s1 = a.y * b.zzz;
    auto s2 = b.y * a.zzz;
    auto pp_ = u2 - u1;
    auto rr = s2 - s1;
    if (pp_.is_zero()) {
        if (rr.is_zero()) return xyzz_dbl(a);
        return X::infinity();
    }
    auto pp = sqr(pp_);
    auto ppp = pp_ * pp;
    auto qq = u1 * pp;
    X r;
    auto x3 = normed(sqr(rr) - ppp - dbl(qq));  // a long lazy chain ends here (a no-op with 29-bit limbs)
    r.x = x3;
    r.y = mul_sub(rr, qq - x3, s1, ppp);
    r.zz = a.zz * b.zz * pp;
    r.zzz = a.zzz * b.zzz * ppp;
    return r;
}

LWK_HD G1Affine affine_neg(const G1Affine &p) {
    G1Affine r;
    r.x = p.x;
    r.y = neg(p.y);
    return r;
}

// requires !p.is_inf(); one field inversion
LWK_HD G1Affine xyzz_to_affine(const G1Xyzz &p) {
    Fp i = inv(p.zz * p.zzz);
    Fp izz = i * p.zzz;   // 1/ZZ
    Fp izzz = i * p.zz;   // 1/ZZZ
    G1Affine r;
    r.x = p.x * izz;
    r.y = p.y * izzz;
    return r;
}

// same, from the hot-loop representation to canonical saturated coordinates
LWK_HD G1Affine xyzz_to_affine(const G1Xyzz29 &p) {
    auto i = f29_inv(p.zz * p.zzz);
    auto izz = i * p.zzz;
    auto izzz = i * p.zz;
    G1Affine r;
    r.x = f29_to_fp(p.x * izz);
    r.y = f29_to_fp(p.y * izzz);
    return r;
}

// same, staying in the hot-loop representation
LWK_HD G1Affine29 xyzz29_to_affine29(const G1Xyzz29 &p) {
    auto i = f29_inv(p.zz * p.zzz);
    auto izz = i * p.zzz;
    auto izzz = i * p.zz;
    G1Affine29 r;
    r.x = p.x * izz;
    r.y = p.y * izzz;
    return r;
}

LWK_HD G1Affine29 affine_to_29(const G1Affine &p) {
    G1Affine29 r;
    r.x = f29_from_fp(p.x);
    r.y = f29_from_fp(p.y);
    return r;
}

// decompress_g1_point (/root/reference/src/compression.rs:62-103) WITHOUT the subgroup check, in the hot-loop
// representation: 0 = (x, y) is the point, 1 = point at infinity, 2 = invalid (not flagged compressed, or x^3 + 4 is
// not a square). want_greater = the ZCash sign bit (select_sqrt_value_from_third_bit).
// `root`: F29<2> -> its ((p + 1) / 4)-th power given the exponent's words (the default: f29_pow; k_decompress_points passes a chain
// whose window table lives in LDS, sha256.hip)
template <class Root>
LWK_HD int g1_decompress29_nocheck_t(const uint8_t *in48, F29<2> &x, F29<2> &y, bool &want_greater, Root root) {
    uint8_t b[48];
    for (int k = 0; k < 48; k++) b[k] = in48[k];
    const uint8_t prefix = b[0] >> 5;
    want_greater = (prefix & 1) != 0;
    if (!(prefix & 4)) return 2;  // not flagged compressed
    if (prefix & 2) return 1;     // infinity; remaining input bits are not inspected (compression.rs:73-75)
    b[0] &= 0x1f;
    uint32_t raw[12];
    raw_from_be<12>(raw, b);
    x = f29_from_raw32(raw);  // x >= p is reduced, as upstream from_bytes_be is believed to
    uint32_t four[12] = {4};
    F29<2> y2 = (sqr(x) * x + f29_from_raw32(four)) * F29<1>::one();  // the product by R mod p reduces < 4p back to < 2p
    const uint32_t e[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                            0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};
    F29<2> r = root(y2, e);  // y2^((p+1)/4)
    if (!(sqr(r) - y2).is_zero()) return 2;  // x^3 + 4 is not a square: not on the curve
    uint32_t ry[12], half[12], one[12] = {1};
    f29_to_raw32(ry, r);
    // r is the greater root  <=>  r > (p - 1) / 2
    raw_sub<12>(half, FpParams::MOD, one);
#pragma unroll
    for (int k = 0; k < 11; k++) half[k] = (half[k] >> 1) | (half[k + 1] << 31);
    half[11] >>= 1;
    const bool r_greater = !raw_geq<12>(half, ry);
    y = cneg(r, want_greater != r_greater) * F29<1>::one();  // back to < 2p
    return 0;
}
LWK_HD int g1_decompress29_nocheck(const uint8_t *in48, F29<2> &x, F29<2> &y, bool &want_greater) {
    return g1_decompress29_nocheck_t(in48, x, y, want_greater, [](const F29<2> &a, const uint32_t *e) { return f29_pow<12>(a, e); });
}

// [k]P, k = NK little-endian 32-bit limbs (plain integer, not reduced); left-to-right double-and-add
template <int NK>
LWK_HD G1Xyzz xyzz_mul_affine(const G1Affine &p, const uint32_t *k) {
    G1Xyzz acc = G1Xyzz::infinity();
    for (int i = NK * 32 - 1; i >= 0; i--) {
        acc = xyzz_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) acc = xyzz_madd(acc, p.x, p.y);
    }
    return acc;
}

// check_point_is_in_subgroup, /root/reference/src/compression.rs:22-27, as the reference computes it:
// [r]P == O by a 255-bit double-and-add. Kept as the definition (and used by tests); the production
// paths use the equivalent endomorphism test below, which is three times shorter.
LWK_HD bool g1_in_subgroup_by_order(const G1Affine &p) {
    uint32_t r[8];
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = FrParams::MOD[i];
    return xyzz_mul_affine<8>(p, r).is_inf();
}

// [|z|]T for the curve parameter |z| = 0xd201000000010000 (bits 63, 62, 60, 57, 48, 16), T in XYZZ
template <class X>
LWK_HD X xyzz_mul_by_z(const X &t) {
    X acc = t;  // bit 63
    for (int i = 62; i >= 0; i--) {
        acc = xyzz_dbl(acc);
        if (i == 62 || i == 60 || i == 57 || i == 48 || i == 16) acc = xyzz_add(acc, t);
    }
    return acc;
}

// Same accept/reject as [r]P == O for every point of E(Fp), by the endomorphism phi(x, y) = (beta x, y),
// which acts on G1 as multiplication by lambda = -z^2 (lambda^2 + lambda + 1 = r-multiple, since
// r = z^4 - z^2 + 1):  P in G1  <=>  [z^2]P + phi(P) == O   (M. Scott, "A note on group membership tests for
// G1, G2 and GT on BLS pairing-friendly curves", 2021; the test blst uses). 126 doublings + 10 additions
// instead of 255 + ~128. X is the XYZZ type to compute in; (px, py) affine, beta in the same field.
template <class X, class FX, class FY, class FB>
LWK_HD bool g1_in_subgroup_endo(const FX &px, const FY &py, const FB &beta) {
    X p = X::from_affine(px, py);
    X q = xyzz_mul_by_z(xyzz_mul_by_z(p));  // [z^2]P  (z < 0, squared)
    if (q.is_inf()) return false;           // P has small order dividing z^2: not in G1 (r is prime, r !| z^2)
    // q == -phi(P) = (beta x, -y)  <=>  X_q == beta x ZZ_q  and  Y_q == -y ZZZ_q
    auto dx = q.x - (beta * px) * q.zz;
    auto sy = q.y + py * q.zzz;
    return dx.is_zero() && sy.is_zero();
}

// beta = 0x5f19672fdf76ce51ba69c6076a0f77eaddb3a93be6f89688de17d813620a00022e01fffffffefffe (canonical limbs)
LWK_HD void g1_beta_raw(uint32_t raw[12]) {
    const uint32_t b[12] = {0xfffefffeu, 0x2e01ffffu, 0x620a0002u, 0xde17d813u, 0xe6f89688u, 0xddb3a93bu,
                            0x6a0f77eau, 0xba69c607u, 0xdf76ce51u, 0x5f19672fu, 0x00000000u, 0x00000000u};
#pragma unroll
    for (int i = 0; i < 12; i++) raw[i] = b[i];
}

LWK_HD bool g1_in_subgroup(const G1Affine &p) {
    uint32_t raw[12];
    g1_beta_raw(raw);
    return g1_in_subgroup_endo<G1Xyzz>(p.x, p.y, fe_from_raw<FpParams>(raw));
}

// compress_g1_point, /root/reference/src/compression.rs:33-60 (ZCash format):
// bit7 = compressed, bit6 = infinity (x = 0), bit5 = (p - y < y).
LWK_HD void g1_compress_affine(uint8_t out[48], const G1Affine &a) {
    uint32_t rx[12], ry[12], ryn[12];
    fe_to_raw<FpParams>(rx, a.x);
    fe_to_raw<FpParams>(ry, a.y);
    fe_to_raw<FpParams>(ryn, neg(a.y));
    raw_to_be<12>(out, rx);
    out[0] |= 0x80;
    // y_neg.representative() < y.representative()
    if (!raw_geq<12>(ryn, ry)) out[0] |= 0x20;
}

template <class X>
LWK_HD void g1_compress(uint8_t out[48], const X &p) {
    if (p.is_inf()) {
        for (int i = 0; i < 48; i++) out[i] = 0;
        out[0] = 0xc0;
        return;
    }
    g1_compress_affine(out, xyzz_to_affine(p));
}

// decompress_g1_point, /root/reference/src/compression.rs:62-103, without the subgroup check
// (callers decide where to run it). Returns 0 = ok affine, 1 = ok infinity, 2 = invalid.
LWK_HD int g1_decompress_nocheck(G1Affine &out, const uint8_t in[48]) {
    uint8_t prefix = in[0] >> 5;
    if (!(prefix & 4)) return 2;  // not flagged compressed
    if (prefix & 2) return 1;     // infinity (remaining bits are not inspected by the reference)
    uint8_t b[48];
    for (int i = 0; i < 48; i++) b[i] = in[i];
    b[0] &= 0x1f;
    uint32_t raw[12];
    raw_from_be<12>(raw, b);
    Fp x = fe_from_raw<FpParams>(raw);  // x >= p is reduced, as upstream from_bytes_be is believed to
    Fp y2 = sqr(x) * x + fp_from_u32(4);
    // p = 3 mod 4: sqrt = y2^((p+1)/4)
    const uint32_t e[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                            0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};
    Fp y = fe_pow<FpParams, 12>(y2, e);
    if (sqr(y) != y2) return 2;
    Fp yn = neg(y);
    uint32_t ry[12], ryn[12];
    fe_to_raw<FpParams>(ry, y);
    fe_to_raw<FpParams>(ryn, yn);
    bool y_greater = raw_geq<12>(ry, ryn);
    // select_sqrt_value_from_third_bit: the greater root iff bit5 is set
    bool want_greater = (prefix & 1) != 0;
    out.x = x;
    out.y = (want_greater == y_greater) ? y : yn;
    return 0;
}

}  // namespace lwk
