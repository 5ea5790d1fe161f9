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

namespace lwk {

struct G1Affine {
    Fp x, y;  // Montgomery form; the point at infinity is not representable (never needed in tables)
};

struct G1Xyzz {
    Fp x, y, zz, zzz;  // infinity <=> zz == 0

    LWK_HD static G1Xyzz infinity() {
        G1Xyzz r;
        r.x = Fp::zero();
        r.y = Fp::zero();
        r.zz = Fp::zero();
        r.zzz = Fp::zero();
        return r;
    }
    LWK_HD bool is_inf() const { return zz.is_zero(); }
    LWK_HD static G1Xyzz from_affine(const G1Affine &p) {
        G1Xyzz r;
        r.x = p.x;
        r.y = p.y;
        r.zz = Fp::one();
        r.zzz = Fp::one();
        return r;
    }
};

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

// 2P for affine P (mdbl-2008-s-1, a = 0)
LWK_HD G1Xyzz xyzz_dbl_affine(const G1Affine &p) {
    if (p.y.is_zero()) return G1Xyzz::infinity();  // order-2 point: cannot occur in G1, kept for completeness
    Fp u = dbl(p.y);
    Fp v = sqr(u);
    Fp w = u * v;
    Fp s = p.x * v;
    Fp xx = sqr(p.x);
    Fp m = dbl(xx) + xx;
    G1Xyzz r;
    r.x = sqr(m) - dbl(s);
    r.y = m * (s - r.x) - w * p.y;
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2P (dbl-2008-s-1, a = 0)
LWK_HD G1Xyzz xyzz_dbl(const G1Xyzz &p) {
    if (p.is_inf() || p.y.is_zero()) return G1Xyzz::infinity();
    Fp u = dbl(p.y);
    Fp v = sqr(u);
    Fp w = u * v;
    Fp s = p.x * v;
    Fp xx = sqr(p.x);
    Fp m = dbl(xx) + xx;
    G1Xyzz r;
    r.x = sqr(m) - dbl(s);
    r.y = m * (s - r.x) - w * p.y;
    r.zz = v * p.zz;
    r.zzz = w * p.zzz;
    return r;
}

// acc + q for affine q (madd-2008-s), complete
LWK_HD G1Xyzz xyzz_madd(const G1Xyzz &acc, const G1Affine &q) {
    if (acc.is_inf()) return G1Xyzz::from_affine(q);
    Fp u2 = q.x * acc.zz;
    Fp s2 = q.y * acc.zzz;
    Fp pp_ = u2 - acc.x;
    Fp rr = s2 - acc.y;
    if (pp_.is_zero()) {
        if (rr.is_zero()) return xyzz_dbl_affine(q);
        return G1Xyzz::infinity();
    }
    Fp pp = sqr(pp_);
    Fp ppp = pp_ * pp;
    Fp qq = acc.x * pp;
    G1Xyzz r;
    r.x = sqr(rr) - ppp - dbl(qq);
    r.y = rr * (qq - r.x) - acc.y * ppp;
    r.zz = acc.zz * pp;
    r.zzz = acc.zzz * ppp;
    return r;
}

// a + b (add-2008-s), complete
LWK_HD G1Xyzz xyzz_add(const G1Xyzz &a, const G1Xyzz &b) {
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    Fp u1 = a.x * b.zz;
    Fp u2 = b.x * a.zz;
    Fp s1 = a.y * b.zzz;
    Fp s2 = b.y * a.zzz;
    Fp pp_ = u2 - u1;
    Fp rr = s2 - s1;
    if (pp_.is_zero()) {
        if (rr.is_zero()) return xyzz_dbl(a);
        return G1Xyzz::infinity();
    }
    Fp pp = sqr(pp_);
    Fp ppp = pp_ * pp;
    Fp qq = u1 * pp;
    G1Xyzz r;
    r.x = sqr(rr) - ppp - dbl(qq);
    r.y = rr * (qq - r.x) - s1 * ppp;
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

// [k]P, k = NK little-endian 32-bit limbs (plain integer, not reduced); left-to-right double-and-add
template <int NK>
LWK_HD G1Xyzz xyzz_mul_affine(const G1Affine &p, const uint32_t *k) {
    G1Xyzz acc = G1Xyzz::infinity();
    for (int i = NK * 32 - 1; i >= 0; i--) {
        acc = xyzz_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) acc = xyzz_madd(acc, p);
    }
    return acc;
}

// check_point_is_in_subgroup, /root/reference/src/compression.rs:22-27: [r]P == O
LWK_HD bool g1_in_subgroup(const G1Affine &p) {
    uint32_t r[8];
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = FrParams::MOD[i];
    return xyzz_mul_affine<8>(p, r).is_inf();
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

LWK_HD void g1_compress(uint8_t out[48], const G1Xyzz &p) {
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
