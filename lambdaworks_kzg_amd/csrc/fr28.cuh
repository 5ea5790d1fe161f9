// fr28.cuh -- the scalar field Fr in 10 limbs of 28 bits, Montgomery radix 2^280: the NTT's arithmetic.
//
// Same recipe as the hot loop's Fp (field29.cuh), for the same reasons: a column of the product-scanning multiplication
// (<= 20 partial products of 2^56, lazy limbs included) fits one 64-bit accumulator, so a product is 200 back-to-back
// v_mad_u64_u32 with no carry handling (the 8 x 32-bit CIOS form of field.cuh spends two thirds of its instructions on
// carries and their wait states), and sums / differences are limb-wise with no conditional subtraction: 25 bits of
// headroom over the modulus and 4 bits over the limb let the twelve butterfly stages of a 4096-point transform run
// with one carry ripple on half of the elements.
//
// Bounds a caller must keep (checked for the transform in fr_ops.hip: k_ntt4096):
//   product a*b:  (limb bound of a) x (limb bound of b) <= 24 (units of 2^28), (value bound of a) x (value bound of b) <= 2^25 (units of r);
//                 the result has normalised limbs and is < 2r
//   a + b:        bounds add;   a - b (b a product's result): a + 4r - b, value bound + 4, limb bound + 2;
//   a limb holds 15 units at most (2^32 / 2^28): fr28_norm brings the limbs back to one unit each
// r = 1 mod 2^32, so -r^-1 mod 2^28 is 2^28 - 1 and the reduction digit of a column is the negated low limb.
#pragma once
#include "field.cuh"

namespace lwk {

struct R28 {
    static constexpr int L = 10;
    static constexpr int W = 28;
    static constexpr uint32_t MASK = (1u << W) - 1;
#include "fr28_consts.inc"
};
static_assert(R28::INV == R28::MASK && R28::MOD[0] == 1, "the reduction digit is taken as -acc mod 2^28");

struct Fr28 {
    uint32_t l[10];
};

// Montgomery product, product scanning (as mont_mul29): result limbs normalised, value < 2r. `b(i)` = limb i of the
// second factor (a value, or one of the constants below by index: constexpr tables are read by value in device code).
template <class B>
LWK_HD Fr28 fr28_mul_t(const Fr28 &a, B b) {
    u64 acc = 0;
    uint32_t m[10];
    Fr28 r;
#pragma unroll
    for (int k = 0; k < 10; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (u64)a.l[i] * b(k - i);
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * R28::MOD[k - i];
        m[k] = (0u - (uint32_t)acc) & R28::MASK;  // * INV = * (2^28 - 1)
        acc += m[k];                              // * MOD[0] = * 1
        acc >>= R28::W;
    }
#pragma unroll
    for (int k = 10; k < 19; k++) {
#pragma unroll
        for (int i = k - 9; i < 10; i++) acc += (u64)a.l[i] * b(k - i);
#pragma unroll
        for (int i = k - 9; i < 10; i++) acc += (u64)m[i] * R28::MOD[k - i];
        r.l[k - 10] = (uint32_t)acc & R28::MASK;
        acc >>= R28::W;
    }
    r.l[9] = (uint32_t)acc;
    return r;
}
LWK_HD Fr28 fr28_mul(const Fr28 &a, const Fr28 &b) {
    return fr28_mul_t(a, [&b](int i) { return b.l[i]; });
}
#define LWK_FR28_MUL_CONST(a, NAME) fr28_mul_t(a, [](int i) { return R28::NAME[i]; })

LWK_HD Fr28 fr28_add(const Fr28 &a, const Fr28 &b) {
    Fr28 r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// a - b for a b with normalised limbs and value < 4r (a product's result): no limb goes negative
LWK_HD Fr28 fr28_sub(const Fr28 &a, const Fr28 &b) {
    Fr28 r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = a.l[i] + R28::OFF4[i] - b.l[i];
    return r;
}

// one carry ripple: limbs 0..8 back under 2^28, the value's top in limb 9
LWK_HD Fr28 fr28_norm(const Fr28 &a) {
    Fr28 r = a;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.l[i + 1] += r.l[i] >> R28::W;
        r.l[i] &= R28::MASK;
    }
    return r;
}

// 8 x 32-bit little-endian words (a value < 2^256) <-> 10 x 28-bit limbs
LWK_HD Fr28 fr28_pack(const uint32_t *w) {
    Fr28 r;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const int bit = 28 * i, k = bit >> 5, sh = bit & 31;
        uint32_t v = k < 8 ? w[k] >> sh : 0u;
        if (sh > 4 && k + 1 < 8) v |= w[k + 1] << (32 - sh);
        r.l[i] = v & R28::MASK;
    }
    return r;
}
// limbs normalised, value < 2^256
LWK_HD void fr28_unpack(uint32_t *w, const Fr28 &a) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        // word k holds bits 32k .. 32k+31: limb i0 = floor(32k / 28) from bit offset (32k mod 28), then the next one or two limbs
        const int bit = 32 * k, i0 = bit / 28, off = bit % 28;
        uint32_t v = a.l[i0] >> off;
        if (i0 + 1 < 10) v |= a.l[i0 + 1] << (28 - off);
        if (off > 24 && i0 + 2 < 10) v |= a.l[i0 + 2] << (56 - off);
        w[k] = v;
    }
}

// value < 2r with normalised limbs -> canonical (< r)
LWK_HD Fr28 fr28_canonical(const Fr28 &a) {
    Fr28 d;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        uint32_t t = a.l[i] - R28::MOD[i] - borrow;
        borrow = t >> 31;  // limbs are < 2^28 (the top one < 2^5): a wrap shows in bit 31
        d.l[i] = t & R28::MASK;
    }
    Fr28 r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
    return r;
}

// Fe<FrParams> in its Montgomery form (x 2^256, canonical) -> x 2^280 in 28-bit limbs (< 2r, normalised)
LWK_HD Fr28 fr28_from_mont256(const Fr &x) { return LWK_FR28_MUL_CONST(fr28_pack(x.l), TO28); }
// (y 2^280), lazy -> y 2^256 as an Fe<FrParams>
LWK_HD Fr fr28_to_mont256(const Fr28 &y) {
    Fr r;
    fr28_unpack(r.l, fr28_canonical(LWK_FR28_MUL_CONST(y, TO256)));
    return r;
}
// (y 2^280), lazy -> y / 4096 as canonical little-endian words (what the MSM's digit extraction reads)
LWK_HD void fr28_to_raw_scaled(uint32_t *w, const Fr28 &y) { fr28_unpack(w, fr28_canonical(LWK_FR28_MUL_CONST(y, NINV_RAW))); }

}  // namespace lwk
