// field29.cuh -- carry-free Fp for the MSM hot loop on gfx950: 14 limbs of W = 28 (default) or 29 bits, lazy ranges.
// (The file and the type keep the name of the first version, which had 29-bit limbs; LWK_LIMB_BITS selects W.)
//
// Why a second representation (field.cuh's 12x32-bit one stays for the host, setup and Fr):
//   * gfx950 multiplies with v_mad_u64_u32 (32x32+64->64) at ~4.7 cycles per wave-instruction, the same
//     as any 3-operand VALU op, but a carry costs TWO wait states between the instruction that
//     writes VCC and the one that reads it (the compiler emits `s_nop 1` after every v_add_co), and
//     saturated 32-bit limbs need a 64-bit carry add plus zero-extension moves per product:
//     profiles/r01_ubench_instruction_rates.jsonl -- the 288 multiply-adds of a 12-limb CIOS were
//     only ~35 % of its 3660 cycles.
//   * With limbs of W <= 29 bits a whole column of a 14x14 product (28 partial products < 2^(2W)) fits a 64-bit
//     accumulator: the product is 392 back-to-back v_mad_u64_u32 with no carry handling at all, and
//     additions/subtractions are limb-wise, with no conditional subtraction.
//   * 14 W bits leave 11 (W = 28) or 25 (W = 29) bits of headroom over the 381-bit modulus, so values may grow to
//     many multiples of p between multiplications. The multiple of p a value is bounded by is tracked in
//     the TYPE (F29<B>: value < B*p), so every bound is checked at compile time and no reduction is
//     ever executed on the hot path.
//   * W = 28 leaves three spare bits in the 64-bit column as well, so the LIMBS may be lazy too: sums and differences
//     keep their carries (limbs < LB * 2^28, LB tracked in the type next to B) and a product accepts factors with
//     LA * LB <= 17. A mixed addition then needs ONE carry ripple (where X3's three-term difference is stored)
//     instead of seven: -3.4 % on the accumulate kernel, same-box A/B against W = 29.
//
// Invariants of F29<B, INL, LB>: integer value < B*p; limbs 0..12 < LB * 2^W; Montgomery radix R = 2^(14 W).
#pragma once
#include "field.cuh"

namespace lwk {

#ifndef LWK_LIMB_BITS
#define LWK_LIMB_BITS 28
#endif

struct P29 {
    static constexpr int L = 14;
    static constexpr int W = LWK_LIMB_BITS;  // limb width: 29 (every sum is renormalised) or 28 (lazy limbs, see F29)
    static constexpr uint32_t MASK = (1u << W) - 1;
    // a 64-bit column holds 14 (La Lb + 1) products of 2^(2W): the largest admissible limb-bound product of two factors
    static constexpr int MAXLL = (1 << (64 - 2 * W)) / 14 - 1;  // W = 29: 3, W = 28: 17
    static constexpr bool LAZY = W == 28;                       // sums keep their carries until a product needs them gone
#include "field29_consts.inc"
};

// offset multiple used by a subtraction whose subtrahend is < B*p: the next power of two >= B + 1
constexpr int sub_offset(int B) { return B < 2 ? 2 : B < 4 ? 4 : B < 8 ? 8 : B < 16 ? 16 : 32; }
// borrow (in units of 2^W per limb) of that multiple for a subtrahend whose limbs are < LB * 2^W
constexpr int sub_borrow(int LB) { return LB <= 1 ? 1 : LB + 1 <= 4 ? 4 : 8; }

template <int K, int BR>
LWK_HD uint32_t kp_limb(int i) {
    static_assert(K == 2 || K == 4 || K == 8 || K == 16 || K == 32, "offset table");
    static_assert(BR == 1 || BR == 2 || BR == 4 || (BR == 8 && P29::W == 28), "borrow table");
#define LWK_KP(k, b) if constexpr (K == k && BR == b) return P29::KP##k##_##b[i];
    LWK_KP(2, 1) LWK_KP(4, 1) LWK_KP(8, 1) LWK_KP(16, 1) LWK_KP(32, 1)
    LWK_KP(2, 2) LWK_KP(4, 2) LWK_KP(8, 2) LWK_KP(16, 2) LWK_KP(32, 2)
    LWK_KP(2, 4) LWK_KP(4, 4) LWK_KP(8, 4) LWK_KP(16, 4) LWK_KP(32, 4)
#if LWK_LIMB_BITS == 28
    LWK_KP(2, 8) LWK_KP(4, 8) LWK_KP(8, 8) LWK_KP(16, 8) LWK_KP(32, 8)
#endif
#undef LWK_KP
    return 0;
}

// one carry ripple: limbs 0..12 back under 2^W (inputs < 2^32 per limb)
LWK_HD void norm29(uint32_t *l) {
#pragma unroll
    for (int i = 0; i < 13; i++) {
        l[i + 1] += l[i] >> P29::W;
        l[i] &= P29::MASK;
    }
}

// F29<B, INL, LB>: integer value < B*p, limbs 0..12 < LB * 2^W (LB = 1: normalised).
// INL selects how operator* is emitted on the device: false = call of the shared mont_mul29_call
// function (small code, used where many group operations are instantiated), true = inlined into the
// caller (no call overhead or forced s_waitcnt at call boundaries; used by the accumulate loop).
// With 29-bit limbs LB is always 1 (every sum is renormalised at once); with 28-bit limbs sums and differences keep
// their carries (LB grows) and are renormalised only where a product's 64-bit column would overflow or where they
// are stored into a normalised slot -- both checked at compile time.
template <int B, bool INL = false, int LB = 1>
struct alignas(8) F29 {
    static constexpr int BOUND = B;
    static constexpr int LIMB = LB;
    uint32_t l[14];

    F29() = default;
    // widening the bounds is free; storing a lazier value into a tighter slot renormalises it
    template <int A, int LA>
    LWK_HD F29(const F29<A, INL, LA> &o) {
        static_assert(A <= B, "bound would shrink: a value < A*p is not known to be < B*p");
#pragma unroll
        for (int i = 0; i < 14; i++) l[i] = o.l[i];
        if constexpr (LA > LB) norm29(l);
    }
    LWK_HD static F29 zero() {
        F29 r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.l[i] = 0;
        return r;
    }
    LWK_HD static F29 one() {
        F29 r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.l[i] = P29::R1[i];
        return r;
    }
    // the integer is exactly 0 (used for the infinity marker zz == 0, which is stored as literal zeros)
    LWK_HD bool is_literal_zero() const {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) x |= l[i];
        return x == 0;
    }
    // value == 0 (mod p): the integer is one of 0, p, 2p, ..., (B-1)p. The low limb filters out all
    // but a 2^-W-ish fraction of non-zero values; candidates get the exact comparison (on normalised limbs).
    LWK_HD bool is_zero() const {
        static_assert(B <= 64, "is_zero scans B multiples");
        const uint32_t low = l[0] & P29::MASK;  // limb 0 has no incoming carry: its low W bits are final
        bool cand = false;
        uint32_t kk = 0;
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)B; k++) {
            if (low == ((k * P29::MOD[0]) & P29::MASK)) {
                cand = true;
                kk = k;
            }
        }
        if (!cand) return false;
        uint32_t n[14];
        for (int i = 0; i < 14; i++) n[i] = l[i];
        if (LB > 1) norm29(n);
        u64 c = 0;
        uint32_t diff = 0;
        for (int i = 0; i < 14; i++) {
            c += (u64)kk * P29::MOD[i];
            uint32_t want = (i < 13) ? ((uint32_t)c & P29::MASK) : (uint32_t)c;
            c >>= P29::W;
            diff |= want ^ n[i];
        }
        return diff == 0;
    }
};

// explicit renormalisation (a no-op type change for values that already are)
template <int A, bool I, int LA>
LWK_HD F29<A, I, 1> normed(const F29<A, I, LA> &a) {
    return F29<A, I, 1>(a);
}

template <int A, int B, bool I, int LA, int LB>
LWK_HD auto operator+(const F29<A, I, LA> &a, const F29<B, I, LB> &b) {
    static_assert(A + B <= 4096, "value bound");
    constexpr int LR = P29::LAZY ? LA + LB : 1;
    static_assert(LR <= 14, "limb bound");
    F29<A + B, I, LR> r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + b.l[i];
    if constexpr (!P29::LAZY) norm29(r.l);
    return r;
}

template <int A, int B, bool I, int LA, int LB>
LWK_HD auto operator-(const F29<A, I, LA> &a, const F29<B, I, LB> &b) {
    constexpr int K = sub_offset(B);
    constexpr int BR = sub_borrow(LB);
    static_assert(LB == 1 || LB + 1 <= BR, "subtrahend too lazy for the borrow tables: renormalise it (normed())");
    static_assert(A + K <= 4096, "value bound");
    constexpr int LR = P29::LAZY ? LA + BR + 1 : 1;
    static_assert(LR <= 14, "limb bound");
    F29<A + K, I, LR> r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + kp_limb<K, BR>(i) - b.l[i];
    if constexpr (!P29::LAZY) norm29(r.l);
    return r;
}

template <int A, bool I, int LA>
LWK_HD auto dbl(const F29<A, I, LA> &a) {
    return a + a;
}

template <int A, bool I, int LA>
LWK_HD auto neg(const F29<A, I, LA> &a) {
    constexpr int K = sub_offset(A);
    constexpr int BR = sub_borrow(LA);
    constexpr int LR = P29::LAZY ? BR + 1 : 1;
    F29<K, I, LR> r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = kp_limb<K, BR>(i) - a.l[i];
    if constexpr (!P29::LAZY) norm29(r.l);
    return r;
}

// flag ? -a : a, one type for both outcomes
template <int A, bool I, int LA>
LWK_HD auto cneg(const F29<A, I, LA> &a, bool flag) {
    constexpr int K = sub_offset(A);
    constexpr int BR = sub_borrow(LA);
    constexpr int LR = P29::LAZY ? (BR + 1 > LA ? BR + 1 : LA) : 1;
    F29<K, I, LR> r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = flag ? kp_limb<K, BR>(i) - a.l[i] : a.l[i];
    if constexpr (!P29::LAZY) norm29(r.l);
    return r;
}

// Montgomery product, product scanning: column k of a*b + m*p accumulates in ONE 64-bit register
// (at most 28 products plus the running carry: 14 (La Lb + 1) 2^(2W) < 2^64, checked by the callers' static_asserts),
// is cleared in its low W bits by the choice of m[k], and shifts down. Result < p + a*b/2^(14W) < 2p.
LWK_HD void mont_mul29(uint32_t *r, const uint32_t *a, const uint32_t *b) {
    u64 acc = 0;
    uint32_t m[14];
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (u64)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * P29::MOD[k - i];
        m[k] = ((uint32_t)acc * P29::INV) & P29::MASK;
        acc += (u64)m[k] * P29::MOD[0];
        acc >>= P29::W;
    }
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (u64)a[i] * b[k - i];
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (u64)m[i] * P29::MOD[k - i];
        r[k - 14] = (uint32_t)acc & P29::MASK;
        acc >>= P29::W;
    }
    r[13] = (uint32_t)acc;
}

// by-value operands of the device function below: a 14-lane vector travels in 14 VGPRs (a struct of
// the same size is passed through scratch memory by the AMDGPU calling convention)
typedef uint32_t Raw29 __attribute__((ext_vector_type(14)));

#if defined(__HIP_DEVICE_COMPILE__)
// a real function on the device, as fe_mul_call (field.cuh): operands by value in VGPRs
__device__ __noinline__ Raw29 mont_mul29_call(Raw29 a, Raw29 b) {
    uint32_t x[14], y[14], z[14];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        x[i] = a[i];
        y[i] = b[i];
    }
    mont_mul29(z, x, y);
    Raw29 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r[i] = z[i];
    return r;
}
#endif

// value-bound budget of a product: a*b / R + p < 2p  <=>  A*B*p < R = 2^(14 W)
constexpr long long kProdBound = P29::W == 29 ? (1ll << 22) : 2048;  // R / p = 2^26 / 1.625 (W = 29), 2^12 / 1.625 = 2520 (W = 28)

template <int A, int B, bool I, int LA, int LB>
LWK_HD F29<2, I> operator*(const F29<A, I, LA> &a, const F29<B, I, LB> &b) {
    static_assert((long long)A * B <= kProdBound, "product of bounds too large for the Montgomery radix");
    static_assert(LA * LB <= P29::MAXLL, "limbs too lazy for a 64-bit column: renormalise one factor (normed())");
    F29<2, I> r;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (I) {
        mont_mul29(r.l, a.l, b.l);
        return r;
    }
    Raw29 x, y;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        x[i] = a.l[i];
        y[i] = b.l[i];
    }
    Raw29 z = mont_mul29_call(x, y);
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = z[i];
#else
    mont_mul29(r.l, a.l, b.l);
#endif
    return r;
}

// Montgomery square: the 91 cross products are taken once against a doubled operand, so a column holds the same
// total as the full product's. 105 + 196 multiply-adds instead of 392.
LWK_HD void mont_sqr29(uint32_t *r, const uint32_t *a) {
    u64 acc = 0;
    uint32_t m[14], a2[14];
#pragma unroll
    for (int i = 0; i < 14; i++) a2[i] = a[i] << 1;
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (u64)a2[i] * a[k - i];
        if ((k & 1) == 0) acc += (u64)a[k >> 1] * a[k >> 1];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * P29::MOD[k - i];
        m[k] = ((uint32_t)acc * P29::INV) & P29::MASK;
        acc += (u64)m[k] * P29::MOD[0];
        acc >>= P29::W;
    }
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; 2 * i < k; i++) acc += (u64)a2[i] * a[k - i];
        if ((k & 1) == 0) acc += (u64)a[k >> 1] * a[k >> 1];
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (u64)m[i] * P29::MOD[k - i];
        r[k - 14] = (uint32_t)acc & P29::MASK;
        acc >>= P29::W;
    }
    r[13] = (uint32_t)acc;
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __noinline__ Raw29 mont_sqr29_call(Raw29 a) {
    uint32_t x[14], z[14];
#pragma unroll
    for (int i = 0; i < 14; i++) x[i] = a[i];
    mont_sqr29(z, x);
    Raw29 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r[i] = z[i];
    return r;
}
#endif

template <int A, bool I, int LA>
LWK_HD F29<2, I> sqr(const F29<A, I, LA> &a) {
    static_assert((long long)A * A <= kProdBound, "square of the bound too large for the Montgomery radix");
    static_assert(LA * LA <= P29::MAXLL && LA <= 7, "limbs too lazy for a 64-bit column (or for the doubled operand)");
    F29<2, I> r;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (I) {
        mont_sqr29(r.l, a.l);
        return r;
    }
    Raw29 x;
#pragma unroll
    for (int i = 0; i < 14; i++) x[i] = a.l[i];
    Raw29 z = mont_sqr29_call(x);
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = z[i];
#else
    mont_sqr29(r.l, a.l);
#endif
    return r;
}

// a*b + c*d with ONE Montgomery reduction (588 multiply-adds instead of 784). A column holds at most
// 14 (La Lb + Lc Ld + 1) products of 2^(2W).
LWK_HD void mont_mul_add29(uint32_t *r, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d) {
    u64 acc = 0;
    uint32_t m[14];
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (u64)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (u64)c[i] * d[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * P29::MOD[k - i];
        m[k] = ((uint32_t)acc * P29::INV) & P29::MASK;
        acc += (u64)m[k] * P29::MOD[0];
        acc >>= P29::W;
    }
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (u64)a[i] * b[k - i];
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (u64)c[i] * d[k - i];
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (u64)m[i] * P29::MOD[k - i];
        r[k - 14] = (uint32_t)acc & P29::MASK;
        acc >>= P29::W;
    }
    r[13] = (uint32_t)acc;
}

// a*b - c*d. Inlined flavour: one fused product pair over a negated c; call flavour: two products and a subtraction.
template <int A, int B, int C, int D, bool I, int LA, int LB, int LC, int LD>
LWK_HD auto mul_sub(const F29<A, I, LA> &a, const F29<B, I, LB> &b, const F29<C, I, LC> &c, const F29<D, I, LD> &d) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (I) {
        static_assert((long long)A * B + (long long)sub_offset(C) * D <= kProdBound, "bounds too large for the Montgomery radix");
        auto nc = neg(c);
        static_assert(LA * LB + decltype(nc)::LIMB * LD <= P29::MAXLL, "limbs too lazy for a 64-bit column");
        F29<2, I> r;
        mont_mul_add29(r.l, a.l, b.l, nc.l, d.l);
        return r;
    } else {
        return F29<6, I>(a * b - c * d);
    }
#else
    return F29<6, I>(a * b - c * d);
#endif
}

// ---- conversions -----------------------------------------------------------------------------------

// canonical integer (12 x u32, < 2^384) -> F29<2> in Montgomery form (radix 2^(14 W))
LWK_HD F29<2> f29_from_raw32(const uint32_t raw[12]) {
    F29<32> t;  // any integer < 2^384 < 8.3 p... declare generously: < 32p
#pragma unroll
    for (int i = 0; i < 14; i++) {
        int bit = P29::W * i;
        int w = bit >> 5, sh = bit & 31;
        uint32_t v = 0;
        if (w < 12) {
            v = raw[w] >> sh;
            if (sh + P29::W > 32 && w + 1 < 12) v |= raw[w + 1] << (32 - sh);
        }
        t.l[i] = v & P29::MASK;
    }
    F29<1> r2;
#pragma unroll
    for (int i = 0; i < 14; i++) r2.l[i] = P29::R2[i];
    return t * r2;
}

// F29<B> -> canonical integer in [0, p) as 12 x u32
template <int B, int LB>
LWK_HD void f29_to_raw32(uint32_t raw[12], const F29<B, false, LB> &a) {
    F29<1> one;
#pragma unroll
    for (int i = 0; i < 14; i++) one.l[i] = (i == 0) ? 1u : 0u;
    F29<2> v = a * one;  // leaves Montgomery form; v < 2p
    // v >= p ? v - p : v   (signed limb-wise difference with arithmetic carries)
    uint32_t d[14];
    long long c = 0;
    for (int i = 0; i < 14; i++) {
        c += (long long)v.l[i] - (long long)P29::MOD[i];
        d[i] = (i < 13) ? ((uint32_t)c & P29::MASK) : (uint32_t)c;
        c >>= P29::W;
    }
    bool ge = c >= 0 && (int32_t)d[13] >= 0;
    uint32_t w[14];
    for (int i = 0; i < 14; i++) w[i] = ge ? d[i] : v.l[i];
    for (int i = 0; i < 12; i++) raw[i] = 0;
    for (int i = 0; i < 14; i++) {
        int bit = P29::W * i;
        int k = bit >> 5, sh = bit & 31;
        if (k < 12) raw[k] |= w[i] << sh;
        if (sh + P29::W > 32 && k + 1 < 12) raw[k + 1] |= w[i] >> (32 - sh);
    }
}

LWK_HD F29<2> f29_from_fp(const Fp &a) {
    uint32_t raw[12];
    fe_to_raw<FpParams>(raw, a);
    return f29_from_raw32(raw);
}

template <int B, int LB>
LWK_HD Fp f29_to_fp(const F29<B, false, LB> &a) {
    uint32_t raw[12];
    f29_to_raw32(raw, a);
    return fe_from_raw<FpParams>(raw);
}

// a^e, public exponent, NE little-endian 32-bit limbs. Fixed 4-bit windows (r05): 14 products of table, then four squares and at most
// one product per window -- 380 + 95 + 14 products for the 380-bit exponent of the square root instead of the 380 + ~190 of
// square-and-multiply (the chain is what a commitment's validation waits for: k_decompress_points, k_validate_commitments)
template <int NE, int B, int LB>
LWK_HD F29<2> f29_pow(const F29<B, false, LB> &a, const uint32_t *e) {
    F29<2> tab[16];
    tab[0] = F29<2>::one();
    tab[1] = a * F29<1>::one();  // a * R / R = a, but weakly reduced to < 2p
#pragma unroll 1
    for (int k = 2; k < 16; k++) tab[k] = tab[k - 1] * tab[1];
    F29<2> acc = F29<2>::one();
    bool started = false;
#pragma unroll 1
    for (int w = NE * 8 - 1; w >= 0; w--) {
        const uint32_t d = (e[w >> 3] >> (4 * (w & 7))) & 15u;
        if (started) {
            acc = sqr(acc);
            acc = sqr(acc);
            acc = sqr(acc);
            acc = sqr(acc);
        }
        if (d) {
            acc = started ? acc * tab[d] : tab[d];
            started = true;
        }
    }
    return acc;
}

// Fermat inversion a^(p-2): ~480 dependent products. Kept as the cross-check of f29_inv (tools/host_check.hip).
template <int B, int LB>
LWK_HD F29<2> f29_inv_fermat(const F29<B, false, LB> &a) {
    uint32_t e[12], two[12];
#pragma unroll
    for (int i = 0; i < 12; i++) two[i] = (i == 0) ? 2u : 0u;
    raw_sub<12>(e, FpParams::MOD, two);
    return f29_pow<12>(a, e);
}

template <int B, int LB>
LWK_HD F29<2> f29_inv(const F29<B, false, LB> &a) {
    uint32_t x[12], y[12];
    f29_to_raw32(x, a);       // out of Montgomery form, canonical
    fp_inv_raw32(y, x);
    return f29_from_raw32(y);  // back into Montgomery form
}

}  // namespace lwk
