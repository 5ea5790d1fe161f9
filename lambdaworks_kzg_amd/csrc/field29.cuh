// field29.cuh -- carry-free Fp for the MSM hot loop on gfx950: 14 limbs of 29 bits, lazy range.
//
// Why a second representation (field.cuh's 12x32-bit one stays for the host, setup and Fr):
//   * gfx950 multiplies with v_mad_u64_u32 (32x32+64->64) at ~4.7 cycles per wave-instruction, the same
//     as any 3-operand VALU op, but a carry costs TWO wait states between the instruction that
//     writes VCC and the one that reads it (the compiler emits `s_nop 1` after every v_add_co), and
//     saturated 32-bit limbs need a 64-bit carry add plus zero-extension moves per product:
//     profiles/r01_ubench_instruction_rates.jsonl -- the 288 multiply-adds of a 12-limb CIOS were
//     only ~35 % of its 3660 cycles.
//   * With 29-bit limbs a whole column of a 14x14 product (28 partial products < 2^58) fits a 64-bit
//     accumulator: the product is 392 back-to-back v_mad_u64_u32 with no carry handling at all, and
//     additions/subtractions are limb-wise with one cheap carry ripple, no conditional subtraction.
//   * 14 x 29 = 406 bits leaves 25 bits of headroom over the 381-bit modulus, so values may grow to
//     thousands of p between multiplications. The multiple of p a value is bounded by is tracked in
//     the TYPE (F29<B>: value < B*p), so every bound is checked at compile time and no reduction is
//     ever executed on the hot path.
//
// Invariants of F29<B>: integer value < B*p; limbs 0..12 < 2^29; Montgomery radix R = 2^406.
#pragma once
#include "field.cuh"

namespace lwk {

struct P29 {
    static constexpr int L = 14;
    static constexpr int W = 29;
    static constexpr uint32_t MASK = (1u << 29) - 1;
    static constexpr uint32_t INV = 0x1ffcfffdu;  // -p^-1 mod 2^29
    static constexpr uint32_t MOD[14] = {0x1fffaaabu, 0x0ff7ffffu, 0x14ffffeeu, 0x17fffd62u, 0x0f6241eau,
                                         0x09507b58u, 0x0afd9cc3u, 0x109e70a2u, 0x1764774bu, 0x121a5d66u,
                                         0x12c6e9edu, 0x12ffcd34u, 0x00111ea3u, 0x0000000du};
    static constexpr uint32_t R1[14] = {0x03a9fb84u, 0x0ba00690u, 0x071288f1u, 0x0f59bcc5u, 0x126cb614u,
                                        0x0585bf36u, 0x1b85ac3du, 0x1cf856fau, 0x1891ecbdu, 0x1a7eec05u,
                                        0x155a88f0u, 0x0741ac6du, 0x1317c30fu, 0x00000009u};
    static constexpr uint32_t R2[14] = {0x15bef7aeu, 0x1031cd0eu, 0x02dd93e8u, 0x09226323u, 0x0e6e2cd2u,
                                        0x11684daau, 0x1170e5dbu, 0x088e25b1u, 0x1b366399u, 0x1c536f47u,
                                        0x0d1f9cbcu, 0x0278b67fu, 0x1ea66a2bu, 0x0000000cu};
    // K*p with every limb but the top one "borrowed" up by 2^29 (and the next one down by 1), so that
    // a + K*p - b never goes negative in any limb for normalised b < (K-1)*p.
    static constexpr uint32_t KP2[14] = {0x3fff5556u, 0x3feffffeu, 0x29ffffdbu, 0x2ffffac4u, 0x3ec483d4u,
                                         0x32a0f6afu, 0x35fb3985u, 0x213ce143u, 0x2ec8ee96u, 0x2434baccu,
                                         0x258dd3dau, 0x25ff9a68u, 0x20223d46u, 0x00000019u};
    static constexpr uint32_t KP4[14] = {0x3ffeaaacu, 0x3fdffffeu, 0x33ffffb8u, 0x3ffff589u, 0x3d8907a9u,
                                         0x2541ed60u, 0x2bf6730cu, 0x2279c288u, 0x3d91dd2du, 0x28697599u,
                                         0x2b1ba7b5u, 0x2bff34d1u, 0x20447a8du, 0x00000033u};
    static constexpr uint32_t KP8[14] = {0x3ffd5558u, 0x3fbffffeu, 0x27ffff72u, 0x3fffeb14u, 0x3b120f54u,
                                         0x2a83dac2u, 0x37ece619u, 0x24f38511u, 0x3b23ba5bu, 0x30d2eb34u,
                                         0x36374f6bu, 0x37fe69a3u, 0x2088f51bu, 0x00000067u};
    static constexpr uint32_t KP16[14] = {0x3ffaaab0u, 0x3f7ffffeu, 0x2ffffee6u, 0x3fffd629u, 0x36241eaau,
                                          0x3507b586u, 0x2fd9cc33u, 0x29e70a24u, 0x364774b7u, 0x21a5d66au,
                                          0x2c6e9ed8u, 0x2ffcd348u, 0x2111ea38u, 0x000000cfu};
    static constexpr uint32_t KP32[14] = {0x3ff55560u, 0x3efffffeu, 0x3ffffdceu, 0x3fffac53u, 0x2c483d56u,
                                          0x2a0f6b0eu, 0x3fb39868u, 0x33ce1449u, 0x2c8ee96fu, 0x234bacd6u,
                                          0x38dd3db1u, 0x3ff9a691u, 0x2223d471u, 0x0000019fu};
};

// offset multiple used by a subtraction whose subtrahend is < B*p: the next power of two >= B + 1
constexpr int sub_offset(int B) { return B < 2 ? 2 : B < 4 ? 4 : B < 8 ? 8 : B < 16 ? 16 : 32; }

template <int K>
LWK_HD uint32_t kp_limb(int i) {
    static_assert(K == 2 || K == 4 || K == 8 || K == 16 || K == 32, "offset table");
    if constexpr (K == 2) return P29::KP2[i];
    else if constexpr (K == 4) return P29::KP4[i];
    else if constexpr (K == 8) return P29::KP8[i];
    else if constexpr (K == 16) return P29::KP16[i];
    else return P29::KP32[i];
}

// INL selects how operator* is emitted on the device: false = call of the shared mont_mul29_call
// function (small code, used where many group operations are instantiated), true = inlined into the
// caller (no call overhead or forced s_waitcnt at call boundaries; used by the accumulate loop).
template <int B, bool INL = false>
struct alignas(8) F29 {
    static constexpr int BOUND = B;
    uint32_t l[14];

    F29() = default;
    // widening the bound is free
    template <int A>
    LWK_HD F29(const F29<A, INL> &o) {
        static_assert(A <= B, "bound would shrink: a value < A*p is not known to be < B*p");
#pragma unroll
        for (int i = 0; i < 14; i++) l[i] = o.l[i];
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
    // but a 2^-29-ish fraction of non-zero values; candidates get the exact comparison.
    LWK_HD bool is_zero() const {
        static_assert(B <= 64, "is_zero scans B multiples");
        const uint32_t low = l[0];
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
        u64 c = 0;
        uint32_t diff = 0;
        for (int i = 0; i < 14; i++) {
            c += (u64)kk * P29::MOD[i];
            uint32_t want = (i < 13) ? ((uint32_t)c & P29::MASK) : (uint32_t)c;
            c >>= 29;
            diff |= want ^ l[i];
        }
        return diff == 0;
    }
};

// one carry ripple: limbs 0..12 back under 2^29 (inputs < 2^32 per limb)
LWK_HD void norm29(uint32_t *l) {
#pragma unroll
    for (int i = 0; i < 13; i++) {
        l[i + 1] += l[i] >> 29;
        l[i] &= P29::MASK;
    }
}

template <int A, int B, bool I>
LWK_HD F29<A + B, I> operator+(const F29<A, I> &a, const F29<B, I> &b) {
    static_assert(A + B <= 4096, "value bound");
    F29<A + B, I> r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + b.l[i];
    norm29(r.l);
    return r;
}

template <int A, int B, bool I>
LWK_HD F29<A + sub_offset(B), I> operator-(const F29<A, I> &a, const F29<B, I> &b) {
    constexpr int K = sub_offset(B);
    static_assert(A + K <= 4096, "value bound");
    F29<A + K, I> r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + kp_limb<K>(i) - b.l[i];
    norm29(r.l);
    return r;
}

template <int A, bool I>
LWK_HD F29<2 * A, I> dbl(const F29<A, I> &a) {
    return a + a;
}

template <int A, bool I>
LWK_HD F29<sub_offset(A), I> neg(const F29<A, I> &a) {
    constexpr int K = sub_offset(A);
    F29<K, I> r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = kp_limb<K>(i) - a.l[i];
    norm29(r.l);
    return r;
}

// flag ? -a : a, one type for both outcomes
template <int A, bool I>
LWK_HD F29<sub_offset(A), I> cneg(const F29<A, I> &a, bool flag) {
    constexpr int K = sub_offset(A);
    F29<K, I> r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = flag ? kp_limb<K>(i) - a.l[i] : a.l[i];
    norm29(r.l);
    return r;
}

// Montgomery product, product scanning: column k of a*b + m*p accumulates in ONE 64-bit register
// (at most 28 products < 2^58 plus the running carry), is cleared in its low 29 bits by the choice
// of m[k], and shifts down. Result < p + a*b/2^406 < 2p for a*b < 2^22 p^2.
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
        acc >>= 29;
    }
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (u64)a[i] * b[k - i];
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (u64)m[i] * P29::MOD[k - i];
        r[k - 14] = (uint32_t)acc & P29::MASK;
        acc >>= 29;
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

template <int A, int B, bool I>
LWK_HD F29<2, I> operator*(const F29<A, I> &a, const F29<B, I> &b) {
    static_assert((long long)A * B <= (1ll << 22), "product of bounds too large for the Montgomery radix");
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

// Montgomery square: the 91 cross products are taken once against a doubled operand (2 a_i < 2^30,
// products < 2^59), so a column holds at most 7 * 2^59 + 2^58 + 14 * 2^58 < 2^63.  105 + 196 multiply-adds
// instead of 392.
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
        acc >>= 29;
    }
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; 2 * i < k; i++) acc += (u64)a2[i] * a[k - i];
        if ((k & 1) == 0) acc += (u64)a[k >> 1] * a[k >> 1];
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (u64)m[i] * P29::MOD[k - i];
        r[k - 14] = (uint32_t)acc & P29::MASK;
        acc >>= 29;
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

template <int A, bool I>
LWK_HD F29<2, I> sqr(const F29<A, I> &a) {
    static_assert((long long)A * A <= (1ll << 22), "square of the bound too large for the Montgomery radix");
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

// a*b + c*d with ONE Montgomery reduction (588 multiply-adds instead of 784). A column holds at most 28 products
// < 2^58 plus 14 reduction products plus the carry: < 2^63.5.
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
        acc >>= 29;
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
        acc >>= 29;
    }
    r[13] = (uint32_t)acc;
}

// a*b - c*d. Inlined flavour: one fused product pair over a negated c; call flavour: two products and a subtraction.
template <int A, int B, int C, int D, bool I>
LWK_HD auto mul_sub(const F29<A, I> &a, const F29<B, I> &b, const F29<C, I> &c, const F29<D, I> &d) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (I) {
        static_assert((long long)A * B + (long long)sub_offset(C) * D <= (1ll << 22), "bounds too large for the Montgomery radix");
        auto nc = neg(c);
        F29<2, I> r;
        mont_mul_add29(r.l, a.l, b.l, nc.l, d.l);
        return r;
    } else {
        return a * b - c * d;
    }
#else
    return a * b - c * d;
#endif
}

// ---- conversions -----------------------------------------------------------------------------------

// canonical integer (12 x u32, < 2^384) -> F29<2> in Montgomery form (radix 2^406)
LWK_HD F29<2> f29_from_raw32(const uint32_t raw[12]) {
    F29<32> t;  // any integer < 2^384 < 8.3 p... declare generously: < 32p
#pragma unroll
    for (int i = 0; i < 14; i++) {
        int bit = 29 * i;
        int w = bit >> 5, sh = bit & 31;
        uint32_t v = 0;
        if (w < 12) {
            v = raw[w] >> sh;
            if (sh + 29 > 32 && w + 1 < 12) v |= raw[w + 1] << (32 - sh);
        }
        t.l[i] = v & P29::MASK;
    }
    F29<1> r2;
#pragma unroll
    for (int i = 0; i < 14; i++) r2.l[i] = P29::R2[i];
    return t * r2;
}

// F29<B> -> canonical integer in [0, p) as 12 x u32
template <int B>
LWK_HD void f29_to_raw32(uint32_t raw[12], const F29<B> &a) {
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
        c >>= 29;
    }
    bool ge = c >= 0 && (int32_t)d[13] >= 0;
    uint32_t w[14];
    for (int i = 0; i < 14; i++) w[i] = ge ? d[i] : v.l[i];
    for (int i = 0; i < 12; i++) raw[i] = 0;
    for (int i = 0; i < 14; i++) {
        int bit = 29 * i;
        int k = bit >> 5, sh = bit & 31;
        if (k < 12) raw[k] |= w[i] << sh;
        if (sh + 29 > 32 && k + 1 < 12) raw[k + 1] |= w[i] >> (32 - sh);
    }
}

LWK_HD F29<2> f29_from_fp(const Fp &a) {
    uint32_t raw[12];
    fe_to_raw<FpParams>(raw, a);
    return f29_from_raw32(raw);
}

template <int B>
LWK_HD Fp f29_to_fp(const F29<B> &a) {
    uint32_t raw[12];
    f29_to_raw32(raw, a);
    return fe_from_raw<FpParams>(raw);
}

// a^e, public exponent, NE little-endian 32-bit limbs
template <int NE, int B>
LWK_HD F29<2> f29_pow(const F29<B> &a, const uint32_t *e) {
    F29<2> acc = F29<2>::one();
    F29<2> base = a * F29<1>::one();  // a * R / R = a, but weakly reduced to < 2p
    bool started = false;
    for (int i = NE * 32 - 1; i >= 0; i--) {
        if (started) acc = sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            acc = started ? acc * base : base;
            started = true;
        }
    }
    return acc;
}

// Fermat inversion a^(p-2): ~480 dependent products. Kept as the cross-check of f29_inv (tools/host_check.hip).
template <int B>
LWK_HD F29<2> f29_inv_fermat(const F29<B> &a) {
    uint32_t e[12], two[12];
#pragma unroll
    for (int i = 0; i < 12; i++) two[i] = (i == 0) ? 2u : 0u;
    raw_sub<12>(e, FpParams::MOD, two);
    return f29_pow<12>(a, e);
}

template <int B>
LWK_HD F29<2> f29_inv(const F29<B> &a) {
    uint32_t x[12], y[12];
    f29_to_raw32(x, a);       // out of Montgomery form, canonical
    fp_inv_raw32(y, x);
    return f29_from_raw32(y);  // back into Montgomery form
}

}  // namespace lwk
