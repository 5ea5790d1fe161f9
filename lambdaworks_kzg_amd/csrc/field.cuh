// field.cuh -- BLS12-381 base field Fp (381 bit) and scalar field Fr (255 bit) for gfx950.
//
// Replaces, for the blob-commitment hot path, what the reference gets from the un-vendored
// lambdaworks-math crate: FieldElement<MontgomeryBackendPrimeField<..>> (call sites
// /root/reference/src/lib.rs:11-30, src/utils.rs:36,153, src/compression.rs:84-98).
//
// Representation: Montgomery form, little-endian 32-bit limbs (12 for Fp, 8 for Fr), always
// fully reduced to [0, p).  gfx950 has no 64-bit integer multiplier; the primitive is
// v_mad_u64_u32 (32x32+64 -> 64), which the (u64)a*b+c expressions below lower to.
// Both moduli leave the top limb's high bit clear, so the CIOS accumulator needs N+1 limbs.
//
// The same code is compiled for the host (setup-time helpers, G2/pairing side) and the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lwk {

typedef unsigned long long u64;

#define LWK_HD __host__ __device__ __forceinline__

struct FpParams {
    static constexpr int N = 12;
    static constexpr uint32_t INV = 0xfffcfffdu;  // -p^-1 mod 2^32
    static constexpr uint32_t MOD[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                         0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    static constexpr uint32_t R1[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                        0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
    static constexpr uint32_t R2[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                        0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
};

struct FrParams {
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xffffffffu;  // -r^-1 mod 2^32
    static constexpr uint32_t MOD[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                        0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    static constexpr uint32_t R1[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                       0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
    static constexpr uint32_t R2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                       0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
};

template <class P>
struct alignas(16) Fe {
    static constexpr int N = P::N;
    uint32_t l[P::N];

    LWK_HD static Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    LWK_HD static Fe one() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::R1[i];
        return r;
    }
    LWK_HD bool is_zero() const {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < N; i++) x |= l[i];
        return x == 0;
    }
    LWK_HD bool operator==(const Fe &o) const {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < N; i++) x |= l[i] ^ o.l[i];
        return x == 0;
    }
    LWK_HD bool operator!=(const Fe &o) const { return !(*this == o); }
};

// raw (non-modular) helpers on N-limb little-endian integers -------------------------------

template <int N>
LWK_HD uint32_t raw_add(uint32_t *o, const uint32_t *a, const uint32_t *b) {
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        c += (u64)a[i] + b[i];
        o[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
}

template <int N>
LWK_HD uint32_t raw_sub(uint32_t *o, const uint32_t *a, const uint32_t *b) {
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        u64 d = (u64)a[i] - b[i] - br;
        o[i] = (uint32_t)d;
        br = (uint32_t)(d >> 63);
    }
    return br;
}

// a >= b
template <int N>
LWK_HD bool raw_geq(const uint32_t *a, const uint32_t *b) {
    uint32_t t[N];
    return raw_sub<N>(t, a, b) == 0;
}

// o = (t >= m) ? t - m : t   (t < 2m)
template <class P>
LWK_HD void cond_sub_mod(uint32_t *o, const uint32_t *t) {
    uint32_t d[P::N];
    uint32_t br = raw_sub<P::N>(d, t, P::MOD);
#pragma unroll
    for (int i = 0; i < P::N; i++) o[i] = br ? t[i] : d[i];
}

// modular ops --------------------------------------------------------------------------------

template <class P>
LWK_HD Fe<P> fe_add(const Fe<P> &a, const Fe<P> &b) {
    uint32_t t[P::N];
    raw_add<P::N>(t, a.l, b.l);  // no carry out: both moduli are < 2^(32N-1)
    Fe<P> r;
    cond_sub_mod<P>(r.l, t);
    return r;
}

template <class P>
LWK_HD Fe<P> fe_sub(const Fe<P> &a, const Fe<P> &b) {
    uint32_t d[P::N], e[P::N];
    uint32_t br = raw_sub<P::N>(d, a.l, b.l);
    raw_add<P::N>(e, d, P::MOD);
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = br ? e[i] : d[i];
    return r;
}

template <class P>
LWK_HD Fe<P> fe_neg(const Fe<P> &a) {
    uint32_t d[P::N];
    raw_sub<P::N>(d, P::MOD, a.l);
    bool z = a.is_zero();
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = z ? 0u : d[i];
    return r;
}

template <class P>
LWK_HD Fe<P> fe_dbl(const Fe<P> &a) {
    return fe_add<P>(a, a);
}

// Montgomery product a*b*R^-1 mod p, CIOS, R = 2^(32N).  `a` must be < p; `b` may be any
// N-limb integer (the accumulator stays < 2p either way).
template <class P>
LWK_HD Fe<P> fe_mul_inl(const Fe<P> &a, const Fe<P> &b) {
    constexpr int N = P::N;
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            u64 s = (u64)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (uint32_t)s;
            c = s >> 32;
        }
        u64 s = (u64)t[N] + c;
        t[N] = (uint32_t)s;
        uint32_t m = t[0] * P::INV;
        s = (u64)m * P::MOD[0] + t[0];
        c = s >> 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            s = (u64)m * P::MOD[j] + t[j] + c;
            t[j - 1] = (uint32_t)s;
            c = s >> 32;
        }
        s = (u64)t[N] + c;
        t[N - 1] = (uint32_t)s;
        t[N] = (uint32_t)(s >> 32);
    }
    Fe<P> r;
    cond_sub_mod<P>(r.l, t);  // t < 2p, t[N] == 0
    return r;
}

// On the device the product is a real function (s_swappc): a madd is ten of them, and inlining
// ten 1.2k-instruction bodies per group operation overflows the 64 KiB instruction cache and takes
// minutes to compile. Operands travel by value in VGPRs (48 B aggregates are passed directly).
#if defined(__HIP_DEVICE_COMPILE__)
template <class P>
__device__ __noinline__ Fe<P> fe_mul_call(Fe<P> a, Fe<P> b) {
    return fe_mul_inl<P>(a, b);
}
#endif

template <class P>
LWK_HD Fe<P> fe_mul(const Fe<P> &a, const Fe<P> &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fe_mul_call<P>(a, b);
#else
    return fe_mul_inl<P>(a, b);
#endif
}

template <class P>
LWK_HD Fe<P> fe_sqr(const Fe<P> &a) {
    return fe_mul<P>(a, a);
}

// canonical integer (little-endian limbs, any value < 2^(32N)) -> Montgomery form, reducing mod p
template <class P>
LWK_HD Fe<P> fe_from_raw(const uint32_t *raw) {
    Fe<P> t, r2;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        t.l[i] = raw[i];
        r2.l[i] = P::R2[i];
    }
    // CIOS keeps its accumulator < 2p for ANY limb-scanned operand (2nd argument) < 2^(32N)
    // as long as the other operand (R2 here) is < p, so t needs no pre-reduction.
    return fe_mul<P>(r2, t);
}

template <class P>
LWK_HD void fe_to_raw(uint32_t *raw, const Fe<P> &a) {
    Fe<P> one;
#pragma unroll
    for (int i = 0; i < P::N; i++) one.l[i] = (i == 0) ? 1u : 0u;
    Fe<P> r = fe_mul<P>(a, one);
#pragma unroll
    for (int i = 0; i < P::N; i++) raw[i] = r.l[i];
}

// a^e for a public exponent given as NE little-endian 32-bit limbs (left-to-right)
template <class P, int NE>
LWK_HD Fe<P> fe_pow(const Fe<P> &a, const uint32_t *e) {
    Fe<P> acc = Fe<P>::one();
    bool started = false;
    for (int i = NE * 32 - 1; i >= 0; i--) {
        if (started) acc = fe_sqr<P>(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            acc = started ? fe_mul<P>(acc, a) : a;
            started = true;
        }
    }
    return acc;
}

// inverse by Fermat (a^(p-2)); a == 0 -> 0
template <class P>
LWK_HD Fe<P> fe_inv(const Fe<P> &a) {
    uint32_t e[P::N], two[P::N];
#pragma unroll
    for (int i = 0; i < P::N; i++) two[i] = (i == 0) ? 2u : 0u;
    raw_sub<P::N>(e, P::MOD, two);  // r ends in ...00000001: the borrow must propagate
    return fe_pow<P, P::N>(a, e);
}

typedef Fe<FpParams> Fp;
typedef Fe<FrParams> Fr;

LWK_HD Fp operator+(const Fp &a, const Fp &b) { return fe_add<FpParams>(a, b); }
LWK_HD Fp operator-(const Fp &a, const Fp &b) { return fe_sub<FpParams>(a, b); }
LWK_HD Fp operator*(const Fp &a, const Fp &b) { return fe_mul<FpParams>(a, b); }
LWK_HD Fr operator+(const Fr &a, const Fr &b) { return fe_add<FrParams>(a, b); }
LWK_HD Fr operator-(const Fr &a, const Fr &b) { return fe_sub<FrParams>(a, b); }
LWK_HD Fr operator*(const Fr &a, const Fr &b) { return fe_mul<FrParams>(a, b); }
LWK_HD Fp sqr(const Fp &a) { return fe_sqr<FpParams>(a); }
LWK_HD Fr sqr(const Fr &a) { return fe_sqr<FrParams>(a); }
LWK_HD Fp neg(const Fp &a) { return fe_neg<FpParams>(a); }
LWK_HD Fp mul_sub(const Fp &a, const Fp &b, const Fp &c, const Fp &d) { return a * b - c * d; }
LWK_HD Fr neg(const Fr &a) { return fe_neg<FrParams>(a); }
LWK_HD Fp dbl(const Fp &a) { return fe_dbl<FpParams>(a); }
LWK_HD Fp inv(const Fp &a) { return fe_inv<FpParams>(a); }
LWK_HD Fr inv(const Fr &a) { return fe_inv<FrParams>(a); }

// byte conversions (canonical big-endian, as the reference's to_bytes_be / from_bytes_be) -------

// 4*N big-endian bytes -> raw little-endian limbs
template <int N>
LWK_HD void raw_from_be(uint32_t *raw, const uint8_t *b) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint8_t *q = b + 4 * (N - 1 - i);
        raw[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
    }
}
template <int N>
LWK_HD void raw_to_be(uint8_t *b, const uint32_t *raw) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint8_t *q = b + 4 * (N - 1 - i);
        uint32_t v = raw[i];
        q[0] = (uint8_t)(v >> 24);
        q[1] = (uint8_t)(v >> 16);
        q[2] = (uint8_t)(v >> 8);
        q[3] = (uint8_t)v;
    }
}
template <int N>
LWK_HD void raw_from_le(uint32_t *raw, const uint8_t *b) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint8_t *q = b + 4 * i;
        raw[i] = ((uint32_t)q[3] << 24) | ((uint32_t)q[2] << 16) | ((uint32_t)q[1] << 8) | q[0];
    }
}
template <int N>
LWK_HD void raw_to_le(uint8_t *b, const uint32_t *raw) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint8_t *q = b + 4 * i;
        uint32_t v = raw[i];
        q[0] = (uint8_t)v;
        q[1] = (uint8_t)(v >> 8);
        q[2] = (uint8_t)(v >> 16);
        q[3] = (uint8_t)(v >> 24);
    }
}

}  // namespace lwk
