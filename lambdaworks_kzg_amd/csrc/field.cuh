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
LWK_HD Fp normed(const Fp &a) { return a; }  // the lazy field's renormalisation point (field29.cuh); nothing to do here
LWK_HD Fr neg(const Fr &a) { return fe_neg<FrParams>(a); }
LWK_HD Fp dbl(const Fp &a) { return fe_dbl<FpParams>(a); }
// ---- inversion by division steps (Bernstein-Yang "safegcd", 30 steps per batch on 13 signed 30-bit limbs) -------
// A lane that inverts is a lane on a serial dependency chain (the final step of every commitment, k_finalize_compress),
// so what counts is the length of that chain: ~28 batches x (30 cheap word steps + two 2x2-matrix-times-vector
// updates) is ~15x shorter than the ~480 Montgomery products of Fermat's a^(p-2).
struct Gcd30 {   // Fp: 381 bits
    static constexpr int L = 13;
    static constexpr int NW = 12;      // 32-bit words of a canonical value
    static constexpr int BATCHES = 40; // 1103 steps bound the worst case; typical inputs finish in <= 28 batches
    static constexpr int32_t M30 = (1 << 30) - 1;
    static constexpr uint32_t PINV = 0x30003u;  // p^-1 mod 2^30
    static constexpr int32_t P[13] = {0x3fffaaab, 0x27fbffff, 0x153ffffb, 0x2affffac, 0x30f6241e, 0x034a83da, 0x112bf673,
                                      0x12e13ce1, 0x2cd76477, 0x1ed90d2e, 0x29a4b1ba, 0x3a8e5ff9, 0x001a0111};
};
struct Gcd30Fr {  // Fr: 255 bits (r05: the evaluation-form quotient's one inversion per blob, fr_ops.hip)
    static constexpr int L = 9;
    static constexpr int NW = 8;
    static constexpr int BATCHES = 26; // 741 steps bound a 256-bit modulus from delta = 1
    static constexpr int32_t M30 = (1 << 30) - 1;
    static constexpr uint32_t PINV = 0x1u;      // r = 1 mod 2^32
    static constexpr int32_t P[9] = {0x1, 0x3ffffffc, 0x3fe5bfef, 0x2f6900bf, 0x21d80553, 0x27602026, 0x17d48333, 0x29d4ca67, 0x73ed};
};

// 30 division steps on the low words; t = (u, v, q, r) with 2^30 (f', g') = (u f + v g, q f + r g)
LWK_HD int32_t gcd30_divsteps(int32_t eta, uint32_t f, uint32_t g, int32_t t[4]) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
    for (int i = 0; i < 30; i++) {
        uint32_t c1 = (uint32_t)(eta >> 31);  // eta < 0  (delta > 0)
        uint32_t c2 = 0u - (g & 1u);          // g odd
        uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
        g += x & c2;
        q += y & c2;
        r += z & c2;
        c1 &= c2;
        eta = (int32_t)(((uint32_t)eta ^ c1) - 1u - c1);  // c1 ? -eta - 1 : eta - 1
        f += g & c1;
        u += q & c1;
        v += r & c1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t[0] = (int32_t)u;
    t[1] = (int32_t)v;
    t[2] = (int32_t)q;
    t[3] = (int32_t)r;
    return eta;
}

// (f, g) <- t (f, g) / 2^30, exact
template <class G = Gcd30>
LWK_HD void gcd30_update_fg(int32_t *f, int32_t *g, const int32_t t[4]) {
    const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
    int64_t cf = u * f[0] + v * g[0], cg = q * f[0] + r * g[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < G::L; i++) {
        cf += u * f[i] + v * g[i];
        cg += q * f[i] + r * g[i];
        f[i - 1] = (int32_t)cf & G::M30;
        g[i - 1] = (int32_t)cg & G::M30;
        cf >>= 30;
        cg >>= 30;
    }
    f[G::L - 1] = (int32_t)cf;
    g[G::L - 1] = (int32_t)cg;
}

// (d, e) <- t (d, e) / 2^30 mod p, both kept in (-2p, p)
template <class G = Gcd30>
LWK_HD void gcd30_update_de(int32_t *d, int32_t *e, const int32_t t[4]) {
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    const int32_t sd = d[G::L - 1] >> 31, se = e[G::L - 1] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int64_t cd = (int64_t)u * d[0] + (int64_t)v * e[0], ce = (int64_t)q * d[0] + (int64_t)r * e[0];
    // multiples of p that clear the low 30 bits
    md -= (int32_t)((G::PINV * (uint32_t)cd + (uint32_t)md) & (uint32_t)G::M30);
    me -= (int32_t)((G::PINV * (uint32_t)ce + (uint32_t)me) & (uint32_t)G::M30);
    cd += (int64_t)G::P[0] * md;
    ce += (int64_t)G::P[0] * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < G::L; i++) {
        cd += (int64_t)u * d[i] + (int64_t)v * e[i] + (int64_t)G::P[i] * md;
        ce += (int64_t)q * d[i] + (int64_t)r * e[i] + (int64_t)G::P[i] * me;
        d[i - 1] = (int32_t)cd & G::M30;
        e[i - 1] = (int32_t)ce & G::M30;
        cd >>= 30;
        ce >>= 30;
    }
    d[G::L - 1] = (int32_t)cd;
    e[G::L - 1] = (int32_t)ce;
}

// canonical x in [0, p) as NW x u32 -> x^-1 mod p in [0, p) (0 -> 0)
template <class G>
LWK_HD void gcd30_inv_raw32(uint32_t *out, const uint32_t *x) {
    constexpr int L = G::L, NW = G::NW;
    int32_t f[L], g[L], d[L], e[L];
#pragma unroll
    for (int i = 0; i < L; i++) {
        const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
        uint32_t lo = x[w] >> sh;
        if (sh + 30 > 32 && w + 1 < NW) lo |= x[w + 1] << (32 - sh);
        g[i] = (int32_t)(lo & (uint32_t)G::M30);
        f[i] = G::P[i];
        d[i] = 0;
        e[i] = (i == 0) ? 1 : 0;
    }
    int32_t eta = -1;
    for (int it = 0; it < G::BATCHES; it++) {
        int32_t nz = 0;
#pragma unroll
        for (int i = 0; i < L; i++) nz |= g[i];
        if (nz == 0) break;
        int32_t t[4];
        eta = gcd30_divsteps(eta, (uint32_t)f[0], (uint32_t)g[0], t);
        gcd30_update_de<G>(d, e, t);
        gcd30_update_fg<G>(f, g, t);
    }
    // f = +-1 (or +-p for x = 0, where d = 0): d <- sign(f) d, into [0, p)
    const int32_t fneg = f[L - 1] >> 31;
    int32_t add = d[L - 1] >> 31;
#pragma unroll
    for (int i = 0; i < L; i++) d[i] = ((d[i] + (G::P[i] & add)) ^ fneg) - fneg;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
        d[i + 1] += d[i] >> 30;
        d[i] &= G::M30;
    }
    add = d[L - 1] >> 31;
#pragma unroll
    for (int i = 0; i < L; i++) d[i] += G::P[i] & add;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
        d[i + 1] += d[i] >> 30;
        d[i] &= G::M30;
    }
#pragma unroll
    for (int i = 0; i < NW; i++) out[i] = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const int bit = 30 * i, k = bit >> 5, sh = bit & 31;
        out[k] |= (uint32_t)d[i] << sh;
        if (sh + 30 > 32 && k + 1 < NW) out[k + 1] |= (uint32_t)d[i] >> (32 - sh);
    }
}

LWK_HD void fp_inv_raw32(uint32_t out[12], const uint32_t x[12]) { gcd30_inv_raw32<Gcd30>(out, x); }
LWK_HD void fr_inv_raw32(uint32_t out[8], const uint32_t x[8]) { gcd30_inv_raw32<Gcd30Fr>(out, x); }

// Fp inversion: division steps; fe_inv<FpParams> (Fermat) stays as the cross-check
LWK_HD Fp inv(const Fp &a) {
    uint32_t x[12], y[12];
    fe_to_raw<FpParams>(x, a);
    fp_inv_raw32(y, x);
    return fe_from_raw<FpParams>(y);
}
LWK_HD Fp inv_fermat(const Fp &a) { return fe_inv<FpParams>(a); }
LWK_HD Fr inv(const Fr &a) { return fe_inv<FrParams>(a); }
// the same by division steps (~15x shorter as a dependency chain: one lane inverts per blob in fr_ops.hip: k_eval_quotient_evalform)
LWK_HD Fr inv_divsteps(const Fr &a) {
    uint32_t x[8], y[8];
    fe_to_raw<FrParams>(x, a);
    fr_inv_raw32(y, x);
    return fe_from_raw<FrParams>(y);
}

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
