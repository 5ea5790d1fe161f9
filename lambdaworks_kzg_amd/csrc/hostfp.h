// hostfp.h -- host-only Fp for the pairing: 6 x 64-bit limbs, Montgomery radix 2^384, `unsigned __int128`
// products. Same residues as the 12 x 32-bit Fe<FpParams> (same radix), so conversion is limb packing; the
// 32-bit representation exists for the GPU, and on an x86 core it costs 3-4x this one.
#pragma once
#include "field.cuh"

namespace lwk {

struct HFp {
    uint64_t l[6];

    static constexpr uint64_t P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                      0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    static constexpr uint64_t N0 = 0x89f3fffcfffcfffdull;  // -p^-1 mod 2^64

    static HFp from_fe(const Fp &a) {
        HFp r;
        for (int k = 0; k < 6; k++) r.l[k] = (uint64_t)a.l[2 * k] | ((uint64_t)a.l[2 * k + 1] << 32);
        return r;
    }
    Fp to_fe() const {
        Fp r;
        for (int k = 0; k < 6; k++) {
            r.l[2 * k] = (uint32_t)l[k];
            r.l[2 * k + 1] = (uint32_t)(l[k] >> 32);
        }
        return r;
    }
    static HFp zero() {
        HFp r;
        for (int k = 0; k < 6; k++) r.l[k] = 0;
        return r;
    }
    static HFp one() { return from_fe(Fp::one()); }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3] | l[4] | l[5]) == 0; }
    bool operator==(const HFp &o) const {
        uint64_t d = 0;
        for (int k = 0; k < 6; k++) d |= l[k] ^ o.l[k];
        return d == 0;
    }
};

// r = t - p if t >= p (t < 2p, 6 limbs)
inline void hfp_cond_sub(uint64_t t[6]) {
    uint64_t d[6];
    unsigned __int128 br = 0;
    for (int k = 0; k < 6; k++) {
        unsigned __int128 v = (unsigned __int128)t[k] - HFp::P[k] - (uint64_t)br;
        d[k] = (uint64_t)v;
        br = (v >> 64) & 1;
    }
    if (!br)
        for (int k = 0; k < 6; k++) t[k] = d[k];
}

inline HFp operator+(const HFp &a, const HFp &b) {
    HFp r;
    unsigned __int128 c = 0;
    for (int k = 0; k < 6; k++) {
        c += (unsigned __int128)a.l[k] + b.l[k];
        r.l[k] = (uint64_t)c;
        c >>= 64;
    }
    hfp_cond_sub(r.l);  // a + b < 2p < 2^382: no carry out
    return r;
}

inline HFp operator-(const HFp &a, const HFp &b) {
    HFp r;
    unsigned __int128 br = 0;
    for (int k = 0; k < 6; k++) {
        unsigned __int128 v = (unsigned __int128)a.l[k] - b.l[k] - (uint64_t)br;
        r.l[k] = (uint64_t)v;
        br = (v >> 64) & 1;
    }
    if (br) {
        unsigned __int128 c = 0;
        for (int k = 0; k < 6; k++) {
            c += (unsigned __int128)r.l[k] + HFp::P[k];
            r.l[k] = (uint64_t)c;
            c >>= 64;
        }
    }
    return r;
}

inline HFp neg(const HFp &a) { return a.is_zero() ? a : HFp::zero() - a; }

// Montgomery product, coarsely integrated operand scanning
inline HFp operator*(const HFp &a, const HFp &b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < 6; j++) {
            c += (unsigned __int128)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[6] = (uint64_t)c;
        t[7] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * HFp::N0;
        c = (unsigned __int128)m * HFp::P[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 6; j++) {
            c += (unsigned __int128)m * HFp::P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[5] = (uint64_t)c;
        t[6] = t[7] + (uint64_t)(c >> 64);
    }
    HFp r;
    for (int k = 0; k < 6; k++) r.l[k] = t[k];
    hfp_cond_sub(r.l);
    return r;
}

inline HFp sqr(const HFp &a) { return a * a; }
inline HFp dbl(const HFp &a) { return a + a; }
inline HFp mul_sub(const HFp &a, const HFp &b, const HFp &c, const HFp &d) { return a * b - c * d; }
inline HFp normed(const HFp &a) { return a; }
inline bool literal_zero(const HFp &a) { return a.is_zero(); }  // g1.cuh's generic point types: infinity <=> zz == 0
inline HFp inv(const HFp &a) { return HFp::from_fe(inv(a.to_fe())); }  // division steps (field.cuh)

}  // namespace lwk
