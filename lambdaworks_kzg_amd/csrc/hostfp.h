// hostfp.h -- host-only Fp (pairing, host-side G1 arithmetic) and Fr (random-linear-combination scalars): 64-bit limbs,
// Montgomery radix 2^(64 N), `unsigned __int128` products. Same residues as the 32-bit-limb Fe<Params> of field.cuh
// (same radix), so conversion is limb packing; the 32-bit representation exists for the GPU, and on an x86 core it
// costs 3-4x this one.
#pragma once
#include "field.cuh"
#include "knobs.h"

namespace lwk {

struct HFpPar {
    static constexpr int N = 6;
    typedef Fp FeT;
    static constexpr uint64_t P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                      0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    static constexpr uint64_t N0 = 0x89f3fffcfffcfffdull;  // -p^-1 mod 2^64
};
struct HFrPar {
    static constexpr int N = 4;
    typedef Fr FeT;
    static constexpr uint64_t P[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    static constexpr uint64_t N0 = 0xfffffffeffffffffull;  // -r^-1 mod 2^64
};

template <class Par>
struct HostField {
    static constexpr int N = Par::N;
    uint64_t l[N];

    static HostField from_fe(const typename Par::FeT &a) {
        HostField r;
        for (int k = 0; k < N; k++) r.l[k] = (uint64_t)a.l[2 * k] | ((uint64_t)a.l[2 * k + 1] << 32);
        return r;
    }
    typename Par::FeT to_fe() const {
        typename Par::FeT r;
        for (int k = 0; k < N; k++) {
            r.l[2 * k] = (uint32_t)l[k];
            r.l[2 * k + 1] = (uint32_t)(l[k] >> 32);
        }
        return r;
    }
    static HostField zero() {
        HostField r;
        for (int k = 0; k < N; k++) r.l[k] = 0;
        return r;
    }
    static HostField one() { return from_fe(Par::FeT::one()); }
    bool is_zero() const {
        uint64_t d = 0;
        for (int k = 0; k < N; k++) d |= l[k];
        return d == 0;
    }
    bool operator==(const HostField &o) const {
        uint64_t d = 0;
        for (int k = 0; k < N; k++) d |= l[k] ^ o.l[k];
        return d == 0;
    }
};
typedef HostField<HFpPar> HFp;
typedef HostField<HFrPar> HFr;

// t -= modulus if t >= modulus (t < 2 * modulus). Branch-free: on field data the comparison is a coin flip, and a mispredicted
// branch costs more than the whole subtraction.
template <class Par>
inline void hf_cond_sub(uint64_t *t) {
    uint64_t d[Par::N];
    uint64_t br = 0;
    for (int k = 0; k < Par::N; k++) {
        unsigned __int128 v = (unsigned __int128)t[k] - Par::P[k] - br;
        d[k] = (uint64_t)v;
        br = (uint64_t)(v >> 64) & 1;
    }
    const uint64_t keep = 0 - br;  // all ones: t < modulus, keep t
    for (int k = 0; k < Par::N; k++) t[k] = (t[k] & keep) | (d[k] & ~keep);
}

template <class Par>
inline HostField<Par> operator+(const HostField<Par> &a, const HostField<Par> &b) {
    HostField<Par> r;
    unsigned __int128 c = 0;
    for (int k = 0; k < Par::N; k++) {
        c += (unsigned __int128)a.l[k] + b.l[k];
        r.l[k] = (uint64_t)c;
        c >>= 64;
    }
    hf_cond_sub<Par>(r.l);  // a + b < 2 * modulus < 2^(64 N): no carry out
    return r;
}

template <class Par>
inline HostField<Par> operator-(const HostField<Par> &a, const HostField<Par> &b) {
    HostField<Par> r;
    uint64_t br = 0;
    for (int k = 0; k < Par::N; k++) {
        unsigned __int128 v = (unsigned __int128)a.l[k] - b.l[k] - br;
        r.l[k] = (uint64_t)v;
        br = (uint64_t)(v >> 64) & 1;
    }
    const uint64_t back = 0 - br;  // all ones: the difference wrapped, add the modulus back
    unsigned __int128 c = 0;
    for (int k = 0; k < Par::N; k++) {
        c += (unsigned __int128)r.l[k] + (Par::P[k] & back);
        r.l[k] = (uint64_t)c;
        c >>= 64;
    }
    return r;
}

template <class Par>
inline HostField<Par> neg(const HostField<Par> &a) {
    return a.is_zero() ? a : HostField<Par>::zero() - a;
}

// Montgomery product, coarsely integrated operand scanning
template <class Par>
inline HostField<Par> hf_mul_portable(const HostField<Par> &a, const HostField<Par> &b) {
    constexpr int N = Par::N;
    uint64_t t[N + 2];
    for (int k = 0; k < N + 2; k++) t[k] = 0;
    for (int i = 0; i < N; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < N; j++) {
            c += (unsigned __int128)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[N];
        t[N] = (uint64_t)c;
        t[N + 1] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * Par::N0;
        c = (unsigned __int128)m * Par::P[0] + t[0];
        c >>= 64;
        for (int j = 1; j < N; j++) {
            c += (unsigned __int128)m * Par::P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[N];
        t[N - 1] = (uint64_t)c;
        t[N] = t[N + 1] + (uint64_t)(c >> 64);
    }
    HostField<Par> r;
    for (int k = 0; k < N; k++) r.l[k] = t[k];
    hf_cond_sub<Par>(r.l);
    return r;
}

template <class Par>
inline HostField<Par> operator*(const HostField<Par> &a, const HostField<Par> &b) {
    return hf_mul_portable<Par>(a, b);
}
#if defined(__x86_64__)
// Fp on a core with BMI2 + ADX: the same algorithm on MULX and the two carry chains (fp_x86.S, generated by
// tools/gen_fp_x86.py; 37 ns against 59 ns for the C above on the build host). The C path stays for other cores and
// is what tests/test_host_fp_asm_cpu.py compares it with. Non-template, so overload resolution prefers it for HFp.
extern "C" void lwk_fp_mul_adx(uint64_t *r, const uint64_t *a, const uint64_t *b);
extern "C" void lwk_fp2_mul_adx(uint64_t *r, const uint64_t *a, const uint64_t *b);  // 12 limbs each: (c0, c1) of Fp[u]/(u^2 + 1)
extern "C" int lwk_cpu_has_bmi2_adx(void);
inline bool hf_fp_on_adx() {
    static const bool yes = lwk_cpu_has_bmi2_adx() != 0 && knobs().host_fp_portable != 1;
    return yes;
}
inline bool hf_fp2_on_adx() {
    static const bool yes = hf_fp_on_adx() && knobs().host_fp_portable != 2;
    return yes;
}
inline HFp operator*(const HFp &a, const HFp &b) {
    if (!hf_fp_on_adx()) return hf_mul_portable<HFpPar>(a, b);
    HFp r;
    lwk_fp_mul_adx(r.l, a.l, b.l);
    return r;
}
#endif
template <class Par>
inline HostField<Par> sqr(const HostField<Par> &a) { return a * a; }
template <class Par>
inline HostField<Par> dbl(const HostField<Par> &a) { return a + a; }
template <class Par>
inline HostField<Par> mul_sub(const HostField<Par> &a, const HostField<Par> &b, const HostField<Par> &c, const HostField<Par> &d) {
    return a * b - c * d;
}
template <class Par>
inline HostField<Par> normed(const HostField<Par> &a) { return a; }
template <class Par>
inline bool literal_zero(const HostField<Par> &a) { return a.is_zero(); }  // g1.cuh's generic point types: infinity <=> zz == 0
inline HFp inv(const HFp &a) { return HFp::from_fe(inv(a.to_fe())); }  // division steps (field.cuh)

}  // namespace lwk
