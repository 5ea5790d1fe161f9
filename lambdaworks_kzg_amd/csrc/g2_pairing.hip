// g2_pairing.hip -- host-side G2 handling for the trusted setup.
//
// SURVEY section 8a marks G2 out of the GPU hot path (65 points at load, 2 pairings per verify):
// it stays on the host, as in the reference. This file restates decompress_g2_point
// (/root/reference/src/compression.rs:105-139) and g2_point_to_blst_p2
// (/root/reference/src/srs.rs:175-212) on top of the shared field code.
//
// Deliberate difference, documented in DESIGN.md: the reference ignores the ZCash sign bit of a
// compressed G2 point (it takes whatever root upstream sqrt_qfe(.., 0) returns) and does no subgroup
// check. Here the sign bit is honoured, which is the only reading under which verification against
// [tau]G2 is meaningful; for a setup file the reference handles correctly both agree.
#include "engine.h"
#include "fp2.h"

#include <string.h>

#include <functional>
#include <vector>

namespace lwk {
void host_parallel_for(size_t n, const std::function<void(size_t)> &fn);  // sha256_host.hip


static inline Fp2 fp2_add(const Fp2 &a, const Fp2 &b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
static inline Fp2 fp2_sub(const Fp2 &a, const Fp2 &b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
static inline Fp2 fp2_mul(const Fp2 &a, const Fp2 &b) {
    Fp t0 = a.c0 * b.c0, t1 = a.c1 * b.c1;
    Fp m = (a.c0 + a.c1) * (b.c0 + b.c1);
    return {t0 - t1, m - t0 - t1};
}
static inline Fp2 fp2_sqr(const Fp2 &a) { return fp2_mul(a, a); }
static inline Fp2 fp2_conj(const Fp2 &a) { return {a.c0, neg(a.c1)}; }
static inline bool fp2_eq(const Fp2 &a, const Fp2 &b) { return a.c0 == b.c0 && a.c1 == b.c1; }
static inline Fp2 fp2_one() { return {Fp::one(), Fp::zero()}; }

template <int NE>
static Fp2 fp2_pow(const Fp2 &a, const uint32_t *e) {
    Fp2 acc = fp2_one();
    for (int i = NE * 32 - 1; i >= 0; i--) {
        acc = fp2_sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = fp2_mul(acc, a);
    }
    return acc;
}

// square root in Fp2 for p = 3 mod 4 (Adj, Rodriguez-Henriquez, "Square root computation over even
// extension fields", Alg. 9). Returns false when `a` is not a square.
static bool fp2_sqrt(Fp2 &out, const Fp2 &a) {
    uint32_t p[12], one[12] = {1}, three[12] = {3}, e34[12], e12[12];
    for (int i = 0; i < 12; i++) p[i] = FpParams::MOD[i];
    raw_sub<12>(e34, p, three);  // (p - 3) / 4
    for (int k = 0; k < 2; k++) {
        for (int i = 0; i < 11; i++) e34[i] = (e34[i] >> 1) | (e34[i + 1] << 31);
        e34[11] >>= 1;
    }
    raw_sub<12>(e12, p, one);  // (p - 1) / 2
    for (int i = 0; i < 11; i++) e12[i] = (e12[i] >> 1) | (e12[i + 1] << 31);
    e12[11] >>= 1;

    if (a.c0.is_zero() && a.c1.is_zero()) {
        out = a;
        return true;
    }
    Fp2 a1 = fp2_pow<12>(a, e34);
    Fp2 alpha = fp2_mul(fp2_sqr(a1), a);
    Fp2 a0 = fp2_mul(fp2_conj(alpha), alpha);  // alpha^(p+1)
    Fp2 minus_one = {neg(Fp::one()), Fp::zero()};
    if (fp2_eq(a0, minus_one)) return false;
    Fp2 x0 = fp2_mul(a1, a);
    if (fp2_eq(alpha, minus_one)) {
        out = {neg(x0.c1), x0.c0};  // i * x0
    } else {
        Fp2 b = fp2_pow<12>(fp2_add(fp2_one(), alpha), e12);
        out = fp2_mul(b, x0);
    }
    return fp2_eq(fp2_sqr(out), a);
}

// ZCash ordering for Fp2: compare c1 first, then c0; "greater" means y > -y
static bool fp2_lex_greater(const Fp2 &y) {
    uint32_t a[12], b[12];
    Fp2 yn = {neg(y.c0), neg(y.c1)};
    if (!y.c1.is_zero()) {
        fe_to_raw<FpParams>(a, y.c1);
        fe_to_raw<FpParams>(b, yn.c1);
    } else {
        fe_to_raw<FpParams>(a, y.c0);
        fe_to_raw<FpParams>(b, yn.c0);
    }
    return !raw_geq<12>(b, a);  // yn < y
}

static void fp_to_blst(blst_fp *o, const Fp &v) {
    uint32_t raw[12];
    fe_to_raw<FpParams>(raw, v);
    for (int k = 0; k < 6; k++) o->l[5 - k] = (uint64_t)raw[2 * k] | ((uint64_t)raw[2 * k + 1] << 32);
}

// 96 bytes = x.c1 (48, flags in the top three bits) | x.c0 (48)   (compression.rs:127-131)
bool g2_decompress(Fp2 &x, Fp2 &y, bool &inf, const uint8_t in[96]) {
    uint8_t prefix = in[0] >> 5;
    if (!(prefix & 4)) return false;
    inf = (prefix & 2) != 0;
    if (inf) return true;
    uint8_t b[48];
    memcpy(b, in, 48);
    b[0] &= 0x1f;
    uint32_t raw[12];
    raw_from_be<12>(raw, b);
    x.c1 = fe_from_raw<FpParams>(raw);
    raw_from_be<12>(raw, in + 48);
    x.c0 = fe_from_raw<FpParams>(raw);
    Fp four = fp_from_u32(4);
    Fp2 bcoef = {four, four};  // y^2 = x^3 + 4(1 + i)   (compression.rs:133-134)
    Fp2 y2 = fp2_add(fp2_mul(fp2_sqr(x), x), bcoef);
    if (!fp2_sqrt(y, y2)) return false;
    bool want_greater = (prefix & 1) != 0;
    if (fp2_lex_greater(y) != want_greater) y = {neg(y.c0), neg(y.c1)};
    return true;
}

bool g2_fill_values(g2_t *out, const uint8_t *g2_bytes, size_t n2) {
    // one Fp2 square root per point on the 32-bit host field (~1.5 ms): spread over the host threads
    std::vector<int> bad(n2, 0);
    host_parallel_for(n2, [&](size_t i) {
        Fp2 x, y;
        bool inf = false;
        if (!g2_decompress(x, y, inf, g2_bytes + 96 * i)) {
            bad[i] = 1;
            return;
        }
        memset(&out[i], 0, sizeof(g2_t));
        if (inf) return;  // g2_point_to_blst_p2 of the neutral element goes through to_affine upstream; keep (0, 0, z = 0)
        fp_to_blst(&out[i].x.fp[0], x.c0);
        fp_to_blst(&out[i].x.fp[1], x.c1);
        fp_to_blst(&out[i].y.fp[0], y.c0);
        fp_to_blst(&out[i].y.fp[1], y.c1);
        out[i].z.fp[0].l[5] = 1;  // z = 1 + 0 i, canonical, most-significant limb first
    });
    for (size_t i = 0; i < n2; i++)
        if (bad[i]) {
            set_error("g2 point %zu: invalid compressed point", i);
            return false;
        }
    return true;
}

}  // namespace lwk
