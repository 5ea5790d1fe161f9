/* oracle/ref_pairing.c -- TEST INFRASTRUCTURE (never linked into, imported by or executed from the product): a second, deliberately
 * plain implementation of the BLS12-381 optimal ate pairing, so that the product's host pairing (csrc/pairing.hip: tower Fp2-Fp6-Fp12,
 * projective Miller loop with cached lines, cyclotomic final exponentiation) has a DIFFERENT implementation to be held against on
 * random inputs (Missing 7). What it stands in for on the reference's side: `BLS12381AtePairing::compute_batch` behind
 * `KateZaveruchaGoldberg::verify` (/root/reference/src/lib.rs:407-453, 639-692, src/utils.rs:224-236) -- lambdaworks-math, un-vendored.
 *
 * Everything here is the textbook form:
 *   Fp2  = Fp[i] / (i^2 + 1)
 *   Fp12 = Fp2[w] / (w^6 - xi), xi = 1 + i          (ONE extension step, six Fp2 coefficients, schoolbook product: no tower)
 *   G2 on the twist E': y^2 = x^3 + 4 xi, affine coordinates, one Fp2 inversion per group operation
 *   untwist psi(x', y') = (x' / w^2, y' / w^3); a line through psi(T) with twist-slope l' evaluated at P = (xP, yP), scaled by w^3
 *   (an element of a proper subfield, removed by the final exponentiation):   yP w^3 - l' xP w^2 + (l' xT' - yT')
 *   Miller loop over the 64 bits of |z| = 0xd201000000010000, affine, verticals omitted (they lie in Fp6)
 *   final exponentiation: f^((p^12 - 1) / r) by square-and-multiply over the 4314-bit exponent itself (no easy / hard split, no Frobenius)
 * z is negative, so this computes e(P, Q)^-1 for every pair alike; "the product of the pairings is one" is unaffected.
 *
 * Also here, for the differential test's inputs: [k]G2 on the twist, ZCash compression of a G2 point INCLUDING the sign bit that the
 * reference's decompress_g2_point ignores (/root/reference/src/compression.rs:105-139). */
#include "ref_field.h"

#define ORC_EXPORT __attribute__((visibility("default")))

typedef struct { fp_t c0, c1; } fp2_t;
typedef struct { fp2_t c[6]; } fp12_t;
typedef struct { fp2_t x, y; int inf; } g2a_t;

/* ---- Fp2 ---- */
static void fp2_add(fp2_t *o, const fp2_t *a, const fp2_t *b) { fp_add(&o->c0, &a->c0, &b->c0); fp_add(&o->c1, &a->c1, &b->c1); }
static void fp2_sub(fp2_t *o, const fp2_t *a, const fp2_t *b) { fp_sub(&o->c0, &a->c0, &b->c0); fp_sub(&o->c1, &a->c1, &b->c1); }
static void fp2_neg(fp2_t *o, const fp2_t *a) { fp_neg(&o->c0, &a->c0); fp_neg(&o->c1, &a->c1); }
static void fp2_mul(fp2_t *o, const fp2_t *a, const fp2_t *b) {
    fp_t t0, t1, t2, t3;
    fp_mul(&t0, &a->c0, &b->c0);
    fp_mul(&t1, &a->c1, &b->c1);
    fp_mul(&t2, &a->c0, &b->c1);
    fp_mul(&t3, &a->c1, &b->c0);
    fp_sub(&o->c0, &t0, &t1);
    fp_add(&o->c1, &t2, &t3);
}
static void fp2_mul_fp(fp2_t *o, const fp2_t *a, const fp_t *k) { fp_mul(&o->c0, &a->c0, k); fp_mul(&o->c1, &a->c1, k); }
static void fp2_mul_xi(fp2_t *o, const fp2_t *a) { /* (a0 + a1 i)(1 + i) */
    fp_t t0, t1;
    fp_sub(&t0, &a->c0, &a->c1);
    fp_add(&t1, &a->c0, &a->c1);
    o->c0 = t0;
    o->c1 = t1;
}
static void fp2_inv(fp2_t *o, const fp2_t *a) { /* conj(a) / (a0^2 + a1^2) */
    fp_t n, t, ni;
    fp_sqr(&n, &a->c0);
    fp_sqr(&t, &a->c1);
    fp_add(&n, &n, &t);
    fp_inv(&ni, &n);
    fp_mul(&o->c0, &a->c0, &ni);
    fp_neg(&t, &a->c1);
    fp_mul(&o->c1, &t, &ni);
}
static int fp2_is_zero(const fp2_t *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static int fp2_eq(const fp2_t *a, const fp2_t *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static void fp2_set_zero(fp2_t *o) { fp_set_zero(&o->c0); fp_set_zero(&o->c1); }

/* ---- Fp12 = Fp2[w] / (w^6 - xi) ---- */
static void fp12_set_one(fp12_t *o) {
    for (int k = 0; k < 6; k++) fp2_set_zero(&o->c[k]);
    fp_set_one(&o->c[0].c0);
}
static void fp12_mul(fp12_t *o, const fp12_t *a, const fp12_t *b) {
    fp2_t d[11], t;
    for (int k = 0; k < 11; k++) fp2_set_zero(&d[k]);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            fp2_mul(&t, &a->c[i], &b->c[j]);
            fp2_add(&d[i + j], &d[i + j], &t);
        }
    for (int k = 0; k < 5; k++) { /* w^(6 + k) = xi w^k */
        fp2_mul_xi(&t, &d[6 + k]);
        fp2_add(&d[k], &d[k], &t);
    }
    for (int k = 0; k < 6; k++) o->c[k] = d[k];
}
static int fp12_is_one(const fp12_t *a) {
    fp_t one;
    fp_set_one(&one);
    if (!fp_eq(&a->c[0].c0, &one) || !fp_is_zero(&a->c[0].c1)) return 0;
    for (int k = 1; k < 6; k++)
        if (!fp2_is_zero(&a->c[k])) return 0;
    return 1;
}

/* (p^12 - 1) / r, 68 little-endian 64-bit words: python3 -c "p=0x1a0111ea...aaab; r=0x73eda753...00000001; print(hex((p**12-1)//r))";
 * tests/test_oracle_pairing.py recomputes it and compares (orc_final_exponent). */
static const uint64_t FINAL_EXP[68] = {
    0xc0bcb9b55df57510ull, 0x25f98630e68bfb24ull, 0x4406fbc8fbd5f489ull, 0x8e2f8491d12191a0ull,
    0x3e9d71650a6f8069ull, 0x226c2f011d4cab80ull, 0x67f67c4717489119ull, 0xaf3f881bd88592d7ull,
    0x1a67e49eeed2161dull, 0xe5b78c7869aeb218ull, 0xf6539314043f7bbcull, 0x73f62537f2701aaeull,
    0xaff1c910e9622d2aull, 0x6283313492caa9d4ull, 0x2e2f3ec2bea83d19ull, 0xa4c7e79fb02faa73ull,
    0x6c49637fd7961be1ull, 0x08e88adce8817745ull, 0x35de3f7a36399917ull, 0x9c1d9f7c31759c36ull,
    0xfa9e13c24ea820b0ull, 0x3fc56947a403577dull, 0xa4c1b6dcfc5cceb7ull, 0x1bbd81367066bca6ull,
    0x0418a3ef0bc62775ull, 0x49bf9b71a9f9e010ull, 0x511291097db60b17ull, 0x498345c6e5308f1cull,
    0x6d8823b19dadd7c2ull, 0x92004cedd556952cull, 0x4c6bec3ec03ef195ull, 0x0a1fad20044ce6adull,
    0xc55d3109cd15948dull, 0x334f46c02c3f0bd0ull, 0x3b5a62eb34c05739ull, 0x724538411d1676a5ull,
    0x127a1b5ad0463434ull, 0x61a474c5c85b0129ull, 0x8dfc8e2886ef965eull, 0x96532fef459f1243ull,
    0x40ee7169cdc10412ull, 0x9c40a68eb74bb22aull, 0x25118790f4684d0bull, 0x596bc293c8d4c01full,
    0x1064837f27611212ull, 0x077ffb10bf24dde4ull, 0xc49f570bcd2b01f3ull, 0x1a0c5bf24c374693ull,
    0x350da5359bc73ab6ull, 0xd2670d93e4d7acddull, 0xd39099b86e1ab656ull, 0x19328148978e2b0dull,
    0xb113f414386b0e88ull, 0x07a0dce2630d9aa4ull, 0xa927e7bb93753318ull, 0xe347aa68ad49466full,
    0x1c0ad0d6106feaf4ull, 0xc872ee83ff3a0f0full, 0x074e43b9a660835cull, 0xc0aadff5e9cfee9aull,
    0x30698e8cc7deada9ull, 0xd1073776ab353f2cull, 0x17848517badc3a43ull, 0x7363baa13f8d14a9ull,
    0xd4977b3f7d4507d0ull, 0x496a1c0a89ee0193ull, 0xdcc825b7e1bda9c0ull, 0x0000000002ee1db5ull,
};

static void fp12_final_exp(fp12_t *o, const fp12_t *f) {
    fp12_t acc;
    fp12_set_one(&acc);
    int started = 0;
    for (int i = 68 * 64 - 1; i >= 0; i--) {
        if (started) fp12_mul(&acc, &acc, &acc);
        if ((FINAL_EXP[i / 64] >> (i % 64)) & 1) {
            fp12_mul(&acc, &acc, f);
            started = 1;
        }
    }
    *o = acc;
}

/* ---- G2 on the twist y^2 = x^3 + 4 xi, affine ---- */
static void g2_b(fp2_t *b) { /* 4 xi = 4 + 4 i */
    fp_set_u64(&b->c0, 4);
    fp_set_u64(&b->c1, 4);
}
static int g2_on_curve(const g2a_t *p) {
    if (p->inf) return 1;
    fp2_t l, r, b;
    fp2_mul(&l, &p->y, &p->y);
    fp2_mul(&r, &p->x, &p->x);
    fp2_mul(&r, &r, &p->x);
    g2_b(&b);
    fp2_add(&r, &r, &b);
    return fp2_eq(&l, &r);
}
/* slope of the tangent at t (t not of order 2) / of the chord through t and q (t.x != q.x) */
static void g2_slope_dbl(fp2_t *lam, const g2a_t *t) {
    fp2_t n, d, di;
    fp_t three;
    fp_set_u64(&three, 3);
    fp2_mul(&n, &t->x, &t->x);
    fp2_mul_fp(&n, &n, &three);
    fp2_add(&d, &t->y, &t->y);
    fp2_inv(&di, &d);
    fp2_mul(lam, &n, &di);
}
static void g2_slope_add(fp2_t *lam, const g2a_t *t, const g2a_t *q) {
    fp2_t n, d, di;
    fp2_sub(&n, &q->y, &t->y);
    fp2_sub(&d, &q->x, &t->x);
    fp2_inv(&di, &d);
    fp2_mul(lam, &n, &di);
}
static void g2_from_slope(g2a_t *o, const fp2_t *lam, const g2a_t *t, const g2a_t *q) { /* x3 = l^2 - xt - xq, y3 = l (xt - x3) - yt */
    g2a_t r;
    fp2_t t0;
    fp2_mul(&r.x, lam, lam);
    fp2_sub(&r.x, &r.x, &t->x);
    fp2_sub(&r.x, &r.x, &q->x);
    fp2_sub(&t0, &t->x, &r.x);
    fp2_mul(&r.y, lam, &t0);
    fp2_sub(&r.y, &r.y, &t->y);
    r.inf = 0;
    *o = r;
}
static void g2_double(g2a_t *o, const g2a_t *t) {
    if (t->inf || fp2_is_zero(&t->y)) { o->inf = 1; fp2_set_zero(&o->x); fp2_set_zero(&o->y); return; }
    fp2_t lam;
    g2_slope_dbl(&lam, t);
    g2_from_slope(o, &lam, t, t);
}
static void g2_add(g2a_t *o, const g2a_t *t, const g2a_t *q) {
    if (t->inf) { *o = *q; return; }
    if (q->inf) { *o = *t; return; }
    if (fp2_eq(&t->x, &q->x)) {
        if (fp2_eq(&t->y, &q->y)) { g2_double(o, t); return; }
        o->inf = 1; fp2_set_zero(&o->x); fp2_set_zero(&o->y);
        return;
    }
    fp2_t lam;
    g2_slope_add(&lam, t, q);
    g2_from_slope(o, &lam, t, q);
}
static void g2_generator(g2a_t *g) {
    static const uint8_t X0[48] = {0x02,0x4a,0xa2,0xb2,0xf0,0x8f,0x0a,0x91,0x26,0x08,0x05,0x27,0x2d,0xc5,0x10,0x51,0xc6,0xe4,0x7a,0xd4,0xfa,0x40,0x3b,0x02,
                                   0xb4,0x51,0x0b,0x64,0x7a,0xe3,0xd1,0x77,0x0b,0xac,0x03,0x26,0xa8,0x05,0xbb,0xef,0xd4,0x80,0x56,0xc8,0xc1,0x21,0xbd,0xb8};
    static const uint8_t X1[48] = {0x13,0xe0,0x2b,0x60,0x52,0x71,0x9f,0x60,0x7d,0xac,0xd3,0xa0,0x88,0x27,0x4f,0x65,0x59,0x6b,0xd0,0xd0,0x99,0x20,0xb6,0x1a,
                                   0xb5,0xda,0x61,0xbb,0xdc,0x7f,0x50,0x49,0x33,0x4c,0xf1,0x12,0x13,0x94,0x5d,0x57,0xe5,0xac,0x7d,0x05,0x5d,0x04,0x2b,0x7e};
    static const uint8_t Y0[48] = {0x0c,0xe5,0xd5,0x27,0x72,0x7d,0x6e,0x11,0x8c,0xc9,0xcd,0xc6,0xda,0x2e,0x35,0x1a,0xad,0xfd,0x9b,0xaa,0x8c,0xbd,0xd3,0xa7,
                                   0x6d,0x42,0x9a,0x69,0x51,0x60,0xd1,0x2c,0x92,0x3a,0xc9,0xcc,0x3b,0xac,0xa2,0x89,0xe1,0x93,0x54,0x86,0x08,0xb8,0x28,0x01};
    static const uint8_t Y1[48] = {0x06,0x06,0xc4,0xa0,0x2e,0xa7,0x34,0xcc,0x32,0xac,0xd2,0xb0,0x2b,0xc2,0x8b,0x99,0xcb,0x3e,0x28,0x7e,0x85,0xa7,0x63,0xaf,
                                   0x26,0x74,0x92,0xab,0x57,0x2e,0x99,0xab,0x3f,0x37,0x0d,0x27,0x5c,0xec,0x1d,0xa1,0xaa,0xa9,0x07,0x5f,0xf0,0x5f,0x79,0xbe};
    fp_from_be(&g->x.c0, X0);
    fp_from_be(&g->x.c1, X1);
    fp_from_be(&g->y.c0, Y0);
    fp_from_be(&g->y.c1, Y1);
    g->inf = 0;
}
static void g2_mul(g2a_t *o, const g2a_t *p, const uint64_t *k, int nl) { /* left-to-right double-and-add */
    g2a_t acc;
    acc.inf = 1;
    fp2_set_zero(&acc.x);
    fp2_set_zero(&acc.y);
    for (int i = nl * 64 - 1; i >= 0; i--) {
        g2_double(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) g2_add(&acc, &acc, p);
    }
    *o = acc;
}

/* ---- the pairing ---- */
/* f *= yP w^3 - lam xP w^2 + (lam xT - yT) */
static void mul_by_line(fp12_t *f, const fp2_t *lam, const g2a_t *t, const fp_t *xp, const fp_t *yp) {
    fp12_t l;
    fp2_t t0;
    for (int k = 0; k < 6; k++) fp2_set_zero(&l.c[k]);
    fp2_mul(&t0, lam, &t->x);
    fp2_sub(&l.c[0], &t0, &t->y);
    fp2_mul_fp(&t0, lam, xp);
    fp2_neg(&l.c[2], &t0);
    l.c[3].c0 = *yp;
    fp12_mul(f, f, &l);
}
/* f *= f_{|z|, Q}(P) (Miller function of the loop parameter's absolute value); P, Q finite */
static void miller_into(fp12_t *f, const fp_t *xp, const fp_t *yp, const g2a_t *q) {
    const uint64_t Z = 0xd201000000010000ull;
    fp12_t m;
    fp12_set_one(&m);
    g2a_t t = *q;
    for (int i = 62; i >= 0; i--) {
        fp2_t lam;
        fp12_mul(&m, &m, &m);
        g2_slope_dbl(&lam, &t);
        mul_by_line(&m, &lam, &t, xp, yp);
        g2_from_slope(&t, &lam, &t, &t);
        if ((Z >> i) & 1) {
            g2_slope_add(&lam, &t, q);
            mul_by_line(&m, &lam, &t, xp, yp);
            g2_from_slope(&t, &lam, &t, q);
        }
    }
    fp12_mul(f, f, &m);
}

/* prod_i e(P_i, Q_i) == 1 ?  g1: n x (x | y), 48-byte big-endian canonical each, all 96 bytes zero = the point at infinity;
 * g2: n x (x.c0 | x.c1 | y.c0 | y.c1), all 192 bytes zero = infinity. Returns 0, or 1 when a point is not on its curve / a coordinate
 * is not canonical (subgroup membership is NOT checked: the caller builds its inputs as multiples of the generators). */
ORC_EXPORT int orc_pairing_product_is_one(int *ok, const uint8_t *g1, const uint8_t *g2, int n) {
    fp12_t f, e;
    fp12_set_one(&f);
    *ok = 0;
    for (int i = 0; i < n; i++) {
        const uint8_t *a = g1 + 96 * i, *b = g2 + 192 * i;
        int z1 = 1, z2 = 1;
        for (int k = 0; k < 96; k++) z1 &= a[k] == 0;
        for (int k = 0; k < 192; k++) z2 &= b[k] == 0;
        uint64_t raw[6];
        for (int c = 0; c < 2 && !z1; c++) {
            fp_raw_from_be(raw, a + 48 * c);
            if (fp_raw_geq(raw, fp_MOD)) return 1;
        }
        for (int c = 0; c < 4 && !z2; c++) {
            fp_raw_from_be(raw, b + 48 * c);
            if (fp_raw_geq(raw, fp_MOD)) return 1;
        }
        fp_t xp, yp;
        g2a_t q;
        q.inf = z2;
        if (!z1) {
            fp_from_be(&xp, a);
            fp_from_be(&yp, a + 48);
            if (!g1_on_curve_affine(&xp, &yp)) return 1;
        }
        if (!z2) {
            fp_from_be(&q.x.c0, b);
            fp_from_be(&q.x.c1, b + 48);
            fp_from_be(&q.y.c0, b + 96);
            fp_from_be(&q.y.c1, b + 144);
            if (!g2_on_curve(&q)) return 1;
        }
        if (z1 || z2) continue; /* e(O, Q) = e(P, O) = 1 */
        miller_into(&f, &xp, &yp, &q);
    }
    fp12_final_exp(&e, &f);
    *ok = fp12_is_one(&e);
    return 0;
}

/* [k]G2, k = 32 bytes big-endian (any integer): affine x.c0 | x.c1 | y.c0 | y.c1 (192 bytes, zero = infinity) */
ORC_EXPORT void orc_g2_generator_mul(uint8_t out192[192], const uint8_t k_be[32]) {
    uint64_t k[4];
    fr_raw_from_be(k, k_be);
    g2a_t g, r;
    g2_generator(&g);
    g2_mul(&r, &g, k, 4);
    memset(out192, 0, 192);
    if (r.inf) return;
    fp_to_be(out192, &r.x.c0);
    fp_to_be(out192 + 48, &r.x.c1);
    fp_to_be(out192 + 96, &r.y.c0);
    fp_to_be(out192 + 144, &r.y.c1);
}

/* ZCash compression of an affine G2 point (192 bytes as above): x.c1 | x.c0, bit 7 = compressed, bit 6 = infinity, bit 5 = y is the
 * lexicographically larger of (y, -y), c1 compared first (the bit /root/reference/src/compression.rs:105-139 does not read) */
ORC_EXPORT void orc_g2_compress(uint8_t out96[96], const uint8_t in192[192]) {
    int z = 1;
    for (int k = 0; k < 192; k++) z &= in192[k] == 0;
    memset(out96, 0, 96);
    if (z) { out96[0] = 0xc0; return; }
    memcpy(out96, in192 + 48, 48);
    memcpy(out96 + 48, in192, 48);
    out96[0] |= 0x80;
    fp_t y0, y1, n;
    fp_from_be(&y0, in192 + 96);
    fp_from_be(&y1, in192 + 144);
    const fp_t *c = fp_is_zero(&y1) ? &y0 : &y1;
    uint64_t a[6], b[6];
    fp_neg(&n, c);
    fp_to_raw(a, c);
    fp_to_raw(b, &n);
    if (!fp_raw_geq(b, a)) out96[0] |= 0x20; /* -y < y */
}

/* is the affine G2 point (192 bytes) on the twist? and the generator's coordinates, for the test's own checks */
ORC_EXPORT int orc_g2_on_curve(const uint8_t in192[192]) {
    g2a_t q;
    q.inf = 0;
    fp_from_be(&q.x.c0, in192);
    fp_from_be(&q.x.c1, in192 + 48);
    fp_from_be(&q.y.c0, in192 + 96);
    fp_from_be(&q.y.c1, in192 + 144);
    return g2_on_curve(&q);
}

/* the final exponent (p^12 - 1) / r as 544 big-endian bytes (tests recompute it) */
ORC_EXPORT void orc_final_exponent(uint8_t out[544]) {
    for (int i = 0; i < 68; i++)
        for (int k = 0; k < 8; k++) out[543 - 8 * i - k] = (uint8_t)(FINAL_EXP[i] >> (8 * k));
}
