/*
 * oracle/ref_g1.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). See ref_field.h header.
 *
 * Constants and the G1 group law. The group law restates lambdaworks-math's
 * ShortWeierstrassProjectivePoint::{operate_with, operate_with_self, neg, to_affine}
 * (un-vendored dependency; call sites /root/reference/src/lib.rs:664-688,
 * /root/reference/src/compression.rs:22-27,42,98): classical homogeneous-projective
 * addition with explicit neutral / doubling / inverse branches, a = 0, b = 4.
 */
#include "ref_field.h"

const uint64_t fp_MOD[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                            0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
const uint64_t fp_R1[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                           0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};
const uint64_t fp_R2[6] = {0xf4df1f341c341746ull, 0x0a76e6a609d104f1ull, 0x8de5476c4c95b6d5ull,
                           0x67eb88a9939d83c0ull, 0x9a793e85b519952dull, 0x11988fe592cae3aaull};
const uint64_t fp_INV = 0x89f3fffcfffcfffdull;

const uint64_t fr_MOD[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull,
                            0x73eda753299d7d48ull};
const uint64_t fr_R1[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull,
                           0x1824b159acc5056full};
const uint64_t fr_R2[4] = {0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full,
                           0x0748d9d99f59ff11ull};
const uint64_t fr_INV = 0xfffffffeffffffffull;

/* generator, canonical big-endian hex split into raw little-endian limbs
 * (/root/reference/tests/lib_test.rs:276 pins its compressed form) */
static const uint64_t G1_GEN_X[6] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull,
                                     0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull};
static const uint64_t G1_GEN_Y[6] = {0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull,
                                     0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};

void g1_set_neutral(g1_t *o) {
    fp_set_zero(&o->x);
    fp_set_one(&o->y);
    fp_set_zero(&o->z);
}

int g1_is_neutral(const g1_t *a) { return fp_is_zero(&a->z); }

void g1_from_affine(g1_t *o, const fp_t *x, const fp_t *y) {
    o->x = *x;
    o->y = *y;
    fp_set_one(&o->z);
}

int g1_on_curve_affine(const fp_t *x, const fp_t *y) {
    fp_t l, r, four;
    fp_sqr(&l, y);
    fp_sqr(&r, x);
    fp_mul(&r, &r, x);
    fp_set_u64(&four, 4);
    fp_add(&r, &r, &four);
    return fp_eq(&l, &r);
}

void g1_generator(g1_t *o) {
    fp_t x, y;
    fp_from_raw(&x, G1_GEN_X);
    fp_from_raw(&y, G1_GEN_Y);
    g1_from_affine(o, &x, &y);
}

void g1_to_affine(fp_t *x, fp_t *y, const g1_t *a) {
    fp_t zi;
    fp_inv(&zi, &a->z);
    fp_mul(x, &a->x, &zi);
    fp_mul(y, &a->y, &zi);
}

void g1_neg(g1_t *o, const g1_t *p) {
    o->x = p->x;
    fp_neg(&o->y, &p->y);
    o->z = p->z;
}

int g1_eq(const g1_t *a, const g1_t *b) {
    int na = g1_is_neutral(a), nb = g1_is_neutral(b);
    if (na || nb) return na && nb;
    fp_t l, r;
    fp_mul(&l, &a->x, &b->z);
    fp_mul(&r, &b->x, &a->z);
    if (!fp_eq(&l, &r)) return 0;
    fp_mul(&l, &a->y, &b->z);
    fp_mul(&r, &b->y, &a->z);
    return fp_eq(&l, &r);
}

/* doubling, homogeneous projective, a = 0:
 *   w = 3 X^2 ; s = Y Z ; b = X Y s ; h = w^2 - 8 b
 *   X' = 2 h s ; Y' = w (4 b - h) - 8 Y^2 s^2 ; Z' = 8 s^3 */
void g1_double(g1_t *o, const g1_t *p) {
    if (g1_is_neutral(p) || fp_is_zero(&p->y)) {
        g1_set_neutral(o);
        return;
    }
    fp_t w, s, b, h, t, t2, ss, x3, y3, z3;
    fp_sqr(&t, &p->x);
    fp_add(&w, &t, &t);
    fp_add(&w, &w, &t); /* 3 X^2 */
    fp_mul(&s, &p->y, &p->z);
    fp_mul(&b, &p->x, &p->y);
    fp_mul(&b, &b, &s);
    fp_sqr(&h, &w);
    fp_add(&t, &b, &b);
    fp_add(&t, &t, &t); /* 4b */
    fp_add(&t2, &t, &t); /* 8b */
    fp_sub(&h, &h, &t2);
    fp_mul(&x3, &h, &s);
    fp_add(&x3, &x3, &x3);
    fp_sub(&t, &t, &h); /* 4b - h */
    fp_mul(&y3, &w, &t);
    fp_sqr(&ss, &s);
    fp_sqr(&t, &p->y);
    fp_mul(&t, &t, &ss);
    fp_add(&t, &t, &t);
    fp_add(&t, &t, &t);
    fp_add(&t, &t, &t); /* 8 Y^2 s^2 */
    fp_sub(&y3, &y3, &t);
    fp_mul(&z3, &ss, &s);
    fp_add(&z3, &z3, &z3);
    fp_add(&z3, &z3, &z3);
    fp_add(&z3, &z3, &z3);
    o->x = x3;
    o->y = y3;
    o->z = z3;
}

/* addition:
 *   u1 = Yq Zp ; u2 = Yp Zq ; v1 = Xq Zp ; v2 = Xp Zq
 *   v1 == v2: same x  -> double if u1 == u2 (and Yp != 0) else neutral
 *   u = u1 - u2 ; v = v1 - v2 ; w = Zp Zq ; a = u^2 w - v^3 - 2 v^2 v2
 *   X' = v a ; Y' = u (v^2 v2 - a) - v^3 u2 ; Z' = v^3 w */
void g1_add(g1_t *o, const g1_t *p, const g1_t *q) {
    if (g1_is_neutral(q)) {
        *o = *p;
        return;
    }
    if (g1_is_neutral(p)) {
        *o = *q;
        return;
    }
    fp_t u1, u2, v1, v2;
    fp_mul(&u1, &q->y, &p->z);
    fp_mul(&u2, &p->y, &q->z);
    fp_mul(&v1, &q->x, &p->z);
    fp_mul(&v2, &p->x, &q->z);
    if (fp_eq(&v1, &v2)) {
        if (!fp_eq(&u1, &u2) || fp_is_zero(&p->y)) {
            g1_set_neutral(o);
        } else {
            g1_double(o, p);
        }
        return;
    }
    fp_t u, v, w, vv, vvv, a, t, x3, y3, z3;
    fp_sub(&u, &u1, &u2);
    fp_sub(&v, &v1, &v2);
    fp_mul(&w, &p->z, &q->z);
    fp_sqr(&vv, &v);
    fp_mul(&vvv, &vv, &v);
    fp_mul(&t, &vv, &v2); /* v^2 v2 */
    fp_sqr(&a, &u);
    fp_mul(&a, &a, &w);
    fp_sub(&a, &a, &vvv);
    fp_sub(&a, &a, &t);
    fp_sub(&a, &a, &t);
    fp_mul(&x3, &v, &a);
    fp_sub(&t, &t, &a);
    fp_mul(&y3, &u, &t);
    fp_mul(&t, &vvv, &u2);
    fp_sub(&y3, &y3, &t);
    fp_mul(&z3, &vvv, &w);
    o->x = x3;
    o->y = y3;
    o->z = z3;
}

void g1_mul_raw(g1_t *o, const g1_t *p, const uint64_t *k, int nlimbs) {
    g1_t acc;
    g1_set_neutral(&acc);
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        g1_double(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) g1_add(&acc, &acc, p);
    }
    *o = acc;
}
