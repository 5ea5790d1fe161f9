/*
 * oracle/ref_field.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of the BLS12-381 arithmetic that lambdaworks_kzg gets from
 * the un-vendored, un-pinned git dependency `lambdaworks-math`
 * (/root/reference/Cargo.toml:15-16; no Cargo.lock, /root/reference/.gitignore:8).
 * The upstream source is NOT under /root/reference, so what is restated here is the
 * published algorithm (Montgomery CIOS, homogeneous-projective short-Weierstrass
 * group law with explicit doubling/inverse/neutral branches, double-and-add) and
 * parity is anchored on the reference's own call sites, tests and golden vectors
 * (see oracle/README.md for the pin list).
 *
 * Nothing in the product (lambdaworks_kzg_amd/) may include, link or call this file.
 * 64-bit limbs + unsigned __int128: deliberately a different implementation from the
 * product's 32-bit-limb device code so the two cannot share a bug.
 */
#ifndef ORACLE_REF_FIELD_H
#define ORACLE_REF_FIELD_H

#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ generic */

#define DEFINE_FIELD(PFX, NL, TYPE)                                                      \
    typedef struct { uint64_t l[NL]; } TYPE;                                             \
    extern const uint64_t PFX##_MOD[NL];                                                 \
    extern const uint64_t PFX##_R1[NL];                                                  \
    extern const uint64_t PFX##_R2[NL];                                                  \
    extern const uint64_t PFX##_INV;                                                     \
    static inline int PFX##_raw_geq(const uint64_t *a, const uint64_t *b) {              \
        for (int i = NL - 1; i >= 0; i--) {                                              \
            if (a[i] > b[i]) return 1;                                                   \
            if (a[i] < b[i]) return 0;                                                   \
        }                                                                                \
        return 1;                                                                        \
    }                                                                                    \
    static inline uint64_t PFX##_raw_sub(uint64_t *o, const uint64_t *a, const uint64_t *b) { \
        uint64_t br = 0;                                                                 \
        for (int i = 0; i < NL; i++) {                                                   \
            u128 d = (u128)a[i] - b[i] - br;                                             \
            o[i] = (uint64_t)d;                                                          \
            br = (uint64_t)(d >> 64) & 1;                                                \
        }                                                                                \
        return br;                                                                       \
    }                                                                                    \
    static inline uint64_t PFX##_raw_add(uint64_t *o, const uint64_t *a, const uint64_t *b) { \
        uint64_t c = 0;                                                                  \
        for (int i = 0; i < NL; i++) {                                                   \
            u128 s = (u128)a[i] + b[i] + c;                                              \
            o[i] = (uint64_t)s;                                                          \
            c = (uint64_t)(s >> 64);                                                     \
        }                                                                                \
        return c;                                                                        \
    }                                                                                    \
    static inline void PFX##_add(TYPE *o, const TYPE *a, const TYPE *b) {                \
        uint64_t t[NL];                                                                  \
        uint64_t c = PFX##_raw_add(t, a->l, b->l);                                       \
        if (c || PFX##_raw_geq(t, PFX##_MOD)) PFX##_raw_sub(t, t, PFX##_MOD);            \
        memcpy(o->l, t, sizeof t);                                                       \
    }                                                                                    \
    static inline void PFX##_sub(TYPE *o, const TYPE *a, const TYPE *b) {                \
        uint64_t t[NL];                                                                  \
        if (PFX##_raw_sub(t, a->l, b->l)) PFX##_raw_add(t, t, PFX##_MOD);                \
        memcpy(o->l, t, sizeof t);                                                       \
    }                                                                                    \
    static inline int PFX##_is_zero(const TYPE *a) {                                     \
        uint64_t x = 0;                                                                  \
        for (int i = 0; i < NL; i++) x |= a->l[i];                                       \
        return x == 0;                                                                   \
    }                                                                                    \
    static inline int PFX##_eq(const TYPE *a, const TYPE *b) {                           \
        return memcmp(a->l, b->l, sizeof a->l) == 0;                                     \
    }                                                                                    \
    static inline void PFX##_neg(TYPE *o, const TYPE *a) {                               \
        if (PFX##_is_zero(a)) { *o = *a; return; }                                       \
        PFX##_raw_sub(o->l, PFX##_MOD, a->l);                                            \
    }                                                                                    \
    /* Montgomery product, CIOS (Koc-Acar-Kaliski), R = 2^(64*NL) */                     \
    static inline void PFX##_mul(TYPE *o, const TYPE *a, const TYPE *b) {                \
        uint64_t t[NL + 2];                                                              \
        memset(t, 0, sizeof t);                                                          \
        for (int i = 0; i < NL; i++) {                                                   \
            uint64_t c = 0;                                                              \
            for (int j = 0; j < NL; j++) {                                               \
                u128 s = (u128)a->l[j] * b->l[i] + t[j] + c;                             \
                t[j] = (uint64_t)s;                                                      \
                c = (uint64_t)(s >> 64);                                                 \
            }                                                                            \
            u128 s = (u128)t[NL] + c;                                                    \
            t[NL] = (uint64_t)s;                                                         \
            t[NL + 1] = (uint64_t)(s >> 64);                                             \
            uint64_t m = t[0] * PFX##_INV;                                               \
            s = (u128)m * PFX##_MOD[0] + t[0];                                           \
            c = (uint64_t)(s >> 64);                                                     \
            for (int j = 1; j < NL; j++) {                                               \
                s = (u128)m * PFX##_MOD[j] + t[j] + c;                                   \
                t[j - 1] = (uint64_t)s;                                                  \
                c = (uint64_t)(s >> 64);                                                 \
            }                                                                            \
            s = (u128)t[NL] + c;                                                         \
            t[NL - 1] = (uint64_t)s;                                                     \
            t[NL] = t[NL + 1] + (uint64_t)(s >> 64);                                     \
        }                                                                                \
        if (t[NL] || PFX##_raw_geq(t, PFX##_MOD)) PFX##_raw_sub(t, t, PFX##_MOD);        \
        memcpy(o->l, t, sizeof o->l);                                                    \
    }                                                                                    \
    static inline void PFX##_sqr(TYPE *o, const TYPE *a) { PFX##_mul(o, a, a); }         \
    /* canonical little-endian-limb integer -> Montgomery form (reduces if >= modulus,  \
       matching what from_bytes_be is believed to do upstream) */     \
    static inline void PFX##_from_raw(TYPE *o, const uint64_t *raw) {                    \
        TYPE t, r2;                                                                      \
        memcpy(t.l, raw, sizeof t.l);                                                    \
        while (PFX##_raw_geq(t.l, PFX##_MOD)) PFX##_raw_sub(t.l, t.l, PFX##_MOD);        \
        memcpy(r2.l, PFX##_R2, sizeof r2.l);                                             \
        PFX##_mul(o, &t, &r2);                                                           \
    }                                                                                    \
    static inline void PFX##_to_raw(uint64_t *raw, const TYPE *a) {                      \
        TYPE one, t;                                                                     \
        memset(one.l, 0, sizeof one.l);                                                  \
        one.l[0] = 1;                                                                    \
        PFX##_mul(&t, a, &one);                                                          \
        memcpy(raw, t.l, sizeof t.l);                                                    \
    }                                                                                    \
    static inline void PFX##_set_one(TYPE *o) { memcpy(o->l, PFX##_R1, sizeof o->l); }   \
    static inline void PFX##_set_zero(TYPE *o) { memset(o->l, 0, sizeof o->l); }         \
    static inline void PFX##_set_u64(TYPE *o, uint64_t v) {                              \
        uint64_t raw[NL];                                                                \
        memset(raw, 0, sizeof raw);                                                      \
        raw[0] = v;                                                                      \
        PFX##_from_raw(o, raw);                                                          \
    }                                                                                    \
    /* a^e, e given as NE little-endian 64-bit limbs; left-to-right square-and-multiply */\
    static inline void PFX##_pow(TYPE *o, const TYPE *a, const uint64_t *e, int ne) {    \
        TYPE acc;                                                                        \
        PFX##_set_one(&acc);                                                             \
        int started = 0;                                                                 \
        for (int i = ne * 64 - 1; i >= 0; i--) {                                         \
            if (started) PFX##_sqr(&acc, &acc);                                          \
            if ((e[i / 64] >> (i % 64)) & 1) {                                           \
                PFX##_mul(&acc, &acc, a);                                                \
                started = 1;                                                             \
            }                                                                            \
        }                                                                                \
        *o = acc;                                                                        \
    }                                                                                    \
    /* inverse by Fermat: a^(m-2) */                                                     \
    static inline void PFX##_inv(TYPE *o, const TYPE *a) {                               \
        uint64_t e[NL];                                                                  \
        uint64_t two[NL];                                                                \
        memset(two, 0, sizeof two);                                                      \
        two[0] = 2;                                                                      \
        PFX##_raw_sub(e, PFX##_MOD, two);                                                \
        PFX##_pow(o, a, e, NL);                                                          \
    }                                                                                    \
    /* big-endian bytes (8*NL of them) <-> canonical raw limbs */                        \
    static inline void PFX##_raw_from_be(uint64_t *raw, const uint8_t *b) {              \
        for (int i = 0; i < NL; i++) {                                                   \
            uint64_t v = 0;                                                              \
            for (int k = 0; k < 8; k++) v = (v << 8) | b[(NL - 1 - i) * 8 + k];          \
            raw[i] = v;                                                                  \
        }                                                                                \
    }                                                                                    \
    static inline void PFX##_raw_to_be(uint8_t *b, const uint64_t *raw) {                \
        for (int i = 0; i < NL; i++)                                                     \
            for (int k = 0; k < 8; k++)                                                  \
                b[(NL - 1 - i) * 8 + k] = (uint8_t)(raw[i] >> (56 - 8 * k));             \
    }                                                                                    \
    static inline void PFX##_from_be(TYPE *o, const uint8_t *b) {                        \
        uint64_t raw[NL];                                                                \
        PFX##_raw_from_be(raw, b);                                                       \
        PFX##_from_raw(o, raw);                                                          \
    }                                                                                    \
    static inline void PFX##_to_be(uint8_t *b, const TYPE *a) {                          \
        uint64_t raw[NL];                                                                \
        PFX##_to_raw(raw, a);                                                            \
        PFX##_raw_to_be(b, raw);                                                         \
    }

DEFINE_FIELD(fp, 6, fp_t)
DEFINE_FIELD(fr, 4, fr_t)

/* ------------------------------------------------------------------ G1 */

/* Homogeneous projective (X:Y:Z), x = X/Z, y = Y/Z; neutral = (0:1:0).
 * Mirrors lambdaworks' ShortWeierstrassProjectivePoint (SURVEY Appendix C;
 * call sites /root/reference/src/lib.rs:664-688, src/compression.rs:25,42,98). */
typedef struct { fp_t x, y, z; } g1_t;

void g1_set_neutral(g1_t *o);
int g1_is_neutral(const g1_t *a);
void g1_from_affine(g1_t *o, const fp_t *x, const fp_t *y);
int g1_on_curve_affine(const fp_t *x, const fp_t *y);
void g1_to_affine(fp_t *x, fp_t *y, const g1_t *a); /* a must not be neutral */
void g1_add(g1_t *o, const g1_t *p, const g1_t *q);   /* operate_with */
void g1_double(g1_t *o, const g1_t *p);
void g1_neg(g1_t *o, const g1_t *p);
int g1_eq(const g1_t *a, const g1_t *b);              /* cross-multiplied compare */
/* operate_with_self: left-to-right double-and-add over raw little-endian limbs */
void g1_mul_raw(g1_t *o, const g1_t *p, const uint64_t *k, int nlimbs);
void g1_generator(g1_t *o);

#endif
