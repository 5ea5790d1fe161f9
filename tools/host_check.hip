// host_check.hip -- the product's own arithmetic (the LWK_HD sources the kernels are built from), compiled for the HOST
// and cross-checked against itself along independent routes. Pure host code: runs without a GPU (tests/test_capi_cpu.py
// builds and runs it under -m "not gpu").
//   1. division-step inversion (f29_inv, inv(Fp)) == Fermat's a^(p-2), and a * a^-1 == 1
//   2. hot-loop field (14 x 28-bit limbs, lazy value and limb bounds): a*b, a^2, fused a*b - c*d, sums/differences == the 12 x 32-bit
//      CIOS field on the same values
//   3. hot-loop group law: [k]G by double-and-add over F29 (XYZZ, mixed additions) == [k]G over the CIOS field, on
//      compressed bytes; [r]G = O; [r-1]G = -G; P + P through the addition formula's doubling branch; P + (-P) = O
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -I lambdaworks_kzg_amd/csrc tools/host_check.hip -o /tmp/host_check
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include "g1.cuh"
using namespace lwk;

static uint64_t sm(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static bool same(const Fp &a, const Fp &b) {
    for (int i = 0; i < 12; i++)
        if (a.l[i] != b.l[i]) return false;
    return true;
}

static G1Xyzz29 mul29(const G1Affine29 &p, const uint32_t k[8]) {
    G1Xyzz29 acc = G1Xyzz29::infinity();
    for (int bit = 255; bit >= 0; bit--) {
        acc = xyzz_dbl(acc);
        if ((k[bit >> 5] >> (bit & 31)) & 1) acc = xyzz_madd(acc, p.x, p.y);
    }
    return acc;
}

int main() {
    uint64_t seed = 12345;
    int bad = 0;
    // ---- 1. inversion
    for (int k = 0; k < 2000; k++) {
        uint32_t raw[12];
        for (int i = 0; i < 12; i++) raw[i] = (uint32_t)sm(seed);
        raw[11] &= 0x0fffffffu;  // < 2^380 < p
        if (k == 0) for (int i = 0; i < 12; i++) raw[i] = 0;
        if (k == 1) for (int i = 0; i < 12; i++) raw[i] = i == 0;
        if (k == 2) { for (int i = 0; i < 12; i++) raw[i] = FpParams::MOD[i]; raw[0] -= 1; }
        F29<2> a = f29_from_raw32(raw);
        F29<2> i1 = f29_inv(a), i2 = f29_inv_fermat(a);
        uint32_t r1[12], r2[12], pr[12];
        f29_to_raw32(r1, i1);
        f29_to_raw32(r2, i2);
        f29_to_raw32(pr, a * i1);
        bool ok = true;
        for (int i = 0; i < 12; i++) ok &= r1[i] == r2[i] && pr[i] == ((i == 0 && k != 0) ? 1u : 0u);
        Fp fa = f29_to_fp(a);
        ok &= same(inv(fa), inv_fermat(fa));
        if (!ok && bad++ < 5) printf("inversion mismatch at case %d\n", k);
    }
    // ---- 2. field operations, lazy bounds included
    for (int k = 0; k < 2000; k++) {
        uint32_t ra[12], rb[12], rc[12], rd[12];
        for (int i = 0; i < 12; i++) { ra[i] = (uint32_t)sm(seed); rb[i] = (uint32_t)sm(seed); rc[i] = (uint32_t)sm(seed); rd[i] = (uint32_t)sm(seed); }
        ra[11] &= 0x0fffffffu; rb[11] &= 0x0fffffffu; rc[11] &= 0x0fffffffu; rd[11] &= 0x0fffffffu;
        if (k == 0) { memset(ra, 0, sizeof ra); memset(rc, 0, sizeof rc); }
        if (k == 1) { for (int i = 0; i < 12; i++) ra[i] = rb[i] = FpParams::MOD[i]; ra[0] -= 1; rb[0] -= 1; }
        F29<2> a = f29_from_raw32(ra), b = f29_from_raw32(rb), c = f29_from_raw32(rc), d = f29_from_raw32(rd);
        Fp fa = f29_to_fp(a), fb = f29_to_fp(b), fc = f29_to_fp(c), fd = f29_to_fp(d);
        bool ok = same(f29_to_fp(a * b), fa * fb) && same(f29_to_fp(sqr(a)), sqr(fa)) &&
                  same(f29_to_fp(mul_sub(a, b, c, d)), fa * fb - fc * fd) && same(f29_to_fp(a + b), fa + fb) &&
                  same(f29_to_fp(a - b), fa - fb) && same(f29_to_fp((a - b) * (c + d + a)), (fa - fb) * (fc + fd + fa)) &&
                  same(f29_to_fp(sqr(normed(a - b - c))), sqr(fa - fb - fc)) && same(f29_to_fp(neg(a)), neg(fa)) &&
                  same(f29_to_fp(cneg(a, true)), neg(fa)) && same(f29_to_fp(cneg(a, false)), fa) &&
                  (a - a).is_zero() && ((a * b) - (b * a)).is_zero() && !(a - b).is_zero() == !same(fa, fb);
        if (!ok && bad++ < 5) printf("field mismatch at case %d\n", k);
    }
    // ---- 3. group law on the generator (compressed form from the reference's tests, tests/lib_test.rs:262-291)
    const char *ghex = "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb";
    uint8_t gb[48];
    for (int i = 0; i < 48; i++) {
        unsigned v;
        sscanf(ghex + 2 * i, "%2x", &v);
        gb[i] = (uint8_t)v;
    }
    G1Affine g;
    if (g1_decompress_nocheck(g, gb) != 0) { printf("generator does not decompress\n"); return 1; }
    const G1Affine29 g29 = affine_to_29(g);
    for (int k = 0; k < 24; k++) {
        uint32_t sc[8];
        for (int i = 0; i < 8; i++) sc[i] = (uint32_t)sm(seed);
        sc[7] &= 0x3fffffffu;
        if (k == 0) { memset(sc, 0, sizeof sc); sc[0] = 1; }
        if (k == 1) { memset(sc, 0, sizeof sc); sc[0] = 2; }
        if (k == 2) for (int i = 0; i < 8; i++) sc[i] = FrParams::MOD[i];                 // [r]G = O
        if (k == 3) { for (int i = 0; i < 8; i++) sc[i] = FrParams::MOD[i]; sc[0] -= 1; }  // [r-1]G = -G
        uint8_t o1[48], o2[48];
        g1_compress(o1, xyzz_mul_affine<8>(g, sc));
        g1_compress(o2, mul29(g29, sc));
        bool ok = memcmp(o1, o2, 48) == 0;
        if (k == 0) ok &= memcmp(o1, gb, 48) == 0;
        if (k == 2) ok &= o1[0] == 0xc0;
        if (k == 3) { uint8_t ng[48]; memcpy(ng, gb, 48); ng[0] ^= 0x20; ok &= memcmp(o1, ng, 48) == 0; }
        if (!ok && bad++ < 5) printf("group mismatch at case %d\n", k);
    }
    {
        // branches of the complete addition: doubling through madd, and P + (-P)
        G1Xyzz29 p = G1Xyzz29::from_affine(g29.x, g29.y);
        uint8_t o1[48], o2[48];
        g1_compress(o1, xyzz_madd(p, g29.x, g29.y));
        g1_compress(o2, xyzz_dbl(p));
        bool ok = memcmp(o1, o2, 48) == 0 && xyzz_madd(p, g29.x, neg(g29.y) * F29<1>::one()).is_inf() &&
                  xyzz_add(p, p).is_inf() == false && xyzz_add(xyzz_dbl(p), p).is_inf() == false;
        uint8_t o3[48], o4[48];
        uint32_t three[8] = {3};
        g1_compress(o3, xyzz_add(xyzz_dbl(p), p));
        g1_compress(o4, xyzz_mul_affine<8>(g, three));
        ok &= memcmp(o3, o4, 48) == 0;
        if (!ok && bad++ < 5) printf("addition branches mismatch\n");
    }
    printf(bad ? "FAIL %d\n" : "ok: inversion, field and group-law cross-checks agree\n", bad);
    return bad != 0;
}
