// host_check.hip -- the product's own arithmetic (the LWK_HD sources the kernels are built from), compiled for the HOST
// and cross-checked against itself along independent routes. Pure host code: runs without a GPU (tests/test_capi_cpu.py
// builds and runs it under -m "not gpu").
//   1. division-step inversion (f29_inv, inv(Fp)) == Fermat's a^(p-2), and a * a^-1 == 1
//   2. hot-loop field (14 x 28-bit limbs, lazy value and limb bounds): a*b, a^2, fused a*b - c*d, sums/differences == the 12 x 32-bit
//      CIOS field on the same values
//   3. hot-loop group law: [k]G by double-and-add over F29 (XYZZ, mixed additions) == [k]G over the CIOS field, on
//      compressed bytes; [r]G = O; [r-1]G = -G; P + P through the addition formula's doubling branch; P + (-P) = O
//   4. the transform's scalar field (fr28.cuh: 10 lazy limbs of 28 bits): products, butterflies and a whole 4096-point
//      transform in the kernel's stage order (one carry ripple at stage 6, nothing else reduced) == the 8 x 32-bit CIOS field
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -I lambdaworks_kzg_amd/csrc tools/host_check.hip -o /tmp/host_check
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include "g1.cuh"
#include "fr28.cuh"
#include <vector>
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
    for (int k = 0; k < 2000; k++) {   // the scalar field's division-step inversion (9 limbs of 30 bits) against Fermat
        uint32_t raw[8];
        for (int i = 0; i < 8; i++) raw[i] = (uint32_t)sm(seed);
        raw[7] &= 0x3fffffffu;  // < 2^254 < r
        if (k == 0) for (int i = 0; i < 8; i++) raw[i] = 0;
        if (k == 1) for (int i = 0; i < 8; i++) raw[i] = i == 0;
        if (k == 2) { for (int i = 0; i < 8; i++) raw[i] = FrParams::MOD[i]; raw[0] -= 1; }
        if (k == 3) { for (int i = 0; i < 8; i++) raw[i] = 0; raw[0] = 2; }
        if (k >= 4 && k < 260) { for (int i = 0; i < 8; i++) raw[i] = 0; raw[(k - 4) >> 5] = 1u << ((k - 4) & 31); if (k - 4 >= 254) raw[7] = 1; }
        const Fr a = fe_from_raw<FrParams>(raw);
        const Fr i1 = inv_divsteps(a), i2 = inv(a);
        bool ok = true;
        for (int i = 0; i < 8; i++) ok &= i1.l[i] == i2.l[i];
        if (!ok && bad++ < 5) printf("Fr inversion mismatch at case %d\n", k);
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
    // ---- 4. fr28: the NTT's arithmetic against the CIOS scalar field
    {
        auto same_fr = [](const Fr &a, const Fr &b) {
            for (int i = 0; i < 8; i++)
                if (a.l[i] != b.l[i]) return false;
            return true;
        };
        auto rand_fr = [&]() {
            uint32_t raw[8];
            for (int i = 0; i < 8; i++) raw[i] = (uint32_t)sm(seed);
            raw[7] &= 0x3fffffffu;  // < 2^254 < r
            return fe_from_raw<FrParams>(raw);
        };
        for (int k = 0; k < 2000; k++) {
            Fr x = rand_fr(), y = rand_fr(), z = rand_fr();
            if (k == 0) x = Fr::zero();
            if (k == 1) { uint32_t raw[8]; for (int i = 0; i < 8; i++) raw[i] = FrParams::MOD[i]; raw[0] -= 1; x = y = fe_from_raw<FrParams>(raw); }
            Fr28 X = fr28_from_mont256(x), Y = fr28_canonical(fr28_from_mont256(y)), Z = fr28_from_mont256(z);
            uint32_t w1[8], w2[8];
            fr28_to_raw_scaled(w1, fr28_mul(X, Y));
            uint32_t ninv_raw[8] = {0x00100001u, 0x400fffffu, 0xbfce5c19u, 0xd3686828u, 0x89213de7u, 0x5eb6a46au, 0xb46ae370u, 0x73e66878u};
            fe_to_raw<FrParams>(w2, x * y * fe_from_raw<FrParams>(ninv_raw));
            bool ok = same_fr(fr28_to_mont256(fr28_mul(X, Y)), x * y) && memcmp(w1, w2, 32) == 0 &&
                      same_fr(fr28_to_mont256(fr28_add(Z, fr28_mul(X, Y))), z + x * y) &&
                      same_fr(fr28_to_mont256(fr28_sub(Z, fr28_mul(X, Y))), z - x * y) &&
                      same_fr(fr28_to_mont256(fr28_norm(fr28_sub(fr28_add(Z, Z), fr28_mul(X, Y)))), z + z - x * y);
            if (!ok && bad++ < 5) printf("fr28 mismatch at case %d\n", k);
        }
        // a whole transform, the kernel's loop (fr_ops.hip: k_ntt4096) on both fields; twiddles w^k from the CIOS side
        const uint32_t omega_raw[8] = {0xa5d36306u, 0xe206da11u, 0x378fbf96u, 0x0ad1347bu, 0xe0f8245fu, 0xfc3e8acfu, 0xa0f704f4u, 0x564c0a11u};
        const Fr w = fe_from_raw<FrParams>(omega_raw);
        std::vector<Fr> tw(2048), a(4096);
        std::vector<Fr28> tw28(2048), a28(4096);
        tw[0] = Fr::one();
        for (int k = 1; k < 2048; k++) tw[k] = tw[k - 1] * w;
        for (int k = 0; k < 2048; k++) tw28[k] = fr28_canonical(fr28_from_mont256(tw[k]));
        for (int i = 0; i < 4096; i++) {
            a[i] = rand_fr();
            if (i % 7 == 0) { uint32_t raw[8]; for (int j = 0; j < 8; j++) raw[j] = FrParams::MOD[j]; raw[0] -= 1 + (i & 3); a[i] = fe_from_raw<FrParams>(raw); }  // large values
            a28[i] = fr28_from_mont256(a[i]);
        }
        uint32_t max_limb = 0;
        for (int s = 0; s < 12; s++) {
            const int half = 1 << s, tshift = 11 - s;
            for (int bfly = 0; bfly < 2048; bfly++) {
                int k = bfly & (half - 1), i0 = ((bfly >> s) << (s + 1)) + k, i1 = i0 + half;
                Fr u = a[i0], v = a[i1] * tw[k << tshift];
                a[i0] = u + v;
                a[i1] = u - v;
                Fr28 u28 = a28[i0];
                if (s == 6) u28 = fr28_norm(u28);
                Fr28 v28 = a28[i1];
                if (s != 0) v28 = fr28_mul(v28, tw28[k << tshift]);
                a28[i0] = fr28_add(u28, v28);
                a28[i1] = fr28_sub(u28, v28);
                for (int j = 0; j < 10; j++) {
                    if (a28[i0].l[j] > max_limb) max_limb = a28[i0].l[j];
                    if (a28[i1].l[j] > max_limb) max_limb = a28[i1].l[j];
                }
            }
        }
        int wrong = 0;
        for (int i = 0; i < 4096; i++) wrong += !same_fr(fr28_to_mont256(a28[i]), a[i]);
        if (wrong || max_limb >= (14u << 28)) {  // the analysis says 13 units of 2^28 at most
            printf("fr28 transform: %d of 4096 outputs differ, largest limb %u units\n", wrong, max_limb >> 28);
            bad++;
        }
    }
    printf(bad ? "FAIL %d\n" : "ok: inversion, field, group-law and transform-field cross-checks agree\n", bad);
    return bad != 0;
}
