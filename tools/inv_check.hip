// inv_check.hip -- host-side cross-check of the division-step inversion (field29.cuh: f29_inv) against Fermat's
// a^(p-2) (f29_inv_fermat) and against a * a^-1 == 1. Pure host code: runs without a GPU.
//   hipcc -O2 -std=c++17 -I lambdaworks_kzg_amd/csrc tools/inv_check.hip -o /tmp/inv_check && /tmp/inv_check
#include <stdio.h>
#include <stdint.h>
#include "field29.cuh"
using namespace lwk;

static uint64_t sm(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main() {
    uint64_t seed = 12345;
    int bad = 0;
    for (int k = 0; k < 3000; k++) {
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
        ok &= inv(fa) == inv_fermat(fa);   // the 12x32-bit representation's entry point (host pairing, setup kernels)
        if (!ok) { bad++; if (bad < 5) printf("mismatch at case %d\n", k); }
    }
    printf(bad ? "FAIL %d\n" : "ok 3000 inversions match Fermat\n", bad);
    return bad != 0;
}
