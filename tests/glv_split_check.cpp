// CPU check of csrc/glv.cuh (compiled by tests/test_capi_cpu.py with g++): the Barrett split of k = lo + hi z^2 that k_vmsm_scalars uses
// against the bitwise restoring division k_lincomb3 keeps, on edge values and on random integers below 2^255 (the scalars are < r < 2^255;
// beyond z^2 2^128 the quotient no longer fits the four limbs either function returns).
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../lambdaworks_kzg_amd/csrc/glv.cuh"

static uint64_t sm = 0x9e3779b97f4a7c15ull;
static uint64_t splitmix() {
    uint64_t z = (sm += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

static int check(const uint32_t k[8]) {
    uint32_t lo1[4], hi1[4], lo2[4], hi2[4];
    lwk::split_by_z2(lo1, hi1, k);
    lwk::split_by_z2_barrett(lo2, hi2, k);
    return memcmp(lo1, lo2, 16) == 0 && memcmp(hi1, hi2, 16) == 0;
}

int main() {
    const uint32_t d[4] = {0x00000000u, 0x00000001u, 0x0001a402u, 0xac45a401u};
    long bad = 0, n = 0;
    // multiples of z^2 and their neighbours, the ends of the range
    for (int m = 0; m < 64; m++) {
        for (int delta = -2; delta <= 2; delta++) {
            uint32_t k[8] = {0};
            // k = (hi_m) * d + delta with hi_m a patterned 128-bit value (k < 2^256)
            uint32_t h[4] = {(uint32_t)splitmix(), (uint32_t)splitmix(), (uint32_t)splitmix(), (uint32_t)splitmix()};
            if (m == 0) h[0] = h[1] = h[2] = h[3] = 0;
            if (m == 0 && delta < 0) continue;  // (would wrap to 2^256 - 1)
            h[3] &= 0x7fffffffu;  // k < 2^255
            if (m == 1) { h[0] = h[1] = h[2] = 0xffffffffu; h[3] = 0x7fffffffu; }
            if (m == 2) { h[0] = 1; h[1] = h[2] = h[3] = 0; }
            uint64_t acc[9] = {0};
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) {
                    const uint64_t p = (uint64_t)h[i] * d[j];
                    acc[i + j] += (uint32_t)p;
                    acc[i + j + 1] += p >> 32;
                }
            uint64_t c = 0;
            for (int i = 0; i < 8; i++) {
                c += acc[i];
                k[i] = (uint32_t)c;
                c >>= 32;
            }
            // + delta (mod 2^256)
            int64_t cc = delta;
            for (int i = 0; i < 8 && cc != 0; i++) {
                const int64_t v = (int64_t)k[i] + cc;
                k[i] = (uint32_t)v;
                cc = v >> 32;
            }
            n++;
            if (!check(k)) bad++;
        }
    }
    for (long t = 0; t < 1000000; t++) {
        uint32_t k[8];
        for (int i = 0; i < 8; i += 2) {
            const uint64_t v = splitmix();
            k[i] = (uint32_t)v;
            k[i + 1] = (uint32_t)(v >> 32);
        }
        k[7] &= 0x7fffffffu;
        n++;
        if (!check(k)) bad++;
    }
    printf("%ld %ld\n", n, bad);
    return bad != 0;
}
