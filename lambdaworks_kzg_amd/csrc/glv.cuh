// glv.cuh -- the scalar side of the G1 endomorphism split used by the verification's linear combinations (setup.hip: k_lincomb3,
// vmsm.hip: k_vmsm_scalars):  [k]P = [lo]P + [hi](-phi(P)),  phi(x, y) = (beta x, y) acting on G1 as multiplication by -z^2.
#pragma once
#include <stdint.h>

// also compiled as plain C++ by tests/glv_split_check.cpp (g++: the CPU suite holds the two splits against each other)
#if defined(__HIPCC__)
#define LWK_GLV_FN __host__ __device__ __forceinline__
#else
#define LWK_GLV_FN inline
#endif

namespace lwk {

// k = lo + hi * z^2 with z^2 = 0xac45a4010001a4020000000100000000 (the curve parameter squared, 128 bits);
// k < r = z^4 - z^2 + 1, so both halves fit 128 bits. Bitwise restoring division, once per lane.
LWK_GLV_FN void split_by_z2(uint32_t lo[4], uint32_t hi[4], const uint32_t k[8]) {
    const uint32_t d[4] = {0x00000000u, 0x00000001u, 0x0001a402u, 0xac45a401u};
    uint32_t rem[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++) hi[i] = 0;
    for (int bit = 255; bit >= 0; bit--) {
        const uint32_t top = rem[3] >> 31;
        rem[3] = (rem[3] << 1) | (rem[2] >> 31);
        rem[2] = (rem[2] << 1) | (rem[1] >> 31);
        rem[1] = (rem[1] << 1) | (rem[0] >> 31);
        rem[0] = (rem[0] << 1) | ((k[bit >> 5] >> (bit & 31)) & 1u);
        uint32_t t[4];
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint64_t v = (uint64_t)rem[i] - d[i] - br;
            t[i] = (uint32_t)v;
            br = (v >> 32) & 1u;
        }
        if (top || !br) {
#pragma unroll
            for (int i = 0; i < 4; i++) rem[i] = t[i];
            if (bit < 128) hi[bit >> 5] |= 1u << (bit & 31);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) lo[i] = rem[i];
}

// The same split by Barrett's reduction (r06: k_vmsm_scalars sits on the critical path behind r, and the 256 rounds of the restoring
// division above were most of its 82 us): q' = ((k >> 127) mu) >> 129 with mu = floor(2^256 / z^2) (129 bits) is floor(k / z^2) or one
// less for every k < 2^256, so ONE conditional correction finishes it (tests/glv_split_check.cpp: edge values and 10^6 random ones
// against the restoring division; tools' arithmetic: max corrections 1).
LWK_GLV_FN void split_by_z2_barrett(uint32_t lo[4], uint32_t hi[4], const uint32_t k[8]) {
    const uint32_t d[4] = {0x00000000u, 0x00000001u, 0x0001a402u, 0xac45a401u};
    const uint32_t mu[4] = {0xf6cfee2eu, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u};  // + 2^128
    uint32_t t[4];  // k >> 127
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = (k[3 + i] >> 31) | (k[4 + i] << 1);
    // p = t * (2^128 + mu'), 9 limbs; only p >> 129 is wanted, but every column carries
    uint32_t p[9];
    {
        uint64_t acc = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            uint64_t carry = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = c - i;
                if (j < 0 || j > 3) continue;
                const uint64_t m = (uint64_t)t[i] * mu[j];
                acc += (uint32_t)m;
                carry += m >> 32;
            }
            if (c >= 4) acc += t[c - 4];  // the 2^128 term
            p[c] = (uint32_t)acc;
            acc = (acc >> 32) + carry;
        }
        p[8] = (uint32_t)acc;
    }
    uint32_t q[4];  // p >> 129
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = (p[4 + i] >> 1) | (p[5 + i] << 31);
    // rem = k - q d, low 5 limbs (rem < 2 d < 2^129)
    uint32_t qd[5];
    {
        uint64_t acc = 0;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            uint64_t carry = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = c - i;
                if (j < 0 || j > 3) continue;
                const uint64_t m = (uint64_t)q[i] * d[j];
                acc += (uint32_t)m;
                carry += m >> 32;
            }
            qd[c] = (uint32_t)acc;
            acc = (acc >> 32) + carry;
        }
    }
    uint32_t rem[5];
    {
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint64_t v = (uint64_t)k[i] - qd[i] - br;
            rem[i] = (uint32_t)v;
            br = (v >> 32) & 1u;
        }
    }
    // rem >= d ? one correction
    uint32_t s[4];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint64_t v = (uint64_t)rem[i] - d[i] - br;
        s[i] = (uint32_t)v;
        br = (v >> 32) & 1u;
    }
    const bool ge = rem[4] != 0 || br == 0;
    uint64_t c = ge ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        lo[i] = ge ? s[i] : rem[i];
        c += q[i];
        hi[i] = (uint32_t)c;
        c >>= 32;
    }
}

}  // namespace lwk
