// fp2.h -- Fp2 = Fp[u]/(u^2 + 1) element shared by the host-side G2 / pairing code.
#pragma once
#include "field.cuh"

namespace lwk {

struct Fp2 {
    Fp c0, c1;  // c0 + c1 * u
};

// g2_pairing.hip
bool g2_decompress(Fp2 &x, Fp2 &y, bool &inf, const uint8_t in[96]);
}  // namespace lwk
