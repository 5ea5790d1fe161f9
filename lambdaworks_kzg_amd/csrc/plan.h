// plan.h -- compile-time shape of the blob-commitment pipeline on one MI355X.
//
// One unit of work = one 4096-element blob -> one 48-byte commitment = one 4096-term G1 MSM
// (reference: KZG::commit -> msm::pippenger::msm, call site /root/reference/src/lib.rs:270).
//
// The SRS is fixed, so the engine precomputes T[j][i] = 2^(c*j) * P_i for every window j at
// load time ("fixed-base" Pippenger). All 20 windows of a blob then share ONE bucket set:
//   digits   : 4096 scalars x 20 signed 13-bit digits  -> <= 81,920 (point, sign) entries
//   buckets  : 2^12 = 4096 per blob (digit magnitudes 1..4096)
//   reduce   : sum_k k * B_k once per blob (instead of once per window, and no doublings)
// Table size: 20 x 4096 x 96 B = 7.5 MiB, resident in the 256 MiB Infinity Cache.
#pragma once
#include <stdint.h>

namespace lwk {

constexpr int kBlobElems = 4096;
constexpr int kBlobBytes = kBlobElems * 32;
constexpr int kWindowBits = 13;
constexpr int kNumWindows = 20;  // 20 * 13 = 260 >= 256: room for the signed-digit carry
constexpr int kNumBuckets = 1 << (kWindowBits - 1);
constexpr int kMaxEntries = kBlobElems * kNumWindows;
constexpr int kTablePoints = kNumWindows * kBlobElems;
constexpr uint32_t kEntryNegBit = 0x80000000u;

static_assert(kNumWindows * kWindowBits >= 256 + 1, "signed digits need one spare bit");

// status words written by kernels (values of C_KZG_RET)
constexpr int kStatusOk = 0;
constexpr int kStatusBadArgs = 1;
constexpr int kStatusError = 2;

}  // namespace lwk
