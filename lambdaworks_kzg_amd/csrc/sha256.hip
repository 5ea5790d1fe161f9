// sha256.hip -- Fiat-Shamir challenge on the GPU.
//
// compute_challenge (/root/reference/src/utils.rs:120-144) hashes
//     "FSBLOBVERIFY_V1_" | usize(4096) LE | u64(0) LE | blob (131072 B) | compress(commitment) (48 B)
// = 131,152 bytes with SHA-256 and reads the digest as a field element (hash_field_unsafe,
// utils.rs:148-154: big-endian, reduced mod r; the c-kzg-4844 vectors read it little-endian).
// The reference builds that 131 KB Vec byte by byte on the host for every blob; here the blob is
// already in HBM, so one lane per blob streams it through the compression function (16 B loads).
// SHA-256 is sequential per message: the parallelism is across the blobs of the batch.
#include <stdlib.h>
#include <atomic>
#include "kernels.h"
#include "knobs.h"

namespace lwk {

__device__ __constant__ uint32_t kShaK[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }

// w[16] holds the block as big-endian words
__device__ __forceinline__ void sha256_compress(uint32_t h[8], uint32_t w[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        uint32_t wi;
        if (i < 16) {
            wi = w[i];
        } else {
            uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
            uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
            wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
            w[i & 15] = wi;
        }
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + kShaK[i] + wi;
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// The 131,152-byte message as 8,200 16-byte chunks (with padding): chunk q ->
//   0..1      header  ("FSBLOBVERIFY_V1_", le64(4096) le64(0))
//   2..8193   blob
//   8194..8196 commitment
//   8197      0x80 then zeros, 8198 zeros, 8199 zeros + 64-bit big-endian bit length
__device__ __forceinline__ uint4 challenge_chunk(int q, const uint4 *blob, const uint4 *comm) {
    if (q >= 2 && q < 2 + kBlobBytes / 16) {
        uint4 v = blob[q - 2];
        return make_uint4(__builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z), __builtin_bswap32(v.w));
    }
    if (q == 0) return make_uint4(0x4653424cu, 0x4f425645u, 0x52494659u, 0x5f56315fu);  // "FSBL" "OBVE" "RIFY" "_V1_"
    if (q == 1) return make_uint4(0x00100000u, 0u, 0u, 0u);                             // 4096 little-endian, then 0
    if (q < 2 + kBlobBytes / 16 + 3) {
        uint4 v = comm[q - 2 - kBlobBytes / 16];
        return make_uint4(__builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z), __builtin_bswap32(v.w));
    }
    constexpr int kPad0 = 2 + kBlobBytes / 16 + 3;
    if (q == kPad0) return make_uint4(0x80000000u, 0u, 0u, 0u);
    if (q == kPad0 + 2) return make_uint4(0u, 0u, 0u, (uint32_t)((kBlobBytes + 80) * 8));  // 1,049,216 bits
    return make_uint4(0u, 0u, 0u, 0u);
}

// SHA-256 is sequential per message and a lane issues one instruction every ~4.5 cycles no matter how idle the chip
// is, so the hash of a batch takes (instructions per lane) x 4.5 cycles however few blobs there are. The kernel
// therefore splits the per-lane instruction stream over TWO cooperating waves per 64 blobs:
//   wave 0 (producer): loads + byte-swaps the next 64-byte block, expands the 64-word message schedule, adds the
//                      round constants, parks W[t] + K[t] in LDS (double-buffered, 16 KB per buffer);
//   wave 1 (consumer): runs only the 64 rounds (14-15 instructions each) on the words it reads back as 16 x b128.
// One barrier per block. 7.8 -> ~4 ms per batch.
//
// only_if_differs_from (optional): the fix-up pass of the optimistic pipeline -- z was already computed from
// these (caller-supplied) commitment bytes while the validation kernel ran; lanes whose canonical bytes are
// identical have nothing to redo, and a workgroup without any lane left exits at once.
__global__ __launch_bounds__(128) void k_challenge(const uint8_t *__restrict__ blobs, const uint8_t *__restrict__ canon48,
                                                   Fr *__restrict__ z_mont, int le, size_t n,
                                                   const uint8_t *__restrict__ only_if_differs_from) {
    __shared__ uint4 wk[2][16][64];
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    size_t i = (size_t)blockIdx.x * 64 + lane;
    bool active = i < n;
    if (!active) i = n - 1;  // keeps every address valid; the result is dropped
    const uint4 *blob = (const uint4 *)(blobs + (size_t)kBlobBytes * i);
    const uint4 *comm = (const uint4 *)(canon48 + 48 * i);
    if (only_if_differs_from) {
        const uint4 *raw = (const uint4 *)(only_if_differs_from + 48 * i);
        bool same = true;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            uint4 a = comm[k], b = raw[k];
            same = same && a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w;
        }
        active = active && !same;
    }
    if (__ballot(active) == 0) return;  // both waves see the same 64 blobs: uniform across the workgroup

    constexpr int kBlocks = (2 + kBlobBytes / 16 + 3 + 3) / 4;  // 2050
    uint32_t a = 0x6a09e667u, b = 0xbb67ae85u, c = 0x3c6ef372u, d = 0xa54ff53au, e = 0x510e527fu, f = 0x9b05688cu,
             g = 0x1f83d9abu, hh = 0x5be0cd19u;
    uint32_t h0 = a, h1 = b, h2 = c, h3 = d, h4 = e, h5 = f, h6 = g, h7 = hh;
    uint4 nxt[4];
    if (role == 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) nxt[q] = challenge_chunk(q, blob, comm);
    }
    for (int it = 0; it <= kBlocks; it++) {
        if (role == 0) {
            if (it < kBlocks) {
                uint32_t w[16];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    w[4 * q] = nxt[q].x; w[4 * q + 1] = nxt[q].y; w[4 * q + 2] = nxt[q].z; w[4 * q + 3] = nxt[q].w;
                }
                if (it + 1 < kBlocks) {  // the next block's loads fly while this one is expanded
#pragma unroll
                    for (int q = 0; q < 4; q++) nxt[q] = challenge_chunk(4 * (it + 1) + q, blob, comm);
                }
                uint4(*dst)[64] = wk[it & 1];
#pragma unroll
                for (int t = 0; t < 64; t += 4) {
                    uint32_t o[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int r = t + u;
                        if (r >= 16) {
                            uint32_t w15 = w[(r - 15) & 15], w2 = w[(r - 2) & 15];
                            uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
                            uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
                            w[r & 15] = w[r & 15] + s0 + w[(r - 7) & 15] + s1;
                        }
                        o[u] = w[r & 15] + kShaK[r];
                    }
                    dst[t >> 2][lane] = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
        } else if (it > 0) {
            const uint4(*src)[64] = wk[(it - 1) & 1];
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const uint4 v = src[q][lane];
                const uint32_t wq[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
                    uint32_t ch = (e & f) ^ (~e & g);
                    uint32_t t1 = hh + S1 + ch + wq[u];
                    uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
                    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
                    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + S0 + mj;
                }
            }
            h0 += a; h1 += b; h2 += c; h3 += d; h4 += e; h5 += f; h6 += g; h7 += hh;
            a = h0; b = h1; c = h2; d = h3; e = h4; f = h5; g = h6; hh = h7;
        }
        __syncthreads();
    }
    if (role != 1 || !active) return;
    const uint32_t h[8] = {h0, h1, h2, h3, h4, h5, h6, h7};
    // digest bytes d[0..32) = big-endian h[0..8)
    uint32_t s[8];
    if (le) {
        // little-endian integer: limb k = bytes 4k..4k+3 little-endian = bswap(h[k])
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = __builtin_bswap32(h[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = h[7 - k];
    }
    z_mont[i] = fe_from_raw<FrParams>(s);  // reduced mod r
}

// ---- lane-pair variant ---------------------------------------------------------------------------------------
// The consumer's 15-16 instructions per round shrink to TEN when a blob's state is split over two neighbouring lanes: the
// "e half" (e, f, g, h) and the "a half" (a, b, c, d) run the SAME instruction stream on different data --
//     Sigma(r0): three rotations with per-lane amounts (6, 11, 25 | 2, 13, 22) joined by ONE v_bitop3_b32 (truth table 0x96 =
//                three-way xor; gfx950 has the three-input boolean instruction),
//     Ch(e, f, g) = bfi(e, f, g) and Maj(a, b, c) = bfi(a ^ c, b, c) as one bfi(r0 ^ (r2 & a_half), r1, r2), the selector
//                again ONE v_bitop3_b32 (0x78 = a ^ (b & c)),
//     t = Sigma + select + u, where u = h + (W + K) on the e half and 0 on the a half is computed ONE ROUND AHEAD by a
//                bank-masked DPP add (the next round's h is this round's g) -- which also is the independent instruction
//                that fills the wait state between the write of t and its DPP read two instructions later,
// and exchange T1 / d through DPP operands: e' = d + T1, a' = T1 + T2. Within a row of 16 lanes, lanes 0-7 are
// the e halves of eight blobs and lanes 8-15 their a halves (partner = lane ^ 8 = row_ror:8; DPP banks 0-1 | 2-3).
// Two producer waves (even / odd blocks, half a block per barrier interval each) feed two consumer waves (32 blobs
// each); the consumers preload the sixteen LDS words of a block before its first round. LWKZG_HASH_PAIRS=0 selects
// the plain kernel. (Round 1: 12 instructions with two xors and a separate + h; 3.3 ms per batch.)
// One round on state registers named R0..R3 (asm operand names); WKN = the NEXT round's W + K; the new r0 lands in R3's
// register (the roles rotate through the four registers, back to the start after four rounds).
#define LWK_SHA_PAIR_RND(R0, R1, R2, R3, WKN)                                                      \
    "v_alignbit_b32 %[t1], %[" R0 "], %[" R0 "], %[s1]\n"                                           \
    "v_alignbit_b32 %[t2], %[" R0 "], %[" R0 "], %[s2]\n"                                           \
    "v_alignbit_b32 %[t3], %[" R0 "], %[" R0 "], %[s3]\n"                                           \
    "v_bitop3_b32 %[t1], %[t1], %[t2], %[t3] bitop3:0x96\n"                                         \
    "v_bitop3_b32 %[t2], %[" R0 "], %[" R2 "], %[bm] bitop3:0x78\n"                                 \
    "v_bfi_b32 %[t2], %[t2], %[" R1 "], %[" R2 "]\n"                                                \
    "v_add3_u32 %[t1], %[t1], %[t2], %[u]\n"                                                       \
    "v_add_u32_dpp %[" R3 "], %[" R3 "], %[t1] row_ror:8 row_mask:0xf bank_mask:0x3\n"              \
    "v_add_u32_dpp %[u], %[" R2 "], %[" WKN "] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n"    \
    "v_add_u32_dpp %[" R3 "], %[t1], %[t1] row_ror:8 row_mask:0xf bank_mask:0xc\n"

// four rounds = one 16-byte word of W + K (V; N = the first word of the next one); state in (r0, r1, r2, r3) before and after
#define LWK_SHA_PAIR_ROUNDS4(V, N)                                                                           \
    {                                                                                                        \
        uint32_t t1, t2, t3;                                                                                 \
        asm volatile(LWK_SHA_PAIR_RND("a0", "a1", "a2", "a3", "w1") LWK_SHA_PAIR_RND("a3", "a0", "a1", "a2", "w2") \
                         LWK_SHA_PAIR_RND("a2", "a3", "a0", "a1", "w3") LWK_SHA_PAIR_RND("a1", "a2", "a3", "a0", "w4") \
                     : [a0] "+v"(r0), [a1] "+v"(r1), [a2] "+v"(r2), [a3] "+v"(r3), [u] "+v"(u), [t1] "=&v"(t1),    \
                       [t2] "=&v"(t2), [t3] "=&v"(t3)                                                         \
                     : [w1] "v"((V).y), [w2] "v"((V).z), [w3] "v"((V).w), [w4] "v"(N), [s1] "v"(s1), [s2] "v"(s2), \
                       [s3] "v"(s3), [bm] "v"(bm));                                                          \
    }

// MID = true: only the 2048 blocks that do not depend on the commitment (domain, degree and all but the last 32 bytes of
// the blob) are absorbed, and the chaining value goes to `midstate` (8 words per blob) instead of a challenge: the fused
// commit-and-prove entry point runs this beside the commitment MSM and finishes with k_challenge_finish once the
// commitments exist.
template <bool MID>
__global__ __launch_bounds__(256) void k_challenge_pairs(const uint8_t *__restrict__ blobs, const uint8_t *__restrict__ canon48,
                                                         Fr *__restrict__ z_mont, int le, size_t n,
                                                         const uint8_t *__restrict__ only_if_differs_from, int prio,
                                                         uint32_t *__restrict__ midstate) {
    // waves 0, 1: producers (all 64 blobs each; wave 0 expands the even blocks, wave 1 the odd ones, half a block
    // per barrier interval, so a block has two intervals to get ready); waves 2, 3: consumers (32 blobs each, two
    // lanes per blob). Block b is written in intervals b and b + 1 and read in interval b + 2: three LDS buffers.
    __shared__ uint4 wk[3][16][64];
    __shared__ uint4 zero4;  // what the a halves read in place of W + K
    // a latency chain of a few dozen waves: where one of them shares a SIMD with the throughput-bound MSM waves of
    // another call (engine.hip: pick_ctx), it should issue whenever it can -- the MSM fills the gaps
    if (prio) __builtin_amdgcn_s_setprio(2);
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const bool producer = role < 2;
    const bool a_half = !producer && ((lane >> 3) & 1);
    const int blob_in_wg = producer ? lane : (role - 2) * 32 + (lane >> 4) * 8 + (lane & 7);
    size_t i = (size_t)blockIdx.x * 64 + blob_in_wg;
    bool active = i < n;
    if (!active) i = n - 1;  // keeps every address valid; the result is dropped
    const uint4 *blob = (const uint4 *)(blobs + (size_t)kBlobBytes * i);
    const uint4 *comm = (const uint4 *)(canon48 + 48 * i);
    if (only_if_differs_from) {
        const uint4 *raw = (const uint4 *)(only_if_differs_from + 48 * i);
        bool same = true;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            uint4 a = comm[k], b = raw[k];
            same = same && a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w;
        }
        active = active && !same;
    }
    if (threadIdx.x == 0) zero4 = make_uint4(0u, 0u, 0u, 0u);
    if (!__syncthreads_or(active)) return;

    constexpr int kBlocks = MID ? 2048 : (2 + kBlobBytes / 16 + 3 + 3) / 4;  // 2050 for the whole message
    // consumer state: (e, f, g, h) on the e half, (a, b, c, d) on the a half; hv = the chaining values of that half
    uint32_t hv0 = a_half ? 0x6a09e667u : 0x510e527fu, hv1 = a_half ? 0xbb67ae85u : 0x9b05688cu,
             hv2 = a_half ? 0x3c6ef372u : 0x1f83d9abu, hv3 = a_half ? 0xa54ff53au : 0x5be0cd19u;
    uint32_t r0 = hv0, r1 = hv1, r2 = hv2, r3 = hv3;
    const uint32_t s1 = a_half ? 2u : 6u, s2 = a_half ? 13u : 11u, s3 = a_half ? 22u : 25u, bm = a_half ? ~0u : 0u;
    uint32_t u = 0;  // h + (W + K) of the coming round on the e half, 0 on the a half (see LWK_SHA_PAIR_RND)
    // producer state: the rolling 16-word window of its current block and the loads of its next one
    uint32_t w[16];
    uint4 nxt[4];
    int pblock = role;  // the block this producer is expanding (role 0: 0, 2, 4, ...; role 1: 1, 3, 5, ...)
    if (producer) {
#pragma unroll
        for (int q = 0; q < 4; q++) nxt[q] = challenge_chunk(4 * pblock + q, blob, comm);
    }
    for (int it = 0; it < kBlocks + 2; it++) {
        if (producer) {
            const int phase = (it - role) & 1;  // 0: first half of the schedule, 1: second half
            if (it >= role && pblock < kBlocks) {
                uint4(*dst)[64] = wk[pblock % 3];
                if (phase == 0) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        w[4 * q] = nxt[q].x; w[4 * q + 1] = nxt[q].y; w[4 * q + 2] = nxt[q].z; w[4 * q + 3] = nxt[q].w;
                    }
                    if (pblock + 2 < kBlocks) {  // this producer's next block: its loads have two intervals to land
#pragma unroll
                        for (int q = 0; q < 4; q++) nxt[q] = challenge_chunk(4 * (pblock + 2) + q, blob, comm);
                    }
#pragma unroll
                    for (int t = 0; t < 32; t += 4) {
                        uint32_t o[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int r = t + u;
                            if (r >= 16) {
                                uint32_t w15 = w[(r - 15) & 15], w2 = w[(r - 2) & 15];
                                uint32_t g0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
                                uint32_t g1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
                                w[r & 15] = w[r & 15] + g0 + w[(r - 7) & 15] + g1;
                            }
                            o[u] = w[r & 15] + kShaK[r];
                        }
                        dst[t >> 2][lane] = make_uint4(o[0], o[1], o[2], o[3]);
                    }
                } else {
#pragma unroll
                    for (int t = 32; t < 64; t += 4) {
                        uint32_t o[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int r = t + u;
                            uint32_t w15 = w[(r - 15) & 15], w2 = w[(r - 2) & 15];
                            uint32_t g0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
                            uint32_t g1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
                            w[r & 15] = w[r & 15] + g0 + w[(r - 7) & 15] + g1;
                            o[u] = w[r & 15] + kShaK[r];
                        }
                        dst[t >> 2][lane] = make_uint4(o[0], o[1], o[2], o[3]);
                    }
                    pblock += 2;
                }
            }
        } else if (it >= 2) {
            // e halves walk the 16 x b128 words of their blob, a halves re-read the zero word
            const char *base = a_half ? (const char *)&zero4 : (const char *)&wk[(it - 2) % 3][0][blob_in_wg];
            const uint32_t stride = a_half ? 0u : (uint32_t)sizeof(uint4) * 64u;
            uint4 v[16];  // all sixteen LDS reads in flight before the first round (the asm rounds are scheduling barriers)
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = *(const uint4 *)(base + q * stride);
            u = a_half ? 0u : r3 + v[0].x;  // the first round's h + (W + K); every later one is computed a round ahead
#pragma unroll
            for (int q = 0; q < 16; q++) LWK_SHA_PAIR_ROUNDS4(v[q], v[q < 15 ? q + 1 : 15].x)  // (the last round's look-ahead is unused)
            hv0 += r0; hv1 += r1; hv2 += r2; hv3 += r3;
            r0 = hv0; r1 = hv1; r2 = hv2; r3 = hv3;
        }
        __syncthreads();
    }
    if (producer) return;
    // the e half collects the a half's four words and writes z
    const uint32_t p0 = __shfl_xor(hv0, 8, 64), p1 = __shfl_xor(hv1, 8, 64), p2 = __shfl_xor(hv2, 8, 64),
                   p3 = __shfl_xor(hv3, 8, 64);
    if (a_half || !active) return;
    const uint32_t h[8] = {p0, p1, p2, p3, hv0, hv1, hv2, hv3};
    if constexpr (MID) {
#pragma unroll
        for (int k = 0; k < 8; k++) midstate[8 * i + k] = h[k];
        return;
    }
    uint32_t sdig[8];
    if (le) {
#pragma unroll
        for (int k = 0; k < 8; k++) sdig[k] = __builtin_bswap32(h[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) sdig[k] = h[7 - k];
    }
    z_mont[i] = fe_from_raw<FrParams>(sdig);  // reduced mod r
}

void launch_challenge(const uint8_t *blobs, const uint8_t *canon48, Fr *z_mont, int le, size_t n, hipStream_t st,
                      const uint8_t *only_if_differs_from) {
    if (n == 0) return;
    const bool pairs = knobs().hash_pairs;
    const int prio = knobs().hash_prio;
    ProfScope p(only_if_differs_from ? "k_challenge_fixup" : "k_challenge", st);
    if (pairs)
        hipLaunchKernelGGL(k_challenge_pairs<false>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, blobs, canon48, z_mont, le, n,
                           only_if_differs_from, prio, (uint32_t *)nullptr);
    else
        hipLaunchKernelGGL(k_challenge, dim3((unsigned)((n + 63) / 64)), dim3(128), 0, st, blobs, canon48, z_mont, le, n,
                           only_if_differs_from);
}

// ---- the challenge in two parts (fused commit-and-prove: engine.hip) ------------------------------------------------
// part 1: everything the commitment does not touch (2048 of the 2050 blocks), beside the commitment MSM
void launch_challenge_midstate(const uint8_t *blobs, uint32_t *midstate, size_t n, hipStream_t st) {
    if (n == 0) return;
    const int prio = knobs().hash_prio;
    ProfScope p("k_challenge_midstate", st);
    hipLaunchKernelGGL(k_challenge_pairs<true>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, blobs, (const uint8_t *)nullptr,
                       (Fr *)nullptr, 0, n, (const uint8_t *)nullptr, prio, midstate);
}

// part 2: the last two blocks (32 bytes of blob, the 48 commitment bytes, padding and length), one lane per blob
__global__ __launch_bounds__(64) void k_challenge_finish(const uint8_t *__restrict__ blobs, const uint8_t *__restrict__ canon48,
                                                         const uint32_t *__restrict__ midstate, Fr *__restrict__ z_mont, int le,
                                                         size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    __builtin_amdgcn_s_setprio(2);
    const uint4 *blob = (const uint4 *)(blobs + (size_t)kBlobBytes * i);
    const uint4 *comm = (const uint4 *)(canon48 + 48 * i);
    uint32_t h[8];
#pragma unroll
    for (int k = 0; k < 8; k++) h[k] = midstate[8 * i + k];
    for (int blk = 2048; blk < 2050; blk++) {
        uint32_t w[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 c = challenge_chunk(4 * blk + q, blob, comm);
            w[4 * q] = c.x; w[4 * q + 1] = c.y; w[4 * q + 2] = c.z; w[4 * q + 3] = c.w;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
        for (int r = 0; r < 64; r++) {
            if (r >= 16) {
                const uint32_t w15 = w[(r - 15) & 15], w2 = w[(r - 2) & 15];
                w[r & 15] += (rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3)) + w[(r - 7) & 15] + (rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10));
            }
            const uint32_t t1 = hh + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + kShaK[r] + w[r & 15];
            const uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    uint32_t sdig[8];
    if (le) {
#pragma unroll
        for (int k = 0; k < 8; k++) sdig[k] = __builtin_bswap32(h[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) sdig[k] = h[7 - k];
    }
    z_mont[i] = fe_from_raw<FrParams>(sdig);  // reduced mod r
}

void launch_challenge_finish(const uint8_t *blobs, const uint8_t *canon48, const uint32_t *midstate, Fr *z_mont, int le, size_t n,
                             hipStream_t st) {
    if (n == 0) return;
    ProfScope p("k_challenge_finish", st);
    hipLaunchKernelGGL(k_challenge_finish, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, blobs, canon48, midstate, z_mont, le, n);
}

// host SHA-256 for the one batch-level hash of verify_blob_kzg_proof_batch (compute_r_powers,
// /root/reference/src/utils.rs:166-206): a few hundred KB once per call, not worth a launch
static const uint32_t kShaKHost[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static void sha256_block_host(uint32_t h[8], const uint8_t *b) {
    auto ror = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + kShaKHost[i] + w[i];
        uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
    }
    h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void sha256_blocks_portable(uint32_t h[8], const uint8_t *blocks, size_t n_blocks) {
    for (size_t k = 0; k < n_blocks; k++) sha256_block_host(h, blocks + 64 * k);
}

void sha256_host(uint8_t out[32], const uint8_t *msg, size_t len) {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    size_t i = 0;
    for (; i + 64 <= len; i += 64) sha256_block_host(h, msg + i);
    uint8_t tail[128] = {0};
    size_t rem = len - i;
    for (size_t k = 0; k < rem; k++) tail[k] = msg[i + k];
    tail[rem] = 0x80;
    size_t tl = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    sha256_block_host(h, tail);
    if (tl == 128) sha256_block_host(h, tail + 64);
    for (int k = 0; k < 8; k++) {
        out[4 * k] = (uint8_t)(h[k] >> 24);
        out[4 * k + 1] = (uint8_t)(h[k] >> 16);
        out[4 * k + 2] = (uint8_t)(h[k] >> 8);
        out[4 * k + 3] = (uint8_t)h[k];
    }
}

// decompress_g1_point (incl. the subgroup check) then compress_g1_point again, as compute_blob_kzg_proof +
// compute_challenge do (/root/reference/src/lib.rs:372-375, src/utils.rs:138). One lane per point, all in the
// lazy-limb field (field29.cuh): square root (p = 3 mod 4), root selection by the sign flag, endomorphism subgroup
// test. Re-compressing an affine point needs no inversion: the canonical bytes are x (reduced) + flags.
// aff_out / kind_out (optional): the validated point in the hot-loop representation and 0 = affine,
// 1 = infinity, 2 = invalid, for the verify side's linear combinations.
__global__ __launch_bounds__(64) void k_validate_commitments(const uint8_t *__restrict__ comm48,
                                                             uint8_t *__restrict__ canon48, int32_t *__restrict__ status,
                                                             int bad_code, size_t n, G1Affine29 *__restrict__ aff_out,
                                                             int32_t *__restrict__ kind_out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    __builtin_amdgcn_s_setprio(2);  // a latency chain, like the hash kernel it runs beside (see there)
    G1Affine29 aff;
    aff.x = F29<2>::zero();
    aff.y = F29<2>::zero();
    uint8_t o[48];
    for (int k = 0; k < 48; k++) o[k] = 0;
    F29<2> x = F29<2>::zero(), y = F29<2>::zero();
    bool want_greater = false;
    int rc = g1_decompress29_nocheck(comm48 + 48 * i, x, y, want_greater);
    if (rc == 1) {
        o[0] = 0xc0;
    } else if (rc == 0) {
        uint32_t braw[12];
        g1_beta_raw(braw);
        if (!g1_in_subgroup_endo<G1Xyzz29>(x, y, f29_from_raw32(braw))) {
            rc = 2;
        } else {
            uint32_t rx[12];
            f29_to_raw32(rx, x);
            raw_to_be<12>(o, rx);
            o[0] |= 0x80;
            if (want_greater) o[0] |= 0x20;
            aff.x = x;
            aff.y = y;
        }
    }
    if (rc == 2) {
        status[i] = bad_code;
        for (int k = 0; k < 48; k++) o[k] = 0;
    }
    for (int k = 0; k < 48; k++) canon48[48 * i + k] = o[k];
    if (aff_out) aff_out[i] = aff;
    if (kind_out) kind_out[i] = rc;
}

static bool validate_coop_enabled() {
    return knobs().validate_coop;
}


void launch_validate_commitments(const uint8_t *comm48, uint8_t *canon48, int32_t *status, int bad_code, size_t n,
                                 hipStream_t st, G1Affine29 *aff_out, int32_t *kind_out, uint32_t *verdict_scratch, bool apart) {
    // r05: with scratch for the points and the verdicts the validation is three launches -- the square root (one lane per point, windowed),
    // the subgroup test on a quad of lanes per point (k_subgroup_coop_asm), canonical bytes + verdicts -- 2.0 -> ~1.0 ms whatever the batch
    if (aff_out && kind_out && verdict_scratch && n && validate_coop_enabled()) {
        launch_decompress_points(comm48, aff_out, kind_out, n, st, apart);
        launch_subgroup_canon(aff_out, kind_out, canon48, status, bad_code, n, st, verdict_scratch, apart);
        return;
    }
    ProfScope p("k_validate_commitments", st);
    // The kernel is a one-wave-per-workgroup latency chain that runs beside other latency chains (the Fiat-Shamir hash of
    // the device-resident proofs, the other point set's validation). Where a wave of each shares a SIMD, both run at
    // about half speed, and the dispatcher likes to start every kernel's workgroups on the same compute units. An LDS
    // footprint the kernel never touches keeps them apart: 112 KB here + the hash kernel's 48 KB (or a second
    // validation workgroup) exceed the 160 KB of a compute unit, so the dispatcher has to pick another one. The hash
    // of 1024 blobs takes 3.2 ms instead of 4.3 ms beside it (LWKZG_VALIDATE_LDS_PAD=0 switches the padding off).
    const unsigned lds_pad = knobs().validate_lds_pad;
    static std::atomic<bool> pad_ok{true};  // a runtime that refuses the footprint gets the plain launch from then on
    if (lds_pad && pad_ok.load(std::memory_order_relaxed)) {
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_validate_commitments, dim3((unsigned)((n + 63) / 64)), dim3(64), lds_pad, st, comm48, canon48,
                           status, bad_code, n, aff_out, kind_out);
        if (hipGetLastError() == hipSuccess) return;
        pad_ok.store(false);
    }
    hipLaunchKernelGGL(k_validate_commitments, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, comm48, canon48, status,
                       bad_code, n, aff_out, kind_out);
}

// ---- the same validation in two launches, for the batch verification ------------------------------------------------
// k_decompress_points: square root only. k_subgroup_canon: subgroup test + canonical bytes + verdicts. What lies
// between them is the point of the split: the multiples the linear combinations want (setup.hip: k_point_multiples) need
// the decompressed point but not the subgroup verdict, so they run BESIDE the second kernel instead of behind it.
// kind carries the sign bit in bit 8 between the two kernels (rc | want_greater << 8) and is final (0 / 1 / 2) after
// the second; readers in between mask with 0xff.

// The square root's chain for a lane that is alone on its SIMD (256 commitments are four waves): f29_pow's 4-bit windows with the products
// INLINED (a call costs the lone wave ~40 instruction slots of moves, 475 times) and the 16-entry window table in LDS, [entry][limb][lane]
// (the exponent is public: every lane reads the same entry, its own column; the table indexed by a run-time digit would otherwise live
// in scratch, a memory round trip per window).
__device__ __forceinline__ F29<2> sqrt_chain_lds(const F29<2> &a, const uint32_t *e, uint32_t (*tab)[14][64], int lane) {
    typedef F29<2, true> Fi;
    Fi t1;
#pragma unroll
    for (int j = 0; j < 14; j++) t1.l[j] = a.l[j];
    t1 = t1 * F29<1, true>::one();
    const Fi one = Fi::one();
#pragma unroll
    for (int j = 0; j < 14; j++) {
        tab[0][j][lane] = one.l[j];
        tab[1][j][lane] = t1.l[j];
    }
    Fi cur = t1;
#pragma unroll 1
    for (int k = 2; k < 16; k++) {
        cur = cur * t1;
#pragma unroll
        for (int j = 0; j < 14; j++) tab[k][j][lane] = cur.l[j];
    }
    Fi acc = one;
    bool started = false;
#pragma unroll 1
    for (int w = 12 * 8 - 1; w >= 0; w--) {
        const uint32_t d = (e[w >> 3] >> (4 * (w & 7))) & 15u;
        if (started) {
            acc = sqr(acc);
            acc = sqr(acc);
            acc = sqr(acc);
            acc = sqr(acc);
        }
        if (d) {
            Fi f;
#pragma unroll
            for (int j = 0; j < 14; j++) f.l[j] = tab[d][j][lane];
            acc = started ? acc * f : f;
            started = true;
        }
    }
    F29<2> r;
#pragma unroll
    for (int j = 0; j < 14; j++) r.l[j] = acc.l[j];
    return r;
}

// blockIdx.y selects one of two point sets (a verification's proofs and commitments in one launch; a single set passes itself twice)
__global__ __launch_bounds__(64) void k_decompress_points(const uint8_t *__restrict__ in48_a, G1Affine29 *__restrict__ pts_a,
                                                          int32_t *__restrict__ kind_a, const uint8_t *__restrict__ in48_b,
                                                          G1Affine29 *__restrict__ pts_b, int32_t *__restrict__ kind_b, size_t n) {
    __shared__ uint32_t tab[16][14][64];   // 56 KiB: two workgroups to a compute unit
    const uint8_t *in48 = blockIdx.y ? in48_b : in48_a;
    G1Affine29 *pts = blockIdx.y ? pts_b : pts_a;
    int32_t *kind = blockIdx.y ? kind_b : kind_a;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F29<2> x = F29<2>::zero(), y = F29<2>::zero();
    bool want_greater = false;
    const int lane = threadIdx.x;
    const int rc = g1_decompress29_nocheck_t(in48 + 48 * i, x, y, want_greater,
                                             [&](const F29<2> &a, const uint32_t *e) { return sqrt_chain_lds(a, e, tab, lane); });
    G1Affine29 aff;
    aff.x = rc == 0 ? x : F29<2>::zero();
    aff.y = rc == 0 ? y : F29<2>::zero();
    pts[i] = aff;
    kind[i] = rc | (want_greater ? 0x100 : 0);
}

// verdict (optional): the cooperative subgroup test's word per point (k_subgroup_coop_asm: 0 = not in G1, 1 = in G1, 2 = undetermined --
// an addition met P = +-Q in its low 56 bits --, which this kernel settles with the complete-branches test)
__global__ __launch_bounds__(64) void k_subgroup_canon(G1Affine29 *__restrict__ pts_a, int32_t *__restrict__ kind_a,
                                                       uint8_t *__restrict__ canon48_a, const uint32_t *__restrict__ verdict_a,
                                                       G1Affine29 *__restrict__ pts_b, int32_t *__restrict__ kind_b,
                                                       uint8_t *__restrict__ canon48_b, const uint32_t *__restrict__ verdict_b,
                                                       int32_t *__restrict__ status, int bad_code, size_t n) {
    G1Affine29 *pts = blockIdx.y ? pts_b : pts_a;
    int32_t *kind = blockIdx.y ? kind_b : kind_a;
    uint8_t *canon48 = blockIdx.y ? canon48_b : canon48_a;
    const uint32_t *verdict = blockIdx.y ? verdict_b : verdict_a;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k0 = kind[i];
    int rc = k0 & 0xff;
    uint8_t o[48];
    for (int k = 0; k < 48; k++) o[k] = 0;
    if (rc == 1) {
        o[0] = 0xc0;
    } else if (rc == 0) {
        const G1Affine29 aff = pts[i];
        const uint32_t vd = verdict ? verdict[i] : 2u;
        bool in_g1 = vd == 1u;
        if (vd >= 2u) {
            uint32_t braw[12];
            g1_beta_raw(braw);
            in_g1 = g1_in_subgroup_endo<G1Xyzz29>(aff.x, aff.y, f29_from_raw32(braw));
        }
        if (!in_g1) {
            rc = 2;
            G1Affine29 z;
            z.x = F29<2>::zero();
            z.y = F29<2>::zero();
            pts[i] = z;
        } else {
            uint32_t rx[12];
            f29_to_raw32(rx, aff.x);
            raw_to_be<12>(o, rx);
            o[0] |= 0x80;
            if (k0 & 0x100) o[0] |= 0x20;
        }
    }
    if (rc == 2) status[i] = bad_code;
    for (int k = 0; k < 48; k++) canon48[48 * i + k] = o[k];
    kind[i] = rc;
}

// LWKZG_VERIFY_PAD_KB (experiment, knobs.h): an LDS footprint the validation kernels never touch, so that the dispatcher cannot put their
// workgroups on the compute units the hash kernel's workgroups occupy (profiles/r06_experiments.md section 1)
unsigned verify_pad_bytes(int which, const void *kernel) {
    static std::atomic<bool> allowed[3];  // a footprint above the default limit needs the attribute once per kernel
    const unsigned b = (unsigned)knobs().verify_pad_kb[which] * 1024u;
    if (b > 48u * 1024u && !allowed[which].exchange(true)) (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
    return b;
}

void launch_decompress_points(const uint8_t *in48, G1Affine29 *pts, int32_t *kind, size_t n, hipStream_t st, bool apart) {
    ProfScope p("k_decompress_points", st);
    hipLaunchKernelGGL(k_decompress_points, dim3((unsigned)((n + 63) / 64)), dim3(64), apart ? verify_pad_bytes(0, (const void *)k_decompress_points) : 0u, st,
                       in48, pts, kind, in48, pts, kind, n);
}

void launch_decompress_points2(const uint8_t *in48_a, G1Affine29 *pts_a, int32_t *kind_a, const uint8_t *in48_b, G1Affine29 *pts_b,
                               int32_t *kind_b, size_t n, hipStream_t st, bool apart) {
    ProfScope p("k_decompress_points", st);
    hipLaunchKernelGGL(k_decompress_points, dim3((unsigned)((n + 63) / 64), 2), dim3(64), apart ? verify_pad_bytes(0, (const void *)k_decompress_points) : 0u, st,
                       in48_a, pts_a, kind_a, in48_b, pts_b, kind_b, n);
}

// The subgroup test on a QUAD of lanes per point (tools/gen_subgroup_asm.py writes subgroup_asm.inc and explains it): doublings in three
// rounds of one product per lane, the cooperative MSM kernel's addition, the public bits of |z| as a scalar loop. Workgroups of four
// unrelated waves (one per SIMD of a compute unit), 16 points per wave.
__global__ __launch_bounds__(256) void k_subgroup_coop_asm(const G1Affine29 *__restrict__ pts_a, const int32_t *__restrict__ kind_a,
                                                           uint32_t *__restrict__ verdict_a, const G1Affine29 *__restrict__ pts_b,
                                                           const int32_t *__restrict__ kind_b, uint32_t *__restrict__ verdict_b, uint32_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    const G1Affine29 *pts = blockIdx.y ? pts_b : pts_a;
    const int32_t *kind = blockIdx.y ? kind_b : kind_a;
    uint32_t *verdict = blockIdx.y ? verdict_b : verdict_a;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t first = (blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 16;
    __builtin_amdgcn_s_setprio(2);
    asm volatile(
#include "subgroup_asm.inc"
        :
        : "s"(pts), "s"(kind), "s"(verdict), "s"(n), "s"(first), "v"(lane)
        :
#include "subgroup_asm_clobbers.inc"
    );
#endif
}

void launch_subgroup_canon(G1Affine29 *pts, int32_t *kind, uint8_t *canon48, int32_t *status, int bad_code, size_t n,
                           hipStream_t st, uint32_t *verdict_scratch, bool apart) {
    const uint32_t *verdict = nullptr;
    if (verdict_scratch && validate_coop_enabled()) {
        ProfScope p("k_subgroup_coop_asm", st);
        hipLaunchKernelGGL(k_subgroup_coop_asm, dim3((unsigned)((n + 63) / 64)), dim3(256), apart ? verify_pad_bytes(1, (const void *)k_subgroup_coop_asm) : 0u, st,
                           pts, kind, verdict_scratch, pts, kind, verdict_scratch, (uint32_t)n);
        verdict = verdict_scratch;
    }
    ProfScope p("k_subgroup_canon", st);
    hipLaunchKernelGGL(k_subgroup_canon, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, pts, kind, canon48, verdict, pts, kind, canon48, verdict,
                       status, bad_code, n);
}

// both point sets of a verification in one launch each (the quad test, then canonical bytes and verdicts)
void launch_subgroup_canon2(G1Affine29 *pts_a, int32_t *kind_a, uint8_t *canon48_a, uint32_t *verdict_a, G1Affine29 *pts_b, int32_t *kind_b,
                            uint8_t *canon48_b, uint32_t *verdict_b, int32_t *status, int bad_code, size_t n, hipStream_t st, bool apart) {
    {
        ProfScope p("k_subgroup_coop_asm", st);
        hipLaunchKernelGGL(k_subgroup_coop_asm, dim3((unsigned)((n + 63) / 64), 2), dim3(256), apart ? verify_pad_bytes(1, (const void *)k_subgroup_coop_asm) : 0u, st,
                           pts_a, kind_a, verdict_a, pts_b, kind_b, verdict_b, (uint32_t)n);
    }
    ProfScope p("k_subgroup_canon", st);
    hipLaunchKernelGGL(k_subgroup_canon, dim3((unsigned)((n + 63) / 64), 2), dim3(64), 0, st, pts_a, kind_a, canon48_a, verdict_a, pts_b, kind_b, canon48_b,
                       verdict_b, status, bad_code, n);
}

}  // namespace lwk
