// vmsm.hip -- the variable-base side of batch verification: the three size-n linear combinations of
// verify_kzg_proof_batch (/root/reference/src/lib.rs:679-685: g1_lincomb -> msm::pippenger::msm, three times) as ONE bucket MSM built
// for LATENCY, not throughput. 3 x 4096 terms are 0.05 ms of this chip's arithmetic; what a verification waits for is the length of
// the dependent chain behind the Fiat-Shamir scalar r (utils.rs:166-206), which is only known after every blob has been hashed and
// evaluated. So the work is cut at r:
//
//   BEFORE r, beside the 3.2 ms challenge hash (points only; k_vmsm_multiples): for every validated point P the 16 byte-spaced
//   multiples [2^(8 j)]P, j = 0..15, and the same for -phi(P) = (beta x, -y) (phi acts on G1 as multiplication by -z^2), all affine
//   (one inversion per point, Montgomery's trick over the 15 new rows). 120 dependent doublings per lane -- the whole doubling chain
//   of a 255-bit scalar multiplication -- are paid here, where nothing waits for them.
//
//   AFTER r (k_vmsm_scalars, k_vmsm_accumulate, k_vmsm_bucket_sums, k_vmsm_weighted): lane i takes a_i = r^(first + i) from a table
//   of r^(2^k), b_i = a_i z_i, splits both as lo + hi z^2 (two 128-bit halves), and every BYTE of a half is a digit d of one row:
//   [k]P = sum_j lo_j [2^(8 j)]P + sum_j hi_j [2^(8 j)](-phi(P)). All 32 rows of all terms of a set then share ONE set of 255
//   buckets (digit value -> bucket), because the shift is in the row, not in the window: no doubling is left behind r. A workgroup
//   takes a slice of terms, counting-sorts its <= 2048 (row, digit) entries by digit in LDS and lane b sums bucket b's rows (mean 8,
//   the chain that matters: ~16 mixed additions on the slowest lane of a wave); the slices' bucket sums are added per bucket (a wave
//   per bucket), and sum_b b B_b is a suffix scan plus a tree (16 levels) in one workgroup per set, which also leaves the sum affine.
//
// Replaces r05's k_point_multiples (96 doublings + an inversion per lane, three rows per point) + k_lincomb3 (a 32-bit
// double-and-add per lane and an 8-level tree) + three k_sum_points + k_xyzz29_to_affine_be: 1.41 + 0.41 + 0.1 ms behind r -> see
// profiles/r06_experiments.md section 2. LWKZG_VERIFY_MSM=0 (experiment knob) keeps r05's kernels as the A/B arm.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "knobs.h"
#include "glv.cuh"

namespace lwk {

unsigned verify_pad_bytes(int which, const void *kernel);  // sha256.hip

// rows of one point: row = half * 16 + j holds [2^(8 j)] of (half ? -phi(P) : P); table[row * n + i]
static_assert(kVmsmRows == 32 && kVmsmSteps == 15, "two 128-bit halves of sixteen byte digits");

// ---- before r: the rows ---------------------------------------------------------------------------------------------------------
// One lane per point, workgroups of four unrelated waves (one per SIMD of a compute unit, which the launch's LDS footprint keeps to itself);
// blockIdx.y selects the point set (proofs / commitments). tmp (XYZZ) and pre (prefix products) are global
// scratch, [step][lane] -- 15 x (224 + 56) bytes per lane would otherwise be private memory.
__global__ __launch_bounds__(256) void k_vmsm_multiples(const G1Affine29 *__restrict__ pts_a, const int32_t *__restrict__ kind_a,
                                                       G1Affine29 *__restrict__ tab_a, const G1Affine29 *__restrict__ pts_b,
                                                       const int32_t *__restrict__ kind_b, G1Affine29 *__restrict__ tab_b,
                                                       G1Xyzz29 *__restrict__ tmp, F29<2> *__restrict__ pre, uint32_t n) {
    const G1Affine29 *pts = blockIdx.y ? pts_b : pts_a;
    const int32_t *kind = blockIdx.y ? kind_b : kind_a;
    G1Affine29 *tab = blockIdx.y ? tab_b : tab_a;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if ((kind[i] & 0xff) != 0) return;  // infinity or invalid: k_vmsm_accumulate never reads this point's rows (bit 8: a sign bit in flight)
    const size_t lanes = (size_t)2 * n, g = (size_t)blockIdx.y * n + i;
    uint32_t braw[12];
    g1_beta_raw(braw);
    const F29<2> beta = f29_from_raw32(braw);
    const G1Affine29 p = pts[i];
    auto put = [&](int j, const F29<2> &x, const F29<2> &y) {
        G1Affine29 a, b;
        a.x = x;
        a.y = y;
        b.x = x * beta;
        b.y = neg(y) * F29<1>::one();  // back to the < 2p form the affine slots carry
        tab[(size_t)j * n + i] = a;
        tab[(size_t)(kVmsmDigits + j) * n + i] = b;
    };
    put(0, p.x, p.y);
    G1Xyzz29i acc = G1Xyzz29i::from_affine(*(const F29<2, true> *)&p.x, *(const F29<2, true> *)&p.y);
    F29<2, true> run = F29<2, true>::one();
#pragma unroll 1
    for (int j = 1; j <= kVmsmSteps; j++) {
#pragma unroll 1
        for (int k = 0; k < 8; k++) acc = xyzz_dbl(acc);
        // a point of G1 never doubles away (odd prime order); one that does is outside G1, its batch is rejected by the validation and
        // nothing below is read: the zero it leaves in the running product only spoils this lane's own rows
        tmp[(size_t)(j - 1) * lanes + g] = *(const G1Xyzz29 *)&acc;
        const F29<2, true> t = acc.zz * acc.zzz;
        run = j == 1 ? t : run * t;
        pre[(size_t)(j - 1) * lanes + g] = *(const F29<2> *)&run;
    }
    F29<2> inv = f29_inv(*(const F29<2> *)&run);  // 1 / (t_1 ... t_15)
#pragma unroll 1
    for (int j = kVmsmSteps; j >= 1; j--) {
        const G1Xyzz29 a = tmp[(size_t)(j - 1) * lanes + g];
        const F29<2> t = a.zz * a.zzz;
        F29<2> it = inv;  // 1 / t_j
        if (j > 1) it = inv * pre[(size_t)(j - 2) * lanes + g];
        inv = inv * t;
        const F29<2> izz = it * a.zzz, izzz = it * a.zz;
        put(j, a.x * izz, a.y * izzz);
    }
}

void launch_vmsm_multiples2(const G1Affine29 *pts_a, const int32_t *kind_a, G1Affine29 *tab_a, const G1Affine29 *pts_b,
                            const int32_t *kind_b, G1Affine29 *tab_b, G1Xyzz29 *tmp, F29<2> *pre, size_t n, hipStream_t st, bool apart) {
    ProfScope p("k_vmsm_multiples", st);
    hipLaunchKernelGGL(k_vmsm_multiples, dim3((unsigned)((n + 255) / 256), 2), dim3(256), apart ? verify_pad_bytes(2, (const void *)k_vmsm_multiples) : 0u, st,
                       pts_a, kind_a, tab_a, pts_b, kind_b, tab_b, tmp, pre, (uint32_t)n);
}

// ---- behind r: the scalars --------------------------------------------------------------------------------------------------------
// pw[k] = r^(2^k), k = 0..31, pw[32] = r^first (Montgomery form, the host's 64-bit field packs into the same limbs). Lane i:
// a = r^(first + i), b = a z_i (z_i as the per-blob pass left its bytes), both split as lo + hi z^2; sc_a / sc_b hold lo | hi, 32
// little-endian bytes per term: byte 16 h + j is the digit of row 16 h + j.
__global__ __launch_bounds__(64) void k_vmsm_scalars(const uint8_t *__restrict__ z_bytes, int le, const Fr *__restrict__ pw,
                                                     uint32_t *__restrict__ sc_a, uint32_t *__restrict__ sc_b, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr a = pw[32];
#pragma unroll 1
    for (int k = 0; k < 32; k++) {
        if ((i >> k) == 0) break;
        if ((i >> k) & 1u) a = a * pw[k];
    }
    Fr zraw;  // the canonical integer itself: a Montgomery product with a raw factor is the raw product
    if (le) raw_from_le<8>(zraw.l, z_bytes + 32 * (size_t)i);
    else raw_from_be<8>(zraw.l, z_bytes + 32 * (size_t)i);
    const Fr b = a * zraw;
    uint32_t ka[8], lo[4], hi[4];
    fe_to_raw<FrParams>(ka, a);
    split_by_z2_barrett(lo, hi, ka);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        sc_a[8 * (size_t)i + q] = lo[q];
        sc_a[8 * (size_t)i + 4 + q] = hi[q];
    }
    split_by_z2_barrett(lo, hi, b.l);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        sc_b[8 * (size_t)i + q] = lo[q];
        sc_b[8 * (size_t)i + 4 + q] = hi[q];
    }
}

void launch_vmsm_scalars(const uint8_t *z_bytes, int le, const Fr *pw, uint32_t *sc_a, uint32_t *sc_b, size_t n, hipStream_t st) {
    ProfScope p("k_vmsm_scalars", st);
    hipLaunchKernelGGL(k_vmsm_scalars, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, z_bytes, le, pw, sc_a, sc_b, (uint32_t)n);
}

// ---- behind r: bucket accumulation -------------------------------------------------------------------------------------------------
// grid (slices, 3): set 0 = sum a_i pi_i, set 1 = sum b_i pi_i, set 2 = sum a_i C_i. A workgroup sorts the (row, digit) entries of
// its slice of terms by digit (one LDS counter and one list per digit value) and lane b sums bucket b. A list that overflows (more
// than kVmsmListCap rows of one digit in one slice: 2^-40 for hashed scalars, reachable by chosen ones) sends its lane through the
// slice's entries in order instead. partial[(set * slices + slice) * 256 + b].
__global__ __launch_bounds__(256) void k_vmsm_accumulate(const uint32_t *__restrict__ sc_a, const uint32_t *__restrict__ sc_b,
                                                         const G1Affine29 *__restrict__ tab_p, const int32_t *__restrict__ kind_p,
                                                         const G1Affine29 *__restrict__ tab_c, const int32_t *__restrict__ kind_c,
                                                         G1Xyzz29 *__restrict__ partial, uint32_t n, uint32_t terms, uint32_t list_cap) {
    __shared__ uint32_t dig_w[kVmsmMaxTerms * 8];
    __shared__ uint32_t count[256];
    __shared__ uint16_t list[256][kVmsmListCap];
    __shared__ uint8_t live[kVmsmMaxTerms];
    const uint8_t *dig = (const uint8_t *)dig_w;
    const int set = blockIdx.y, tid = threadIdx.x;
    const uint32_t base = blockIdx.x * terms;
    const uint32_t m = n - base < terms ? n - base : terms;
    const uint32_t *sc = set == 1 ? sc_b : sc_a;
    const G1Affine29i *tab = (const G1Affine29i *)(set == 2 ? tab_c : tab_p);
    const int32_t *kind = set == 2 ? kind_c : kind_p;
    count[tid] = 0;
    for (uint32_t w = tid; w < m * 8; w += 256) dig_w[w] = sc[8 * (size_t)base + w];
    if ((uint32_t)tid < m) live[tid] = kind[base + tid] == 0;
    __syncthreads();
    for (uint32_t e = tid; e < m * 32; e += 256) {
        const uint32_t d = dig[e];
        if (d && live[e >> 5]) {
            const uint32_t slot = atomicAdd(&count[d], 1u);
            if (slot < list_cap) list[d][slot] = (uint16_t)e;
        }
    }
    __syncthreads();
    const uint32_t c = count[tid];
    G1Xyzz29i acc = G1Xyzz29i::infinity();
    if (c <= list_cap) {
#pragma unroll 1
        for (uint32_t s = 0; s < c; s++) {
            const uint32_t e = list[tid][s];
            const G1Affine29i q = tab[(size_t)(e & 31u) * n + base + (e >> 5)];
            acc = xyzz_madd(acc, q.x, q.y);
        }
    } else {
#pragma unroll 1
        for (uint32_t e = 0; e < m * 32; e++) {
            if (dig[e] != (uint32_t)tid || !live[e >> 5]) continue;
            const G1Affine29i q = tab[(size_t)(e & 31u) * n + base + (e >> 5)];
            acc = xyzz_madd(acc, q.x, q.y);
        }
    }
    partial[((size_t)set * gridDim.x + blockIdx.x) * 256 + tid] = *(const G1Xyzz29 *)&acc;
}

// terms per slice: about a chip's worth of waves in all (3 sets x slices x 4 waves ~ 1024), 8 .. kVmsmMaxTerms
uint32_t vmsm_terms_per_slice(size_t n) {
    size_t t = (3 * n + 255) / 256;
    if (t < 8) t = 8;
    if (t > (size_t)kVmsmMaxTerms) t = kVmsmMaxTerms;
    return (uint32_t)t;
}
size_t vmsm_slices(size_t n) {
    const uint32_t t = vmsm_terms_per_slice(n);
    return (n + t - 1) / t;
}
// the most slices any batch of up to `cap` terms cuts into (scratch sizing)
size_t vmsm_max_slices(size_t cap) {
    const size_t knee = ((size_t)256 * 8) / 3 + 1;  // up to here a slice is 8 terms
    size_t worst = (cap < knee ? cap : knee) / 8 + 2;
    const size_t big = cap / kVmsmMaxTerms + 2;
    if (big > worst) worst = big;
    return worst < 96 ? 96 : worst;
}

void launch_vmsm_accumulate(const uint32_t *sc_a, const uint32_t *sc_b, const G1Affine29 *tab_p, const int32_t *kind_p,
                            const G1Affine29 *tab_c, const int32_t *kind_c, G1Xyzz29 *partial, size_t n, hipStream_t st) {
    ProfScope p("k_vmsm_accumulate", st);
    const uint32_t terms = vmsm_terms_per_slice(n);
    const int lc = knobs().vmsm_list_cap;
    const uint32_t list_cap = lc > 0 && lc < kVmsmListCap ? (uint32_t)lc : (uint32_t)kVmsmListCap;
    hipLaunchKernelGGL(k_vmsm_accumulate, dim3((unsigned)vmsm_slices(n), 3), dim3(256), 0, st, sc_a, sc_b, tab_p, kind_p, tab_c, kind_c,
                       partial, (uint32_t)n, terms, list_cap);
}

// ---- behind r: the slices' sums per bucket (a wave per bucket) ----------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_vmsm_bucket_sums(const G1Xyzz29 *__restrict__ partial, G1Xyzz29 *__restrict__ bsum, uint32_t slices) {
    __shared__ G1Xyzz29i sh[64];   // (field products inlined: these kernels are dependent chains, a call boundary per product is pure latency)
    const int b = blockIdx.x + 1, set = blockIdx.y, t = threadIdx.x;
    const G1Xyzz29i *part = (const G1Xyzz29i *)partial;
    G1Xyzz29i acc = G1Xyzz29i::infinity();
    for (uint32_t s = t; s < slices; s += 64) acc = xyzz_add(acc, part[((size_t)set * slices + s) * 256 + b]);
    sh[t] = acc;
    __syncthreads();
    for (int d = 32; d >= 1; d >>= 1) {
        if (t < d) sh[t] = xyzz_add(sh[t], sh[t + d]);
        __syncthreads();
    }
    if (t == 0) bsum[set * 256 + b] = *(const G1Xyzz29 *)&sh[0];
}

// ---- behind r: sum_b b B_b = sum_{k >= 1} (sum_{b >= k} B_b): suffix scan, tree, and the affine big-endian result ---------------------
__global__ __launch_bounds__(256) void k_vmsm_weighted(const G1Xyzz29 *__restrict__ bsum, uint8_t *__restrict__ out96, int32_t *__restrict__ inf) {
    __shared__ G1Xyzz29i sh[256];
    const int set = blockIdx.x, b = threadIdx.x;
    G1Xyzz29i v = b >= 1 ? ((const G1Xyzz29i *)bsum)[set * 256 + b] : G1Xyzz29i::infinity();
    sh[b] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        G1Xyzz29i o = G1Xyzz29i::infinity();
        if (b + d < 256) o = sh[b + d];
        __syncthreads();
        v = xyzz_add(v, o);
        sh[b] = v;
        __syncthreads();
    }
    // sh[b] = sum_{k >= b} B_k; lane 0 holds the same as lane 1 (B_0 is empty) and stays out of the tree
    if (b == 0) sh[0] = G1Xyzz29i::infinity();
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (b < d) sh[b] = xyzz_add(sh[b], sh[b + d]);
        __syncthreads();
    }
    if (b != 0) return;
    const G1Xyzz29 total = *(const G1Xyzz29 *)&sh[0];
    uint8_t *o = out96 + 96 * set;
    if (total.is_inf()) {
        inf[set] = 1;
        for (int k = 0; k < 96; k++) o[k] = 0;
        return;
    }
    inf[set] = 0;
    const G1Affine a = xyzz_to_affine(total);
    uint32_t raw[12];
    fe_to_raw<FpParams>(raw, a.x);
    raw_to_be<12>(o, raw);
    fe_to_raw<FpParams>(raw, a.y);
    raw_to_be<12>(o + 48, raw);
}

void launch_vmsm_reduce(const G1Xyzz29 *partial, G1Xyzz29 *bsum, uint8_t *out96, int32_t *inf, size_t n, hipStream_t st) {
    {
        ProfScope p("k_vmsm_bucket_sums", st);
        hipLaunchKernelGGL(k_vmsm_bucket_sums, dim3(255, 3), dim3(64), 0, st, partial, bsum, (uint32_t)vmsm_slices(n));
    }
    ProfScope p("k_vmsm_weighted", st);
    hipLaunchKernelGGL(k_vmsm_weighted, dim3(3), dim3(256), 0, st, bsum, out96, inf);
}

// ---- the transcript, assembled where its pieces are (device-resident verification) -----------------------------------------------------
// records[160 i] = C_i (canonical, 48) | z_i (32) | y_i (32) | pi_i (canonical, 48): the message of the batch challenge r
// (/root/reference/src/utils.rs:166-206) in ONE buffer, so that one copy into pinned memory replaces four into pageable vectors plus the
// host's interleaving; first_bad = the lowest index whose status word is set (0xffffffff: none). One lane per 16-byte piece.
__global__ __launch_bounds__(256) void k_verify_records(const uint8_t *__restrict__ canon_c, const uint8_t *__restrict__ z32,
                                                        const uint8_t *__restrict__ y32, const uint8_t *__restrict__ canon_p,
                                                        const int32_t *__restrict__ status, uint4 *__restrict__ records,
                                                        uint32_t *__restrict__ first_bad, uint32_t n) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = g / 10, piece = g % 10;
    if (i >= n) return;
    const uint4 *src = piece < 3   ? (const uint4 *)(canon_c + 48 * (size_t)i) + piece
                       : piece < 5 ? (const uint4 *)(z32 + 32 * (size_t)i) + (piece - 3)
                       : piece < 7 ? (const uint4 *)(y32 + 32 * (size_t)i) + (piece - 5)
                                   : (const uint4 *)(canon_p + 48 * (size_t)i) + (piece - 7);
    records[10 * (size_t)i + piece] = *src;
    if (piece == 0 && status[i] != 0) atomicMin(first_bad, i);
}

void launch_verify_records(const uint8_t *canon_c, const uint8_t *z32, const uint8_t *y32, const uint8_t *canon_p, const int32_t *status,
                           uint8_t *records, uint32_t *first_bad, size_t n, hipStream_t st) {
    ProfScope p("k_verify_records", st);
    hipLaunchKernelGGL(k_verify_records, dim3((unsigned)((10 * n + 255) / 256)), dim3(256), 0, st, canon_c, z32, y32, canon_p, status,
                       (uint4 *)records, first_bad, (uint32_t)n);
}

}  // namespace lwk
