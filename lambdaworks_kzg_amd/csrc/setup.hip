// setup.hip -- trusted-setup load path on the GPU.
//
// Replaces the reference's per-point CPU loop in load_trusted_setup_file / load_trusted_setup
// (/root/reference/src/srs.rs:56-79, src/lib.rs:728-741): 4096 x (Fp sqrt + 255-bit scalar
// multiplication by r) is the slowest thing the reference does after the MSM and is embarrassingly
// parallel -- one lane per point here. Also replaces the per-call SRS rebuild
// kzgsettings_to_structured_reference_string (/root/reference/src/srs.rs:258-280): the device
// table is built once and cached in the settings' context.
#include "kernels.h"
#include "knobs.h"
#include "glv.cuh"

namespace lwk {

__global__ __launch_bounds__(64) void k_g1_decompress(const uint8_t *__restrict__ in48, G1Affine *__restrict__ out,
                                                      int32_t *__restrict__ status, size_t n, int subgroup_check) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t b[48];
    const uint32_t *src = (const uint32_t *)(in48 + 48 * i);
#pragma unroll
    for (int k = 0; k < 12; k++) {
        uint32_t w = src[k];
        b[4 * k] = (uint8_t)w;
        b[4 * k + 1] = (uint8_t)(w >> 8);
        b[4 * k + 2] = (uint8_t)(w >> 16);
        b[4 * k + 3] = (uint8_t)(w >> 24);
    }
    G1Affine p;
    p.x = Fp::zero();
    p.y = Fp::zero();
    int rc = g1_decompress_nocheck(p, b);
    if (rc == 0 && subgroup_check && !g1_in_subgroup(p)) rc = 2;
    out[i] = p;
    status[i] = rc;
}

void launch_g1_decompress(const uint8_t *in48, G1Affine *out, int32_t *status, size_t n, int subgroup_check,
                          hipStream_t st) {
    ProfScope p("k_g1_decompress", st);
    hipLaunchKernelGGL(k_g1_decompress, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, in48, out, status, n,
                       subgroup_check);
}

// reference blst_fp: 6 x u64, most-significant limb first, canonical integer
__device__ __forceinline__ void blst_fp_to_raw(uint32_t raw[12], const uint64_t *l) {
#pragma unroll
    for (int k = 0; k < 6; k++) {
        uint64_t v = l[5 - k];
        raw[2 * k] = (uint32_t)v;
        raw[2 * k + 1] = (uint32_t)(v >> 32);
    }
}
__device__ __forceinline__ void raw_to_blst_fp(uint64_t *l, const uint32_t raw[12]) {
#pragma unroll
    for (int k = 0; k < 6; k++) l[5 - k] = (uint64_t)raw[2 * k] | ((uint64_t)raw[2 * k + 1] << 32);
}

__global__ __launch_bounds__(64) void k_g1_from_blst(const uint64_t *__restrict__ in, G1Affine *__restrict__ out,
                                                     int32_t *__restrict__ status, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t *p = in + 18 * i;
    uint32_t rx[12], ry[12];
    blst_fp_to_raw(rx, p);
    blst_fp_to_raw(ry, p + 6);
    G1Affine a;
    a.x = fe_from_raw<FpParams>(rx);
    a.y = fe_from_raw<FpParams>(ry);
    out[i] = a;
    status[i] = g1_on_curve(a) ? 0 : 2;  // from_affine's curve check; z is ignored (srs.rs:155-172)
}

void launch_g1_from_blst(const uint64_t *blst_p1, G1Affine *out, int32_t *status, size_t n, hipStream_t st) {
    ProfScope p("k_g1_from_blst", st);
    hipLaunchKernelGGL(k_g1_from_blst, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, blst_p1, out, status, n);
}

__global__ __launch_bounds__(64) void k_g1_to_blst(const G1Affine *__restrict__ in, const int32_t *__restrict__ status,
                                                   uint64_t *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t *o = out + 18 * i;
    if (status && status[i] == 1) {  // neutral element: x = y = 0, z = [0,0,0,0,0,1] (srs.rs:132-140)
        for (int k = 0; k < 18; k++) o[k] = 0;
        o[17] = 1;
        return;
    }
    uint32_t raw[12];
    fe_to_raw<FpParams>(raw, in[i].x);
    raw_to_blst_fp(o, raw);
    fe_to_raw<FpParams>(raw, in[i].y);
    raw_to_blst_fp(o + 6, raw);
    for (int k = 12; k < 17; k++) o[k] = 0;
    o[17] = 1;  // z = 1 (from_affine)
}

void launch_g1_to_blst(const G1Affine *in, const int32_t *status, uint64_t *blst_p1, size_t n, hipStream_t st) {
    ProfScope p("k_g1_to_blst", st);
    hipLaunchKernelGGL(k_g1_to_blst, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, in, status, blst_p1, n);
}

// T[j][i] = 2^(13 j) * P_i, affine. One lane per point walks its 20 windows.
// Entries are stored in the hot loop's 29-bit-limb Montgomery form (field29.cuh), 112 B each.
__global__ __launch_bounds__(64) void k_build_table(const G1Affine *__restrict__ points,
                                                    G1Affine29 *__restrict__ table) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kBlobElems) return;
    G1Affine a = points[i];
    table[i] = affine_to_29(a);
    G1Xyzz cur = G1Xyzz::from_affine(a.x, a.y);
    for (int j = 1; j < kNumWindows; j++) {
        for (int d = 0; d < kWindowBits; d++) cur = xyzz_dbl(cur);
        // P has prime order r and 2^(13 j) is a unit mod r, so cur is never the point at infinity
        table[(size_t)j * kBlobElems + i] = affine_to_29(xyzz_to_affine(cur));
    }
}

constexpr int kLincombThreads = 256;
constexpr int kLincombPieces = 4;                               // 32-bit pieces of each 128-bit half scalar
constexpr int kLincombLanes = 2 * kLincombPieces;               // lanes per term: endomorphism split x pieces
constexpr int kLincombTerms = kLincombThreads / kLincombLanes;  // 32 terms per workgroup
constexpr int kPieceBits = 128 / kLincombPieces;

// mult[(j - 1) n + i] = [2^(32 j)] pts[i], j = 1 .. 3, affine, for the points the validation kernel accepted. Needs the
// points only, not the scalars: it runs right behind the validation, beside the per-blob pass, and lets k_lincomb3 cut
// every scalar into 32-bit pieces on lanes of their own (a lone wave needs ~8-10 us per group operation however idle
// the chip is, so the length of the serial chain is all that matters there).
// blockIdx.y selects one of two point sets (proofs / commitments in one launch).
__global__ __launch_bounds__(64) void k_point_multiples(const G1Affine29 *__restrict__ pts_a, const int32_t *__restrict__ kind_a,
                                                        G1Affine29 *__restrict__ mult_a, const G1Affine29 *__restrict__ pts_b,
                                                        const int32_t *__restrict__ kind_b, G1Affine29 *__restrict__ mult_b,
                                                        size_t n) {
    const G1Affine29 *pts = blockIdx.y ? pts_b : pts_a;
    const int32_t *kind = blockIdx.y ? kind_b : kind_a;
    G1Affine29 *mult = blockIdx.y ? mult_b : mult_a;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (kLincombPieces - 1) * n) return;
    const size_t i = gid % n;
    const int j = (int)(gid / n) + 1;
    if ((kind[i] & 0xff) != 0) {  // (bit 8 may carry a sign bit while the split validation is under way)
        mult[gid].x = F29<2>::zero();
        mult[gid].y = F29<2>::zero();
        return;
    }
    G1Affine29i p = ((const G1Affine29i *)pts)[i];
    G1Xyzz29i acc = G1Xyzz29i::from_affine(p.x, p.y);
    for (int k = 0; k < kPieceBits * j; k++) acc = xyzz_dbl(acc);
    if (acc.is_inf()) {  // only a point off the subgroup can double away; its batch is rejected by the validation
        mult[gid].x = F29<2>::zero();
        mult[gid].y = F29<2>::zero();
        return;
    }
    mult[gid] = xyzz29_to_affine29(*(G1Xyzz29 *)&acc);
}

unsigned verify_pad_bytes(int which, const void *kernel);  // sha256.hip

void launch_point_multiples(const G1Affine29 *pts, const int32_t *kind, G1Affine29 *mult, size_t n, hipStream_t st) {
    ProfScope p("k_point_multiples", st);
    hipLaunchKernelGGL(k_point_multiples, dim3((unsigned)(((kLincombPieces - 1) * n + 63) / 64)), dim3(64), verify_pad_bytes(2, (const void *)k_point_multiples), st, pts, kind, mult,
                       pts, kind, mult, n);
}

void launch_point_multiples2(const G1Affine29 *pts_a, const int32_t *kind_a, G1Affine29 *mult_a, const G1Affine29 *pts_b,
                             const int32_t *kind_b, G1Affine29 *mult_b, size_t n, hipStream_t st) {
    ProfScope p("k_point_multiples", st);
    hipLaunchKernelGGL(k_point_multiples, dim3((unsigned)(((kLincombPieces - 1) * n + 63) / 64), 2), dim3(64), verify_pad_bytes(2, (const void *)k_point_multiples), st, pts_a, kind_a,
                       mult_a, pts_b, kind_b, mult_b, n);
}

// The three linear combinations of verify_kzg_proof_batch (/root/reference/src/lib.rs:679-685) in ONE launch, on points
// the validation kernel already decompressed into the hot-loop representation. A 255-bit double-and-add is a serial
// chain of ~383 group operations on one lane, so the chain is cut twice. The curve endomorphism phi(x, y) = (beta x, y)
// acts on G1 as multiplication by -z^2:  [k]P = [lo]P + [hi](-phi(P)),  k = lo + hi z^2, two 128-bit halves. Each half
// is cut into four 32-bit pieces that multiply [2^(32 j)]P (k_point_multiples; phi commutes with the multiples). Eight
// lanes per term, a 32-bit double-and-add each (field products inlined: no call boundaries on the chain), then the
// workgroup sums its 256 results in LDS.
__global__ __launch_bounds__(kLincombThreads) void k_lincomb3(const G1Affine29 *__restrict__ proofs,
                                                              const int32_t *__restrict__ proof_kind,
                                                              const G1Affine29 *__restrict__ proof_mult,
                                                              const G1Affine29 *__restrict__ comms,
                                                              const int32_t *__restrict__ comm_kind,
                                                              const G1Affine29 *__restrict__ comm_mult,
                                                              const uint8_t *__restrict__ sc_r, const uint8_t *__restrict__ sc_rz,
                                                              G1Xyzz29 *__restrict__ partial, size_t n) {
    __shared__ G1Xyzz29 sh[kLincombThreads];
    const int tid = threadIdx.x, set = blockIdx.y, half = tid & 1, piece = (tid >> 1) % kLincombPieces;
    const size_t i = (size_t)blockIdx.x * kLincombTerms + tid / kLincombLanes;
    const G1Affine29 *pts = set == 2 ? comms : proofs;
    const G1Affine29 *mult = set == 2 ? comm_mult : proof_mult;
    const int32_t *kind = set == 2 ? comm_kind : proof_kind;
    const uint8_t *sc = set == 1 ? sc_rz : sc_r;
    G1Xyzz29i acc = G1Xyzz29i::infinity();
    if (i < n && kind[i] == 0) {
        G1Affine29i p = piece == 0 ? ((const G1Affine29i *)pts)[i] : ((const G1Affine29i *)mult)[(size_t)(piece - 1) * n + i];
        uint32_t k[8], lo[4], hi[4];
        raw_from_be<8>(k, sc + 32 * i);
        split_by_z2(lo, hi, k);
        if (half) {  // -phi(P) = (beta x, -y)
            uint32_t braw[12];
            g1_beta_raw(braw);
            const F29<2> b = f29_from_raw32(braw);
            F29<2, true> bi;
#pragma unroll
            for (int q = 0; q < 14; q++) bi.l[q] = b.l[q];
            p.x = p.x * bi;
            p.y = neg(p.y) * F29<1, true>::one();  // back to the < 2p form the affine slots carry
#pragma unroll
            for (int q = 0; q < 4; q++) lo[q] = hi[q];
        }
        static_assert(kPieceBits == 32, "one 32-bit word per lane");
        const uint32_t w = lo[piece];
        int bit = 31;
        while (bit >= 0 && !((w >> bit) & 1)) bit--;  // leading zeros: nothing to double yet
        for (; bit >= 0; bit--) {
            acc = xyzz_dbl(acc);
            if ((w >> bit) & 1) acc = xyzz_madd(acc, p.x, p.y);
        }
    }
    sh[tid] = *(G1Xyzz29 *)&acc;
    __syncthreads();
    for (int d = kLincombThreads / 2; d >= 1; d >>= 1) {
        if (tid < d) sh[tid] = xyzz_add(sh[tid], sh[tid + d]);
        __syncthreads();
    }
    if (tid == 0) partial[(size_t)set * gridDim.x + blockIdx.x] = sh[0];
}

size_t lincomb3_blocks(size_t n) { return (n + kLincombTerms - 1) / kLincombTerms; }

void launch_lincomb3(const G1Affine29 *proofs, const int32_t *proof_kind, const G1Affine29 *proof_mult, const G1Affine29 *comms,
                     const int32_t *comm_kind, const G1Affine29 *comm_mult, const uint8_t *sc_r_be, const uint8_t *sc_rz_be,
                     G1Xyzz29 *partial, size_t n, hipStream_t st) {
    ProfScope p("k_lincomb3", st);
    unsigned grid = (unsigned)lincomb3_blocks(n);
    hipLaunchKernelGGL(k_lincomb3, dim3(grid, 3), dim3(kLincombThreads), 0, st, proofs, proof_kind, proof_mult, comms, comm_kind,
                       comm_mult, sc_r_be, sc_rz_be, partial, n);
}

__global__ __launch_bounds__(64) void k_xyzz29_to_affine_be(const G1Xyzz29 *__restrict__ in, uint8_t *__restrict__ out96,
                                                            int32_t *__restrict__ inf, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Xyzz29 p = in[i];
    uint8_t *o = out96 + 96 * i;
    if (p.is_inf()) {
        inf[i] = 1;
        for (int k = 0; k < 96; k++) o[k] = 0;
        return;
    }
    inf[i] = 0;
    G1Affine a = xyzz_to_affine(p);
    uint32_t raw[12];
    fe_to_raw<FpParams>(raw, a.x);
    raw_to_be<12>(o, raw);
    fe_to_raw<FpParams>(raw, a.y);
    raw_to_be<12>(o + 48, raw);
}

void launch_xyzz29_to_affine_be(const G1Xyzz29 *in, uint8_t *out96, int32_t *inf, size_t n, hipStream_t st) {
    ProfScope p("k_xyzz29_to_affine_be", st);
    hipLaunchKernelGGL(k_xyzz29_to_affine_be, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, in, out96, inf, n);
}

void launch_build_table(const G1Affine *points, G1Affine29 *table, hipStream_t st) {
    ProfScope p("k_build_table", st);
    hipLaunchKernelGGL(k_build_table, dim3(kBlobElems / 64), dim3(64), 0, st, points, table);
}

}  // namespace lwk
