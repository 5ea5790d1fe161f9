// kernels.h -- host-side launchers of the HIP kernels (one per pipeline stage).
// Every launcher only enqueues work on `st`; nothing here synchronises or allocates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "g1.cuh"
#include "fr28.cuh"
#include "plan.h"

namespace lwk {

// ---- profiling hook (engine.hip): wraps a launch in hipEvents when profiling is on
struct ProfScope {
    ProfScope(const char *name, hipStream_t st);
    ~ProfScope();
    const char *name;
    hipStream_t st;
    hipEvent_t e0, e1;
    bool on;
};

// ---- scalar ingest (msm.hip)
// Mode R: 32-byte big-endian elements -> canonical 8xu32 little-endian limbs, reduced mod r
// (blob_to_polynomial, /root/reference/src/utils.rs:27-41).
void launch_parse_be_reduce(const uint8_t *blobs, uint32_t *scalars_raw, size_t n_elems, hipStream_t st, uint32_t *zero = nullptr, uint32_t zero_words = 0);
// Mode C: 32-byte little-endian elements, must be canonical (else status[blob] = BADARGS) -> Montgomery Fr

// ---- MSM (msm.hip)
void launch_digit_sort(const uint32_t *scalars_raw, uint32_t *sorted, uint32_t *bucket_start, uint32_t *perm,
                       size_t n_blobs, hipStream_t st);
void launch_bucket_accumulate(const G1Affine29 *table, const uint32_t *sorted, const uint32_t *bucket_start,
                              const uint32_t *perm, G1Xyzz29 *buckets, size_t n_blobs, hipStream_t st);
void launch_bucket_reduce(const G1Xyzz29 *buckets, G1Xyzz29 *sums, size_t n_blobs, hipStream_t st);
// total (+)= sum of n points (tiled MSM partial results)
void launch_sum_points(const G1Xyzz29 *in, size_t n, G1Xyzz29 *total, int accumulate, hipStream_t st);
// sums -> 48-byte compressed points (compress_g1_point, /root/reference/src/compression.rs:33-60)
void launch_finalize_compress(const G1Xyzz29 *sums, uint8_t *out48, size_t n, hipStream_t st);

// ---- direct fixed-base MSM (direct.hip): every multiple d * 2^(bits j) * P_i precomputed, no buckets.
// bits in 10 .. 16; the table has direct_table_entries(bits) rows of 112 bytes, kDirectRowAligned or kDirectRowPacked apart
// (packed: 6 / 11 / 21 / 36 / 68 / 135 / 240 GB); 0 rows for any other width.
constexpr int kDirectMinBits = 10, kDirectMaxBits = 16;
// bytes from one row of the direct table to the next: 128 = every row in a 128-byte line of its own (a gather touches one
// line; +2.5 % at 13 bits, +5 % at 15, +1.4 % at 16, profiles/r02_experiments.md section 9), or 112 = the affine point
// itself, packed (7 of 8 rows straddle two lines), when the aligned table would not leave headroom in HBM
constexpr size_t kDirectRowAligned = 128, kDirectRowPacked = 112;
constexpr size_t kDirectAlignedHeadroom = (size_t)8 << 30;  // what an aligned table must leave free (two workspaces, verification scratch, the caller's buffers)
size_t direct_table_entries(int bits);
int direct_num_windows(int bits);
// The table is ONE ALLOCATION PER WINDOW: a hipMalloc that needs memory released shortly before (by the previous process, or by the table
// this one replaces) waits for the driver's scrub of it, ~25 ms per GB (profiles/r03_alloc_pieces.txt), so window j + 1 is being allocated
// by the host while the GPU builds window j, and the build costs max(wait, kernels) instead of their sum. The kernels index `win_dev`, the windows' base addresses in device memory.
constexpr int kDirectMaxWindows = 26;  // 10-bit windows
struct DirectTable {
    void *win[kDirectMaxWindows] = {};  // rows of window j: 4096 * 2^(bits-1) (the top window: 4096 * 2^wtop) rows of row_bytes
    uint64_t *win_dev = nullptr;        // the same addresses on the device
    int nw = 0;
    size_t bytes = 0;
};
// ms[0] = scratch allocations, ms[1] = the windows' hipMallocs (summed), ms[2] = what was left of the build kernels after the
// last allocation returned, ms[3] = freeing the scratch (host wall clock; may be null). On failure nothing stays allocated.
// in_place: keep the window allocations `table` already has (same width, same row size) and only run the build kernels over them
hipError_t build_direct_table(int bits, const G1Affine *points, DirectTable &table, size_t row_bytes, hipStream_t st, double *ms = nullptr,
                              bool in_place = false);
void free_direct_table(DirectTable &table);
// sums[b] = sum_i scalars[b][i] * P_i. `partials` needs up to 64 * n_blobs entries when a blob is spread over several workgroups (unused
// otherwise); `lane_scratch` 4096 * n_blobs entries (the per-lane sums of the hand-scheduled kernel) and `redo` n_blobs words (its flags).
// fill: workgroups to aim for (0 = 512, the right number when the kernel has the chip alone; see direct.hip).
void launch_direct_msm(int bits, const uint64_t *win_dev, size_t row_bytes, const uint32_t *scalars_raw, G1Xyzz29 *lane_scratch,
                       G1Xyzz29 *partials, uint32_t *redo, G1Xyzz29 *sums, size_t n_blobs, hipStream_t st, int fill = 0,
                       uint32_t *redo_flag_out = nullptr);
uint32_t direct_one_blob_counter_words(int bits);   // r06: see direct.hip
// sums[b] recomputed for the blobs with only_if[b] != 0 (one workgroup each, complete branches; exits at once for the others)
void launch_direct_msm_only(int bits, const uint64_t *win_dev, size_t row_bytes, const uint32_t *scalars_raw, G1Xyzz29 *sums,
                            const uint32_t *only_if, size_t n_blobs, hipStream_t st);

// ---- setup (setup.hip)
// 48-byte compressed -> affine Montgomery + status (0 ok, 1 infinity, 2 invalid); optional [r]P check
void launch_g1_decompress(const uint8_t *in48, G1Affine *out, int32_t *status, size_t n, int subgroup_check,
                          hipStream_t st);
// reference blst_p1 (canonical big-endian-limb x, y; z ignored) -> affine Montgomery, curve check
// (blst_p1_to_g1_point, /root/reference/src/srs.rs:155-172)
void launch_g1_from_blst(const uint64_t *blst_p1, G1Affine *out, int32_t *status, size_t n, hipStream_t st);
// affine Montgomery -> reference blst_p1 layout (g1_point_to_blst_p1, /root/reference/src/srs.rs:131-153)
void launch_g1_to_blst(const G1Affine *in, const int32_t *status, uint64_t *blst_p1, size_t n, hipStream_t st);
// T[j][i] = 2^(13 j) P_i
void launch_build_table(const G1Affine *points, G1Affine29 *table, hipStream_t st);

// ---- Fr (fr_ops.hip)
// twiddle table: w^-k (inverse) and w^k (forward), k < 2048, Montgomery; built once on device
void launch_build_twiddles(Fr *tw_fwd, Fr *tw_inv, hipStream_t st);
// in-place-in-LDS 4096-point radix-2 DIT over Fr. Input is consumed in the order given
// (bit-reversed-order input -> natural-order output). `scale_raw_out`: if set, output is multiplied by
// 4096^-1 and written as canonical raw limbs; otherwise Montgomery.
void launch_twiddles_to28(const Fr *tw, Fr28 *tw28, hipStream_t st);
void launch_ntt4096(const Fr *in, Fr *out, const Fr28 *tw28, int inverse_scale_to_raw, size_t n_blobs, hipStream_t st);
void launch_blob_evaluations_to_coefficients(const uint8_t *blobs, uint32_t *coeffs_raw, const Fr28 *tw28_inv, int32_t *status,
                                             size_t n_blobs, hipStream_t st);
void launch_bitrev_permute(const Fr *in, Fr *out, size_t n_blobs, hipStream_t st);
// c-kzg mode on the Lagrange form of the setup (fr_ops.hip): the 4096 rows of inverse-DFT coefficients whose
// commitments are the Lagrange points; the copy + range check that replaces the transform in front of a commitment; and the forward
// transform of a proof's quotient when only the Lagrange-form table exists
void launch_idft_columns(uint32_t *coeffs_raw, const Fr *tw_inv, uint32_t first, size_t n_rows, hipStream_t st);
void launch_copy_le_check(const uint8_t *blobs, uint32_t *scalars_raw, int32_t *status, size_t n_blobs, hipStream_t st);
void launch_coefficients_to_evaluations(uint32_t *coeffs_raw, Fr *scratch, Fr *scratch2, const Fr28 *tw28_fwd, size_t n_blobs, hipStream_t st);
void launch_fr_be_to_mont(const uint8_t *in_be, Fr *out, size_t n_elems, hipStream_t st);
void launch_fr_mont_to_be(const Fr *in, uint8_t *out_be, size_t n_elems, hipStream_t st);
void launch_raw_to_be(const uint32_t *raw, uint8_t *out_be, size_t n_elems, hipStream_t st);
void launch_fr_mont_to_bytes(const Fr *in, uint8_t *out, int le, size_t n, hipStream_t st);
// y = p(z) and q = (p - y)/(x - z) per blob (Polynomial::evaluate + ruffini_division, call sites
// /root/reference/src/lib.rs:320,329,389,394). coeffs_raw/quot_raw: canonical limbs. y_out: 32 bytes,
// big-endian (le = 0) or little-endian (le = 1); may be NULL.
// only_if (optional): one word per blob; blobs whose word is zero are left as they are (a second pass over a few blobs)
// quot_raw = NULL: only y is computed (batch verification: a third of the products less)
void launch_eval_quotient(const uint32_t *coeffs_raw, const Fr *z_mont, uint32_t *quot_raw, uint8_t *y_out, int le,
                          size_t n_blobs, hipStream_t st, const uint32_t *only_if = nullptr);
// y = p(z) of reference-mode blobs straight from their bytes, any number in one launch (batch verification)
void launch_eval_y_from_blobs_be(const uint8_t *blobs, const Fr *z_mont, uint8_t *y_out, size_t n_blobs, hipStream_t st);
void launch_eval_y_from_blobs_evalform(const uint8_t *blobs, const Fr *z_mont, const Fr28 *roots_brp28, uint8_t *y_out, int32_t *status,
                                       size_t n_blobs, hipStream_t st);   // c-kzg mode: little-endian evaluations, range-checked
// the same in evaluation form (c-kzg mode on the Lagrange form): evaluations in, the quotient's evaluations out, y = p(z) by the barycentric formula
// roots_brp28: the 4096 domain points in element order (launch_roots_brp28 writes them from the transform's twiddles, once per setup)
void launch_roots_brp28(const Fr28 *tw28_fwd, Fr28 *roots_brp28, hipStream_t st);
void launch_eval_quotient_evalform(const uint32_t *evals_raw, const Fr *z_mont, const Fr28 *roots_brp28, uint32_t *quot_raw, uint8_t *y_out, int le,
                                   size_t n_blobs, hipStream_t st, const uint32_t *only_if = nullptr);
// flags[i] = (a[48 i ..] != b[48 i ..])
void launch_flag_differs48(const uint8_t *a, const uint8_t *b, uint32_t *flags, size_t n, hipStream_t st);
// z bytes -> Montgomery. le = 0: big-endian, reduced. le = 1: little-endian, must be canonical else BADARGS.
void launch_z_from_bytes(const uint8_t *z_bytes, Fr *z_mont, int32_t *status, int le, size_t n, hipStream_t st);

// ---- Fiat-Shamir (sha256.hip)
// validate + canonicalise commitments (decompress incl. subgroup check, recompress), then
// z = sha256("FSBLOBVERIFY_V1_" | le64(4096) | le64(0) | blob | commitment) as Fr
// (compute_challenge, /root/reference/src/utils.rs:120-154).
// verdict_scratch (n words, with aff_out and kind_out): the validation runs as three launches with the subgroup test on a quad of
// lanes per point (k_subgroup_coop_asm) instead of one lane per point for the whole chain
// apart (r06): the launches carry an LDS footprint they never touch, sized so that the dispatcher cannot place their workgroups on the
// compute units of a challenge-hash kernel running beside them (knobs.h: verify_pad_kb; profiles/r06_experiments.md section 1)
void launch_validate_commitments(const uint8_t *comm48, uint8_t *canon48, int32_t *status, int bad_code, size_t n,
                                 hipStream_t st, G1Affine29 *aff_out = nullptr, int32_t *kind_out = nullptr, uint32_t *verdict_scratch = nullptr,
                                 bool apart = false);
// verify side: three variable-base linear combinations in one launch (setup.hip).
//   set 0 = sum r_i P_i, set 1 = sum rz_i P_i (P = proofs), set 2 = sum r_i C_i (C = commitments);
// per-block partial sums to partial[set * nblk + block]
size_t lincomb3_blocks(size_t n);  // workgroups (= partial sums) per set
// *_mult: [2^32]P, [2^64]P, [2^96]P of every point (3 n entries, launch_point_multiples), so that each scalar is cut
// into 32-bit pieces on lanes of their own
void launch_point_multiples(const G1Affine29 *pts, const int32_t *kind, G1Affine29 *mult, size_t n, hipStream_t st);
void launch_point_multiples2(const G1Affine29 *pts_a, const int32_t *kind_a, G1Affine29 *mult_a, const G1Affine29 *pts_b,
                             const int32_t *kind_b, G1Affine29 *mult_b, size_t n, hipStream_t st);  // two sets, one launch
// launch_validate_commitments in two launches (sha256.hip): the multiples above can start after the first
void launch_decompress_points(const uint8_t *in48, G1Affine29 *pts, int32_t *kind, size_t n, hipStream_t st, bool apart = false);
void launch_subgroup_canon(G1Affine29 *pts, int32_t *kind, uint8_t *canon48, int32_t *status, int bad_code, size_t n,
                           hipStream_t st, uint32_t *verdict_scratch = nullptr, bool apart = false);
// the same two steps for both point sets of a verification at once (one launch per kernel: r06)
void launch_decompress_points2(const uint8_t *in48_a, G1Affine29 *pts_a, int32_t *kind_a, const uint8_t *in48_b, G1Affine29 *pts_b,
                               int32_t *kind_b, size_t n, hipStream_t st, bool apart = false);
void launch_subgroup_canon2(G1Affine29 *pts_a, int32_t *kind_a, uint8_t *canon48_a, uint32_t *verdict_a, G1Affine29 *pts_b, int32_t *kind_b,
                            uint8_t *canon48_b, uint32_t *verdict_b, int32_t *status, int bad_code, size_t n, hipStream_t st, bool apart = false);
void launch_lincomb3(const G1Affine29 *proofs, const int32_t *proof_kind, const G1Affine29 *proof_mult, const G1Affine29 *comms,
                     const int32_t *comm_kind, const G1Affine29 *comm_mult, const uint8_t *sc_r_be, const uint8_t *sc_rz_be,
                     G1Xyzz29 *partial, size_t n, hipStream_t st);
// ---- the verification's linear combinations as one latency-shaped bucket MSM (vmsm.hip, r06) ----------------------------------
constexpr int kVmsmDigits = 16;                  // byte digits of a 128-bit half scalar
constexpr int kVmsmRows = 2 * kVmsmDigits;       // rows per point: [2^(8 j)]P and [2^(8 j)](-phi(P)), j = 0..15
constexpr int kVmsmSteps = kVmsmDigits - 1;      // rows per half beyond the point itself
constexpr int kVmsmMaxTerms = 64;                // terms per slice (one workgroup of k_vmsm_accumulate), at most
constexpr size_t kVmsmPinBytes = 33 * 32 + 3 * 96 + 3 * 4 + 52;  // the pinned host block of a verification: powers up, sums down
constexpr int kVmsmListCap = 24;                 // rows of one digit value a slice lists before its lane falls back to a scan
// rows of two point sets in one launch, beside the challenge hash: tab_*[row * n + i]; tmp: 15 x 2n XYZZ, pre: 15 x 2n field elements
void launch_vmsm_multiples2(const G1Affine29 *pts_a, const int32_t *kind_a, G1Affine29 *tab_a, const G1Affine29 *pts_b,
                            const int32_t *kind_b, G1Affine29 *tab_b, G1Xyzz29 *tmp, F29<2> *pre, size_t n, hipStream_t st, bool apart = false);
// pw: 33 Fr in Montgomery form, r^(2^k) for k = 0..31 and r^first; sc_a / sc_b: 8 words per term (lo | hi of a_i = r^(first+i), b_i = a_i z_i)
void launch_vmsm_scalars(const uint8_t *z_bytes, int le, const Fr *pw, uint32_t *sc_a, uint32_t *sc_b, size_t n, hipStream_t st);
uint32_t vmsm_terms_per_slice(size_t n);
size_t vmsm_slices(size_t n);
size_t vmsm_max_slices(size_t cap);
// partial: 3 x vmsm_slices(n) x 256 XYZZ; list_cap <= kVmsmListCap (smaller only to force the overflow path in tests)
void launch_vmsm_accumulate(const uint32_t *sc_a, const uint32_t *sc_b, const G1Affine29 *tab_p, const int32_t *kind_p,
                            const G1Affine29 *tab_c, const int32_t *kind_c, G1Xyzz29 *partial, size_t n, hipStream_t st);
// bsum: 3 x 256 XYZZ; out96 / inf: the three sums, affine big-endian x | y and an infinity flag each (what launch_xyzz29_to_affine_be leaves)
void launch_vmsm_reduce(const G1Xyzz29 *partial, G1Xyzz29 *bsum, uint8_t *out96, int32_t *inf, size_t n, hipStream_t st);
// records[160 i] = C_i | z_i | y_i | pi_i from the device-resident pieces; *first_bad = lowest index with a status word set (pre-set to 0xffffffff)
void launch_verify_records(const uint8_t *canon_c, const uint8_t *z32, const uint8_t *y32, const uint8_t *canon_p, const int32_t *status,
                           uint8_t *records, uint32_t *first_bad, size_t n, hipStream_t st);
void launch_xyzz29_to_affine_be(const G1Xyzz29 *in, uint8_t *out96, int32_t *inf, size_t n, hipStream_t st);
void launch_challenge(const uint8_t *blobs, const uint8_t *canon48, Fr *z_mont, int le, size_t n, hipStream_t st,
                      const uint8_t *only_if_differs_from = nullptr);
void launch_challenge_midstate(const uint8_t *blobs, uint32_t *midstate, size_t n, hipStream_t st);
void launch_challenge_finish(const uint8_t *blobs, const uint8_t *canon48, const uint32_t *midstate, Fr *z_mont, int le, size_t n,
                             hipStream_t st);
void sha256_host(uint8_t out[32], const uint8_t *msg, size_t len);
// sha256_host.hip: digests[i] = SHA-256("FSBLOBVERIFY_V1_" | le64(4096) | le64(0) | blobs[i] | comms[i]) on host threads
void challenge_digests_host(uint8_t *digests32, const uint8_t *blobs, const uint8_t *comms48, size_t n);
double host_hash_rate();   // bytes per second the host threads hashed in their recent long jobs (sha256_host.hip)
void challenge_midstates_host(uint32_t *mid, const uint8_t *blobs, size_t n);   // 8 words per blob: the hash state k_challenge_finish continues from
void sha256_blocks_portable(uint32_t h[8], const uint8_t *blocks, size_t n_blocks);

}  // namespace lwk
