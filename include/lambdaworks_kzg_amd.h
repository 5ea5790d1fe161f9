/*
 * lambdaworks_kzg_amd.h -- C ABI of the MI355X-native KZG / EIP-4844 blob-commitment engine.
 *
 * Drop-in boundary for lambdaclass/lambdaworks_kzg's hot path. The first nine functions are
 * exactly the reference's `#[no_mangle] extern "C"` symbols (the c-kzg-4844 surface); a
 * maintainer links this library instead of liblambdaworks_kzg and keeps calling them.
 * Header of record on the reference side: src/c_kzg_4844.h:85-231 (NOT the stale
 * src/lambdaworks_kzg.h). Each declaration cites the Rust definition it replaces.
 *
 * Everything after "extensions" is additive: batched and device-resident entry points (a
 * synchronous one-blob call cannot reach 10k ops/s), the R/C semantics switch, multi-GPU setup
 * hand-off, and profiling hooks used by bench.py.
 *
 * Struct layouts follow the reference's #[repr(C)] types (src/lib.rs:45-232). NOTE the
 * reference's blst_fp convention, which this library reproduces bit for bit and which is NOT
 * blst's: limbs hold the canonical (non-Montgomery) integer, most-significant limb first
 * (src/srs.rs:131-172; the reference's own test compares these structs, tests/lib_test.rs:153-163).
 */
#ifndef LAMBDAWORKS_KZG_AMD_H
#define LAMBDAWORKS_KZG_AMD_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden; these are its only exported symbols */
#pragma GCC visibility push(default)

#ifndef FIELD_ELEMENTS_PER_BLOB
#define FIELD_ELEMENTS_PER_BLOB 4096 /* fixed in the reference: src/lib.rs:76 */
#endif
#if FIELD_ELEMENTS_PER_BLOB != 4096
#error this engine is built for FIELD_ELEMENTS_PER_BLOB = 4096 (reference src/lib.rs:76)
#endif

#define BYTES_PER_COMMITMENT 48    /* src/lib.rs:67 */
#define BYTES_PER_PROOF 48         /* src/lib.rs:70 */
#define BYTES_PER_FIELD_ELEMENT 32 /* src/lib.rs:73 */
#define BYTES_PER_BLOB (FIELD_ELEMENTS_PER_BLOB * BYTES_PER_FIELD_ELEMENT) /* src/lib.rs:81 */
#define TRUSTED_SETUP_NUM_G1_POINTS FIELD_ELEMENTS_PER_BLOB                /* src/lib.rs:78 */
#define TRUSTED_SETUP_NUM_G2_POINTS 65                                     /* src/lib.rs:92 */

/* ---- blst-shaped types (src/lib.rs:100-166). Guarded so a real blst.h may come first. ---- */
#ifndef LWKZG_HAVE_BLST_TYPES
#ifndef __BLST_H__
typedef uint64_t limb_t;
typedef struct { limb_t l[256 / 8 / sizeof(limb_t)]; } blst_fr;
typedef struct { limb_t l[384 / 8 / sizeof(limb_t)]; } blst_fp;
typedef struct { blst_fp fp[2]; } blst_fp2;
typedef struct { blst_fp x, y, z; } blst_p1;
typedef struct { blst_fp x, y; } blst_p1_affine;
typedef struct { blst_fp2 x, y, z; } blst_p2;
typedef struct { blst_fp2 x, y; } blst_p2_affine;
#endif
#endif

typedef blst_p1 g1_t;
typedef blst_p2 g2_t;
typedef blst_fr fr_t;

typedef struct { uint8_t bytes[32]; } Bytes32;             /* src/lib.rs:94 */
typedef struct { uint8_t bytes[48]; } Bytes48;             /* src/lib.rs:95 */
typedef struct { uint8_t bytes[BYTES_PER_BLOB]; } Blob;    /* src/lib.rs:98 */
typedef Bytes48 KZGCommitment;                             /* src/lib.rs:96 */
typedef Bytes48 KZGProof;                                  /* src/lib.rs:97 */

typedef enum {
    C_KZG_OK = 0,  /* success */
    C_KZG_BADARGS, /* the supplied data is invalid in some way */
    C_KZG_ERROR,   /* internal error; the reference maps EVERY failure to this (src/lib.rs:263,267,272,...) */
    C_KZG_MALLOC,  /* could not allocate memory */
} C_KZG_RET;       /* src/lib.rs:45-57 */

typedef struct {
    uint64_t max_width;
    fr_t *expanded_roots_of_unity; /* w^i, i = 0..max_width (max_width + 1 entries) */
    fr_t *reverse_roots_of_unity;  /* w^-i, max_width + 1 entries */
    fr_t *roots_of_unity;          /* w^bitrev(i), max_width entries */
} FFTSettings;                     /* src/lib.rs:173-197 */

typedef struct {
    FFTSettings *fs;  /* reference: always NULL (src/lib.rs:754-758). Here: the engine's context; its
                         first member is a genuine FFTSettings (max_width = 4096). */
    g1_t *g1_values;  /* 4096 monomial points [tau^i]G, reference blst_fp convention, malloc'd */
    g2_t *g2_values;  /* 65 points [tau^i]G2, malloc'd */
} KZGSettings;        /* src/lib.rs:206-222 */

/* ============================ the reference's nine symbols ============================ */

/* src/lib.rs:709-776. n1 != 4096 or n2 != 65 -> C_KZG_BADARGS (the reference's only BADARGS). */
C_KZG_RET load_trusted_setup(KZGSettings *out, const uint8_t *g1_bytes /* n1*48 */, size_t n1,
                             const uint8_t *g2_bytes /* n2*96 */, size_t n2);

/* src/lib.rs:779-802 + src/srs.rs:25-128: line-based text format, decimal n1, n2, then hex points. */
C_KZG_RET load_trusted_setup_file(KZGSettings *out, FILE *in);

/* src/lib.rs:821-829. (c_kzg_4844.h:186 declares void; the Rust symbol returns C_KZG_OK.) */
C_KZG_RET free_trusted_setup(KZGSettings *s);

/* src/lib.rs:253-283 */
C_KZG_RET blob_to_kzg_commitment(KZGCommitment *out, const Blob *blob, const KZGSettings *s);

/* src/lib.rs:300-344 */
C_KZG_RET compute_kzg_proof(KZGProof *proof_out, Bytes32 *y_out, const Blob *blob, const Bytes32 *z_bytes,
                            const KZGSettings *s);

/* src/lib.rs:361-404 */
C_KZG_RET compute_blob_kzg_proof(KZGProof *out, const Blob *blob, const Bytes48 *commitment_bytes,
                                 const KZGSettings *s);

/* src/lib.rs:407-453 */
C_KZG_RET verify_kzg_proof(bool *ok, const Bytes48 *commitment_bytes, const Bytes32 *z_bytes, const Bytes32 *y_bytes,
                           const Bytes48 *proof_bytes, const KZGSettings *s);

/* src/lib.rs:456-505 */
C_KZG_RET verify_blob_kzg_proof(bool *ok, const Blob *blob, const Bytes48 *commitment_bytes,
                                const Bytes48 *proof_bytes, const KZGSettings *s);

/* src/lib.rs:525-614. n == 0 -> C_KZG_OK with *ok = false in reference mode (src/lib.rs:538-543) and *ok = true in
 * c-kzg mode (the reference's own vector verify_blob_kzg_proof_batch_case_a271b78b8e869d69).
 * Batches of more than 1024 blobs are uploaded into a device buffer of n * 131072 bytes that the settings object keeps (grow-only, freed by
 * free_trusted_setup): 4096 blobs 12.1 ms, of which 9.6 are the upload; lwkzg_verify_blob_kzg_proof_batch_device takes blobs that are
 * already in HBM (5.3 ms). */
C_KZG_RET verify_blob_kzg_proof_batch(bool *ok, const Blob *blobs, const Bytes48 *commitments_bytes,
                                      const Bytes48 *proofs_bytes, size_t n, const KZGSettings *s);

/* ==================================== extensions ==================================== */

/* Semantics switch.
 *   LWKZG_MODE_REFERENCE (default): what lambdaworks_kzg computes -- big-endian scalars are monomial
 *       coefficients (src/utils.rs:27-41), big-endian z/y and Fiat-Shamir digest, every failure is
 *       C_KZG_ERROR, scalars >= r are reduced.
 *   LWKZG_MODE_CKZG: what the c-kzg-4844 vectors under the reference's tests/ encode -- canonical
 *       little-endian scalars are evaluations on the bit-reversed 4096th roots of unity (inverse NTT
 *       in front of the same MSM), little-endian z/y/digest, invalid input is C_KZG_BADARGS.
 * lwkzg_set_mode sets the process-wide DEFAULT (initial value from the environment variable LWKZG_MODE,
 * "reference"|"ckzg"); lwkzg_settings_set_mode gives ONE settings object a mode of its own, which wins over the default
 * (mode -1 hands it back to the default), so two consumers in one process can use different semantics. Every entry
 * point resolves its mode once, when it is entered: a change does not affect calls already in flight.
 * Cost of lwkzg_settings_set_mode: it brings the settings' MSM tables to the new mode's form (lwkzg_direct_table_forms below). The
 * first switch to c-kzg mode derives the Lagrange form of the setup on the device (about 50 ms) and builds a second direct table
 * beside the first if it fits (default engine: 0.2-0.3 s and 41 GB more); at 15 / 16 bits, where only one table fits, the table is
 * REBUILT in the new form (what lwkzg_enable_direct_table of that width costs: 0.9 s of kernels at 16 bits plus the driver's
 * hipMalloc waits). Switching back and forth between two forms that both exist is free. lwkzg_set_mode never touches a table: a
 * settings object that merely follows the default gets the Lagrange form on its first c-kzg call, and only if it fits beside. */
#define LWKZG_MODE_REFERENCE 0
#define LWKZG_MODE_CKZG 1
int lwkzg_set_mode(int mode); /* returns the previous default, or -1 if `mode` is invalid */
int lwkzg_get_mode(void);
int lwkzg_settings_set_mode(const KZGSettings *s, int mode); /* returns the mode calls on `s` answered in before, -1 on error */
int lwkzg_settings_get_mode(const KZGSettings *s);           /* the mode a call on `s` would answer in now */

/* For a KZGSettings filled in by hand (fs == NULL and caller-owned g1_values / g2_values, the reference's own layout,
 * src/lib.rs:754-758): the library builds a device context for it on first use and caches it by the g1_values pointer
 * (re-checked against a digest of the first and last point). free_trusted_setup() would free() the caller's arrays; this
 * releases only the cached device context. C_KZG_BADARGS for a setup the library loaded itself. */
C_KZG_RET lwkzg_release_context(const KZGSettings *s);

/* Batched host-pointer forms of src/lib.rs:253 / :361 / :300: n blobs in, n results out, one launch
 * set. On failure nothing useful is in `out`; `first_bad` (may be NULL) receives the index of the
 * first offending blob. */
C_KZG_RET lwkzg_blob_to_kzg_commitment_batch(KZGCommitment *out, const Blob *blobs, size_t n, const KZGSettings *s,
                                             size_t *first_bad);
C_KZG_RET lwkzg_compute_blob_kzg_proof_batch(KZGProof *out, const Blob *blobs, const Bytes48 *commitments, size_t n,
                                             const KZGSettings *s, size_t *first_bad);
C_KZG_RET lwkzg_compute_kzg_proof_batch(KZGProof *proofs_out, Bytes32 *ys_out, const Blob *blobs, const Bytes32 *zs,
                                        size_t n, const KZGSettings *s, size_t *first_bad);

/* Device-resident forms: every pointer is a DEVICE pointer on the settings' GPU; `stream` is a
 * hipStream_t (NULL = the engine's own stream). Asynchronous: kernels are enqueued and the call
 * returns; per-blob status words (0 = ok, C_KZG_RET otherwise) are written to status_dev (n x int32,
 * may be NULL). No allocation happens here once lwkzg_reserve() has been called for >= n (proof calls of more than
 * 1024 blobs allocate n x 84 bytes once; the first blob-proof call of up to LWKZG_SMALL_PROOF_HOST = 128 blobs allocates its
 * pinned staging, 128 KiB per blob, and synchronises the device once -- such a call takes its Fiat-Shamir challenges and its
 * commitment validation from the host threads, in stream order through a host function, because the GPU's two latency
 * chains cost 3.6 ms however few the blobs are; the call still returns without waiting. Calls of up to LWKZG_MID_PROOF_HOST = 384
 * blobs on a settings object whose other caller stream is idle hash on the host too, in chunks beside the copy out, while the GPU
 * validates: 4.9 instead of 6.1 ms at 256 blobs). Calls on one settings object share its workspace: whatever `stream` each is
 * given, the library orders their GPU work one after the other (event dependencies, no host blocking), so two calls
 * in flight on two streams are safe and serial. A proof call hashes and validates ALL its blobs up front (two latency
 * chains of ~3 ms whose duration does not depend on n), then runs the MSMs chunk by chunk: one call of 4096 blobs
 * pays them once, four calls of 1024 four times. */
C_KZG_RET lwkzg_blob_to_kzg_commitment_batch_device(void *out48_dev, const void *blobs_dev, size_t n,
                                                    const KZGSettings *s, void *stream, int32_t *status_dev);
C_KZG_RET lwkzg_compute_blob_kzg_proof_batch_device(void *out48_dev, const void *blobs_dev, const void *commitments48_dev,
                                                    size_t n, const KZGSettings *s, void *stream, int32_t *status_dev);
/* Commitment AND blob proof of n device-resident blobs in one pass: commitments48_dev[i] = blob_to_kzg_commitment(blob_i),
 * proofs48_dev[i] = compute_blob_kzg_proof(blob_i, commitments48_dev[i]) -- src/lib.rs:253-283 followed by src/lib.rs:361-404
 * on its own output, byte for byte what the two calls above return. What a blob producer needs, ~15 % faster than the two
 * calls: the part of the Fiat-Shamir hash that does not depend on the commitment runs beside the commitment MSM, and the
 * blob is parsed once. status_dev as above. */
C_KZG_RET lwkzg_commit_and_prove_batch_device(void *commitments48_dev, void *proofs48_dev, const void *blobs_dev, size_t n,
                                              const KZGSettings *s, void *stream, int32_t *status_dev);
C_KZG_RET lwkzg_reserve(const KZGSettings *s, size_t max_batch);
/* lwkzg_reserve covers calls on ONE caller stream. A device-resident call that arrives on a second stream while the
 * workspace is busy runs on a second context of the settings object (own streams and workspace, about 1.8 MB per blob of
 * max_batch, over the same tables); unless reserved here it is created inside that first overlapped call, which then
 * allocates and synchronises the device once. caller_streams >= 2 creates and reserves it now. Footprint of a load beside
 * this: the default engine's table (41 GB when the device is empty, see lwkzg_enable_direct_table below). */
C_KZG_RET lwkzg_reserve_streams(const KZGSettings *s, size_t max_batch, int caller_streams);

/* "Direct" fixed-base MSM for this settings object: trade HBM capacity for arithmetic. With every
 * multiple d * 2^(window_bits * j) * P_i of every setup point resident (window_bits 10 .. 16: 6, 11, 21, 36, 68,
 * 135, 240 GB with packed rows, 8/7 of that with the rows aligned to 128-byte lines, see lwkzg_direct_row_bytes), the 4096-term MSM behind every entry point above becomes 4096 * ceil(255 / window_bits) gathered
 * mixed additions (26, 24, 22, 20, 19, 17, 16 per scalar): no digit sort, no buckets, no bucket reduction. Results
 * are bit-identical between all widths and the bucket engine. window_bits = 0 frees the table and selects the bucket
 * engine (9 MB table, 20 additions per scalar plus sort and reduction).
 * Returns C_KZG_MALLOC (engine unchanged and still usable) when the table does not fit, C_KZG_BADARGS for
 * other widths. Replaces nothing in the reference: lambdaworks' pippenger::msm (call sites src/lib.rs:242,270,329,394)
 * has no precomputation at all.
 * What a load_trusted_setup* call selects by itself (for consumers that only know the nine reference symbols):
 *   LWKZG_DIRECT_BITS unset  the DEFAULT engine: the widest table of 13 .. 10 bits that takes at most a quarter of the
 *                            device memory free at load time (13 bits on an empty MI355X), else the bucket engine;
 *   LWKZG_DIRECT_BITS=0      the bucket engine;   =10..16  that width (bucket engine if it does not fit);
 *   LWKZG_DIRECT_BITS=auto   the widest of 16 .. 10 that fits. */
C_KZG_RET lwkzg_enable_direct_table(const KZGSettings *s, int window_bits);
int lwkzg_direct_table_bits(const KZGSettings *s);   /* 0 = bucket engine, 10..16 = direct table live, -1 = bad settings */
int lwkzg_direct_num_windows(int window_bits);       /* additions per scalar on the direct path (0 for other widths) */
/* The forms the live direct table(s) are in: bit 0 = monomial ([tau^i]G: reference mode, and every proof's quotient), bit 1 =
 * Lagrange ([l_i(tau)]G in the blob's own order: a c-kzg commitment is then an MSM over the blob's evaluations as they stand, no
 * transform -- SURVEY Appendix D; the reference left this conversion commented out, src/lib.rs:760-770, src/srs.rs:117-124).
 * The table is built in the form of the mode the settings answer in at that moment (lwkzg_enable_direct_table, the load's own
 * choice, lwkzg_settings_set_mode) and in the other form as well when that fits beside it with 8 GiB to spare (always up to 14 bits
 * on an empty MI355X, never at 15 / 16); the Lagrange form only once the settings have answered in c-kzg mode (a settings object that
 * merely follows the process default gets it at its first c-kzg call, and the second table then only within a quarter of the free
 * memory, the rule the load picks its own table by). No result depends on
 * it: without a Lagrange table a c-kzg commitment pays the inverse transform (k_ntt4096, +4 %); without a monomial table a c-kzg proof
 * pays one forward transform of its quotient, and reference mode runs on the monomial buckets. 0 = bucket engine, -1 = bad settings. */
int lwkzg_direct_table_forms(const KZGSettings *s);
/* lwkzg_enable_direct_table with the forms named: 1 = monomial only, 2 = Lagrange only (a consumer that only ever speaks c-kzg and wants
 * the widest table: 16 bits = 275 GB in ONE form), 3 = both or C_KZG_MALLOC. C_KZG_BADARGS for anything else. */
C_KZG_RET lwkzg_enable_direct_table_forms(const KZGSettings *s, int window_bits, int forms);
/* Bytes from one table row to the next: 128 when every row got a 128-byte line of its own (chosen whenever that table
 * leaves 8 GiB of the device free: one line per gather, +2.5 % / +5 % / +1.4 % at 13 / 15 / 16 bits), 112 when the rows are
 * packed (the sizes quoted above); 0 = bucket engine, -1 = bad settings. LWKZG_DIRECT_ROW=112|128 forces one. */
int lwkzg_direct_row_bytes(const KZGSettings *s);

/* The Fiat-Shamir challenges of device-resident blobs: z_i = compute_challenge(blob_i, commitment_i)
 * (src/utils.rs:120-154) as n x 32 bytes in the mode's byte order, canonical. This is the hash kernel of
 * lwkzg_compute_blob_kzg_proof_batch_device on its own (test / pipeline hook); the commitment bytes are hashed as given. */
C_KZG_RET lwkzg_compute_challenges_device(void *z32_dev, const void *blobs_dev, const void *commitments48_dev, size_t n,
                                          const KZGSettings *s, void *stream);

/* General G1 multi-scalar multiplication against the first `npoints` setup points:
 * scalars_dev = n_msm x npoints x 32 bytes (big-endian, reduced mod r), out = n_msm x 48 bytes compressed.
 * (reference: g1_lincomb, src/lib.rs:234-243, over srs.powers_main_group.) */
C_KZG_RET lwkzg_g1_lincomb_setup_device(void *out48_dev, const void *scalars_be_dev, size_t n_msm, const KZGSettings *s,
                                        void *stream);

/* Long MSM over the setup tiled end to end: out = sum_k scalars[k] * g1[k mod 4096], n_terms a positive multiple of
 * 4096 (e.g. 2^20), scalars big-endian 32-byte, device-resident; out = 48 bytes compressed (device pointer).
 * One fixed-base 4096-term MSM per tile plus one final sum. */
C_KZG_RET lwkzg_g1_msm_tiled_device(void *out48_dev, const void *scalars_be_dev, size_t n_terms, const KZGSettings *s,
                                    void *stream);
/* Host helper for sharded long MSMs: out = sum of n compressed G1 points (48 bytes each; infinity allowed). */
C_KZG_RET lwkzg_g1_sum_compressed(uint8_t out48[48], const uint8_t *points48, size_t n);

/* Batched Fr transform on device-resident data: n vectors of 4096 big-endian 32-byte elements,
 * natural order in and out; inverse != 0 scales by 4096^-1. */
C_KZG_RET lwkzg_fr_ntt4096_device(void *out_dev, const void *in_dev, size_t n, int inverse, const KZGSettings *s,
                                  void *stream);

/* Multi-GPU hand-off of a loaded setup (one process per GPU; rank 0 loads, the others import what an
 * RCCL broadcast delivered). The image is one contiguous DEVICE buffer. */
size_t lwkzg_setup_image_bytes(void);
C_KZG_RET lwkzg_setup_export_device(const KZGSettings *s, void *image_dev, void *stream);
C_KZG_RET lwkzg_setup_import_device(KZGSettings *out, const void *image_dev);

/* ONE process, SEVERAL GPUs (csrc/multi.hip): the node-level form of the batch entry points for the reference's own kind of caller
 * -- plain C calls, /root/reference/fuzz/base_fuzz.h:17-34, src/lib.rs:253-283 -- which has no torch.distributed. lwkzg_multi_load*
 * parses, validates and prepares the setup on devices[0] exactly as load_trusted_setup* does (src/lib.rs:709-802), delivers the
 * 10.3 MB setup image to the other devices device-to-device (hipMemcpyPeer: xGMI on an MI355X node) and lets every device build
 * its own MSM engine (all at once). The batch calls cut the batch into contiguous shards, blob k of n -> device floor(k G / n)
 * (SURVEY 8e), run each shard through the single-device entry point above on a host thread of its own, and write the results
 * in place: no reduction, no data-path collective. The batch verification is the reference's single check (ONE Fiat-Shamir r,
 * ONE linear combination, ONE pairing; src/lib.rs:639-692) via the lwkzg_verify_shard_* steps below. Bytes equal to the
 * single-device calls. A device ordinal may appear more than once (several contexts on one GPU: how the one-GPU tests run it).
 * lwkzg_multi_settings(m, k) is the k-th device's KZGSettings, for everything that is per device (lwkzg_reserve, timing, ...). */
typedef struct LwkzgMulti LwkzgMulti;
C_KZG_RET lwkzg_multi_load(LwkzgMulti **out, const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2, const int *devices,
                           size_t n_devices);
C_KZG_RET lwkzg_multi_load_file(LwkzgMulti **out, FILE *in, const int *devices, size_t n_devices);
void lwkzg_multi_free(LwkzgMulti *m);
size_t lwkzg_multi_device_count(const LwkzgMulti *m);
int lwkzg_multi_device(const LwkzgMulti *m, size_t k);                    /* ordinal of the k-th entry, -1 out of range */
const KZGSettings *lwkzg_multi_settings(const LwkzgMulti *m, size_t k);  /* NULL out of range */
C_KZG_RET lwkzg_multi_set_mode(const LwkzgMulti *m, int mode);           /* lwkzg_settings_set_mode on every device */
C_KZG_RET lwkzg_multi_enable_direct_table(const LwkzgMulti *m, int window_bits); /* every device at once; first failure returned */
C_KZG_RET lwkzg_multi_blob_to_kzg_commitment_batch(KZGCommitment *out, const Blob *blobs, size_t n, const LwkzgMulti *m, size_t *first_bad);
C_KZG_RET lwkzg_multi_compute_blob_kzg_proof_batch(KZGProof *out, const Blob *blobs, const Bytes48 *commitments, size_t n, const LwkzgMulti *m,
                                                   size_t *first_bad);
C_KZG_RET lwkzg_multi_compute_kzg_proof_batch(KZGProof *proofs_out, Bytes32 *ys_out, const Blob *blobs, const Bytes32 *zs, size_t n,
                                              const LwkzgMulti *m, size_t *first_bad);
C_KZG_RET lwkzg_multi_verify_blob_kzg_proof_batch(bool *ok, const Blob *blobs, const Bytes48 *commitments, const Bytes48 *proofs, size_t n,
                                                  const LwkzgMulti *m);
/* The same three calls for shards that are ALREADY in HBM: device k's shard is n_per_device[k] blobs behind blobs_dev[k] (a pointer
 * on device k), results in place behind out48_dev[k] (48 bytes per blob). Nothing crosses PCIe but verdicts and, for the
 * verification, the 160-byte records. first_bad counts through the shards in device order. Synchronous. */
C_KZG_RET lwkzg_multi_blob_to_kzg_commitment_batch_device(void *const *out48_dev, const void *const *blobs_dev, const size_t *n_per_device,
                                                          const LwkzgMulti *m, size_t *first_bad);
C_KZG_RET lwkzg_multi_compute_blob_kzg_proof_batch_device(void *const *out48_dev, const void *const *blobs_dev, const void *const *commitments48_dev,
                                                          const size_t *n_per_device, const LwkzgMulti *m, size_t *first_bad);
C_KZG_RET lwkzg_multi_verify_blob_kzg_proof_batch_device(bool *ok, const void *const *blobs_dev, const void *const *commitments48_dev,
                                                         const void *const *proofs48_dev, const size_t *n_per_device, const LwkzgMulti *m);
/* THE shard rule (multi.hip and lambdaworks_kzg_amd/dist.py both use this function): item i of n_items belongs to part
 * floor(i * parts / n_items); part k owns [*first, *first + *count) = [ceil(k n / parts), ceil((k + 1) n / parts)). */
C_KZG_RET lwkzg_shard_range(size_t n_items, size_t parts, size_t k, size_t *first, size_t *count);
/* BASELINE configs[4] over the devices: out = sum_k scalars[k] * g1[k mod 4096], n_terms a positive multiple of 4096, HOST-resident
 * big-endian scalars; whole tiles per device, one 48-byte partial sum back from each, added on the host. */
C_KZG_RET lwkzg_multi_g1_msm_tiled(uint8_t out48[48], const uint8_t *scalars_be, size_t n_terms, const LwkzgMulti *m);

/* verify_blob_kzg_proof_batch (src/lib.rs:525-614, 639-692) for a batch that is ALREADY on the device: blobs (n * 131072 bytes),
 * commitments and proofs (n * 48 each) are device pointers, produced on `stream` (a hipStream_t; NULL = already complete). The call
 * is synchronous, as the reference's: *ok is a host bool and the pairing check runs on the host; nothing but the 160-byte records
 * crosses PCIe. Same verdicts, return codes and empty-batch rule as verify_blob_kzg_proof_batch. */
C_KZG_RET lwkzg_verify_blob_kzg_proof_batch_device(bool *ok, const void *blobs_dev, const void *commitments48_dev, const void *proofs48_dev,
                                                   size_t n, const KZGSettings *s, void *stream);

/* verify_blob_kzg_proof_batch (src/lib.rs:525-692) for a batch SHARDED over several processes / GPUs, as the
 * reference computes it: ONE Fiat-Shamir scalar r over the whole batch (compute_r_powers, src/utils.rs:166-206), ONE
 * random linear combination, ONE pairing check. Rank k holds blobs [first_k, first_k + n_k) of the n_total:
 *   1. lwkzg_verify_shard_begin: per blob on this rank's GPU -- validate C_i and pi_i, z_i = compute_challenge,
 *      y_i = p_i(z_i); writes the shard's n_k transcript records (C 48 | z 32 | y 32 | pi 48 = 160 bytes each) and keeps
 *      the decompressed points on the device behind *shard_out;
 *   2. all-gather the records (160 bytes per blob; lambdaworks_kzg_amd/dist.py does it with torch.distributed);
 *   3. lwkzg_verify_shard_partial: r from ALL n_total records; this shard's terms of sum r^i pi_i, sum r^i z_i pi_i,
 *      sum r^i C_i (three affine points) and of sum r^i y_i (one scalar) -> LWKZG_VERIFY_PARTIAL_BYTES bytes;
 *   4. all-gather the partial sums; lwkzg_verify_shards_finish adds them and does the pairing check (on every rank, or
 *      on one).
 * verify_blob_kzg_proof_batch itself is steps 1, 3, 4 with one shard. Errors as there (invalid point / non-canonical
 * field element: C_KZG_ERROR in reference mode, C_KZG_BADARGS in c-kzg mode); n_total == 0 gives ok = false. A shard may
 * be empty (n_local = 0). Several shards may be open on one settings object (each owns its device scratch). */
#define LWKZG_VERIFY_RECORD_BYTES 160
#define LWKZG_VERIFY_PARTIAL_BYTES 328
typedef struct LwkzgVerifyShard LwkzgVerifyShard;
C_KZG_RET lwkzg_verify_shard_begin(LwkzgVerifyShard **shard_out, uint8_t *records_out /* n_local * 160 */, const Blob *blobs,
                                   const Bytes48 *commitments, const Bytes48 *proofs, size_t n_local, const KZGSettings *s);
/* Step 1 for a shard that is already in HBM: DEVICE pointers (produced on `stream`, NULL = already complete); records to the host. */
C_KZG_RET lwkzg_verify_shard_begin_device(LwkzgVerifyShard **shard_out, uint8_t *records_out /* n_local * 160, host */, const void *blobs_dev,
                                          const void *commitments48_dev, const void *proofs48_dev, size_t n_local, const KZGSettings *s,
                                          void *stream);
C_KZG_RET lwkzg_verify_shard_partial(uint8_t *partial_out /* 328 */, LwkzgVerifyShard *shard, const uint8_t *records_all /* n_total * 160 */,
                                     size_t n_total, size_t first_index);
void lwkzg_verify_shard_free(LwkzgVerifyShard *shard);
C_KZG_RET lwkzg_verify_shards_finish(bool *ok, const uint8_t *partials /* n_shards * 328 */, size_t n_shards, size_t n_total,
                                     const KZGSettings *s);

/* Host-only test hook for the verify side: prod_i e(P_i, Q_i) == 1 for up to 4 pairs of ZCash-compressed
 * points (G1 48 bytes, G2 96 bytes; a pair with a point at infinity contributes 1). No GPU, no settings. */
C_KZG_RET lwkzg_pairing_product_is_one(bool *ok, const uint8_t *g1_compressed, const uint8_t *g2_compressed, size_t n);

/* Host-only test hook: digests[i] = SHA-256("FSBLOBVERIFY_V1_" | le64(4096) | le64(0) | blobs[i] | commitments[i]),
 * the compute_challenge message (src/utils.rs:120-144), as the host-pointer proof entry points compute it. */
C_KZG_RET lwkzg_challenge_digests_host(uint8_t *digests32, const uint8_t *blobs, const uint8_t *commitments48, size_t n);

/* Host-only: the batch challenge of verify_blob_kzg_proof_batch (src/utils.rs:166-206) over a transcript of n_total 160-byte records
 * (lwkzg_verify_shard_begin*'s output, all shards in order): r = SHA-256("RCKZGBATCH___V1_" | le64(4096) | le64(n_total) | records) read as a
 * field element in `mode`'s byte order and reduced; r_out = its canonical value, 32 bytes big-endian. What every shard derives inside
 * lwkzg_verify_shard_partial; exposed so that a sharded verifier can log / cross-check it (tests hold it against hashlib). */
C_KZG_RET lwkzg_batch_challenge_host(uint8_t r_out[32], const uint8_t *records_all, size_t n_total, int mode);

/* Engine introspection / profiling (bench.py). Kernel timings use hipEvents on the launch stream. */
int lwkzg_device_count(void);
int lwkzg_set_device(int ordinal);                 /* device used by subsequent load_* calls */
const char *lwkzg_version(void);
const char *lwkzg_last_error(void);                /* thread-local, human readable */
void lwkzg_profile_enable(int on);
void lwkzg_profile_reset(void);
/* JSON: {"kernel": {"launches": n, "total_ms": t}, ...}; returns bytes needed (incl. NUL) */
size_t lwkzg_profile_report(char *buf, size_t cap);
/* JSON: the wall-clock milliseconds of this settings object's load (context, points + tables, G2 + FFT settings, the
 * default engine's table) and of its last lwkzg_enable_direct_table: freeing the old table, the hipMallocs of the new one
 * (one per window, summed: the GPU builds window j while the host allocates window j + 1), scratch, and what was left of the
 * build kernels after the last allocation returned; returns bytes needed (incl. NUL). The allocations return in a millisecond on idle memory
 * and wait ~25 ms per GB for the driver's scrub of memory released shortly before (tools/alloc_pieces.hip). */
size_t lwkzg_timing_report(const KZGSettings *s, char *buf, size_t cap);
/* JSON: the environment knobs in effect for this process -- read ONCE, in one place (csrc/knobs.h). "operational" and "experimental"
 * list the variable names; experiment knobs (A/B arms of measurements) are honoured only with LWKZG_EXPERIMENTAL=1 and are otherwise
 * ignored (reported under LWKZG_VERBOSE). Returns bytes needed (incl. NUL). INTEGRATION.md section 5 documents the operational ones. */
size_t lwkzg_knob_report(char *buf, size_t cap);
/* Test hook: the schedule (csrc/plan.h: ProofSchedule, 0 = small-host .. 4 = GPU chains) the most recent
 * lwkzg_compute_blob_kzg_proof_batch_device call of this process took, -1 before the first. tests/test_gpu_plan.py forces each schedule
 * once (LWKZG_EXPERIMENTAL=1 LWKZG_PROOF_SCHEDULE=k) and reads it back. */
int lwkzg_last_proof_schedule(void);

/* First use of the HIP runtime by this process (device context + this library's code object), so that a caller can pay
 * and time it apart from its first real call. 0, or -1 without a GPU. */
int lwkzg_runtime_init(void);
/* The shader clock (MHz) the current device holds under ~0.4 ms of dense multiply-adds on every SIMD (clock64 against the 100 MHz
 * wall clock). A measurement aid: bench.py prints it with a hash of the GPU's uuid so that profiles can be matched to boxes. */
C_KZG_RET lwkzg_clock_probe_mhz(double *mhz);
/* MSM plan constants, for roofline arithmetic in bench.py */
int lwkzg_msm_window_bits(void);
int lwkzg_msm_num_windows(void);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* LAMBDAWORKS_KZG_AMD_H */
