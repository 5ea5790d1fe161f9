// knobs.h -- every environment variable the library reads, in ONE place, read ONCE (the first call of knobs(): context creation
// at the latest). Nothing else in the library calls getenv. Two classes:
//
//   * operational knobs (documented in INTEGRATION.md section 5): always honoured;
//   * experiment knobs: A/B arms of measurements under profiles/ and tools/experiments/. Honoured ONLY with LWKZG_EXPERIMENTAL=1 in
//     the environment; without it they keep their defaults, so a stray variable cannot move a production process onto an arm that
//     exists for a measurement.
//
// tests/test_capi_cpu.py pins the two lists against this file and INTEGRATION.md (a knob added here without documentation fails it).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace lwk {

struct Knobs {
    // ---- operational ------------------------------------------------------------------------------------------------------------
    int mode = 0;                    // LWKZG_MODE: process default semantics, "reference" (0) | "ckzg" (1)
    bool has_direct_bits = false;    // LWKZG_DIRECT_BITS: 0 (bucket engine) | 10..16 | "auto" (widest that fits) ; unset: 13..10 within a quarter of free HBM
    int direct_bits = 0;             //   (-1 = auto)
    int direct_row = 0;              // LWKZG_DIRECT_ROW: 0 = choose, 112 = packed rows, 128 = line-aligned rows
    bool coalesce = true;            // LWKZG_COALESCE: merge concurrent single-blob callers into one launch set
    bool twin = true;                // LWKZG_TWIN: a second context for a second caller stream
    size_t small_proof_host = 64;    // LWKZG_SMALL_PROOF_HOST: device-resident proof calls up to this many blobs hash + validate on host threads
    size_t mid_proof_host = 384;     // LWKZG_MID_PROOF_HOST: ... up to this many hash on host threads, pipelined (0 = GPU hash always)
    int host_threads = 0;            // LWKZG_HOST_THREADS: upper bound on the host threads the library uses (0 = 32; the hardware and a cgroup quota bound it as well)
    int host_warm_ms = 2000;         // LWKZG_HOST_WARM_MS: host-assisted paths only while the host threads ran a job this recently
    size_t host_finish = 8;          // LWKZG_HOST_FINISH: host-pointer calls of up to this many results invert + compress on the calling thread
    bool timing = false;             // LWKZG_TIMING: phase wall clock of verification / proof slices to stderr
    bool verbose = false;            // LWKZG_VERBOSE: every set_error() message to stderr
    bool experimental = false;       // LWKZG_EXPERIMENTAL: honour the experiment knobs below

    // ---- experiment (A/B arms; LWKZG_EXPERIMENTAL=1) ----------------------------------------------------------------------------
    bool direct_asm = true;          // LWKZG_DIRECT_ASM=0: compiler-scheduled k_direct_accumulate
    bool fold_asm = true;            // LWKZG_FOLD_ASM=0: compiler-scheduled lane fold
    bool bucket_asm = true;          // LWKZG_BUCKET_ASM=0: compiler-scheduled k_bucket_accumulate
    int direct_fill = 0;             // LWKZG_DIRECT_FILL: workgroups-per-blob geometry override
    int coop = 1;                    // LWKZG_COOP=0: no cooperative kernel for <= 8 blobs
    int coop_max = 8;                // LWKZG_COOP_MAX (clamped 1..8)
    int coop_rpq = 0;                // LWKZG_COOP_RPQ: rows per quad override (clamped so a blob's hand-off fits its workspace slice)
    bool sort_stage = true;          // LWKZG_SORT_STAGE=0: k_digit_sort without the LDS staging buffer
    int reduce_lanes = 0;            // LWKZG_REDUCE_LANES: k_bucket_reduce geometry
    bool hash_pairs = true;          // LWKZG_HASH_PAIRS=0: one lane per blob SHA kernel
    int hash_prio = 1;               // LWKZG_HASH_PRIO=0: no s_setprio in the latency kernels
    bool validate_coop = true;       // LWKZG_VALIDATE_COOP=0: r04's one-lane-per-point validation
    unsigned validate_lds_pad = 112u * 1024u;  // LWKZG_VALIDATE_LDS_PAD: LDS footprint of that r04 kernel
    bool ckzg_eval_proofs = true;    // LWKZG_CKZG_EVAL_PROOFS=0: c-kzg proofs through the inverse transform
    bool mid_proof_pipe = true;      // LWKZG_MID_PROOF_PIPE=0: unpipelined mid-size proof calls
    size_t mid_proof_pipe_min = 192; // LWKZG_MID_PROOF_PIPE_MIN
    size_t mid_proof_parts = 0;      // LWKZG_MID_PROOF_PARTS
    size_t mid_proof_chunks = 4;     // LWKZG_MID_PROOF_CHUNKS
    int proof_schedule = -1;         // LWKZG_PROOF_SCHEDULE=0..4: force that schedule (plan.h: ProofSchedule) for device-resident proof calls of up to 1024 blobs (tests)
    int heavy_serial = -1;           // LWKZG_HEAVY_SERIAL
    int split = 0;                   // LWKZG_SPLIT: windows of a scalar over this many workgroups (tiny batches)
    size_t slice0 = 0;               // LWKZG_SLICE0: first slice of a long host-pointer batch
    bool set_mode_in_place = true;   // LWKZG_SET_MODE_IN_PLACE=0
    int host_fp_portable = 0;        // LWKZG_HOST_FP_PORTABLE: 1 = the C products of hostfp.h on a core that has MULX/ADX, 2 = only the Fp2 product in C (the A/B arms)
    int host_hash_grain = 4;         // LWKZG_HOST_HASH_GRAIN: blobs per thread woken for a host hashing job (1 = wake every parked thread, as before r06)
    int stage_streams[2] = {8, 0};   // LWKZG_STAGE_STREAMS=c,h: which streams carry the uploads (aux index; 8 = a high-priority stream of their own) / the head's hashes (aux index) of a staged verification
    int side_workers = 1;            // LWKZG_SIDE_WORKERS: 0 = a std::thread per SideTask, as before r06 (the A/B arm)
    bool pairing_generic_sqr = false, pairing_naive = false, pairing_no_precomp = false, pairing_one_thread = false;  // LWKZG_PAIRING_*
    // r06, batch verification
    int verify_msm = 1;              // LWKZG_VERIFY_MSM=0: r05's per-point multiples + Straus pieces (k_point_multiples, k_lincomb3)
    int verify_fused = 1;            // LWKZG_VERIFY_FUSED=0: commitments and proofs validated by separate launches on two side streams
    int verify_pad_kb[3] = {60, 116, 116};  // LWKZG_VERIFY_PAD_KB="d,s,m": unused LDS (KiB) per workgroup of the decompression / subgroup / rows kernels of a verification
    int verify_order = 0;            // LWKZG_VERIFY_ORDER=1: the other submission order of hash and validation (shipped: the hash first up to 8192 blobs, last above)
    int verify_cu_mask = 0;          // LWKZG_VERIFY_CU_MASK=k: side streams confined to k compute units per XCD (hipExtStreamCreateWithCUMask)
    int vmsm_list_cap = 0;           // LWKZG_VMSM_LIST_CAP: rows per digit a slice lists before the scan fallback (tests force the fallback with 1)
    bool zero_copy = true;           // LWKZG_ZERO_COPY=0: a one-blob commitment copies its blob up and its sum and verdict back (r05) instead of reading / writing pinned memory from the kernels
    bool host_stage = true;          // LWKZG_HOST_STAGE=0: long host-pointer batches in r05's 512-blob slices on two streams instead of whole chunks from the device-side double buffer
};

const Knobs &knobs();

// the documented list, for lwkzg_knob_names (tests)
const char *knob_names_operational();
const char *knob_names_experimental();

}  // namespace lwk
