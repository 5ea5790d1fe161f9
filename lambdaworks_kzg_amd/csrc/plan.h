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
#include <stddef.h>
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

// ---- schedule selection as a pure function (r06) ------------------------------------------------------------------------------------
// A device-resident compute_blob_kzg_proof call (engine.hip: blob_proof_batch_device; /root/reference/src/lib.rs:361-404 per blob) has
// five schedules for the part in front of its MSM -- who hashes, who validates, what overlaps -- and r05 chose between them inline, from
// the batch, the warmth of the host threads, the other context's state, the table forms and six environment reads (Weak 8).
// The choice is this function now: no HIP, no globals, no clock; tests/plan_table.cpp enumerates it on the CPU and
// tests/test_plan_cpu.py pins the table; tests/test_gpu_plan.py forces every value once on the GPU and compares bytes.
constexpr size_t kPlanMaxChunk = 1024;   // == engine.h: kMaxChunk (static_assert there)

enum ProofSchedule : int {
    kProofSmallHost = 0,   // blobs + commitments out to pinned memory, ONE host function hashes AND validates on the host threads, results back
    kProofMidCold = 1,     // host threads cold: the GPU's hash kernel this once + a wake-up on the side; validation on the GPU's side stream
    kProofMidHost = 2,     // hashing on the host threads chunk by chunk beside the copy out, validation on the GPU; ONE quotient + MSM behind both
    kProofMidPiped = 3,    // the same hashing, and every sub-batch of whole chunks starts its quotient + MSM as soon as ITS chunks are hashed
    kProofGpuChains = 4,   // hash kernel on the call's stream beside the validation on the side stream (any size; the only one above 1024)
};

struct PlanKnobs {         // the part of knobs.h a plan depends on (values after clamping)
    size_t small_proof_host = 64, mid_proof_host = 384, mid_proof_chunks = 4, mid_proof_pipe_min = 192, mid_proof_parts = 0;
    bool mid_proof_pipe = true;
    int heavy_serial = -1;
};

struct ProofPlan {
    ProofSchedule schedule = kProofGpuChains;
    size_t per_chunk = 0, chunks = 0;   // host-hash chunks (kProofMidHost / kProofMidPiped)
    size_t parts = 0;                   // sub-batches of whole chunks (kProofMidPiped)
    bool heavy_serial = false;          // the ALU-bound phase takes turns with the settings' other context
    bool needs_staging() const { return schedule == kProofSmallHost || schedule == kProofMidHost || schedule == kProofMidPiped; }
};

// n: blobs; host_warm: the host threads ran a job recently (or were just woken); peer_busy: the settings' other context has work in
// flight; msm_on_direct_table: the quotient's MSM runs on a direct table in the form the quotient is taken in; staging: pinned staging for
// n blobs can be had (the engine asks with true first and, should the allocation fail, again with false).
inline ProofPlan plan_proof_call(size_t n, bool host_warm, bool peer_busy, bool msm_on_direct_table, bool staging, const PlanKnobs &k) {
    ProofPlan p;
    p.heavy_serial = k.heavy_serial >= 0 ? k.heavy_serial != 0 : n <= kPlanMaxChunk / 2;
    if (staging && n <= k.small_proof_host) {
        p.schedule = kProofSmallHost;
    } else if (n <= k.mid_proof_host && !peer_busy && !host_warm) {
        p.schedule = kProofMidCold;
    } else if (staging && n <= k.mid_proof_host && !peer_busy) {
        const size_t want = k.mid_proof_chunks < 1 ? 1 : k.mid_proof_chunks;
        p.per_chunk = (n + want - 1) / want;
        p.chunks = p.per_chunk ? (n + p.per_chunk - 1) / p.per_chunk : 0;
        const bool piped = k.mid_proof_pipe && msm_on_direct_table && n >= k.mid_proof_pipe_min && p.chunks >= 2 && p.chunks % 2 == 0 && n <= kPlanMaxChunk;
        p.schedule = piped ? kProofMidPiped : kProofMidHost;
        if (piped) {
            p.parts = k.mid_proof_parts ? k.mid_proof_parts : (n >= 320 ? 4 : 2);
            while (p.parts > 1 && p.chunks % p.parts) p.parts--;
        }
    } else {
        p.schedule = kProofGpuChains;
    }
    return p;
}

// status words written by kernels (values of C_KZG_RET)
constexpr int kStatusOk = 0;
constexpr int kStatusBadArgs = 1;
constexpr int kStatusError = 2;


// ---- who hashes which blobs of a long host-pointer verification (r06; engine.hip: verify_prepare_staged) ---------------------------------
// The batch is uploaded in slices; the GPU's hash kernel takes the HEAD (whole slices, as they land), the host threads the TAIL (the rest,
// the ragged end included). Pure: the batch, the two measured rates, the kernel's latency.
//   * the GPU's kernel is a latency chain of `hash_launch_s` per launch whatever its size, so the head's LAST slice has to land that long
//     before the uploads end: the tail is what is uploaded in those last `hash_launch_s` (+ 6 %), in whole slices ...
//   * ... but no more than the host threads hash in 0.8 of the whole upload at their rate -- or they would be the tail of the call instead;
//   * the head's launches share one stream and so run one after the other: one goes out after every `every`-th slice counted back from
//     the head's last (`every` slices take longer to upload than a launch runs), each taking every slice that landed since the previous.
struct StagedSplit {
    size_t n_gpu = 0, n_host = 0;   // blobs hashed by the GPU's kernel (the first n_gpu) / by the host threads (the last n_host)
    size_t slice = 0, every = 0;
    // does a launch go out once `landed` of the head's slices are on the device?
    bool launch_after(size_t landed) const { return every != 0 && n_gpu != 0 && landed * slice <= n_gpu && ((n_gpu / slice - landed) % every) == 0; }
};
inline StagedSplit plan_staged_verification(size_t n, double host_hash_rate, double upload_rate = 56e9, double hash_launch_s = 3.2e-3,
                                            size_t slice = kPlanMaxChunk / 2, size_t blob_bytes = 131072) {
    StagedSplit p;
    p.slice = slice;
    size_t n_host = ((size_t)(1.0625 * hash_launch_s * upload_rate / (double)blob_bytes) + slice - 1) / slice * slice;
    const size_t host_can = (size_t)(0.8 * host_hash_rate / upload_rate * (double)n);
    if (n_host > host_can) n_host = host_can / slice * slice;
    if (n_host > n) n_host = n;
    p.n_gpu = (n - n_host) / slice * slice;
    p.n_host = n - p.n_gpu;
    const double slice_s = (double)slice * (double)blob_bytes / upload_rate;
    p.every = (size_t)(hash_launch_s / slice_s) + 1;
    return p;
}

}  // namespace lwk
