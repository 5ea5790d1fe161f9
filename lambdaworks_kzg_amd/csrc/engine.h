// engine.h -- per-settings device context and the batch pipelines behind the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/lambdaworks_kzg_amd.h"
#include "kernels.h"
#include "front.h"

namespace lwk {

constexpr uint64_t kCtxMagic = 0x4c574b5a47414d44ull;  // "LWKZGAMD"
constexpr size_t kMaxChunk = 1024;                      // blobs per launch set
static_assert(kMaxChunk == kPlanMaxChunk, "plan.h plans in the engine's chunks");
constexpr int kMaxSplit = 8;                            // sub-batches (streams) a launch set may be cut into

void set_error(const char *fmt, ...);
const char *get_error();

#define LWK_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            lwk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return C_KZG_ERROR;                                                               \
        }                                                                                     \
    } while (0)

// A piece of host work that runs beside the caller when a thread can be had, and inline when it cannot. The threads are persistent
// (sha256_host.hip: side workers, created on demand, parked on a condition variable): handing a job to a parked thread and waiting for
// it costs ~7 us where creating and joining a std::thread cost ~50 us on the build host -- and a one-blob verification has four such
// jobs on its critical path. Nothing may unwind across the C ABI: a thread that cannot be created means the job runs inline.
struct SideWorker;
SideWorker *side_worker_acquire();                                   // nullptr: none idle and none could be created
void side_worker_run(SideWorker *w, std::function<void()> fn);       // returns at once
void side_worker_wait(SideWorker *w);                                // until fn has returned; the worker is idle again afterwards
bool side_workers_on();                                              // LWKZG_SIDE_WORKERS=0 (experiment): a std::thread per job, as before r06
struct SideTask {
    SideWorker *w = nullptr;
    std::thread t;                                                   // the A/B arm only
    SideTask() = default;
    template <class F>
    explicit SideTask(F &&f) { start(static_cast<F &&>(f)); }
    template <class F>
    void start(F &&f) {
        join();
        if (!side_workers_on()) {
            try {
                t = std::thread(f);
            } catch (...) {
                f();
            }
            return;
        }
        w = side_worker_acquire();
        if (w) side_worker_run(w, std::function<void()>(f));
        else f();
    }
    void join() {
        if (w) side_worker_wait(w);
        w = nullptr;
        if (t.joinable()) t.join();
    }
    ~SideTask() { join(); }
    SideTask(const SideTask &) = delete;
    SideTask &operator=(const SideTask &) = delete;
};

struct Workspace {
    size_t cap = 0;  // blobs
    uint8_t *blobs = nullptr;        // host-API staging: cap x 131072
    uint32_t *scalars = nullptr;     // cap x 4096 x 8   canonical coefficients
    uint32_t *scalars2 = nullptr;    // cap x 4096 x 8   quotient coefficients
    Fr *fr = nullptr;                // cap x 4096       Montgomery scratch (mode C / NTT)
    uint32_t *sorted = nullptr;      // cap x 81920
    uint32_t *bucket_start = nullptr;  // cap x 4097
    uint32_t *perm = nullptr;        // cap x 4096
    G1Xyzz29 *buckets = nullptr;       // cap x 4096
    G1Xyzz29 *sums = nullptr;          // cap
    uint8_t *out48 = nullptr;        // cap x 48
    uint8_t *comm48 = nullptr;       // cap x 48
    uint8_t *canon48 = nullptr;      // cap x 48
    uint8_t *zbytes = nullptr;       // cap x 32
    uint8_t *ybytes = nullptr;       // cap x 32
    Fr *z = nullptr;                 // cap
    int32_t *status = nullptr;       // cap
    // scratch of the three-launch validation of a proof call's commitments (decompressed points, kinds, the quad test's verdicts)
    G1Affine29 *val_pts = nullptr;   // cap
    int32_t *val_kind = nullptr;     // cap
    uint32_t *val_verdict = nullptr; // cap
    // device-resident proof calls longer than one chunk hash / validate ALL their blobs up front (both kernels are
    // latency chains whose run time does not depend on the batch size); grow-only
    size_t long_cap = 0;             // blobs
    Fr *z_long = nullptr;            // long_cap
    uint8_t *canon_long = nullptr;   // long_cap x 48
    int32_t *status_long = nullptr;  // long_cap (when the caller passes no status array)
    G1Affine29 *val_pts_long = nullptr;      // long_cap: the validation's scratch for such a call
    int32_t *val_kind_long = nullptr;
    uint32_t *val_verdict_long = nullptr;
};

// device-side scratch of one batch verification (points kept between its two GPU phases). The memory belongs to the
// context (grow-only: hipMalloc / hipFree cost more than the kernels at small n); `hold` keeps other verifications
// of the same settings object out until this one is done with it.
constexpr size_t kVerifyApartMax = 8192;   // blobs: up to here the challenge hash leaves half the compute units free (64 blobs per workgroup)
struct VerifyBuffers {
    G1Affine29 *pts_c = nullptr, *pts_p = nullptr;
    G1Affine29 *mult_c = nullptr, *mult_p = nullptr;  // [2^32]P, [2^64]P, [2^96]P of every point (3 n entries each)
    int32_t *kind_c = nullptr, *kind_p = nullptr;
    uint8_t *proof_in = nullptr;  // compressed proofs as uploaded (validated on an auxiliary stream)
    uint8_t *comm_in = nullptr, *canon_dev = nullptr;  // commitments as uploaded; canonical bytes (n commitments, n proofs)
    int32_t *status_all = nullptr;                      // per-blob verdicts of the up-front validation
    uint32_t *verdict_c = nullptr, *verdict_p = nullptr; // the quad subgroup test's words (k_subgroup_coop_asm), per point set
    uint8_t *d_r = nullptr, *d_rz = nullptr, *d_aff = nullptr;
    G1Xyzz29 *d_part = nullptr;
    int32_t *d_inf = nullptr;
    // r06, vmsm.hip: the rows [2^(8 j)]P / [2^(8 j)](-phi(P)) of both point sets (32 n each), the build's scratch, the split scalars,
    // the slices' and the buckets' sums, the powers of r -- ONE allocation (vm_base) carved up -- and a pinned block on the host:
    // the powers on their way up, the three affine sums and their infinity flags on their way down
    uint8_t *vm_base = nullptr;
    G1Affine29 *tab_p = nullptr, *tab_c = nullptr;
    G1Xyzz29 *vm_tmp = nullptr, *vm_partial = nullptr, *vm_bsum = nullptr;
    F29<2> *vm_pre = nullptr;
    uint32_t *sc_a = nullptr, *sc_b = nullptr;
    Fr *vm_pw = nullptr;
    uint8_t *d_rec = nullptr;       // 160 bytes per blob + a first-bad-index word: the transcript assembled on the device (k_verify_records)
    uint8_t *h_rec = nullptr;       // hipHostMalloc: the same, where one copy lands
    size_t rec_cap = 0;             // blobs d_rec / h_rec hold
    uint8_t *h_pin = nullptr;       // hipHostMalloc: 33 Fr | 3 x 96 bytes | 3 x int32
    hipEvent_t vm_done = nullptr;   // recorded behind the results' copy (vmsm_begin), waited for by vmsm_finish
    // small batches validated on the host threads keep their points here instead (n commitments, then n proofs; kind 0 =
    // affine point, 1 = infinity) and the linear combinations run on the host threads too (verify.hip)
    std::vector<G1Affine29> h_aff;
    std::vector<int32_t> h_kind;
    std::unique_lock<std::mutex> hold;
    bool owned = false;  // the device pointers above belong to this object (verify_buffers_free), not to the context
};
void verify_buffers_free(VerifyBuffers &v);

// Up to this many blobs a verification / proof call validates its points on the host threads: ~0.2 ms per point per
// thread against a 2 ms latency-shaped kernel. 4 per usable hardware thread, at most 64.
size_t host_small_batch_limit();
// sha256_host.hip: the host threads' last job (steady-clock ns, 0 = none yet), the clock itself, and a job that wakes every worker
int64_t host_last_active_ns();
int64_t host_now_ns();
void host_pool_warm();

// Coalescing front of the single-blob symbols (engine.hip: combine_commit). The reference's KZGSettings is read-only
// after load, so any number of threads may call blob_to_kzg_commitment on one settings object at once
// (/root/reference/src/lib.rs:253-283, SURVEY 8b "Threading"); a GPU call per blob would serialise them at one launch
// set each. Callers that arrive while a launch set is in flight are merged into the next one: each copies its blob into
// a pinned staging slot, one of them (the "leader") uploads the batch, launches, waits and hands every caller its 48
// bytes. Two lanes (streams + workspace halves) are in flight at most, so one batch uploads while the other computes.
constexpr int kCombineLanes = 2;
constexpr size_t kCombineSlots = 128;     // pinned 128 KiB staging slots (16 MiB)
constexpr size_t kCombineMaxBatch = 64;   // requests per launch set

struct CombineReq {
    enum State { QUEUED, TAKEN, DONE };
    int slot = -1;          // pinned staging slot holding the blob
    int mode = 0;
    uint8_t *out48 = nullptr;
    int rc = C_KZG_OK;      // a C_KZG_RET
    State state = QUEUED;
};

// the threading (queue, leaders, lanes, slots) is front.h's LaneFront; this adds the pinned host memory of the lanes
struct Combiner {
    LaneFront<CombineReq, kCombineLanes> front;
    std::mutex init_m;
    uint8_t *pinned_blobs = nullptr;                  // kCombineSlots x 131072, hipHostMalloc
    uint8_t *pinned_out[kCombineLanes] = {nullptr, nullptr};     // kCombineMaxBatch x 224: compressed results, or the XYZZ sums of a small batch (finished on the host)
    int32_t *pinned_status[kCombineLanes] = {nullptr, nullptr};  // kCombineMaxBatch
    bool ready = false, failed = false;
};

// The same front for compute_blob_kzg_proof and compute_kzg_proof (engine.hip: front_run): one leader at a time runs
// everything queued in its mode as one host-pointer batch. `second` is the commitment (48 bytes) or z (32 bytes).
struct ProofReq {
    enum State { QUEUED, TAKEN, DONE };
    const uint8_t *blob = nullptr, *second = nullptr;
    uint8_t *out = nullptr, *y_out = nullptr;
    int mode = 0;
    int rc = C_KZG_OK;      // a C_KZG_RET
    State state = QUEUED;
};
typedef LeaderFront<ProofReq> ProofFront;   // front.h

// Where the seconds of a load and of a table build went (lwkzg_timing_report): wall-clock milliseconds of the host thread.
struct BuildTiming {
    int bits = 0;
    size_t row_bytes = 0, table_bytes = 0;
    // table_malloc_ms: the windows' hipMallocs, summed (the GPU builds window j meanwhile); kernels_ms: what was left of the build
    // kernels after the last allocation returned
    double free_old_ms = 0, table_malloc_ms = 0, scratch_malloc_ms = 0, kernels_ms = 0, scratch_free_ms = 0, total_ms = 0;
    bool in_place = false;   // the other form's allocations were kept and only the build kernels ran (lwkzg_settings_set_mode)
};
struct LoadTiming {
    double context_ms = 0;           // streams, events, the small device buffers
    double points_and_tables_ms = 0; // upload, decompression + subgroup checks, blst layout, 9 MB fixed-base table, D2H
    double g2_and_fft_ms = 0;        // G2 points on the host, twiddles and FFTSettings
    double default_table_ms = 0;     // the engine the load selects (BuildTiming of it is kept as the last build)
    double total_ms = 0;
};

// Pinned host staging of the host-assisted Fiat-Shamir step of SMALL device-resident proof calls (engine.hip:
// blob_proof_batch_device): the GPU hash is a 3.2 ms latency chain whatever the batch, a host core with SHA extensions needs
// 0.07 ms per blob, so up to a few dozen blobs are copied out, hashed and validated on the host threads in stream order
// (hipLaunchHostFunc) and only the 32-byte digests go back.
struct SmallProofHost {
    uint8_t *blobs = nullptr, *comm = nullptr, *canon = nullptr, *dig = nullptr;  // cap x 131072 / 48 / 48 / 32, hipHostMalloc
    int32_t *code = nullptr;                                                      // cap status words (0 or the mode's rejection code)
    size_t cap = 0;
    static constexpr int kChunks = 16;                                            // mid-size calls: the blobs leave in chunks, hashed as they land
    hipEvent_t chunk_done[kChunks] = {};                                          // chunk k has landed in `blobs`
    hipEvent_t hashed[kChunks] = {};                                              // chunk k's host function has run (the pipelined mid-size path)
};

// The Lagrange form of the setup (c-kzg mode without the transform): L_i = [l_i(tau)]G in the blob's own
// (bit-reversed domain) order, so that a c-kzg blob's evaluations are MSM scalars as they stand. Derived on the device from the
// monomial points (engine.hip: lagrange_prepare), lazily: a settings object that never answers in c-kzg mode never pays for it.
struct LagrangeForm {
    bool ready = false;                  // points + 9 MB bucket table exist (written under the context's locks, both of them)
    G1Affine *points = nullptr;          // 4096 affine Montgomery points
    G1Affine29 *table = nullptr;         // the bucket engine's fixed-base table over them
    G1Affine29 *direct_table = nullptr;  // direct table over them (window 0's rows); nullptr = none
    DirectTable direct_tab;
    int direct_bits = 0;
    size_t direct_row_bytes = 0;
};

// The object KZGSettings.fs points to. Its first member is a genuine FFTSettings.
// Device-side double buffer of the long host-pointer batches (r06; /root/reference/fuzz/base_fuzz.h:17-34 is that kind of caller: plain
// host arrays). r05 cut such a batch into slices of 512 blobs on two streams and two workspace halves, so that the upload of one slice ran
// beside the compute of the other -- and paid for it with 512-blob launches (two workgroups per blob, a fold per slice: 77k ops/s where a
// 1024-blob launch does 92k). What the measurements say (profiles/r06_h2d_bench.txt): a hipMemcpy from pageable memory runs at 56 GB/s once
// the pages have been touched (13 GB/s the first time), as fast as from pinned memory -- the upload was never the bottleneck, the slice
// size was. So: slice k is uploaded into slot k mod 2 of this buffer on a copy stream of its own, and the SAME device-resident pipeline
// that a device pointer would get (commit_batch_device: whole chunks of up to 1024 blobs, one compute stream) reads it from there; the
// upload of slice k + 1 runs beside the compute of slice k, and a slot is reused once the parse of its previous occupant has run.
// 2 x 1024 blobs = 256 MiB of device memory per context, allocated by the first long batch and kept.
struct DevStage {
    uint8_t *slot[2] = {nullptr, nullptr};
    hipEvent_t copied[2] = {nullptr, nullptr}, parsed[2] = {nullptr, nullptr};
    bool ready = false, failed = false;
};

struct Ctx {
    FFTSettings fs;
    uint64_t magic;
    uint64_t generation = 0;        // unique per context of this process (ctx_is_live: a shard must not mistake a new context at an old address for its own)
    int device;
    hipStream_t stream;
    hipStream_t vstream;            // point-validation kernels (a plain stream of their own; see ctx_new on CU masks)
    hipStream_t aux[kMaxSplit];     // sub-batch streams of commit_batch_device
    hipEvent_t ev_fork, ev_join[kMaxSplit];
    // The workspace is shared by every call on this settings object, whatever stream the caller passes: each call
    // makes its stream wait for the event the previous user of the workspace recorded (WsUse, engine.hip).
    hipEvent_t ws_done;
    hipStream_t ws_last;
    std::atomic<bool> ws_recorded{false};  // ws_done has been recorded at least once (peer_busy reads it without mu)
    hipEvent_t lane_done[kCombineLanes];  // last use of a workspace half by a lane of the coalescing front
    Combiner comb;
    ProofFront blob_proof_front, point_proof_front;
    // A second set of streams + workspace over the SAME tables (engine.hip: pick_ctx): device-resident calls that arrive
    // on another caller stream while this context's workspace is still busy run there, so that the latency-shaped head
    // of one call (Fiat-Shamir hash, commitment validation) overlaps the ALU-bound MSM of the other.
    std::atomic<Ctx *> twin{nullptr};   // written under mu (pick_ctx), read without it by calls that run on the twin: release / acquire
    bool is_twin = false;
    // lwkzg_reserve* has been called on this settings object (kept on the primary): its calls never grow pinned staging by themselves
    std::atomic<bool> reserved{false};
    // Left alone, two proof pipelines that share the GPU fall into step (both hash, then both MSMs fight for the chip).
    // The ALU-bound phase of a proof call therefore takes turns across the two contexts: it waits for this event (the
    // previous call's phase, whichever context ran it) and records it again. Lives in the primary context.
    Ctx *primary = nullptr;
    hipEvent_t heavy_done = nullptr;
    std::mutex heavy_mu;
    G1Affine *points;  // 4096 affine Montgomery (== table row 0 source)
    G1Affine29 *table;  // kTablePoints, hot-loop representation
    G1Affine29 *direct_table;  // all multiples of every window base (direct.hip): window 0's rows; nullptr unless enabled
    DirectTable direct_tab;    // the windows (one allocation each) and their addresses on the device; owned by the primary context
    int direct_bits;           // 14 / 15 / 16 when direct_table is live, else 0
    size_t direct_row_bytes;   // 128 (every row in a line of its own) or 112 (packed), see kernels.h
    LagrangeForm lag;          // owned by the primary context; a twin holds copies of the pointers (sync_twin_tables)
    std::atomic<bool> lag_ready{false};  // == lag.ready, readable without the lock (ensure_lagrange's fast path)
    bool lag_failed = false;   // the derivation failed once (out of memory): c-kzg mode stays on the transform path
    Fr *tw_fwd, *tw_inv;
    Fr28 *tw28_fwd, *tw28_inv;  // the same twiddles in the transform's own arithmetic (fr28.cuh)
    Workspace ws;
    SmallProofHost sph;
    DevStage stage;     // under mu
    uint8_t *one_pin = nullptr;    // 4 KiB of pinned memory for a ONE-blob proof call (r06): XYZZ sum 224 | redo flag 4 | digest 32; under mu
    hipStream_t prio_copy = nullptr;   // a high-priority stream for the uploads of verify_prepare_staged (created by its first call, under mu)
    uint8_t *vblobs = nullptr;     // ALL blobs of a long host-pointer verification on the device (grow-only, under mu): verify_prepare_staged
    size_t vblobs_cap = 0;         // blobs
    uint8_t *host_res = nullptr;   // results / verdicts / digests of a long host-pointer batch on the device (grow-only, under mu):
    size_t host_res_cap = 0;       // r05 allocated and freed them per call, 6 ms of a 52 ms call of 4096 blobs (profiles/r06_experiments.md section 6)
    VerifyBuffers vs;   // verify-side scratch, sized for vs_cap blobs
    size_t vs_cap;
    std::mutex verify_mu;
    std::mutex mu;
    std::atomic<int> mode_override{-1};  // lwkzg_settings_set_mode: -1 = follow the process-wide default
    BuildTiming last_build;              // written under mu (enable_direct_table)
    LoadTiming load_timing;
};

Ctx *ctx_of(const KZGSettings *s);  // resolves fs, or the registry for hand-built settings; nullptr + error otherwise
// is `c` still the live context that was given generation `gen`? Looks only at the registry, never at a caller's KZGSettings
bool ctx_is_live(const Ctx *c, uint64_t gen);

int mode_of(const KZGSettings *s);    // the semantics a call on `s` answers in: its own mode if it has one, else the default

C_KZG_RET ctx_reserve(Ctx *c, size_t n);

// device-resident pipelines; all pointers device, async on st
C_KZG_RET commit_batch_device(Ctx *c, uint8_t *out48, const uint8_t *blobs, size_t n, int mode, hipStream_t st,
                              int32_t *status);
C_KZG_RET blob_proof_batch_device(Ctx *c, uint8_t *out48, const uint8_t *blobs, const uint8_t *comm48, size_t n, int mode,
                                  hipStream_t st, int32_t *status);
// (sums_out, one chunk at most: the XYZZ sums stay where they are, *sums_out says where, and the caller inverts and compresses)
C_KZG_RET point_proof_batch_device(Ctx *c, uint8_t *proof48, uint8_t *y32, const uint8_t *blobs, const uint8_t *z32,
                                   size_t n, int mode, hipStream_t st, int32_t *status, const G1Xyzz29 **sums_out = nullptr);
C_KZG_RET msm_scalars_raw_device(Ctx *c, uint8_t *out48, const uint32_t *scalars_raw, size_t n, hipStream_t st);

// verify-side helpers (host pointers in and out; GPU work inside; engine.hip)
C_KZG_RET verify_prepare_host(Ctx *c, const uint8_t *blobs, const uint8_t *comm48, const uint8_t *proofs48, size_t n,
                              int mode, uint8_t *z32, uint8_t *y32, uint8_t *canon_c, uint8_t *canon_p, VerifyBuffers &vb,
                              const uint8_t *trusted_canon_c = nullptr);
// the same for device-resident inputs (d_ pointers; z32 .. canon_p are host memory)
// records_out (r06; 160 n host bytes, C | z | y | pi per blob): when given, the transcript is assembled on the device and comes back in ONE
// copy through pinned memory; z32 .. canon_p are then not written
C_KZG_RET verify_prepare_device(Ctx *c, const uint8_t *d_blobs, const uint8_t *d_comm, const uint8_t *d_proofs, size_t n, int mode,
                                uint8_t *z32, uint8_t *y32, uint8_t *canon_c, uint8_t *canon_p, VerifyBuffers &vb, hipStream_t caller,
                                uint8_t *records_out = nullptr);
C_KZG_RET lincomb3_device_host(Ctx *c, VerifyBuffers &vb, const uint8_t *sc_r, const uint8_t *sc_rz, size_t n,
                               uint8_t sums[3][96], int infs[3]);
// r06: the same three sums from r alone (vmsm.hip). pw33: r^(2^k), k = 0..31, then r^first, Montgomery form. vmsm_begin only
// enqueues (scalars, buckets, sums, the copy of the results into pinned memory) and returns; vmsm_finish waits and hands over the
// sums -- the host computes sum r^i y_i and [that]G in between. vmsm_ready: this verification has rows on the device.
bool vmsm_ready(const VerifyBuffers &vb);
C_KZG_RET vmsm_begin(Ctx *c, VerifyBuffers &vb, const Fr *pw33, int le, size_t n);
C_KZG_RET vmsm_finish(Ctx *c, VerifyBuffers &vb, uint8_t sums[3][96], int infs[3]);

// host-side decompress_g1_point + subgroup check + recompression (verify.hip): 0 = affine, 1 = infinity, 2 = invalid
// (canon48 zeroed); for the few-points paths where a 2 ms validation kernel is the wrong tool
int host_validate_commitment(const uint8_t in48[48], uint8_t canon48[48], G1Affine29 *aff = nullptr);
void host_validate_commitments(const uint8_t *in48, uint8_t *canon48, int *rc, size_t n, G1Affine29 *aff = nullptr);  // on the host threads

// G2 / pairing side (g2_pairing.hip, host only)
bool g2_fill_values(g2_t *out65, const uint8_t *g2_bytes, size_t n2);

}  // namespace lwk
