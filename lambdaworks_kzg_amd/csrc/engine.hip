// engine.hip -- device context, workspace and the batch pipelines (host side, HIP runtime).
//
// One process drives one GPU (bench.py / torch.distributed launch one process per GPU). A loaded
// trusted setup owns: the affine SRS points, the 20-window fixed-base table, the NTT twiddles, a
// stream, and a grow-only workspace sized for up to kMaxChunk blobs per launch set.
#include "engine.h"
#include "knobs.h"
#include "hostfp.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <map>
#include <set>
#include <thread>
#include <vector>

namespace lwk {

// ------------------------------------------------------------------------------------------------
// errors

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    if (knobs().verbose) fprintf(stderr, "[lambdaworks_kzg_amd] %s\n", g_err);
}
const char *get_error() { return g_err; }

// ------------------------------------------------------------------------------------------------
// profiling: hipEvent pairs around every kernel launch, on the launch stream

struct ProfRec {
    const char *name;
    hipEvent_t e0, e1;
};
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof_pending;
struct ProfAgg {
    uint64_t launches = 0;
    double total_ms = 0;
};
static std::map<std::string, ProfAgg> g_prof_agg;

ProfScope::ProfScope(const char *n, hipStream_t s) : name(n), st(s), e0(nullptr), e1(nullptr), on(g_prof_on.load(std::memory_order_relaxed)) {
    if (!on) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        on = false;
        return;
    }
    hipEventRecord(e0, st);
}
ProfScope::~ProfScope() {
    if (!on) return;
    hipEventRecord(e1, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_pending.push_back({name, e0, e1});
}

static void prof_drain() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof_pending) {
        float ms = 0;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            auto &a = g_prof_agg[r.name];
            a.launches++;
            a.total_ms += ms;
        }
        hipEventDestroy(r.e0);
        hipEventDestroy(r.e1);
    }
    g_prof_pending.clear();
}

// ------------------------------------------------------------------------------------------------
// context registry (for KZGSettings built by hand: fs == NULL, caller-owned g1_values)

static std::mutex g_reg_mu;
// hand-built settings: keyed by the g1_values pointer, re-checked against a digest of the first and last point so that
// a caller that frees and reallocates g1_values at the same address does not silently get the stale context
struct RegEntry {
    Ctx *ctx;
    uint64_t digest;
};
static std::map<const void *, RegEntry> g_registry;
static std::set<const void *> g_live_fs;  // every live Ctx (== the fs pointer it handed out)
static std::atomic<int> g_default_device{0};
thread_local int tl_device_override = -1;  // multi.hip: the device the next context of THIS thread is created on (-1: g_default_device)

static uint64_t g1_values_digest(const g1_t *g1) {
    uint64_t h = 0xcbf29ce484222325ull;  // FNV-1a over the first and the last setup point
    const uint8_t *a = (const uint8_t *)&g1[0], *b = (const uint8_t *)&g1[kBlobElems - 1];
    for (size_t i = 0; i < sizeof(g1_t); i++) h = (h ^ a[i]) * 0x100000001b3ull;
    for (size_t i = 0; i < sizeof(g1_t); i++) h = (h ^ b[i]) * 0x100000001b3ull;
    return h;
}

// Every call that touches the shared workspace on stream `st` brackets its enqueues with this: the stream first waits
// for whatever used the workspace last (a no-op on the same stream), and leaves the event for the next user.
struct WsUse {
    Ctx *c;
    hipStream_t st;
    WsUse(Ctx *c_, hipStream_t st_) : c(c_), st(st_) {
        if (c->ws_last != st) hipStreamWaitEvent(st, c->ws_done, 0);
        for (int k = 0; k < kCombineLanes; k++) hipStreamWaitEvent(st, c->lane_done[k], 0);  // (never recorded: no-op)
    }
    ~WsUse() {
        hipEventRecord(c->ws_done, st);
        c->ws_last = st;
        c->ws_recorded.store(true, std::memory_order_release);
    }
};

// Is the settings' OTHER context (the twin of a primary, the primary of a twin) at work on the GPU right now? Its last
// whole-workspace user left ws_done behind: an event that has not completed yet means a call of the other caller stream is
// in flight. This -- not the mere existence of a twin -- is what makes the direct MSM pick finer workgroups (VERDICT r03:
// a settings object that once saw two streams kept the two-stream geometry for the rest of its life).
static bool peer_busy(const Ctx *c) {
    const Ctx *p = c->is_twin ? c->primary : c->primary->twin.load(std::memory_order_acquire);
    if (!p || !p->ws_recorded.load(std::memory_order_acquire)) return false;
    const bool busy = hipEventQuery(p->ws_done) != hipSuccess;
    (void)hipGetLastError();  // hipErrorNotReady is an answer
    return busy;
}

// A lane of the coalescing front uses ONE half of the workspace on its own stream: it waits for the last user of the
// whole workspace and leaves an event of its own; the two lanes do not wait for each other.
struct WsLaneUse {
    Ctx *c;
    int lane;
    WsLaneUse(Ctx *c_, int lane_) : c(c_), lane(lane_) { hipStreamWaitEvent(c->aux[lane], c->ws_done, 0); }
    ~WsLaneUse() {
        hipEventRecord(c->lane_done[lane], c->aux[lane]);
        c->ws_last = nullptr;  // the next whole-workspace user must wait for its predecessor's event again: a lane ran between
    }
};

static C_KZG_RET dev_alloc(void **p, size_t bytes) {
    LWK_HIP(hipMalloc(p, bytes));
    return C_KZG_OK;
}

template <class T>
static void dev_free(T *&p) {
    if (p) hipFree(p);
    p = nullptr;
}

static void ws_free(Workspace &w) {
    dev_free(w.blobs);
    dev_free(w.scalars);
    dev_free(w.scalars2);
    dev_free(w.fr);
    dev_free(w.sorted);
    dev_free(w.bucket_start);
    dev_free(w.perm);
    dev_free(w.buckets);
    dev_free(w.sums);
    dev_free(w.out48);
    dev_free(w.comm48);
    dev_free(w.canon48);
    dev_free(w.zbytes);
    dev_free(w.ybytes);
    dev_free(w.z);
    dev_free(w.status);
    dev_free(w.val_pts);
    dev_free(w.val_kind);
    dev_free(w.val_verdict);
    w.cap = 0;
}

static void ws_long_free(Workspace &w) {
    dev_free(w.z_long);
    dev_free(w.canon_long);
    dev_free(w.status_long);
    w.long_cap = 0;
    dev_free(w.val_pts_long);
    dev_free(w.val_kind_long);
    dev_free(w.val_verdict_long);
}

C_KZG_RET ctx_reserve(Ctx *c, size_t n) {
    if (n > kMaxChunk) n = kMaxChunk;
    if (n == 0) n = 1;
    Workspace &w = c->ws;
    if (w.cap >= n) return C_KZG_OK;
    LWK_HIP(hipSetDevice(c->device));
    LWK_HIP(hipDeviceSynchronize());  // work on ANY stream (the caller's included) may still be using the old buffers
    ws_free(w);
    // round up so that repeated small growth does not reallocate every call
    size_t cap = 1;
    while (cap < n) cap <<= 1;
    C_KZG_RET rc;
#define WS_ALLOC(field, bytes)                                  \
    if ((rc = dev_alloc((void **)&w.field, (bytes))) != C_KZG_OK) { \
        ws_free(w);                                             \
        return C_KZG_MALLOC;                                    \
    }
    WS_ALLOC(blobs, cap * (size_t)kBlobBytes);
    WS_ALLOC(scalars, cap * (size_t)kBlobBytes);
    WS_ALLOC(scalars2, cap * (size_t)kBlobBytes);
    WS_ALLOC(fr, cap * (size_t)kBlobBytes);
    WS_ALLOC(sorted, cap * (size_t)kMaxEntries * 4);
    WS_ALLOC(bucket_start, cap * (size_t)(kNumBuckets + 1) * 4);
    WS_ALLOC(perm, cap * (size_t)(kNumBuckets + 1) * 4);
    WS_ALLOC(buckets, cap * (size_t)kNumBuckets * sizeof(G1Xyzz29));
    WS_ALLOC(sums, (cap + 1) * sizeof(G1Xyzz29));  // + one slot for the running total of a tiled MSM
    WS_ALLOC(out48, cap * 48);
    WS_ALLOC(comm48, cap * 48);
    WS_ALLOC(canon48, cap * 48);
    WS_ALLOC(zbytes, cap * 32);
    WS_ALLOC(ybytes, cap * 32);
    WS_ALLOC(z, cap * sizeof(Fr));
    WS_ALLOC(status, cap * 4);
    WS_ALLOC(val_pts, cap * sizeof(G1Affine29));
    WS_ALLOC(val_kind, cap * 4);
    WS_ALLOC(val_verdict, cap * 4);
#undef WS_ALLOC
    w.cap = cap;
    return C_KZG_OK;
}

static void vs_free(Ctx *c);
void direct_from_env(const KZGSettings *s);

static void sph_free(SmallProofHost &h) {
    if (h.blobs) hipHostFree(h.blobs);
    if (h.comm) hipHostFree(h.comm);
    if (h.canon) hipHostFree(h.canon);
    if (h.dig) hipHostFree(h.dig);
    if (h.code) hipHostFree(h.code);
    for (hipEvent_t e : h.chunk_done)
        if (e) hipEventDestroy(e);
    for (hipEvent_t e : h.hashed)
        if (e) hipEventDestroy(e);
    h = SmallProofHost();
}

// Up to this many blobs a device-resident blob-proof call takes its Fiat-Shamir challenges and its commitment validation from the
// host threads (0 = never). Default 64 since r05 (the GPU's validation went from 2.0 to 1.1 ms: at 128 blobs the mid-size path below takes 2.9 ms where this one
// takes 3.6, at 64 both 2.25, at 16 this one 1.2 against 1.8; gpurun_out r05/gpu14); r04's table, when the default was 128 (default engine, same box, ms per call with / without: 1 blob 0.80 / 3.62, 16: 0.87 / 3.61, 64: 1.96 / 4.29,
// 128: 3.41 / 5.05, 256: 6.43 / 6.68 -- beyond that the copy out and the host threads cost what the GPU chains did; gpurun_out r04c).
static std::atomic<int> g_last_proof_schedule{-1};   // test hook (lwkzg_last_proof_schedule): the schedule the last device-resident proof call took
static size_t small_proof_host_limit() {
    const size_t x = knobs().small_proof_host;
    return x > kMaxChunk ? kMaxChunk : x;
}

// at_reserve: called from lwkzg_reserve* (the place to pay for pinned memory). From inside a call (at_reserve = false) the staging
// only grows on settings objects whose owner never reserved anything -- growing means a device synchronisation and up to 64 MiB of
// hipHostMalloc, which a caller that reserved (a stream-capturing or latency-sensitive one) was promised not to meet:
// there the call takes the GPU hash path instead.
static bool sph_reserve(Ctx *c, size_t n, bool at_reserve = false) {
    SmallProofHost &h = c->sph;
    if (h.cap >= n) return true;
    if (!at_reserve && c->primary->reserved.load(std::memory_order_acquire)) return false;
    hipDeviceSynchronize();  // a host function of an earlier call may still be reading the old buffers
    sph_free(h);
    size_t cap = 16;
    while (cap < n) cap <<= 1;
    const bool ok = hipHostMalloc((void **)&h.blobs, cap * (size_t)kBlobBytes) == hipSuccess && hipHostMalloc((void **)&h.comm, cap * 48) == hipSuccess &&
                    hipHostMalloc((void **)&h.canon, cap * 48) == hipSuccess && hipHostMalloc((void **)&h.dig, cap * 32) == hipSuccess &&
                    hipHostMalloc((void **)&h.code, cap * 4) == hipSuccess;
    bool ev_ok = ok;
    for (int k = 0; ev_ok && k < SmallProofHost::kChunks; k++)
        ev_ok = hipEventCreateWithFlags(&h.chunk_done[k], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&h.hashed[k], hipEventDisableTiming) == hipSuccess;
    if (!ev_ok) {
        (void)hipGetLastError();
        sph_free(h);
        return false;
    }
    h.cap = cap;
    return true;
}

// Between the small calls and the large ones: up to this many blobs the Fiat-Shamir hashing of a device-resident blob-proof call runs
// on the host threads too, but PIPELINED -- the blobs leave in chunks on a side stream and every chunk is hashed while the next one is
// on its way -- and beside the GPU's commitment validation (which stays on the GPU: 2 ms whatever the batch, where the host would need
// 0.2 ms per point per thread). 256 blobs: the challenges in ~1.6 ms instead of the hash kernel's flat 3.2.
static size_t mid_proof_chunks() {   // experiment knob: host functions (= chunks) per mid-size call, 1 .. SmallProofHost::kChunks
    const size_t x = knobs().mid_proof_chunks;
    return x < 1 ? 1 : x > (size_t)SmallProofHost::kChunks ? (size_t)SmallProofHost::kChunks : x;
}
static size_t mid_proof_host_limit() {
    const size_t x = knobs().mid_proof_host;
    return x > kMaxChunk ? kMaxChunk : x;
}

// the knobs a plan depends on, clamped as the engine uses them (plan.h: PlanKnobs)
static PlanKnobs plan_knobs() {
    PlanKnobs k;
    k.small_proof_host = small_proof_host_limit();
    k.mid_proof_host = mid_proof_host_limit();
    k.mid_proof_chunks = mid_proof_chunks();
    k.mid_proof_pipe = knobs().mid_proof_pipe;
    k.mid_proof_pipe_min = knobs().mid_proof_pipe_min;
    k.mid_proof_parts = knobs().mid_proof_parts;
    k.heavy_serial = knobs().heavy_serial;
    return k;
}

// Are the host threads warm? The mid-size host-assisted challenge (hashing on the host threads, pipelined with the copy out) beats
// the GPU's hash kernel only when they are: a pool that does not exist yet, or threads idle for longer than LWKZG_HOST_WARM_MS
// (default 2000; measured: up to a second of idleness costs the first two calls 0.2 ms, tools/experiments/r05_host_cold.py), make the
// call take the GPU kernel this once -- and wake the threads on the side, from a host function, so that the NEXT call finds them warm
// (a cold burst of host-assisted calls ran at 30.7k proofs/s where the GPU path does 41.5k).
static std::atomic<int64_t> g_wake_requested_ns{0};
static void host_warm_fn(void *);
static bool host_assist_warm() {
    const int64_t window_ns = (int64_t)(knobs().host_warm_ms < 0 ? 0 : knobs().host_warm_ms) * 1000000;
    const int64_t last = host_last_active_ns(), woken = g_wake_requested_ns.load(std::memory_order_acquire), now = host_now_ns();
    // (a wake-up that is on its way counts: the calls of a burst are enqueued within microseconds of each other, before the first one's
    // wake-up has run, and their host functions only run after the first call's GPU work -- milliseconds later, on threads that are awake)
    return (last != 0 && now - last <= window_ns) || (woken != 0 && now - woken <= window_ns);
}
// enqueue the wake-up of the host threads on a stream the call does not wait for
static void launch_host_wake(Ctx *c) {
    g_wake_requested_ns.store(host_now_ns(), std::memory_order_release);
    if (hipLaunchHostFunc(c->aux[1], host_warm_fn, nullptr) != hipSuccess) (void)hipGetLastError();
}
static void host_warm_fn(void *) {
    try {
        host_pool_warm();
    } catch (...) {
    }
}

struct ChunkHashArgs {
    SmallProofHost *h;
    size_t first, count;
};
static void chunk_midstate_host_fn(void *p) {   // the one-pass commit-and-prove: the commitment-free 2048 blocks of a chunk's hashes
    ChunkHashArgs *a = (ChunkHashArgs *)p;
    SmallProofHost &h = *a->h;
    try {
        challenge_midstates_host((uint32_t *)(h.dig + 32 * a->first), h.blobs + a->first * (size_t)kBlobBytes, a->count);
    } catch (...) {
        memset(h.dig + 32 * a->first, 0, 32 * a->count);
    }
    delete a;
}
static void chunk_hash_host_fn(void *p) {
    ChunkHashArgs *a = (ChunkHashArgs *)p;
    SmallProofHost &h = *a->h;
    try {
        challenge_digests_host(h.dig + 32 * a->first, h.blobs + a->first * (size_t)kBlobBytes, h.comm + 48 * a->first, a->count);
    } catch (...) {
        memset(h.dig + 32 * a->first, 0, 32 * a->count);   // (host_parallel_for does not throw; a wrong digest would show as a wrong proof, never silently)
    }
    delete a;
}

// Mid-size host-assisted hashing: the blobs leave on `sc` chunk by chunk and every chunk's host function runs on `sf` once its copy
// has landed. All argument blocks are allocated BEFORE anything is enqueued (an allocation failure returns with nothing in flight);
// if a host function cannot be launched, `join` is made to wait for everything this function has put in flight -- the side copies and
// the host functions already queued -- so that the caller can return the error without leaving work running against its buffers and
// the context's staging.
static C_KZG_RET chunk_host_functions(Ctx *c, const uint8_t *blobs, size_t n, hipStream_t sc, hipStream_t sf, hipStream_t join,
                                      void (*fn)(void *), const char *prof_name, bool record_hashed = false) {
    SmallProofHost &h = c->sph;
    const size_t per = (n + mid_proof_chunks() - 1) / mid_proof_chunks();
    std::vector<ChunkHashArgs *> args;
    for (size_t first = 0; first < n; first += per) {
        ChunkHashArgs *a = new (std::nothrow) ChunkHashArgs{&h, first, n - first < per ? n - first : per};
        if (!a) {
            for (ChunkHashArgs *b : args) delete b;
            return C_KZG_MALLOC;
        }
        args.push_back(a);
    }
    for (size_t k = 0; k < args.size(); k++) {
        const size_t first = args[k]->first, cnt = args[k]->count;
        hipError_t e = hipMemcpyAsync(h.blobs + first * (size_t)kBlobBytes, blobs + first * (size_t)kBlobBytes, cnt * (size_t)kBlobBytes,
                                      hipMemcpyDeviceToHost, sc);
        if (e == hipSuccess) e = hipEventRecord(h.chunk_done[k], sc);
        if (e == hipSuccess) e = hipStreamWaitEvent(sf, h.chunk_done[k], 0);
        bool queued = false;   // once the host function is in the queue args[k] is ITS to delete (a failing event record
                               // behind a successful launch freed it here as well -- a use-after-free in the callback, then a double free)
        if (e == hipSuccess) {
            ProfScope p(prof_name, sf);
            e = hipLaunchHostFunc(sf, fn, args[k]);
            queued = e == hipSuccess;
        }
        if (e == hipSuccess && record_hashed) e = hipEventRecord(h.hashed[k], sf);
        if (e != hipSuccess) {
            for (size_t j = queued ? k + 1 : k; j < args.size(); j++) delete args[j];
            set_error("host-assisted challenge: %s", hipGetErrorString(e));
            (void)hipGetLastError();
            if (hipEventRecord(c->ev_join[kMaxSplit - 2], sc) == hipSuccess) hipStreamWaitEvent(join, c->ev_join[kMaxSplit - 2], 0);
            if (sf != join && hipEventRecord(c->ev_join[kMaxSplit - 3], sf) == hipSuccess) hipStreamWaitEvent(join, c->ev_join[kMaxSplit - 3], 0);
            return C_KZG_ERROR;
        }
    }
    return C_KZG_OK;
}

struct SmallProofArgs {
    SmallProofHost *h;
    size_t n;
    int32_t bad;
};

// runs on the runtime's callback thread, in stream order, between the copies out and the copies back: no HIP calls in here
static void small_proof_host_fn(void *p) {
    SmallProofArgs *a = (SmallProofArgs *)p;
    SmallProofHost &h = *a->h;
    try {
        challenge_digests_host(h.dig, h.blobs, h.comm, a->n);     // compute_challenge, src/utils.rs:120-154, over the caller's commitment bytes
        std::vector<int> vrc(a->n);
        host_validate_commitments(h.comm, h.canon, vrc.data(), a->n);  // src/lib.rs:372-375: decompress + subgroup check; canonical re-compression
        for (size_t i = 0; i < a->n; i++) h.code[i] = vrc[i] == 2 ? a->bad : 0;
    } catch (...) {
        for (size_t i = 0; i < a->n; i++) h.code[i] = kStatusError;
    }
    delete a;
}


static void ctx_destroy(Ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();  // every stream of the context, and the callers' streams that ran its work
    if (Ctx *t = c->twin.load(std::memory_order_acquire)) {
        ctx_destroy(t);
        c->twin.store(nullptr, std::memory_order_release);
    }
    ws_free(c->ws);
    ws_long_free(c->ws);
    vs_free(c);
    if (c->is_twin) {  // the tables belong to the context this one shadows
        c->points = nullptr;
        c->table = nullptr;
        c->direct_table = nullptr;
        c->direct_tab = DirectTable();
        c->lag = LagrangeForm();
        c->tw_fwd = c->tw_inv = nullptr;
        c->tw28_fwd = c->tw28_inv = nullptr;
    }
    dev_free(c->points);
    dev_free(c->table);
    free_direct_table(c->direct_tab);
    c->direct_table = nullptr;
    free_direct_table(c->lag.direct_tab);
    dev_free(c->lag.points);
    dev_free(c->lag.table);
    dev_free(c->tw_fwd);
    dev_free(c->tw_inv);
    dev_free(c->tw28_fwd);
    dev_free(c->tw28_inv);
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->vstream) hipStreamDestroy(c->vstream);
    for (int k = 0; k < kMaxSplit; k++) {
        if (c->aux[k]) hipStreamDestroy(c->aux[k]);
        if (c->ev_join[k]) hipEventDestroy(c->ev_join[k]);
    }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ws_done) hipEventDestroy(c->ws_done);
    if (c->heavy_done) hipEventDestroy(c->heavy_done);
    for (int k = 0; k < kCombineLanes; k++) {
        if (c->lane_done[k]) hipEventDestroy(c->lane_done[k]);
        if (c->comb.pinned_out[k]) hipHostFree(c->comb.pinned_out[k]);
        if (c->comb.pinned_status[k]) hipHostFree(c->comb.pinned_status[k]);
    }
    if (c->comb.pinned_blobs) hipHostFree(c->comb.pinned_blobs);
    dev_free(c->host_res);
    dev_free(c->vblobs);
    if (c->prio_copy) hipStreamDestroy(c->prio_copy);
    if (c->one_pin) hipHostFree(c->one_pin);
    for (int k = 0; k < 2; k++) {
        dev_free(c->stage.slot[k]);
        if (c->stage.copied[k]) hipEventDestroy(c->stage.copied[k]);
        if (c->stage.parsed[k]) hipEventDestroy(c->stage.parsed[k]);
    }
    sph_free(c->sph);
    free(c->fs.expanded_roots_of_unity);
    free(c->fs.reverse_roots_of_unity);
    free(c->fs.roots_of_unity);
    c->magic = 0;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        g_live_fs.erase((const void *)c);
    }
    delete c;
}

static bool gpu_available() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device available: this library has no CPU fallback (hipGetDeviceCount)");
        return false;
    }
    return true;
}

// fs tables: canonical integers in the reference's limb convention (most-significant u64 first)
static void fr_raw_to_blst(blst_fr *o, const uint32_t raw[8]) {
    for (int k = 0; k < 4; k++) o->l[3 - k] = (uint64_t)raw[2 * k] | ((uint64_t)raw[2 * k + 1] << 32);
}

static C_KZG_RET ctx_new(Ctx **out, const Ctx *twin_of = nullptr) {
    if (!gpu_available()) return C_KZG_ERROR;
    Ctx *c = new Ctx();
    memset(&c->fs, 0, sizeof c->fs);
    c->magic = kCtxMagic;
    c->device = twin_of ? twin_of->device : tl_device_override >= 0 ? tl_device_override : g_default_device.load();
    c->is_twin = twin_of != nullptr;
    c->primary = twin_of ? const_cast<Ctx *>(twin_of) : c;
    c->stream = nullptr;
    c->ws_done = nullptr;
    c->ws_last = nullptr;
    for (int k = 0; k < kCombineLanes; k++) c->lane_done[k] = nullptr;
    c->vstream = nullptr;
    c->ev_fork = nullptr;
    for (int k = 0; k < kMaxSplit; k++) {
        c->aux[k] = nullptr;
        c->ev_join[k] = nullptr;
    }
    c->points = nullptr;
    c->table = nullptr;
    c->direct_table = nullptr;
    c->direct_bits = 0;
    c->direct_row_bytes = kDirectRowPacked;
    c->vs_cap = 0;
    c->tw_fwd = c->tw_inv = nullptr;
    c->tw28_fwd = c->tw28_inv = nullptr;
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipStreamCreate(&c->stream);
    // validation kernels get a stream of their own. (A CU-masked stream -- hipExtStreamCreateWithCUMask, upper half of
    // every XCD, which keeps these one-wave-per-SIMD kernels off the SIMDs of the Fiat-Shamir kernel and was worth 4 % on
    // 1024-proof batches -- made free_trusted_setup hang in about half the runs whenever a second process was using
    // the GPU, and was removed.)
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->vstream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ws_done, hipEventDisableTiming);
    if (e == hipSuccess && !twin_of) e = hipEventCreateWithFlags(&c->heavy_done, hipEventDisableTiming);
    for (int k = 0; k < kCombineLanes && e == hipSuccess; k++) e = hipEventCreateWithFlags(&c->lane_done[k], hipEventDisableTiming);
    for (int k = 0; k < kMaxSplit && e == hipSuccess; k++) {
        e = hipStreamCreateWithFlags(&c->aux[k], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join[k], hipEventDisableTiming);
    }
    if (twin_of) {  // same read-only tables, own streams / events / workspace / locks
        c->points = twin_of->points;
        c->table = twin_of->table;
        c->direct_table = twin_of->direct_table;
        c->direct_tab.win_dev = twin_of->direct_tab.win_dev;  // (addresses only: the windows belong to the primary context)
        c->direct_bits = twin_of->direct_bits;
        c->direct_row_bytes = twin_of->direct_row_bytes;
        c->lag = twin_of->lag;
        c->lag.direct_tab = DirectTable();
        c->lag.direct_tab.win_dev = twin_of->lag.direct_tab.win_dev;  // (addresses only)
        c->tw_fwd = twin_of->tw_fwd;
        c->tw_inv = twin_of->tw_inv;
        c->tw28_fwd = twin_of->tw28_fwd;
        c->tw28_inv = twin_of->tw28_inv;
    } else {
        if (e == hipSuccess) e = hipMalloc((void **)&c->points, (size_t)kBlobElems * sizeof(G1Affine));
        if (e == hipSuccess) e = hipMalloc((void **)&c->table, (size_t)kTablePoints * sizeof(G1Affine29));
        if (e == hipSuccess) e = hipMalloc((void **)&c->tw_fwd, (size_t)(kBlobElems / 2) * sizeof(Fr));
        if (e == hipSuccess) e = hipMalloc((void **)&c->tw_inv, (size_t)(kBlobElems / 2) * sizeof(Fr));
        // (behind the 2048 forward twiddles: the 4096 domain points in element order, for the evaluation-form quotient)
        if (e == hipSuccess) e = hipMalloc((void **)&c->tw28_fwd, (size_t)(kBlobElems / 2 + kBlobElems) * sizeof(Fr28));
        if (e == hipSuccess) e = hipMalloc((void **)&c->tw28_inv, (size_t)(kBlobElems / 2) * sizeof(Fr28));
    }
    if (e != hipSuccess) {
        set_error("device context allocation failed: %s", hipGetErrorString(e));
        ctx_destroy(c);
        return C_KZG_MALLOC;
    }
    static std::atomic<uint64_t> next_generation{1};
    c->generation = next_generation.fetch_add(1);
    if (!twin_of) {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        g_live_fs.insert((const void *)c);
    }
    *out = c;
    return C_KZG_OK;
}

// Which of the settings' two contexts a device-resident call on caller stream `st` runs on. The engine's own stream and
// a stream that used this context last stay here (stream order is all the synchronisation they need); a call on another
// stream goes to the twin while this context's workspace is still busy, so that the two calls overlap on the GPU.
static bool twin_off() {
    return !knobs().twin;
}

static Ctx *pick_ctx(Ctx *c, hipStream_t st) {
    if (twin_off() || !st || st == c->stream) return c;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->ws_last == st || c->ws_last == nullptr || hipEventQuery(c->ws_done) == hipSuccess) {
        (void)hipGetLastError();
        return c;
    }
    (void)hipGetLastError();  // hipErrorNotReady is an answer
    if (!c->twin.load(std::memory_order_acquire)) {
        Ctx *t = nullptr;
        if (ctx_new(&t, c) != C_KZG_OK) return c;
        c->twin.store(t, std::memory_order_release);
    }
    return c->twin.load(std::memory_order_acquire);
}

// twiddles on device + the genuine FFTSettings tables on the host
static C_KZG_RET ctx_finish_fft(Ctx *c) {
    launch_build_twiddles(c->tw_fwd, c->tw_inv, c->stream);
    launch_twiddles_to28(c->tw_fwd, c->tw28_fwd, c->stream);
    launch_twiddles_to28(c->tw_inv, c->tw28_inv, c->stream);
    launch_roots_brp28(c->tw28_fwd, c->tw28_fwd + kBlobElems / 2, c->stream);
    const int n = kBlobElems;
    std::vector<Fr> h_f(n / 2), h_i(n / 2);
    LWK_HIP(hipMemcpyAsync(h_f.data(), c->tw_fwd, (n / 2) * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
    LWK_HIP(hipMemcpyAsync(h_i.data(), c->tw_inv, (n / 2) * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
    LWK_HIP(hipStreamSynchronize(c->stream));
    c->fs.max_width = n;
    c->fs.expanded_roots_of_unity = (fr_t *)calloc(n + 1, sizeof(fr_t));
    c->fs.reverse_roots_of_unity = (fr_t *)calloc(n + 1, sizeof(fr_t));
    c->fs.roots_of_unity = (fr_t *)calloc(n, sizeof(fr_t));
    if (!c->fs.expanded_roots_of_unity || !c->fs.reverse_roots_of_unity || !c->fs.roots_of_unity) return C_KZG_MALLOC;
    // w^(k + n/2) = -w^k
    for (int k = 0; k <= n; k++) {
        int kk = k % n;
        Fr f = h_f[kk % (n / 2)], b = h_i[kk % (n / 2)];
        if (kk >= n / 2) {
            f = neg(f);
            b = neg(b);
        }
        uint32_t raw[8];
        fe_to_raw<FrParams>(raw, f);
        fr_raw_to_blst(&c->fs.expanded_roots_of_unity[k], raw);
        fe_to_raw<FrParams>(raw, b);
        fr_raw_to_blst(&c->fs.reverse_roots_of_unity[k], raw);
    }
    for (int k = 0; k < n; k++) {
        unsigned r = 0;
        for (int b = 0; b < 12; b++) r |= ((k >> b) & 1u) << (11 - b);
        c->fs.roots_of_unity[k] = c->fs.expanded_roots_of_unity[r];
    }
    return C_KZG_OK;
}

bool ctx_is_live(const Ctx *c, uint64_t gen) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    return c && g_live_fs.count((const void *)c) && c->generation == gen;
}

Ctx *ctx_of(const KZGSettings *s) {
    if (!s) {
        set_error("KZGSettings pointer is NULL");
        return nullptr;
    }
    // fs is only dereferenced when it is one of OUR contexts: a genuine FFTSettings from another c-kzg-style producer
    // is 32 bytes long and has no magic word behind it
    if (s->fs) {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (g_live_fs.count((const void *)s->fs)) return (Ctx *)s->fs;
    }
    // hand-built settings (the reference's own layout: fs == NULL): build and cache a context from g1_values
    if (!s->g1_values) {
        set_error("KZGSettings has neither an engine context nor g1_values");
        return nullptr;
    }
    const uint64_t digest = g1_values_digest(s->g1_values);
    Ctx *stale = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        auto it = g_registry.find(s->g1_values);
        if (it != g_registry.end()) {
            if (it->second.digest == digest) return it->second.ctx;
            stale = it->second.ctx;  // same address, other contents: the caller rebuilt its array
            g_registry.erase(it);
        }
    }
    if (stale) ctx_destroy(stale);
    Ctx *c = nullptr;
    if (ctx_new(&c) != C_KZG_OK) return nullptr;
    uint64_t *d_blst = nullptr;
    int32_t *d_status = nullptr;
    std::vector<int32_t> h_status(kBlobElems);
    bool ok = hipMalloc((void **)&d_blst, (size_t)kBlobElems * 144) == hipSuccess &&
              hipMalloc((void **)&d_status, (size_t)kBlobElems * 4) == hipSuccess &&
              hipMemcpyAsync(d_blst, s->g1_values, (size_t)kBlobElems * 144, hipMemcpyHostToDevice, c->stream) == hipSuccess;
    if (ok) {
        launch_g1_from_blst(d_blst, c->points, d_status, kBlobElems, c->stream);
        launch_build_table(c->points, c->table, c->stream);
        ok = hipMemcpyAsync(h_status.data(), d_status, (size_t)kBlobElems * 4, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
             hipStreamSynchronize(c->stream) == hipSuccess;
    }
    if (d_blst) hipFree(d_blst);
    if (d_status) hipFree(d_status);
    if (ok)
        for (int i = 0; i < kBlobElems; i++)
            if (h_status[i] != 0) {
                set_error("g1_values[%d] is not a point on the curve", i);  // reference: srs.rs:171 error
                ok = false;
                break;
            }
    if (ok) ok = ctx_finish_fft(c) == C_KZG_OK;
    if (!ok) {
        if (!get_error()[0]) set_error("building a device context from g1_values failed");
        ctx_destroy(c);
        return nullptr;
    }
    bool ours = true;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        auto it = g_registry.find(s->g1_values);
        if (it != g_registry.end()) {  // another thread built one meanwhile: keep the first
            Ctx *first = it->second.ctx;
            if (it->second.digest == digest) {
                // (ctx_destroy takes g_reg_mu itself)
                stale = c;
                c = first;
                ours = false;
            } else {
                stale = first;
                it->second = {c, digest};
            }
        } else {
            stale = nullptr;
            g_registry[s->g1_values] = {c, digest};
        }
    }
    if (stale) ctx_destroy(stale);
    if (ours) direct_from_env(s);  // a context of our own making: it gets the engine a loaded setup would get
    return c;
}

// ------------------------------------------------------------------------------------------------
// pipelines (device-resident, asynchronous)

// `base` = first workspace slot (in blobs) this launch set may use: sub-batches running on different streams
// work in disjoint slices of the same workspace.
// `shared_chip`: latency-chain kernels of the same call run beside this MSM (the fused commit-and-prove's hash): finer
// workgroups, as when the settings' other context is busy, so that the compute units they sit on do not set the launch's end
// `lagrange`: the scalars are evaluations on the bit-reversed domain and the MSM runs over the Lagrange form of the setup
static G1Xyzz29 *msm_sums_stage(Ctx *c, const uint32_t *scalars_raw, size_t n, hipStream_t st, size_t base = 0, bool shared_chip = false,
                                bool lagrange = false, G1Xyzz29 *sums_out = nullptr, uint32_t *redo_flag_out = nullptr) {
    Workspace &w = c->ws;
    uint32_t *sorted = w.sorted + base * (size_t)kMaxEntries;
    uint32_t *bstart = w.bucket_start + base * (size_t)(kNumBuckets + 1);
    uint32_t *perm = w.perm + base * (size_t)(kNumBuckets + 1);
    G1Xyzz29 *buckets = w.buckets + base * (size_t)kNumBuckets;
    G1Xyzz29 *sums = sums_out ? sums_out : w.sums + base;   // (sums_out: anywhere the device can write -- pinned host memory for a one-blob call, r06)
    const G1Affine29 *direct = lagrange ? c->lag.direct_table : c->direct_table;
    if (direct) {  // giant-table path: gather + add, nothing else
        // (scratch of the bucket engine, idle on this path: `buckets` holds the per-lane sums of the hand-scheduled kernel,
        // `sorted` the per-workgroup partial sums, `bstart` the redo flags)
        launch_direct_msm(lagrange ? c->lag.direct_bits : c->direct_bits, lagrange ? c->lag.direct_tab.win_dev : c->direct_tab.win_dev,
                          lagrange ? c->lag.direct_row_bytes : c->direct_row_bytes, scalars_raw, buckets, (G1Xyzz29 *)sorted, bstart, sums, n,
                          st, (shared_chip || peer_busy(c)) ? 2048 : 0, redo_flag_out);
        return sums;
    }
    launch_digit_sort(scalars_raw, sorted, bstart, perm, n, st);
    launch_bucket_accumulate(lagrange ? c->lag.table : c->table, sorted, bstart, perm, buckets, n, st);
    launch_bucket_reduce(buckets, sums, n, st);
    return sums;
}

static void msm_stages(Ctx *c, const uint32_t *scalars_raw, uint8_t *out48, size_t n, hipStream_t st, size_t base = 0,
                       bool shared_chip = false, bool lagrange = false) {
    launch_finalize_compress(msm_sums_stage(c, scalars_raw, n, st, base, shared_chip, lagrange), out48, n, st);
}

// ---- the last step of a SMALL host-pointer call on the host -------------------------------------------------------
// One inversion and the compression per result are a dependent chain of ~20k instructions: 0.09-0.1 ms on a lone GPU lane
// (k_finalize_compress) whatever the batch, a few microseconds on a host core -- and the result of a host-pointer call has to
// cross to the host anyway. Calls of up to LWKZG_HOST_FINISH results (default 8, 0 = never) therefore copy the XYZZ sums
// (224 bytes each) instead of the 48 compressed bytes and finish here. Same bytes: x = X / ZZ, y = Y / ZZZ, compress_g1_point
// (/root/reference/src/compression.rs:33-60) through g1.cuh's g1_compress_affine.
static size_t host_finish_limit() {
    const size_t x = knobs().host_finish;
    return x > kCombineMaxBatch ? kCombineMaxBatch : x;
}

// the integer of a hot-loop field value (14 limbs of 28 bits, value < 16p) reduced into [0, p)
static HFp hfp_from_limbs28(const uint32_t *l) {
    uint64_t w[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 14; i++) {
        const int bit = P29::W * i, k = bit >> 6, sh = bit & 63;
        unsigned __int128 v = (unsigned __int128)l[i] << sh;
        for (int j = k; j < 7 && v; j++) {
            v += w[j];
            w[j] = (uint64_t)v;
            v >>= 64;
        }
    }
    // minus 2^16 p, ..., 2p, p where that leaves it non-negative: whatever lazy bound a producer leaves (14 limbs below 2^32 are an integer
    // below 2^396 < 2^17 p), the result is the residue in [0, p) -- ADVICE r05: the four subtractions of r05 silently assumed < 16p
    for (int shift = 16; shift >= 0; shift--) {
        uint64_t kp[7], d[7];
        for (int j = 0; j < 7; j++) {
            const uint64_t lo = j < 6 ? HFpPar::P[j] : 0, below = j > 0 ? HFpPar::P[j - 1] : 0;
            kp[j] = shift ? (lo << shift) | (below >> (64 - shift)) : lo;
        }
        unsigned __int128 br = 0;
        for (int j = 0; j < 7; j++) {
            const unsigned __int128 v = (unsigned __int128)w[j] - kp[j] - (uint64_t)br;
            d[j] = (uint64_t)v;
            br = (v >> 64) & 1;
        }
        if (!br)
            for (int j = 0; j < 7; j++) w[j] = d[j];
    }
    HFp r;
    for (int j = 0; j < 6; j++) r.l[j] = w[j];
    return r;
}

static void host_finish_compress(uint8_t out[48], const G1Xyzz29 &sum) {
    if (sum.is_inf()) {
        memset(out, 0, 48);
        out[0] = 0xc0;
        return;
    }
    // the four limb vectors read as residues all carry the same Montgomery factor, which the two quotients cancel
    const HFp X = hfp_from_limbs28(sum.x.l), Y = hfp_from_limbs28(sum.y.l), ZZ = hfp_from_limbs28(sum.zz.l), ZZZ = hfp_from_limbs28(sum.zzz.l);
    const HFp i = inv(ZZ * ZZZ);
    G1Affine a;
    a.x = (X * (i * ZZZ)).to_fe();
    a.y = (Y * (i * ZZ)).to_fe();
    g1_compress_affine(out, a);
}

// Which form a c-kzg call's MSM runs on. A COMMITMENT's scalars can be had in either form -- the blob's evaluations as they stand,
// or their coefficients behind the transform --, so it takes the Lagrange form whenever that form has a direct table or the monomial
// one has none. A PROOF's quotient is computed in coefficient form (Horner / Ruffini); it moves to the Lagrange form, by one forward
// transform, only when that is where the settings' one direct table is.
static bool commit_on_lagrange(const Ctx *c, int mode) {
    return mode == LWKZG_MODE_CKZG && c->lag.ready && (c->lag.direct_table || !c->direct_table);
}
static bool proof_on_lagrange(const Ctx *c, int mode) {
    return mode == LWKZG_MODE_CKZG && c->lag.ready && c->lag.direct_table && !c->direct_table;
}
// r05: a c-kzg-mode proof whose MSM can run on the Lagrange form at full speed (that form has a direct table, or no form has one)
// computes its quotient in EVALUATION form (fr_ops.hip: k_eval_quotient_evalform): the blob's evaluations as they stand, one batch
// inversion per blob, no transform in front of the quotient and none behind it. LWKZG_CKZG_EVAL_PROOFS=0 is the A/B arm (the inverse
// transform, Horner / Ruffini, and a forward transform where the Lagrange table is the only one).
static bool proof_in_evaluation_form(const Ctx *c, int mode) {
    return knobs().ckzg_eval_proofs && mode == LWKZG_MODE_CKZG && c->lag.ready && (c->lag.direct_table || !c->direct_table);
}
// quotient (and y = p(z)) of n blobs whose scalars coefficients_stage left at `in`, in the form that function chose; quot = nullptr: y only
static void quotient_stage(Ctx *c, int mode, const uint32_t *in, const Fr *z, uint32_t *quot, uint8_t *y_out, int le, size_t n, hipStream_t st,
                           const uint32_t *only_if = nullptr) {
    if (proof_in_evaluation_form(c, mode)) launch_eval_quotient_evalform(in, z, c->tw28_fwd + kBlobElems / 2, quot, y_out, le, n, st, only_if);
    else launch_eval_quotient(in, z, quot, y_out, le, n, st, only_if);
}
// the quotients in w.scalars2 (slots base ..) -> the form their MSM runs on; returns that form (true = Lagrange). The coefficients
// in w.scalars (same slots) are dead by now and serve as scratch.
static bool quotient_to_msm_form(Ctx *c, int mode, size_t n, hipStream_t st, size_t base = 0) {
    if (proof_in_evaluation_form(c, mode)) return true;   // (already evaluations)
    if (!proof_on_lagrange(c, mode)) return false;
    Workspace &w = c->ws;
    const size_t so = base * (size_t)kBlobElems;
    launch_coefficients_to_evaluations(w.scalars2 + so * 8, w.fr + so, (Fr *)(w.scalars + so * 8), c->tw28_fwd, n, st);
    return true;
}

// blob bytes -> the scalars of an MSM in ws.scalars (slots base .. base + n): canonical monomial coefficients, or -- c-kzg mode
// and a usable Lagrange form: `evaluations_ok` (the caller only commits), or a proof call whose quotient is taken in evaluation
// form (proof_in_evaluation_form; quotient_stage reads what this function left) -- the blob's own evaluations, copied and
// range-checked with no transform at all. Returns true in that case: an MSM of these scalars must run on the Lagrange form.
static bool coefficients_stage(Ctx *c, const uint8_t *blobs, size_t n, int mode, int32_t *status, hipStream_t st,
                               size_t base = 0, bool evaluations_ok = false, uint32_t *zero = nullptr, uint32_t zero_words = 0) {
    Workspace &w = c->ws;
    uint32_t *scalars = w.scalars + base * (size_t)kBlobElems * 8;
    if (mode == LWKZG_MODE_REFERENCE) {
        launch_parse_be_reduce(blobs, scalars, n * kBlobElems, st, zero, zero_words);   // (zero: reference mode only -- the caller checks)
    } else if (evaluations_ok ? commit_on_lagrange(c, mode) : proof_in_evaluation_form(c, mode)) {
        launch_copy_le_check(blobs, scalars, status, n, st);
        return true;
    } else {
        launch_blob_evaluations_to_coefficients(blobs, scalars, c->tw28_inv, status, n, st);
    }
    return false;
}

// sub-batches per launch set. The bucket path gains from two (its sort / reduce / inversion tails hide behind the
// other half's accumulation); the direct path has no such tails and runs as one launch.
static int split_ways(bool direct) {
    const int v = knobs().split < 0 ? 0 : knobs().split > kMaxSplit ? kMaxSplit : knobs().split;
    return v ? v : (direct ? 1 : 2);
}

C_KZG_RET msm_scalars_raw_device(Ctx *c, uint8_t *out48, const uint32_t *scalars_raw, size_t n, hipStream_t st) {
    for (size_t off = 0; off < n; off += kMaxChunk) {
        size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        msm_stages(c, scalars_raw + off * (size_t)kBlobElems * 8, out48 + 48 * off, m, st);
    }
    return C_KZG_OK;
}

// One chunk of the commitment pipeline. Large chunks are cut into sub-batches that run on the context's
// auxiliary streams: the latency-shaped tails of one sub-batch (bucket reduction, inversion, the last waves of
// the accumulation) then overlap with the ALU-bound accumulation of the next instead of idling most SIMDs.
C_KZG_RET commit_batch_device(Ctx *c, uint8_t *out48, const uint8_t *blobs, size_t n, int mode, hipStream_t st,
                              int32_t *status) {
    C_KZG_RET rc = ctx_reserve(c, n);
    if (rc != C_KZG_OK) return rc;
    for (size_t off = 0; off < n; off += kMaxChunk) {
        size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        int32_t *stt = status ? status + off : c->ws.status;
        LWK_HIP(hipMemsetAsync(stt, 0, m * 4, st));
        const int ways = m >= 256 ? split_ways((commit_on_lagrange(c, mode) ? c->lag.direct_table : c->direct_table) != nullptr) : 1;
        if (ways == 1) {
            const bool lg = coefficients_stage(c, blobs + off * (size_t)kBlobBytes, m, mode, stt, st, 0, true);
            msm_stages(c, c->ws.scalars, out48 + 48 * off, m, st, 0, false, lg);
            continue;
        }
        LWK_HIP(hipEventRecord(c->ev_fork, st));
        for (int k = 0; k < ways; k++) {
            size_t lo = m * k / ways, hi = m * (k + 1) / ways;
            hipStream_t sk = c->aux[k];
            LWK_HIP(hipStreamWaitEvent(sk, c->ev_fork, 0));
            const bool lg = coefficients_stage(c, blobs + (off + lo) * (size_t)kBlobBytes, hi - lo, mode, stt + lo, sk, lo, true);
            msm_stages(c, c->ws.scalars + lo * (size_t)kBlobElems * 8, out48 + 48 * (off + lo), hi - lo, sk, lo, false, lg);
            LWK_HIP(hipEventRecord(c->ev_join[k], sk));
            LWK_HIP(hipStreamWaitEvent(st, c->ev_join[k], 0));
        }
    }
    return C_KZG_OK;
}

// z and the canonical commitment bytes of a proof call longer than one chunk
static C_KZG_RET ws_long_reserve(Ctx *c, size_t n) {
    Workspace &w = c->ws;
    if (w.long_cap >= n) return C_KZG_OK;
    LWK_HIP(hipDeviceSynchronize());
    ws_long_free(w);
    size_t cap = 2 * kMaxChunk;
    while (cap < n) cap <<= 1;
    if (hipMalloc((void **)&w.z_long, cap * sizeof(Fr)) != hipSuccess || hipMalloc((void **)&w.canon_long, cap * 48) != hipSuccess ||
        hipMalloc((void **)&w.status_long, cap * 4) != hipSuccess || hipMalloc((void **)&w.val_pts_long, cap * sizeof(G1Affine29)) != hipSuccess ||
        hipMalloc((void **)&w.val_kind_long, cap * 4) != hipSuccess || hipMalloc((void **)&w.val_verdict_long, cap * 4) != hipSuccess) {
        (void)hipGetLastError();
        ws_long_free(w);
        set_error("proof batch of %zu blobs: out of device memory for the challenges", n);
        return C_KZG_MALLOC;
    }
    w.long_cap = cap;
    return C_KZG_OK;
}

C_KZG_RET blob_proof_batch_device(Ctx *c, uint8_t *out48, const uint8_t *blobs, const uint8_t *comm48, size_t n, int mode,
                                  hipStream_t st, int32_t *status) {
    C_KZG_RET rc = ctx_reserve(c, n);
    if (rc != C_KZG_OK) return rc;
    const bool longcall = n > kMaxChunk;
    if (longcall) {
        rc = ws_long_reserve(c, n);
        if (rc != C_KZG_OK) return rc;
    }
    Workspace &w = c->ws;
    const int le = mode == LWKZG_MODE_CKZG;
    int32_t *stt = status ? status : longcall ? w.status_long : w.status;
    Fr *z = longcall ? w.z_long : w.z;
    uint8_t *canon = longcall ? w.canon_long : w.canon48;
    LWK_HIP(hipMemsetAsync(stt, 0, n * 4, st));
    // lib.rs:372-375: the commitment is decompressed (and subgroup-checked) first. Validation (a long scalar
    // multiplication per lane) and hashing (131 KB per lane pair) are both latency chains whose run time does not depend
    // on the number of blobs (2 ms and 3.2 ms for anything from 64 blobs to one workgroup per compute unit, 16k), and independent as long as
    // the caller's bytes are the canonical encoding, which they are except for exotic encodings of infinity: ALL
    // blobs of the call are hashed optimistically from the caller's bytes on `st` while the validation stream
    // validates all commitments, then only the lanes whose canonical bytes differ are redone. A call of several chunks
    // therefore pays the two chains once, not once per chunk.
    // WHICH schedule runs in front of the MSM is a pure function of the batch, the host threads' warmth, the other context's state, the
    // table forms and the knobs (plan.h: plan_proof_call; the table is pinned by tests/test_plan_cpu.py). The engine's part: the inputs,
    // the pinned staging the host-assisted schedules need (asked for only when the plan wants it; if it cannot be had the plan is taken
    // again without it), and the launches.
    const bool evf = proof_in_evaluation_form(c, mode);   // (then the quotient's MSM runs on the Lagrange form's table)
    const bool msm_direct = evf ? c->lag.direct_table != nullptr : c->direct_table && !proof_on_lagrange(c, mode);
    const bool warm = host_assist_warm(), busy = peer_busy(c);
    const PlanKnobs pk = plan_knobs();
    ProofPlan plan = plan_proof_call(n, warm, busy, msm_direct, true, pk);
    const int forced = knobs().proof_schedule;   // experiment: tests/test_gpu_plan.py runs every schedule once on the same inputs
    if (forced >= 0 && forced <= kProofGpuChains && n <= kMaxChunk) {
        PlanKnobs fk = pk;
        fk.small_proof_host = forced == kProofSmallHost ? kMaxChunk : 0;
        fk.mid_proof_host = forced == kProofSmallHost || forced == kProofGpuChains ? 0 : kMaxChunk;
        fk.mid_proof_pipe = forced == kProofMidPiped;
        fk.mid_proof_pipe_min = 0;
        plan = plan_proof_call(n, forced != kProofMidCold, false, forced == kProofMidPiped ? msm_direct : false, true, fk);
    }
    SmallProofArgs *host_args = nullptr;
    if (plan.needs_staging()) {
        bool ok = sph_reserve(c, n);
        if (ok && plan.schedule == kProofSmallHost) {
            host_args = new (std::nothrow) SmallProofArgs{&c->sph, n, le ? kStatusBadArgs : kStatusError};
            ok = host_args != nullptr;
        }
        if (!ok) plan = plan_proof_call(n, warm, busy, msm_direct, false, pk);
    }
    g_last_proof_schedule.store((int)plan.schedule, std::memory_order_relaxed);
    const bool piped = plan.schedule == kProofMidPiped;
    switch (plan.schedule) {
    case kProofSmallHost: {
        // A small call: both chains above cost their full 2-3 ms for a handful of blobs, and the host does the same work in a tenth
        // of that (3.63 ms device-resident against 0.82 ms through the host-pointer ABI at one blob). The blobs and
        // commitments go out to pinned memory, a host function hashes and validates them on the host threads IN STREAM ORDER (the
        // call stays asynchronous), digests, verdicts and canonical bytes come back; a lane whose canonical commitment bytes differ
        // from the caller's (an exotic encoding of infinity) is rehashed on the GPU as in the large-batch path.
        SmallProofHost &h = c->sph;
        LWK_HIP(hipMemcpyAsync(h.blobs, blobs, n * (size_t)kBlobBytes, hipMemcpyDeviceToHost, st));
        LWK_HIP(hipMemcpyAsync(h.comm, comm48, n * 48, hipMemcpyDeviceToHost, st));
        {
            ProfScope p("host_challenge_and_validate", st);
            hipError_t e = hipLaunchHostFunc(st, small_proof_host_fn, host_args);
            if (e != hipSuccess) {
                delete host_args;
                set_error("hipLaunchHostFunc failed: %s", hipGetErrorString(e));
                return C_KZG_ERROR;
            }
        }
        LWK_HIP(hipMemcpyAsync(stt, h.code, n * 4, hipMemcpyHostToDevice, st));
        LWK_HIP(hipMemcpyAsync(w.zbytes, h.dig, n * 32, hipMemcpyHostToDevice, st));
        LWK_HIP(hipMemcpyAsync(canon, h.canon, n * 48, hipMemcpyHostToDevice, st));
        launch_z_from_bytes(w.zbytes, z, nullptr, le, n, st);  // digest -> Fr, reduced (utils.rs:148-154)
        launch_challenge(blobs, canon, z, le, n, st, comm48);
        break;
    }
    case kProofMidCold: {
        // mid-size call, host threads cold: the GPU's hash kernel this once, and a nudge for the threads on the side stream
        LWK_HIP(hipEventRecord(c->ev_fork, st));
        LWK_HIP(hipStreamWaitEvent(c->vstream, c->ev_fork, 0));
        launch_validate_commitments(comm48, canon, stt, le ? kStatusBadArgs : kStatusError, n, c->vstream, longcall ? w.val_pts_long : w.val_pts,
                                    longcall ? w.val_kind_long : w.val_kind, longcall ? w.val_verdict_long : w.val_verdict);
        LWK_HIP(hipEventRecord(c->ev_join[0], c->vstream));
        // (the wake-up on a stream this call does not wait for: on the validation's stream, in front of the join event, the first call of a
        // process paid for the creation of the thread pool)
        launch_host_wake(c);
        launch_challenge(blobs, comm48, z, le, n, st);
        LWK_HIP(hipStreamWaitEvent(st, c->ev_join[0], 0));
        launch_challenge(blobs, canon, z, le, n, st, comm48);
        break;
    }
    case kProofMidHost:
    case kProofMidPiped: {
        // mid-size call on a settings object whose other context is idle (a producer that alternates two caller streams hides the GPU's hash
        // behind the other call's MSM at no cost, and two calls' host hashing would queue for the same host threads: 70.9k against 49.7k
        // proofs/s at 256 blobs on two streams -- there the hash kernel stays): the validation on the GPU's side stream as in the large path; the hashing on the host threads, chunk by chunk
        // while the next chunk is still being copied out (side stream c->aux[0], one event per chunk)
        SmallProofHost &h = c->sph;
        hipStream_t sc = c->aux[0], sh = c->aux[1];
        // PIPELINED when the quotient's MSM runs on a monomial direct table (r05): the host functions run on a stream of their own, the
        // first half's quotient and MSM start as soon as ITS chunks are hashed -- beside the hashing of the second half, and without
        // waiting for the validation either (2 ms flat on its side stream): the challenges are taken over the caller's commitment
        // bytes, and once the validation's canonical bytes exist a blob whose bytes were NOT canonical (a valid point in an exotic
        // encoding: infinity with stray bits) gets its challenge, quotient and MSM again in a second pass that exits at its first
        // instruction for every other blob. 256 blobs: hash 2.1 ms -> MSM 2.3 ms in series becomes 1.05 -> 1.2 || 1.05 -> 1.2.
        // (plan.h decides `piped`: the knob, a direct table under the quotient's MSM, at least mid_proof_pipe_min blobs -- 128 blobs: 3.3 ms
        // pipelined, 2.9 not --, an even number of chunks, one launch set)
        const size_t chunks = plan.chunks;
        LWK_HIP(hipEventRecord(c->ev_fork, st));
        LWK_HIP(hipStreamWaitEvent(c->vstream, c->ev_fork, 0));
        LWK_HIP(hipStreamWaitEvent(sc, c->ev_fork, 0));
        if (piped) LWK_HIP(hipStreamWaitEvent(sh, c->ev_fork, 0));
        launch_validate_commitments(comm48, canon, stt, le ? kStatusBadArgs : kStatusError, n, c->vstream, longcall ? w.val_pts_long : w.val_pts,
                                    longcall ? w.val_kind_long : w.val_kind, longcall ? w.val_verdict_long : w.val_verdict);
        LWK_HIP(hipEventRecord(c->ev_join[0], c->vstream));
        LWK_HIP(hipMemcpyAsync(h.comm, comm48, n * 48, hipMemcpyDeviceToHost, sc));
        {
            const C_KZG_RET rch = chunk_host_functions(c, blobs, n, sc, piped ? sh : st, st, chunk_hash_host_fn, "host_challenge_chunk", piped);
            if (rch != C_KZG_OK) {
                hipStreamWaitEvent(st, c->ev_join[0], 0);   // the validation on the side stream: nothing of this call outlives `st`
                return rch;
            }
        }
        if (piped) {
            const size_t per = plan.per_chunk;
            // sub-batches of whole chunks, alternating between the call's stream and a second one: a sub-batch's quotient and MSM start
            // when ITS chunks are hashed, and its latency-shaped folds run beside the next sub-batch's accumulation
            // (plan.h: four measured best at 256 while the validation took 2 ms; with 1.1 ms of it two win there, 61.9k against 59.7k proofs/s; at 384 four: 66.0k against 63.7k)
            const size_t parts = plan.parts;
            const size_t cps = chunks / parts;   // chunks per sub-batch
            uint32_t *differs = w.perm;   // (bucket-engine scratch, idle on a direct table: one word per blob)
            hipStream_t s2 = c->aux[2];
            LWK_HIP(hipStreamWaitEvent(s2, c->ev_fork, 0));
            for (size_t j = 0; j < parts; j++) {
                const size_t off = j * cps * per, end_ = (j + 1) * cps * per < n ? (j + 1) * cps * per : n;
                if (off >= end_) continue;
                const size_t m = end_ - off;
                hipStream_t sk = (j & 1) ? s2 : st;
                LWK_HIP(hipStreamWaitEvent(sk, h.hashed[(j + 1) * cps - 1], 0));
                LWK_HIP(hipMemcpyAsync(w.zbytes + 32 * off, h.dig + 32 * off, m * 32, hipMemcpyHostToDevice, sk));
                launch_z_from_bytes(w.zbytes + 32 * off, z + off, nullptr, le, m, sk);
                coefficients_stage(c, blobs + off * (size_t)kBlobBytes, m, mode, stt + off, sk, off);
                quotient_stage(c, mode, w.scalars + off * (size_t)kBlobElems * 8, z + off, w.scalars2 + off * (size_t)kBlobElems * 8, nullptr, le, m, sk);
                (void)msm_sums_stage(c, w.scalars2 + off * (size_t)kBlobElems * 8, m, sk, off, false, evf);
            }
            LWK_HIP(hipEventRecord(c->ev_join[2], s2));
            LWK_HIP(hipStreamWaitEvent(st, c->ev_join[2], 0));
            // the second pass: only the blobs whose commitment bytes were not the canonical encoding
            LWK_HIP(hipStreamWaitEvent(st, c->ev_join[0], 0));
            launch_flag_differs48(canon, comm48, differs, n, st);
            launch_challenge(blobs, canon, z, le, n, st, comm48);
            quotient_stage(c, mode, w.scalars, z, w.scalars2, nullptr, le, n, st, differs);
            launch_direct_msm_only(evf ? c->lag.direct_bits : c->direct_bits, evf ? c->lag.direct_tab.win_dev : c->direct_tab.win_dev,
                                   evf ? c->lag.direct_row_bytes : c->direct_row_bytes, w.scalars2, w.sums, differs, n, st);
            launch_finalize_compress(w.sums, out48, n, st);
            return C_KZG_OK;
        }
        LWK_HIP(hipMemcpyAsync(w.zbytes, h.dig, n * 32, hipMemcpyHostToDevice, st));
        launch_z_from_bytes(w.zbytes, z, nullptr, le, n, st);  // digest -> Fr, reduced (utils.rs:148-154)
        LWK_HIP(hipStreamWaitEvent(st, c->ev_join[0], 0));
        launch_challenge(blobs, canon, z, le, n, st, comm48);   // only the lanes whose canonical commitment bytes differ from the caller's
        break;
    }
    case kProofGpuChains: {
        LWK_HIP(hipEventRecord(c->ev_fork, st));
        LWK_HIP(hipStreamWaitEvent(c->vstream, c->ev_fork, 0));
        launch_validate_commitments(comm48, canon, stt, le ? kStatusBadArgs : kStatusError, n, c->vstream, longcall ? w.val_pts_long : w.val_pts,
                                    longcall ? w.val_kind_long : w.val_kind, longcall ? w.val_verdict_long : w.val_verdict);
        LWK_HIP(hipEventRecord(c->ev_join[0], c->vstream));
        launch_challenge(blobs, comm48, z, le, n, st);
        LWK_HIP(hipStreamWaitEvent(st, c->ev_join[0], 0));
        launch_challenge(blobs, canon, z, le, n, st, comm48);
        break;
    }
    }
    // the ALU-bound phase: in turns with the settings' other context (engine.h: heavy_done), so that THIS call's hash
    // above ran beside the other call's MSM and the other call's next hash runs beside this one
    Ctx *pr = c->primary;
    // (up to half a chunk: there the MSM is about as long as the hash and the two pipelines would phase-lock; a longer
    // MSM covers the other call's hash by itself, and taking turns only adds bubbles -- 81k against 90k proofs/s at 1024)
    const bool heavy_serial = plan.heavy_serial;
    if (heavy_serial) {
        std::lock_guard<std::mutex> hk(pr->heavy_mu);
        LWK_HIP(hipStreamWaitEvent(st, pr->heavy_done, 0));
    }
    for (size_t off = 0; off < n; off += kMaxChunk) {
        size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        coefficients_stage(c, blobs + off * (size_t)kBlobBytes, m, mode, stt + off, st);
        quotient_stage(c, mode, w.scalars, z + off, w.scalars2, nullptr, le, m, st);
        G1Xyzz29 *sums = msm_sums_stage(c, w.scalars2, m, st, 0, false, quotient_to_msm_form(c, mode, m, st));
        if (heavy_serial && off + kMaxChunk >= n) {  // the last accumulation is in the queue: the other context's phase may follow it
            std::lock_guard<std::mutex> hk(pr->heavy_mu);
            LWK_HIP(hipEventRecord(pr->heavy_done, st));
        }
        launch_finalize_compress(sums, out48 + 48 * off, m, st);
    }
    return C_KZG_OK;
}

// Commitment AND blob proof of every blob in one pass (what a blob producer needs: lib.rs:253-283 followed by
// lib.rs:361-404 on its own output). Two things a pair of separate calls cannot do: the 2048 blocks of the Fiat-Shamir
// hash that do not depend on the commitment (all but the last 32 bytes of the blob) run on the side stream BESIDE the
// commitment MSMs and only the last two blocks wait for the commitments (k_challenge_finish) -- the 3.3 ms hash phase
// that stands in front of a proof call's MSM disappears -- and the blob is parsed (mode C: transformed) once. The
// commitments are the library's own output: nothing to validate. Results are those of the two calls, byte for byte.
C_KZG_RET commit_and_prove_batch_device(Ctx *c, uint8_t *comm_out48, uint8_t *proof_out48, const uint8_t *blobs, size_t n, int mode,
                                        hipStream_t st, int32_t *status) {
    C_KZG_RET rc = ctx_reserve(c, n);
    if (rc != C_KZG_OK) return rc;
    const bool longcall = n > kMaxChunk;
    if (longcall) {
        rc = ws_long_reserve(c, n);
        if (rc != C_KZG_OK) return rc;
    }
    Workspace &w = c->ws;
    const int le = mode == LWKZG_MODE_CKZG;
    int32_t *stt = status ? status : longcall ? w.status_long : w.status;
    Fr *z = longcall ? w.z_long : w.z;
    uint32_t *mid = (uint32_t *)(longcall ? w.canon_long : w.canon48);  // 32 of the 48 bytes per blob this call has no other use for
    LWK_HIP(hipMemsetAsync(stt, 0, n * 4, st));
    LWK_HIP(hipEventRecord(c->ev_fork, st));
    LWK_HIP(hipStreamWaitEvent(c->vstream, c->ev_fork, 0));
    // mid-size calls on a settings object whose other context is idle: the commitment-free part of the hashes on the host threads,
    // chunk by chunk beside the copy out (as blob_proof_batch_device does for whole hashes), instead of the 3.2 ms kernel beside the MSM
    const bool host_cold = n <= mid_proof_host_limit() && !peer_busy(c) && !host_assist_warm();
    if (host_cold) launch_host_wake(c);   // (the GPU kernel this once; the wake-up on a stream this call does not wait for)
    const bool host_mid = !host_cold && n <= mid_proof_host_limit() && !peer_busy(c) && sph_reserve(c, n);
    if (host_mid) {
        SmallProofHost &h = c->sph;
        hipStream_t sc = c->aux[0];
        LWK_HIP(hipStreamWaitEvent(sc, c->ev_fork, 0));
        {
            const C_KZG_RET rch = chunk_host_functions(c, blobs, n, sc, c->vstream, st, chunk_midstate_host_fn, "host_midstate_chunk");
            if (rch != C_KZG_OK) return rch;   // (st has been made to wait for both side streams)
        }
        LWK_HIP(hipMemcpyAsync(mid, h.dig, n * 32, hipMemcpyHostToDevice, c->vstream));
    } else {
        launch_challenge_midstate(blobs, mid, n, c->vstream);  // ALL blobs of the call: a latency chain, as long for 64 blobs as for 16k
    }
    LWK_HIP(hipEventRecord(c->ev_join[0], c->vstream));
    for (size_t off = 0; off < n; off += kMaxChunk) {
        const size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        const uint8_t *b = blobs + off * (size_t)kBlobBytes;
        if (coefficients_stage(c, b, m, mode, stt + off, st)) {  // evaluations: commitment and quotient both on the Lagrange form
            msm_stages(c, w.scalars, comm_out48 + 48 * off, m, st, 0, off == 0 && !host_mid, true);
        } else if (proof_on_lagrange(c, mode)) {  // coefficients for the quotient, but the only direct table is the Lagrange one: the commitment comes from the evaluations as they stand
            launch_copy_le_check(b, (uint32_t *)w.fr, nullptr, m, st);
            msm_stages(c, (const uint32_t *)w.fr, comm_out48 + 48 * off, m, st, 0, off == 0 && !host_mid, true);
        } else {
            msm_stages(c, w.scalars, comm_out48 + 48 * off, m, st, 0, off == 0 && !host_mid);  // the first one has the hash kernel beside it
        }
        if (off == 0) LWK_HIP(hipStreamWaitEvent(st, c->ev_join[0], 0));
        launch_challenge_finish(b, comm_out48 + 48 * off, mid + 8 * off, z + off, le, m, st);
        quotient_stage(c, mode, w.scalars, z + off, w.scalars2, nullptr, le, m, st);
        msm_stages(c, w.scalars2, proof_out48 + 48 * off, m, st, 0, false, quotient_to_msm_form(c, mode, m, st));
    }
    return C_KZG_OK;
}

C_KZG_RET point_proof_batch_device(Ctx *c, uint8_t *proof48, uint8_t *y32, const uint8_t *blobs, const uint8_t *z32,
                                   size_t n, int mode, hipStream_t st, int32_t *status, const G1Xyzz29 **sums_out) {
    C_KZG_RET rc = ctx_reserve(c, n);
    if (rc != C_KZG_OK) return rc;
    Workspace &w = c->ws;
    const int le = mode == LWKZG_MODE_CKZG;
    for (size_t off = 0; off < n; off += kMaxChunk) {
        size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        int32_t *stt = status ? status + off : w.status;
        LWK_HIP(hipMemsetAsync(stt, 0, m * 4, st));
        coefficients_stage(c, blobs + off * (size_t)kBlobBytes, m, mode, stt, st);
        launch_z_from_bytes(z32 + 32 * off, w.z, stt, le, m, st);
        quotient_stage(c, mode, w.scalars, w.z, w.scalars2, y32 + 32 * off, le, m, st);
        const bool lg = quotient_to_msm_form(c, mode, m, st);
        if (sums_out && n <= kMaxChunk) *sums_out = msm_sums_stage(c, w.scalars2, m, st, 0, false, lg);   // the caller finishes on the host
        else msm_stages(c, w.scalars2, proof48 + 48 * off, m, st, 0, false, lg);
    }
    return C_KZG_OK;
}

// ------------------------------------------------------------------------------------------------
// verify-side helpers: host buffers in and out, kernels in between

static C_KZG_RET first_status(Ctx *c, const int32_t *d_status, size_t n, hipStream_t st) {
    std::vector<int32_t> h(n);
    LWK_HIP(hipMemcpyAsync(h.data(), d_status, n * 4, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipStreamSynchronize(st));
    for (size_t i = 0; i < n; i++)
        if (h[i] != 0) {
            set_error("input %zu rejected (status %d)", i, h[i]);
            return (C_KZG_RET)h[i];
        }
    return C_KZG_OK;
}

void verify_buffers_free(VerifyBuffers &v) {
    dev_free(v.pts_c);
    dev_free(v.pts_p);
    dev_free(v.mult_c);
    dev_free(v.mult_p);
    dev_free(v.kind_c);
    dev_free(v.kind_p);
    dev_free(v.proof_in);
    dev_free(v.comm_in);
    dev_free(v.canon_dev);
    dev_free(v.status_all);
    dev_free(v.verdict_c);
    dev_free(v.verdict_p);
    dev_free(v.d_r);
    dev_free(v.d_rz);
    dev_free(v.d_aff);
    dev_free(v.d_part);
    dev_free(v.d_inf);
    dev_free(v.vm_base);
    dev_free(v.d_rec);
    if (v.h_rec) {
        (void)hipHostFree(v.h_rec);
        v.h_rec = nullptr;
    }
    v.rec_cap = 0;
    v.tab_p = v.tab_c = nullptr;
    v.vm_tmp = v.vm_partial = v.vm_bsum = nullptr;
    v.vm_pre = nullptr;
    v.sc_a = v.sc_b = nullptr;
    v.vm_pw = nullptr;
    if (v.h_pin) {
        (void)hipHostFree(v.h_pin);
        v.h_pin = nullptr;
    }
    if (v.vm_done) {
        (void)hipEventDestroy(v.vm_done);
        v.vm_done = nullptr;
    }
}

static void vs_free(Ctx *c) {
    verify_buffers_free(c->vs);
    c->vs_cap = 0;
}

// device scratch of one batch verification of up to `cap` blobs
static C_KZG_RET verify_buffers_alloc(VerifyBuffers &v, size_t cap) {
    const size_t nblk = lincomb3_blocks(cap);
    bool ok = hipMalloc((void **)&v.pts_c, cap * sizeof(G1Affine29)) == hipSuccess &&
              hipMalloc((void **)&v.pts_p, cap * sizeof(G1Affine29)) == hipSuccess &&
              hipMalloc((void **)&v.mult_c, 3 * cap * sizeof(G1Affine29)) == hipSuccess &&
              hipMalloc((void **)&v.mult_p, 3 * cap * sizeof(G1Affine29)) == hipSuccess &&
              hipMalloc((void **)&v.kind_c, cap * 4) == hipSuccess && hipMalloc((void **)&v.kind_p, cap * 4) == hipSuccess &&
              hipMalloc((void **)&v.proof_in, cap * 48) == hipSuccess && hipMalloc((void **)&v.d_r, cap * 32) == hipSuccess &&
              hipMalloc((void **)&v.comm_in, cap * 48) == hipSuccess && hipMalloc((void **)&v.canon_dev, 2 * cap * 48) == hipSuccess &&
              hipMalloc((void **)&v.status_all, cap * 4) == hipSuccess &&
              hipMalloc((void **)&v.verdict_c, cap * 4) == hipSuccess && hipMalloc((void **)&v.verdict_p, cap * 4) == hipSuccess &&
              hipMalloc((void **)&v.d_rz, cap * 32) == hipSuccess &&
              hipMalloc((void **)&v.d_part, (3 * nblk + 3) * sizeof(G1Xyzz29)) == hipSuccess &&
              hipMalloc((void **)&v.d_aff, 3 * 96) == hipSuccess && hipMalloc((void **)&v.d_inf, 3 * 4) == hipSuccess;
    // vmsm.hip's scratch: one allocation, 256-byte aligned pieces
    if (ok) {
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t b_tab = up((size_t)kVmsmRows * cap * sizeof(G1Affine29)), b_tmp = up((size_t)kVmsmSteps * 2 * cap * sizeof(G1Xyzz29)),
                     b_pre = up((size_t)kVmsmSteps * 2 * cap * sizeof(F29<2>)), b_sc = up(32 * cap),
                     b_part = up(3 * vmsm_max_slices(cap) * 256 * sizeof(G1Xyzz29)), b_bsum = up(3 * 256 * sizeof(G1Xyzz29)),
                     b_pw = up(33 * sizeof(Fr));
        ok = hipMalloc((void **)&v.vm_base, 2 * b_tab + b_tmp + b_pre + 2 * b_sc + b_part + b_bsum + b_pw) == hipSuccess &&
             hipHostMalloc((void **)&v.h_pin, kVmsmPinBytes, hipHostMallocDefault) == hipSuccess &&
             hipMalloc((void **)&v.d_rec, 160 * cap + 16) == hipSuccess &&
             hipHostMalloc((void **)&v.h_rec, 160 * cap + 16, hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&v.vm_done, hipEventDisableTiming) == hipSuccess;
        if (ok) {
            v.rec_cap = cap;
            uint8_t *p = v.vm_base;
            v.tab_p = (G1Affine29 *)p; p += b_tab;
            v.tab_c = (G1Affine29 *)p; p += b_tab;
            v.vm_tmp = (G1Xyzz29 *)p; p += b_tmp;
            v.vm_pre = (F29<2> *)p; p += b_pre;
            v.sc_a = (uint32_t *)p; p += b_sc;
            v.sc_b = (uint32_t *)p; p += b_sc;
            v.vm_partial = (G1Xyzz29 *)p; p += b_part;
            v.vm_bsum = (G1Xyzz29 *)p; p += b_bsum;
            v.vm_pw = (Fr *)p;
        }
    }
    if (!ok) {
        (void)hipGetLastError();
        verify_buffers_free(v);
        set_error("verify scratch for %zu blobs: out of device memory", cap);
        return C_KZG_MALLOC;
    }
    return C_KZG_OK;
}

// a verification that runs on the context's scratch (under verify_mu) sees it through its own VerifyBuffers
static void verify_buffers_lend(VerifyBuffers &vb, const VerifyBuffers &v) {
    vb.mult_c = v.mult_c; vb.mult_p = v.mult_p;
    vb.pts_c = v.pts_c; vb.pts_p = v.pts_p; vb.kind_c = v.kind_c; vb.kind_p = v.kind_p; vb.proof_in = v.proof_in;
    vb.d_r = v.d_r; vb.d_rz = v.d_rz; vb.d_aff = v.d_aff; vb.d_part = v.d_part; vb.d_inf = v.d_inf;
    vb.comm_in = v.comm_in; vb.canon_dev = v.canon_dev; vb.status_all = v.status_all; vb.verdict_c = v.verdict_c; vb.verdict_p = v.verdict_p;
    vb.vm_base = nullptr;   // (not this object's to free)
    vb.tab_p = v.tab_p; vb.tab_c = v.tab_c; vb.vm_tmp = v.vm_tmp; vb.vm_pre = v.vm_pre; vb.sc_a = v.sc_a; vb.sc_b = v.sc_b;
    vb.vm_partial = v.vm_partial; vb.vm_bsum = v.vm_bsum; vb.vm_pw = v.vm_pw; vb.h_pin = v.h_pin; vb.vm_done = v.vm_done;
    vb.d_rec = v.d_rec; vb.h_rec = v.h_rec; vb.rec_cap = v.rec_cap;
}

// the rows of both point sets for the linear combinations, on `st` (needs the decompressed points, not the subgroup verdicts):
// vmsm.hip's 32 byte-spaced rows per point, or (LWKZG_VERIFY_MSM=0) r05's three 32-bit-spaced multiples
static void launch_verify_rows(VerifyBuffers &vb, size_t n, hipStream_t st, bool apart) {
    if (knobs().verify_msm)
        launch_vmsm_multiples2(vb.pts_p, vb.kind_p, vb.tab_p, vb.pts_c, vb.kind_c, vb.tab_c, vb.vm_tmp, vb.vm_pre, n, st, apart);
    else
        launch_point_multiples2(vb.pts_p, vb.kind_p, vb.mult_p, vb.pts_c, vb.kind_c, vb.mult_c, n, st);
}

// grow-only verify scratch for n blobs; the caller holds verify_mu
static C_KZG_RET vs_reserve(Ctx *c, size_t n) {
    if (c->vs_cap >= n) return C_KZG_OK;
    LWK_HIP(hipDeviceSynchronize());  // the validation / multiples streams included
    vs_free(c);
    size_t cap = 64;
    while (cap < n) cap <<= 1;
    C_KZG_RET rc = verify_buffers_alloc(c->vs, cap);
    if (rc != C_KZG_OK) return rc;
    c->vs_cap = cap;
    return C_KZG_OK;
}

// device memory for the results of a long host-pointer batch: the context's own grow-only block (no hipMalloc / hipFree inside a call: both
// are synchronous and cost milliseconds). Caller holds c->mu; nullptr when the memory cannot be had.
static uint8_t *host_res_block(Ctx *c, size_t bytes) {
    if (c->host_res_cap >= bytes) return c->host_res;
    (void)hipDeviceSynchronize();
    dev_free(c->host_res);
    c->host_res = nullptr;
    c->host_res_cap = 0;
    size_t cap = (size_t)1 << 20;
    while (cap < bytes) cap <<= 1;
    if (hipMalloc((void **)&c->host_res, cap) != hipSuccess) {
        (void)hipGetLastError();
        c->host_res = nullptr;
        return nullptr;
    }
    c->host_res_cap = cap;
    return c->host_res;
}

// ---- the device-side double buffer of the long host-pointer batches (engine.h: DevStage). Caller holds c->mu. ---------------------------
// the stream the long host-pointer batches upload on: a high-priority one of the context's own (the runtime keeps a hardware queue per
// priority level, so these copies never queue behind a kernel of another stream -- verify_prepare_staged has the measurement); a side
// stream when it cannot be had or when LWKZG_STAGE_STREAMS (experiment) names one
static hipStream_t upload_stream(Ctx *c) {
    if (knobs().stage_streams[0] < 8) return c->aux[knobs().stage_streams[0]];
    if (!c->prio_copy) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&c->prio_copy, hipStreamNonBlocking, hi) != hipSuccess) {
            (void)hipGetLastError();
            c->prio_copy = nullptr;
        }
    }
    return c->prio_copy ? c->prio_copy : c->aux[3];
}
static bool dev_stage_ready(Ctx *c) {
    DevStage &r = c->stage;
    if (r.ready || r.failed || !knobs().host_stage) return r.ready && knobs().host_stage;
    bool ok = true;
    for (int k = 0; k < 2 && ok; k++)
        ok = hipMalloc((void **)&r.slot[k], kMaxChunk * (size_t)kBlobBytes) == hipSuccess &&
             hipEventCreateWithFlags(&r.copied[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&r.parsed[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {   // no memory for it (a 275 GB table): r05's slices, quietly
        (void)hipGetLastError();
        for (int k = 0; k < 2; k++) {
            dev_free(r.slot[k]);
            if (r.copied[k]) hipEventDestroy(r.copied[k]);
            if (r.parsed[k]) hipEventDestroy(r.parsed[k]);
            r.slot[k] = nullptr;
            r.copied[k] = r.parsed[k] = nullptr;
        }
        r.failed = true;
        return false;
    }
    r.ready = true;
    return true;
}
// slice lengths of a staged batch: a short first slice (the GPU is at work after a sixteenth of a 4096-blob upload), then whole chunks
static size_t stage_slice_len(size_t k, size_t remaining) {
    const size_t want = k == 0 ? kMaxChunk / 4 : kMaxChunk;
    return remaining < want ? remaining : want;
}
// upload `cnt` blobs into slot k mod 2 on the copy stream (behind the parse of the slot's previous occupant) and make `compute` wait for it
static C_KZG_RET stage_upload(Ctx *c, size_t k, const uint8_t *src, size_t cnt, hipStream_t compute, uint8_t **d_blobs) {
    DevStage &r = c->stage;
    const int s = (int)(k & 1);
    hipStream_t sc = upload_stream(c);   // (aux[0..2] are the sub-batch streams of the launch sets)
    if (k >= 2) LWK_HIP(hipStreamWaitEvent(sc, r.parsed[s], 0));
    LWK_HIP(hipMemcpyAsync(r.slot[s], src, cnt * (size_t)kBlobBytes, hipMemcpyHostToDevice, sc));
    LWK_HIP(hipEventRecord(r.copied[s], sc));
    LWK_HIP(hipStreamWaitEvent(compute, r.copied[s], 0));
    *d_blobs = r.slot[s];
    return C_KZG_OK;
}
// every kernel that reads slot k mod 2 has been enqueued on `compute`
static C_KZG_RET stage_parsed(Ctx *c, size_t k, hipStream_t compute) {
    LWK_HIP(hipEventRecord(c->stage.parsed[k & 1], compute));
    return C_KZG_OK;
}

// ---- long host-pointer verifications, r06 second form: the device-resident pipeline behind an upload --------------------------------------
// verify_prepare_long below hashes every blob on the host threads, slice by slice beside the upload -- and on a host whose container gets
// ~32 hardware threads that is the longest stage of its pipeline: 1.78 ms of SHA-256 per 512-blob slice against 1.2 ms of upload, 4096 blobs
// in 16 ms of which the upload is 9.6 (profiles/r06_experiments.md section 9). Here ALL blobs go into one device buffer (537 MB for 4096 of the 288 GB), in
// slices on a copy stream, and the hashing is SHARED: the head of the batch is hashed by the GPU's kernel slice by slice as it lands (a
// 3.2 ms latency chain per launch whatever its size, hidden behind the uploads still to come), the tail by the host threads from the
// caller's memory, starting at once (their share is what they hash in 0.8 of the upload time at their measured rate). y = p(z) is then
// read straight from the uploaded blobs (k_eval_quotient_from_blobs / its evaluation-form twin: no coefficient slots to recycle).
// Reference mode and c-kzg mode on the Lagrange form; other forms, no memory for the buffer, or LWKZG_HOST_STAGE=0 (experiment): the
// sliced form below. `taken` says which. Caller holds c->mu.
static uint8_t *vblobs_reserve(Ctx *c, size_t n) {
    if (c->vblobs_cap >= n) return c->vblobs;
    if (hipDeviceSynchronize() != hipSuccess) return nullptr;
    dev_free(c->vblobs);
    c->vblobs = nullptr;
    c->vblobs_cap = 0;
    size_t cap = 2 * kMaxChunk;
    while (cap < n) cap <<= 1;
    if (hipMalloc((void **)&c->vblobs, cap * (size_t)kBlobBytes) != hipSuccess) {
        (void)hipGetLastError();
        c->vblobs = nullptr;
        return nullptr;
    }
    c->vblobs_cap = cap;
    return c->vblobs;
}

static C_KZG_RET verify_prepare_staged(Ctx *c, const uint8_t *blobs, const uint8_t *comm48, const uint8_t *proofs48, size_t n, int mode,
                                       uint8_t *z32, uint8_t *y32, uint8_t *canon_c, uint8_t *canon_p, VerifyBuffers &vb, bool &taken) {
    taken = false;
    const int le = mode == LWKZG_MODE_CKZG;
    const bool evf = proof_in_evaluation_form(c, mode);
    if (!knobs().host_stage || !(mode == LWKZG_MODE_REFERENCE || evf) || n > ((size_t)1 << 17)) return C_KZG_OK;
    uint8_t *d_all = vblobs_reserve(c, n);
    if (!d_all) return C_KZG_OK;
    C_KZG_RET rc = ws_long_reserve(c, n);
    if (rc != C_KZG_OK) return rc;
    taken = true;
    const int bad = le ? kStatusBadArgs : kStatusError;
    // Four streams are at work at once here, and the runtime multiplexes a process's streams onto four hardware queues: a copy that shares
    // its queue with the validation kernels or with a 3.1 ms hash launch simply waits for them (uploads on aux[3]: 15.9 instead of 13.1 ms;
    // which side streams collide depends on what else the process has created). The uploads therefore get a HIGH-PRIORITY stream of their
    // own -- the runtime keeps a queue per priority level -- and every choice of hash stream then measures the same
    // (profiles/r06_experiments.md section 9). LWKZG_STAGE_STREAMS=c,h (experiment) puts them on side streams instead.
    hipStream_t st = c->stream, sv = c->vstream, sc = upload_stream(c), sh = c->aux[knobs().stage_streams[1]];
    Fr *z = c->ws.z_long;
    // who hashes what, and when the head's launches go out: plan.h: plan_staged_verification (pure; tests/test_plan_cpu.py pins its table)
    const StagedSplit split = plan_staged_verification(n, host_hash_rate());
    const size_t slice = split.slice, n_gpu = split.n_gpu, n_host = split.n_host, every = split.every;
    if (knobs().timing)
        fprintf(stderr, "[lambdaworks_kzg_amd] staged verification of %zu blobs: the GPU hashes the first %zu (a launch per %zu slices of %zu), the host threads the last %zu (they hashed %.1f GB/s lately)\n",
                n, n_gpu, every, slice, n_host, host_hash_rate() * 1e-9);
    std::vector<uint8_t> dig(32 * (n_host ? n_host : 1));
    SideTask hasher;   // joined by its destructor on every exit (digests assume canonical commitment bytes; the comparison below confirms or refutes that)
    if (n_host) hasher.start([&, n_gpu, n_host]() { challenge_digests_host(dig.data(), blobs + n_gpu * (size_t)kBlobBytes, comm48 + 48 * n_gpu, n_host); });
    // up-front validation of every commitment (main stream) and every proof (validation stream), the rows of the linear combinations behind them
    LWK_HIP(hipMemcpyAsync(vb.comm_in, comm48, n * 48, hipMemcpyHostToDevice, st));
    LWK_HIP(hipMemsetAsync(vb.status_all, 0, n * 4, st));
    LWK_HIP(hipEventRecord(c->ev_fork, st));
    LWK_HIP(hipStreamWaitEvent(sv, c->ev_fork, 0));
    LWK_HIP(hipStreamWaitEvent(sc, c->ev_fork, 0));
    LWK_HIP(hipStreamWaitEvent(sh, c->ev_fork, 0));
    LWK_HIP(hipMemcpyAsync(vb.proof_in, proofs48, n * 48, hipMemcpyHostToDevice, sv));
    launch_validate_commitments(vb.proof_in, vb.canon_dev + 48 * n, vb.status_all, bad, n, sv, vb.pts_p, vb.kind_p, vb.verdict_p);
    launch_validate_commitments(vb.comm_in, vb.canon_dev, vb.status_all, bad, n, st, vb.pts_c, vb.kind_c, vb.verdict_c);
    LWK_HIP(hipEventRecord(c->ev_join[kMaxSplit - 2], st));
    LWK_HIP(hipStreamWaitEvent(sv, c->ev_join[kMaxSplit - 2], 0));
    launch_verify_rows(vb, n, sv, false);
    LWK_HIP(hipEventRecord(c->ev_join[kMaxSplit - 1], sv));
    // the uploads (this thread is inside a blocking pageable copy most of the time) and, behind each slice of the head, its hash
    size_t hashed = 0, landed = 0;   // slices
    for (size_t off = 0; off < n; off += slice) {
        const size_t m = n - off < slice ? n - off : slice;
        LWK_HIP(hipMemcpyAsync(d_all + off * (size_t)kBlobBytes, blobs + off * (size_t)kBlobBytes, m * (size_t)kBlobBytes, hipMemcpyHostToDevice, sc));
        if (off < n_gpu) {
            landed++;
            if (split.launch_after(landed)) {
                const size_t lo = hashed * slice, cnt = (landed - hashed) * slice;
                LWK_HIP(hipEventRecord(c->ev_join[3], sc));
                LWK_HIP(hipStreamWaitEvent(sh, c->ev_join[3], 0));
                launch_challenge(d_all + lo * (size_t)kBlobBytes, vb.comm_in + 48 * lo, z + lo, le, cnt, sh);
                hashed = landed;
            }
        }
    }
    LWK_HIP(hipEventRecord(c->ev_join[3], sc));
    LWK_HIP(hipStreamWaitEvent(sh, c->ev_join[3], 0));          // every blob is on the device
    LWK_HIP(hipStreamWaitEvent(st, c->ev_join[kMaxSplit - 1], 0));   // the validation's canonical bytes and verdicts
    LWK_HIP(hipMemcpyAsync(canon_c, vb.canon_dev, n * 48, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipMemcpyAsync(canon_p, vb.canon_dev + 48 * n, n * 48, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipEventRecord(c->ev_join[kMaxSplit - 2], st));
    LWK_HIP(hipStreamSynchronize(st));
    LWK_HIP(hipStreamWaitEvent(sh, c->ev_join[kMaxSplit - 2], 0));
    hasher.join();
    // the head: redo the challenges of blobs whose commitment bytes were not canonical (exits at once otherwise)
    if (n_gpu) launch_challenge(d_all, vb.canon_dev, z, le, n_gpu, sh, vb.comm_in);
    // the tail: the host's digests, unless a commitment among them was not in its canonical encoding
    if (n_host) {
        if (memcmp(canon_c + 48 * n_gpu, comm48 + 48 * n_gpu, 48 * n_host) == 0) {
            LWK_HIP(hipMemcpyAsync(vb.d_rz + 32 * n_gpu, dig.data(), 32 * n_host, hipMemcpyHostToDevice, sh));
            launch_z_from_bytes(vb.d_rz + 32 * n_gpu, z + n_gpu, nullptr, le, n_host, sh);
        } else {
            launch_challenge(d_all + n_gpu * (size_t)kBlobBytes, vb.canon_dev + 48 * n_gpu, z + n_gpu, le, n_host, sh);
        }
    }
    if (evf) launch_eval_y_from_blobs_evalform(d_all, z, c->tw28_fwd + kBlobElems / 2, vb.d_r, vb.status_all, n, sh);
    else launch_eval_y_from_blobs_be(d_all, z, vb.d_r, n, sh);
    launch_fr_mont_to_bytes(z, vb.d_rz, le, n, sh);
    LWK_HIP(hipEventRecord(c->ev_join[3], sh));
    LWK_HIP(hipStreamWaitEvent(st, c->ev_join[3], 0));
    LWK_HIP(hipMemcpyAsync(z32, vb.d_rz, n * 32, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipMemcpyAsync(y32, vb.d_r, n * 32, hipMemcpyDeviceToHost, st));
    return first_status(c, vb.status_all, n, st);
}

// Batches longer than one chunk (1024 blobs). All 2n points are validated ONCE up front (two launches side by side; the kernel is a 2 ms
// latency chain whatever n is), and the blobs then go through in slices that alternate between the two halves of the
// workspace and two streams: while the GPU parses / evaluates one slice, this thread is already inside the (blocking,
// pageable) H2D copy of the next and the host threads hash it. A slot is finished (digests uploaded, y = p(z)
// evaluated, z and y copied back, statuses checked) right before it is reused, and at the end. Caller holds c->mu.
static C_KZG_RET verify_prepare_long(Ctx *c, const uint8_t *blobs, const uint8_t *comm48, const uint8_t *proofs48, size_t n,
                                     int mode, uint8_t *z32, uint8_t *y32, uint8_t *canon_c, uint8_t *canon_p,
                                     VerifyBuffers &vb) {
    const int le = mode == LWKZG_MODE_CKZG;
    const int bad = le ? kStatusBadArgs : kStatusError;
    hipStream_t st = c->stream, sv = c->vstream;
    const bool piped = n >= kMaxChunk;
    const size_t step = piped ? kMaxChunk / 2 : n;
    C_KZG_RET rcw = ctx_reserve(c, n < kMaxChunk ? n : kMaxChunk);
    if (rcw != C_KZG_OK) return rcw;
    Workspace &w = c->ws;

    // up-front validation of every commitment (main stream) and every proof (validation stream)
    LWK_HIP(hipMemcpyAsync(vb.comm_in, comm48, n * 48, hipMemcpyHostToDevice, st));
    LWK_HIP(hipMemsetAsync(vb.status_all, 0, n * 4, st));
    LWK_HIP(hipEventRecord(c->ev_fork, st));
    LWK_HIP(hipStreamWaitEvent(sv, c->ev_fork, 0));
    LWK_HIP(hipMemcpyAsync(vb.proof_in, proofs48, n * 48, hipMemcpyHostToDevice, sv));
    launch_validate_commitments(vb.proof_in, vb.canon_dev + 48 * n, vb.status_all, bad, n, sv, vb.pts_p, vb.kind_p, vb.verdict_p);
    launch_validate_commitments(vb.comm_in, vb.canon_dev, vb.status_all, bad, n, st, vb.pts_c, vb.kind_c, vb.verdict_c);
    LWK_HIP(hipEventRecord(c->ev_join[kMaxSplit - 2], st));
    LWK_HIP(hipStreamWaitEvent(sv, c->ev_join[kMaxSplit - 2], 0));
    launch_verify_rows(vb, n, sv, false);  // for the linear combinations; needs the points of both sets and no scalar
    LWK_HIP(hipEventRecord(c->ev_join[kMaxSplit - 1], sv));
    LWK_HIP(hipStreamWaitEvent(st, c->ev_join[kMaxSplit - 1], 0));
    // the canonical bytes come back the first time the host needs them: a device-to-host copy into pageable memory
    // blocks this thread until the stream has reached it, and the first slices should be on their way by then
    bool validated = false;
    auto fetch_canon = [&]() -> C_KZG_RET {
        if (validated) return C_KZG_OK;
        LWK_HIP(hipMemcpyAsync(canon_c, vb.canon_dev, n * 48, hipMemcpyDeviceToHost, st));
        LWK_HIP(hipMemcpyAsync(canon_p, vb.canon_dev + 48 * n, n * 48, hipMemcpyDeviceToHost, st));
        LWK_HIP(hipStreamSynchronize(st));
        validated = true;
        return C_KZG_OK;
    };

    struct Slot {
        bool used = false;
        size_t off = 0, m = 0, base = 0;
        hipStream_t sk = nullptr;
        const uint8_t *hb = nullptr, *hc = nullptr;
        std::vector<uint8_t> dig;
        SideTask hasher;  // joined by its destructor
    } slots[2];

    auto begin = [&](Slot &s, size_t off, size_t m, int idx) -> C_KZG_RET {
        s.used = true;
        s.off = off;
        s.m = m;
        s.base = piped ? (size_t)idx * step : 0;
        s.sk = c->aux[idx];
        s.hb = blobs + off * (size_t)kBlobBytes;
        s.hc = comm48 + 48 * off;
        s.dig.resize(32 * m);
        Slot *sp = &s;  // digests assume the caller's commitment bytes are canonical; finish() confirms or refutes that
        uint8_t *d_blobs = w.blobs + s.base * (size_t)kBlobBytes;
        s.hasher.start([sp]() { challenge_digests_host(sp->dig.data(), sp->hb, sp->hc, sp->m); });
        LWK_HIP(hipMemcpyAsync(d_blobs, s.hb, m * (size_t)kBlobBytes, hipMemcpyHostToDevice, s.sk));
        // (the parser's verdicts go beside the validation's, as in the device-resident form: both only ever write failure codes)
        coefficients_stage(c, d_blobs, m, mode, vb.status_all + off, s.sk, s.base);
        return C_KZG_OK;
    };

    auto finish = [&](Slot &s) -> C_KZG_RET {
        if (!s.used) return C_KZG_OK;
        s.used = false;
        const size_t base = s.base, off = s.off, m = s.m;
        hipStream_t sk = s.sk;
        s.hasher.join();
        {
            C_KZG_RET rcf = fetch_canon();
            if (rcf != C_KZG_OK) return rcf;
        }
        Fr *d_z = w.z + base;
        uint8_t *d_zb = w.zbytes + 32 * base, *d_yb = w.ybytes + 32 * base;
        if (memcmp(canon_c + 48 * off, s.hc, m * 48) == 0) {
            LWK_HIP(hipMemcpyAsync(d_zb, s.dig.data(), m * 32, hipMemcpyHostToDevice, sk));
            launch_z_from_bytes(d_zb, d_z, nullptr, le, m, sk);
        } else {  // a non-canonical (or invalid) encoding in this slice: hash the canonical bytes on the GPU
            launch_challenge(w.blobs + base * (size_t)kBlobBytes, vb.canon_dev + 48 * off, d_z, le, m, sk);
        }
        quotient_stage(c, mode, w.scalars + base * (size_t)kBlobElems * 8, d_z, nullptr /* y only */, d_yb, le, m,
                             sk);
        launch_fr_mont_to_bytes(d_z, d_zb, le, m, sk);
        // r06: z and y of ALL slices collect on the device (the linear combinations' scalar buffers, idle until then; k_vmsm_scalars reads z
        // from there) and come back in one copy each at the end -- r05 copied them to pageable memory slice by slice and fetched the
        // slice's verdicts, two blocking round trips per slice on the submitting thread
        LWK_HIP(hipMemcpyAsync(vb.d_rz + 32 * off, d_zb, m * 32, hipMemcpyDeviceToDevice, sk));
        LWK_HIP(hipMemcpyAsync(vb.d_r + 32 * off, d_yb, m * 32, hipMemcpyDeviceToDevice, sk));
        return C_KZG_OK;
    };

    // the slice streams start after the caller's earlier work on the main stream
    LWK_HIP(hipStreamWaitEvent(c->aux[0], c->ev_fork, 0));
    LWK_HIP(hipStreamWaitEvent(c->aux[1], c->ev_fork, 0));
    int k = 0;
    C_KZG_RET rc_all = C_KZG_OK;
    for (size_t off = 0; off < n && rc_all == C_KZG_OK; off += step, k++) {
        const size_t m = n - off < step ? n - off : step;
        Slot &s = slots[piped ? (k & 1) : 0];
        rc_all = finish(s);  // the slot's previous occupant, if any
        if (rc_all == C_KZG_OK) rc_all = begin(s, off, m, piped ? (k & 1) : 0);
    }
    for (int j = 0; j < 2; j++) {  // drain in submission order
        C_KZG_RET rc = finish(slots[piped ? ((k + j) & 1) : j]);
        if (rc_all == C_KZG_OK) rc_all = rc;
    }
    hipStreamSynchronize(c->aux[0]);
    hipStreamSynchronize(c->aux[1]);
    {
        C_KZG_RET rcf = fetch_canon();
        if (rcf != C_KZG_OK) return rcf;
    }
    if (rc_all != C_KZG_OK) return rc_all;
    LWK_HIP(hipMemcpyAsync(z32, vb.d_rz, n * 32, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipMemcpyAsync(y32, vb.d_r, n * 32, hipMemcpyDeviceToHost, st));
    return first_status(c, vb.status_all, n, st);  // the validation's verdicts and the parser's: the lowest rejected index of the batch
}

// Everything per blob of a batch verification, in one pass over the blobs: validate C_i and pi_i (keeping the
// decompressed points on the device for the linear combinations), z_i = challenge(blob_i, C_i), y_i = p_i(z_i).
// The Fiat-Shamir digests are computed by host threads while the GPU validates and parses (the blobs are host
// memory here); the GPU hash is the fallback for non-canonical commitment encodings.
C_KZG_RET verify_prepare_host(Ctx *c, const uint8_t *blobs, const uint8_t *comm48, const uint8_t *proofs48, size_t n,
                              int mode, uint8_t *z32, uint8_t *y32, uint8_t *canon_c, uint8_t *canon_p, VerifyBuffers &vb,
                              const uint8_t *trusted_canon_c) {
    // vb.owned: the caller's VerifyBuffers brings device scratch of its own (a shard of a sharded verification, which
    // outlives this call and may coexist with others on the same settings object); otherwise the context's scratch is
    // lent out under verify_mu, released when the caller's VerifyBuffers goes away
    if (!vb.owned) vb.hold = std::unique_lock<std::mutex>(c->verify_mu);
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    const int le = mode == LWKZG_MODE_CKZG;
    const int bad = le ? kStatusBadArgs : kStatusError;
    hipStream_t st = c->stream;
    WsUse wsu(c, st);
    // No exit of this function may leave a validation / multiples kernel running on the side streams against scratch
    // that the next verification (or vs_reserve) is about to reuse: an early error return drains them.
    struct SideDrain {
        Ctx *c;
        bool armed = true;
        ~SideDrain() {
            if (!armed) return;
            hipStreamSynchronize(c->vstream);
            hipStreamSynchronize(c->aux[0]);
            hipStreamSynchronize(c->aux[1]);
            for (int k = 2; k < kMaxSplit; k++) hipStreamSynchronize(c->aux[k]);   // (verify_prepare_staged: its copy and hash streams)
            if (c->prio_copy) hipStreamSynchronize(c->prio_copy);
        }
    } drain{c};
    if (vb.owned) {
        if (!vb.pts_c) {
            C_KZG_RET rcv = verify_buffers_alloc(vb, n < 64 ? 64 : n);
            if (rcv != C_KZG_OK) return rcv;
        }
    } else {
        C_KZG_RET rcv = vs_reserve(c, n);
        if (rcv != C_KZG_OK) return rcv;
        verify_buffers_lend(vb, c->vs);
    }
    if (n > kMaxChunk && proofs48 && !trusted_canon_c) {  // up to one chunk the single pass below is ~1 ms shorter
        bool taken = false;
        C_KZG_RET rcs = verify_prepare_staged(c, blobs, comm48, proofs48, n, mode, z32, y32, canon_c, canon_p, vb, taken);
        if (taken || rcs != C_KZG_OK) {
            if (rcs == C_KZG_OK) drain.armed = false;   // every side stream was joined into the main stream
            return rcs;
        }
        return verify_prepare_long(c, blobs, comm48, proofs48, n, mode, z32, y32, canon_c, canon_p, vb);
    }
    std::vector<uint8_t> dig(32 * (n < kMaxChunk ? n : kMaxChunk));
    for (size_t off = 0; off < n; off += kMaxChunk) {
        size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        C_KZG_RET rc = ctx_reserve(c, m);
        if (rc != C_KZG_OK) return rc;
        Workspace &w = c->ws;
        const uint8_t *hb = blobs + off * (size_t)kBlobBytes, *hc = comm48 + 48 * off;
        // the Fiat-Shamir digests only need host memory: host threads start on them now, beside the pageable H2D copy
        // (which blocks this thread for milliseconds) and the GPU's validation / parsing. They assume the caller's
        // commitment bytes are canonical; the comparison below confirms or refutes that.
        const uint8_t *hash_comm = trusted_canon_c ? trusted_canon_c + 48 * off : hc;
        const bool hash_beside = m > 64;  // a few blobs: hashing takes microseconds, a thread and its contention do not pay
        SideTask hasher;  // joined by its destructor on every exit
        if (hash_beside) hasher.start([&, hash_comm]() { challenge_digests_host(dig.data(), hb, hash_comm, m); });
        LWK_HIP(hipMemcpyAsync(w.comm48, hc, m * 48, hipMemcpyHostToDevice, st));
        LWK_HIP(hipMemsetAsync(w.status, 0, m * 4, st));
        // up to 64 blobs: both point sets are validated on the host threads (0.2 ms per point per thread against a 2 ms
        // latency-shaped kernel) and the decompressed points uploaded in the form the kernel would have left
        const bool host_validate = !trusted_canon_c && n <= host_small_batch_limit();
        if (proofs48 && !host_validate) {
            // Both point sets are validated on streams of their own, started before the blobs go up (the copy blocks
            // this thread for milliseconds): decompression + subgroup test, then the multiples the linear combinations
            // will want, are a ~3 ms latency chain per set that nothing on the main stream should queue behind. Both
            // validations only ever write the same failure code into status.
            // The validation is split in two launches here (square root; subgroup test + canonical bytes), and the
            // multiples run on a fourth stream beside the second one. (off == 0: longer batches take the path above.)
            hipStream_t sa = c->vstream, sc = c->aux[0], sm = c->aux[1];
            LWK_HIP(hipEventRecord(c->ev_fork, st));
            LWK_HIP(hipStreamWaitEvent(sa, c->ev_fork, 0));
            LWK_HIP(hipMemcpyAsync(vb.proof_in, proofs48, m * 48, hipMemcpyHostToDevice, sa));
            launch_decompress_points(vb.proof_in, vb.pts_p, vb.kind_p, m, sa);
            LWK_HIP(hipEventRecord(c->ev_join[4], sa));
            launch_subgroup_canon(vb.pts_p, vb.kind_p, w.out48, w.status, bad, m, sa, vb.verdict_p);
            LWK_HIP(hipEventRecord(c->ev_join[0], sa));
            LWK_HIP(hipStreamWaitEvent(sc, c->ev_fork, 0));
            launch_decompress_points(w.comm48, vb.pts_c, vb.kind_c, m, sc);
            LWK_HIP(hipEventRecord(c->ev_join[5], sc));
            launch_subgroup_canon(vb.pts_c, vb.kind_c, w.canon48, w.status, bad, m, sc, vb.verdict_c);
            LWK_HIP(hipEventRecord(c->ev_join[1], sc));
            LWK_HIP(hipStreamWaitEvent(sm, c->ev_join[4], 0));
            LWK_HIP(hipStreamWaitEvent(sm, c->ev_join[5], 0));
            launch_verify_rows(vb, m, sm, false);
            LWK_HIP(hipEventRecord(c->ev_join[2], sm));
        }
        LWK_HIP(hipMemcpyAsync(w.blobs, hb, m * (size_t)kBlobBytes, hipMemcpyHostToDevice, st));
        coefficients_stage(c, w.blobs, m, mode, w.status, st);
        if (trusted_canon_c) {
            // the caller decompressed (and so validated) the commitments itself and hands over their canonical bytes:
            // no 2 ms validation kernel on the single-blob path
            memcpy(canon_c + 48 * off, trusted_canon_c + 48 * off, m * 48);
            hc = trusted_canon_c + 48 * off;
        } else if (!host_validate && !proofs48) {
            launch_validate_commitments(w.comm48, w.canon48, w.status, bad, m, st, vb.pts_c + off, vb.kind_c + off, vb.verdict_c + off);
        }
        std::vector<int32_t> h_code(m, bad), h_kind;
        std::vector<G1Affine29> h_aff;
        if (host_validate) {
            const size_t np = proofs48 ? 2 * m : m;
            std::vector<int> vrc(np);
            h_aff.resize(np);
            h_kind.resize(np);
            host_validate_commitments(hc, canon_c + 48 * off, vrc.data(), m, h_aff.data());
            if (proofs48) host_validate_commitments(proofs48 + 48 * off, canon_p + 48 * off, vrc.data() + m, m, h_aff.data() + m);
            for (size_t i = 0; i < np; i++) {
                h_kind[i] = vrc[i];
                if (vrc[i] == 2) LWK_HIP(hipMemcpyAsync(w.status + (i % m), &h_code[i % m], 4, hipMemcpyHostToDevice, st));
            }
            if (proofs48) {  // the linear combinations of so few points run on the host threads as well
                vb.h_aff = std::move(h_aff);
                vb.h_kind = std::move(h_kind);
            }
        }
        if (proofs48 && !host_validate) {
            LWK_HIP(hipStreamWaitEvent(st, c->ev_join[0], 0));
            LWK_HIP(hipStreamWaitEvent(st, c->ev_join[1], 0));
        }
        if (hash_beside) hasher.join();
        else challenge_digests_host(dig.data(), hb, hash_comm, m);
        // device-to-host copies into pageable memory block this thread until the stream has reached them, so the
        // canonical bytes are fetched only after every launch above has been submitted
        if (!trusted_canon_c && !host_validate)
            LWK_HIP(hipMemcpyAsync(canon_c + 48 * off, w.canon48, m * 48, hipMemcpyDeviceToHost, st));
        if (proofs48 && !host_validate) LWK_HIP(hipMemcpyAsync(canon_p + 48 * off, w.out48, m * 48, hipMemcpyDeviceToHost, st));
        LWK_HIP(hipStreamSynchronize(st));
        if (memcmp(canon_c + 48 * off, hc, m * 48) == 0) {
            LWK_HIP(hipMemcpyAsync(w.zbytes, dig.data(), m * 32, hipMemcpyHostToDevice, st));
            launch_z_from_bytes(w.zbytes, w.z, nullptr, le, m, st);
        } else {
            if (host_validate) LWK_HIP(hipMemcpyAsync(w.canon48, canon_c + 48 * off, m * 48, hipMemcpyHostToDevice, st));
            launch_challenge(w.blobs, w.canon48, w.z, le, m, st);
        }
        quotient_stage(c, mode, w.scalars, w.z, nullptr /* a verification wants y = p(z) only */, w.ybytes, le, m, st);
        launch_fr_mont_to_bytes(w.z, w.zbytes, le, m, st);
        if (proofs48 && !host_validate)  // k_vmsm_scalars reads the z bytes where the device-resident form leaves them
            LWK_HIP(hipMemcpyAsync(vb.d_rz + 32 * off, w.zbytes, m * 32, hipMemcpyDeviceToDevice, st));
        LWK_HIP(hipMemcpyAsync(z32 + 32 * off, w.zbytes, m * 32, hipMemcpyDeviceToHost, st));
        LWK_HIP(hipMemcpyAsync(y32 + 32 * off, w.ybytes, m * 32, hipMemcpyDeviceToHost, st));
        rc = first_status(c, w.status, m, st);
        if (rc != C_KZG_OK) return rc;
        if (proofs48 && !host_validate)  // the linear combinations (main stream, later) read the multiples
            LWK_HIP(hipStreamWaitEvent(st, c->ev_join[2], 0));
    }
    drain.armed = false;  // everything on the side streams has been joined into the main stream
    return C_KZG_OK;
}

// The same per-blob pass for a batch that is ALREADY on the device (lwkzg_verify_blob_kzg_proof_batch_device,
// lwkzg_verify_shard_begin_device; /root/reference/src/lib.rs:525-614, 639-692): nothing crosses PCIe but the 160-byte records. Both point sets are
// validated on side streams (decompression, subgroup test, canonical bytes, the multiples the linear combinations want) while the main
// stream hashes ALL blobs in one launch over the caller's commitment bytes; where the validation's canonical bytes differ from the
// caller's (a valid point in a non-canonical encoding) that blob's challenge is taken again over the canonical ones -- a launch that
// exits at once otherwise. Then chunk by chunk: parse, y = p(z). `caller`: the stream the inputs were produced on (may be null).
C_KZG_RET verify_prepare_device(Ctx *c, const uint8_t *d_blobs, const uint8_t *d_comm, const uint8_t *d_proofs, size_t n, int mode,
                                uint8_t *z32, uint8_t *y32, uint8_t *canon_c, uint8_t *canon_p, VerifyBuffers &vb, hipStream_t caller,
                                uint8_t *records_out) {
    if (!vb.owned) vb.hold = std::unique_lock<std::mutex>(c->verify_mu);
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    const int le = mode == LWKZG_MODE_CKZG;
    const int bad = le ? kStatusBadArgs : kStatusError;
    hipStream_t st = c->stream, sv = c->vstream, sc = c->aux[0];
    if (caller && caller != st) {  // the inputs are whatever the caller's stream has produced by now
        LWK_HIP(hipEventRecord(c->ev_join[3], caller));
        LWK_HIP(hipStreamWaitEvent(st, c->ev_join[3], 0));
    }
    WsUse wsu(c, st);
    struct SideDrain {
        Ctx *c;
        bool armed = true;
        ~SideDrain() {
            if (!armed) return;
            hipStreamSynchronize(c->vstream);
            hipStreamSynchronize(c->aux[0]);
        }
    } drain{c};
    if (vb.owned) {
        if (!vb.pts_c) {
            C_KZG_RET rcv = verify_buffers_alloc(vb, n < 64 ? 64 : n);
            if (rcv != C_KZG_OK) return rcv;
        }
    } else {
        C_KZG_RET rcv = vs_reserve(c, n);
        if (rcv != C_KZG_OK) return rcv;
        verify_buffers_lend(vb, c->vs);
    }
    C_KZG_RET rc = ctx_reserve(c, n < kMaxChunk ? n : kMaxChunk);
    if (rc != C_KZG_OK) return rc;
    if (n > kMaxChunk && (rc = ws_long_reserve(c, n)) != C_KZG_OK) return rc;
    Workspace &w = c->ws;
    Fr *z = n > kMaxChunk ? w.z_long : w.z;
    LWK_HIP(hipMemsetAsync(vb.status_all, 0, n * 4, st));
    LWK_HIP(hipEventRecord(c->ev_fork, st));
    LWK_HIP(hipStreamWaitEvent(sv, c->ev_fork, 0));
    LWK_HIP(hipStreamWaitEvent(sc, c->ev_fork, 0));
    // Beside the hash (64 blobs per workgroup, one workgroup per compute unit: 3.2 ms whatever n is) the validation and the rows of the
    // linear combinations are latency chains of a few hundred waves, and r05 lost 2 ms to where the dispatcher put them: on the hash's
    // own compute units, four of its waves per SIMD-quad at a raised priority (k_decompress_points 0.43 -> 1.3 ms, k_subgroup_coop_asm
    // 0.67 -> 1.8 ms; profiles/r06_verify_b4096_device_timeline_r05_code.txt). `apart`: every such launch carries an LDS footprint that
    // cannot share a compute unit with a hash workgroup (or with each other), as long as the hash leaves half the chip free.
    const bool apart = n <= kVerifyApartMax;
    const bool fused = knobs().verify_fused || knobs().verify_msm;
    // Up to 8192 blobs (the hash on at most half the compute units) the hash is submitted FIRST and takes its compute units; the padded
    // validation workgroups then fill the others, a compute unit each, and queue among themselves where those run out -- submitted first,
    // they would take the whole chip and the hash would wait for them (8192 blobs: the hash 7.6 ms behind 256 exclusive decompression
    // workgroups; profiles/r06_experiments.md section 3). LWKZG_VERIFY_ORDER=1 (experiment) is the other order.
    const bool hash_first = fused ? (apart != (knobs().verify_order != 0)) : knobs().verify_order != 0;
    if (hash_first) launch_challenge(d_blobs, d_comm, z, le, n, st);
    if (fused) {   // r06: ONE launch per kernel over both point sets; the rows start as soon as the points are decompressed
        launch_decompress_points2(d_proofs, vb.pts_p, vb.kind_p, d_comm, vb.pts_c, vb.kind_c, n, sv, apart);
        LWK_HIP(hipEventRecord(c->ev_join[2], sv));
        launch_subgroup_canon2(vb.pts_p, vb.kind_p, vb.canon_dev + 48 * n, vb.verdict_p, vb.pts_c, vb.kind_c, vb.canon_dev, vb.verdict_c,
                               vb.status_all, bad, n, sv, apart);
        LWK_HIP(hipEventRecord(c->ev_join[0], sv));
        LWK_HIP(hipStreamWaitEvent(sc, c->ev_join[2], 0));
        launch_verify_rows(vb, n, sc, apart);
        LWK_HIP(hipEventRecord(c->ev_join[1], sc));
    } else {       // r05's arrangement (LWKZG_VERIFY_FUSED=0 with LWKZG_VERIFY_MSM=0): a side stream per point set
        launch_validate_commitments(d_proofs, vb.canon_dev + 48 * n, vb.status_all, bad, n, sv, vb.pts_p, vb.kind_p, vb.verdict_p);
        launch_point_multiples(vb.pts_p, vb.kind_p, vb.mult_p, n, sv);
        LWK_HIP(hipEventRecord(c->ev_join[0], sv));
        launch_validate_commitments(d_comm, vb.canon_dev, vb.status_all, bad, n, sc, vb.pts_c, vb.kind_c, vb.verdict_c);
        launch_point_multiples(vb.pts_c, vb.kind_c, vb.mult_c, n, sc);
        LWK_HIP(hipEventRecord(c->ev_join[1], sc));
    }
    if (!apart && fused) {
        // More than half the chip's compute units would hold a hash workgroup: the footprints cannot keep anything apart any more, and a
        // hash workgroup is as slow as the slowest of its four barrier-coupled waves -- 16384 blobs: the hash 7.8 ms with the validation's
        // waves among its own, 3.2 ms alone (profiles/r06_experiments.md section 3). The validation and the rows (~1.7 ms on the whole
        // chip) therefore go first and the hash follows them.
        LWK_HIP(hipStreamWaitEvent(st, c->ev_join[0], 0));
        LWK_HIP(hipStreamWaitEvent(st, c->ev_join[1], 0));
    }
    if (!hash_first) launch_challenge(d_blobs, d_comm, z, le, n, st);
    LWK_HIP(hipStreamWaitEvent(st, c->ev_join[0], 0));
    LWK_HIP(hipStreamWaitEvent(st, c->ev_join[1], 0));
    drain.armed = false;  // both side streams are joined into the main stream from here on
    launch_challenge(d_blobs, vb.canon_dev, z, le, n, st, d_comm);  // only the blobs whose commitment bytes were not canonical
    // chunk by chunk with no host round trip in between: y and z bytes of ALL blobs collect in the linear combinations' scalar
    // buffers (idle until lincomb3), the parser's verdicts beside the validation's
    if (mode == LWKZG_MODE_REFERENCE) {   // the blobs are already on the device and a reference-mode parse cannot fail: one launch reads them as they are
        launch_eval_y_from_blobs_be(d_blobs, z, vb.d_r, n, st);
        launch_fr_mont_to_bytes(z, vb.d_rz, le, n, st);
    } else if (proof_in_evaluation_form(c, mode)) {   // c-kzg on the Lagrange form: the blob's elements ARE the evaluations; range check in the same launch
        launch_eval_y_from_blobs_evalform(d_blobs, z, c->tw28_fwd + kBlobElems / 2, vb.d_r, vb.status_all, n, st);
        launch_fr_mont_to_bytes(z, vb.d_rz, le, n, st);
    } else {
        for (size_t off = 0; off < n; off += kMaxChunk) {
            const size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
            coefficients_stage(c, d_blobs + off * (size_t)kBlobBytes, m, mode, vb.status_all + off, st);
            quotient_stage(c, mode, w.scalars, z + off, nullptr /* y only */, vb.d_r + 32 * off, le, m, st);
            launch_fr_mont_to_bytes(z + off, vb.d_rz + 32 * off, le, m, st);
        }
    }
    if (records_out && vb.d_rec && vb.h_rec && vb.rec_cap >= n) {
        // r06: the transcript C | z | y | pi per blob assembled by a kernel and ONE copy into pinned memory, the lowest rejected index in its
        // last word -- where r05 made four copies into pageable vectors, a fifth for the status words, and the host interleaved
        uint32_t *d_flag = (uint32_t *)(vb.d_rec + 160 * n);
        LWK_HIP(hipMemsetAsync(d_flag, 0xff, 4, st));
        launch_verify_records(vb.canon_dev, vb.d_rz, vb.d_r, vb.canon_dev + 48 * n, vb.status_all, vb.d_rec, d_flag, n, st);
        LWK_HIP(hipMemcpyAsync(vb.h_rec, vb.d_rec, 160 * n + 4, hipMemcpyDeviceToHost, st));
        LWK_HIP(hipStreamSynchronize(st));
        uint32_t first_bad;
        memcpy(&first_bad, vb.h_rec + 160 * n, 4);
        if (first_bad != 0xffffffffu) {
            int32_t code = 0;
            LWK_HIP(hipMemcpy(&code, vb.status_all + first_bad, 4, hipMemcpyDeviceToHost));
            set_error("input %zu rejected (status %d)", (size_t)first_bad, code);
            return (C_KZG_RET)code;
        }
        memcpy(records_out, vb.h_rec, 160 * n);
        return C_KZG_OK;
    }
    LWK_HIP(hipMemcpyAsync(canon_c, vb.canon_dev, n * 48, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipMemcpyAsync(canon_p, vb.canon_dev + 48 * n, n * 48, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipMemcpyAsync(z32, vb.d_rz, n * 32, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipMemcpyAsync(y32, vb.d_r, n * 32, hipMemcpyDeviceToHost, st));
    return first_status(c, vb.status_all, n, st);  // the validation's verdicts and the parser's
}

// sums[0] = sum r_i pi_i, sums[1] = sum r_i z_i pi_i, sums[2] = sum r_i C_i on the points verify_prepare_host kept
C_KZG_RET lincomb3_device_host(Ctx *c, VerifyBuffers &vb, const uint8_t *sc_r, const uint8_t *sc_rz, size_t n,
                               uint8_t sums[3][96], int infs[3]) {
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t nblk = lincomb3_blocks(n);
    if (!(vb.hold.owns_lock() || vb.owned) || !vb.d_r) {
        set_error("lincomb3_device_host: called without a prepared verification");
        return C_KZG_ERROR;
    }
    uint8_t *d_r = vb.d_r, *d_rz = vb.d_rz, *d_aff = vb.d_aff;
    G1Xyzz29 *d_part = vb.d_part;
    int32_t *d_inf = vb.d_inf;
    uint8_t h_aff[3 * 96];
    int32_t h_inf[3];
    bool ok = true;
    if (ok) ok = hipMemcpyAsync(d_r, sc_r, 32 * n, hipMemcpyHostToDevice, st) == hipSuccess &&
                 hipMemcpyAsync(d_rz, sc_rz, 32 * n, hipMemcpyHostToDevice, st) == hipSuccess;
    if (ok) {
        launch_lincomb3(vb.pts_p, vb.kind_p, vb.mult_p, vb.pts_c, vb.kind_c, vb.mult_c, d_r, d_rz, d_part, n, st);
        G1Xyzz29 *totals = d_part + 3 * nblk;
        for (int k = 0; k < 3; k++) launch_sum_points(d_part + k * nblk, nblk, totals + k, 0, st);
        launch_xyzz29_to_affine_be(totals, d_aff, d_inf, 3, st);
        ok = hipMemcpyAsync(h_aff, d_aff, sizeof h_aff, hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipMemcpyAsync(h_inf, d_inf, sizeof h_inf, hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    }
    if (!ok) {
        set_error("lincomb3_device_host: device work failed: %s", hipGetErrorString(hipGetLastError()));
        return C_KZG_ERROR;
    }
    for (int k = 0; k < 3; k++) {
        memcpy(sums[k], h_aff + 96 * k, 96);
        infs[k] = h_inf[k];
    }
    return C_KZG_OK;
}

bool vmsm_ready(const VerifyBuffers &vb) { return knobs().verify_msm && vb.tab_p && vb.h_pin && vb.vm_done; }

// The three sums from r alone (vmsm.hip): everything is enqueued on the context's stream and the results travel to the pinned block by
// themselves; the caller (verify.hip: shard_partial) computes its host share meanwhile and collects with vmsm_finish.
C_KZG_RET vmsm_begin(Ctx *c, VerifyBuffers &vb, const Fr *pw33, int le, size_t n) {
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    if (!(vb.hold.owns_lock() || vb.owned) || !vmsm_ready(vb) || !vb.d_rz) {
        set_error("vmsm_begin: called without a prepared verification");
        return C_KZG_ERROR;
    }
    memcpy(vb.h_pin, pw33, 33 * sizeof(Fr));
    LWK_HIP(hipMemcpyAsync(vb.vm_pw, vb.h_pin, 33 * sizeof(Fr), hipMemcpyHostToDevice, st));
    launch_vmsm_scalars(vb.d_rz, le, vb.vm_pw, vb.sc_a, vb.sc_b, n, st);
    launch_vmsm_accumulate(vb.sc_a, vb.sc_b, vb.tab_p, vb.kind_p, vb.tab_c, vb.kind_c, vb.vm_partial, n, st);
    launch_vmsm_reduce(vb.vm_partial, vb.vm_bsum, vb.d_aff, vb.d_inf, n, st);
    LWK_HIP(hipMemcpyAsync(vb.h_pin + 33 * sizeof(Fr), vb.d_aff, 3 * 96, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipMemcpyAsync(vb.h_pin + 33 * sizeof(Fr) + 3 * 96, vb.d_inf, 3 * 4, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipEventRecord(vb.vm_done, st));
    return C_KZG_OK;
}

C_KZG_RET vmsm_finish(Ctx *c, VerifyBuffers &vb, uint8_t sums[3][96], int infs[3]) {
    LWK_HIP(hipSetDevice(c->device));
    if (hipEventSynchronize(vb.vm_done) != hipSuccess) {
        set_error("vmsm_finish: device work failed: %s", hipGetErrorString(hipGetLastError()));
        return C_KZG_ERROR;
    }
    const uint8_t *res = vb.h_pin + 33 * sizeof(Fr);
    for (int k = 0; k < 3; k++) {
        memcpy(sums[k], res + 96 * k, 96);
        int32_t f;
        memcpy(&f, res + 3 * 96 + 4 * k, 4);
        infs[k] = f;
    }
    return C_KZG_OK;
}

}  // namespace lwk

// =================================================================================================
// C ABI
// =================================================================================================

using namespace lwk;

// Process-wide semantics switch. Every entry point reads it ONCE, at its start; lwkzg_set_mode must not race calls
// whose result should be in a particular mode (documented in the header).
static std::atomic<int> g_mode{-1};

static int mode_now() {
    int m = g_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        m = knobs().mode ? LWKZG_MODE_CKZG : LWKZG_MODE_REFERENCE;   // LWKZG_MODE
        int expect = -1;
        if (!g_mode.compare_exchange_strong(expect, m)) m = expect;
    }
    return m;
}

namespace lwk {
int mode_of(const KZGSettings *s) {
    if (s) {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        Ctx *c = nullptr;
        if (s->fs && g_live_fs.count((const void *)s->fs)) c = (Ctx *)s->fs;
        else if (s->g1_values) {
            auto it = g_registry.find(s->g1_values);
            if (it != g_registry.end()) c = it->second.ctx;
        }
        if (c) {
            const int m = c->mode_override.load(std::memory_order_relaxed);
            if (m >= 0) return m;
        }
    }
    return mode_now();
}
}  // namespace lwk

extern "C" {

int lwkzg_set_mode(int mode) {
    int prev = mode_now();
    if (mode != LWKZG_MODE_REFERENCE && mode != LWKZG_MODE_CKZG) return -1;
    g_mode.store(mode);
    return prev;
}
int lwkzg_get_mode(void) { return mode_now(); }

// Per-settings semantics: a settings object that was given a mode of its own answers in it whatever the process-wide
// default says (-1 gives it back to the default); every entry point resolves its mode ONCE, when it is entered.
// An explicit mode for a settings object also brings its tables to that mode's form (settings_follow_mode below): the first switch
// to c-kzg mode derives the Lagrange form (about 50 ms) and builds a Lagrange direct table beside the monomial one if it fits (the
// default engine: 0.2-0.3 s, 41 GB more); when the two do not fit side by side (15 / 16 bits) the ONE table is rebuilt in the new
// mode's form -- the cost of a lwkzg_enable_direct_table call of that width (0.9 s of kernels at 16 bits plus whatever hipMalloc
// waits for). Nothing happens when the tables already suit the mode. The process-wide default (lwkzg_set_mode) never moves a table.
static void settings_follow_mode(Ctx *c, int mode);
static void ensure_lagrange(Ctx *c, int mode);
int lwkzg_settings_set_mode(const KZGSettings *s, int mode) {
    if (mode != LWKZG_MODE_REFERENCE && mode != LWKZG_MODE_CKZG && mode != -1) return -1;
    Ctx *c = ctx_of(s);  // hand-built settings get their context here
    if (!c) return -1;
    const int prev = lwk::mode_of(s);
    c->mode_override.store(mode, std::memory_order_relaxed);
    const int now = lwk::mode_of(s);
    if (now != prev && gpu_available()) settings_follow_mode(c, now);
    return prev;
}
int lwkzg_settings_get_mode(const KZGSettings *s) { return lwk::mode_of(s); }

int lwkzg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int lwkzg_set_device(int ordinal) {
    int n = lwkzg_device_count();
    if (ordinal < 0 || ordinal >= n) {
        set_error("lwkzg_set_device(%d): %d device(s) visible", ordinal, n);
        return -1;
    }
    g_default_device.store(ordinal);
    return 0;
}
const char *lwkzg_version(void) { return "lambdaworks_kzg_amd 0.2 (gfx950; direct-table MSM, 10..16-bit windows; bucket fallback c=13, 20 windows)"; }
const char *lwkzg_last_error(void) { return get_error(); }
int lwkzg_msm_window_bits(void) { return kWindowBits; }
int lwkzg_msm_num_windows(void) { return kNumWindows; }

void lwkzg_profile_enable(int on) {
    if (!on) prof_drain();
    g_prof_on.store(on != 0);
}
void lwkzg_profile_reset(void) {
    prof_drain();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_agg.clear();
}
size_t lwkzg_profile_report(char *buf, size_t cap) {
    prof_drain();
    std::string s = "{";
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        bool first = true;
        for (auto &kv : g_prof_agg) {
            char line[256];
            snprintf(line, sizeof line, "%s\"%s\": {\"launches\": %llu, \"total_ms\": %.6f}", first ? "" : ", ",
                     kv.first.c_str(), (unsigned long long)kv.second.launches, kv.second.total_ms);
            s += line;
            first = false;
        }
    }
    s += "}";
    if (buf && cap) {
        size_t k = s.size() < cap - 1 ? s.size() : cap - 1;
        memcpy(buf, s.data(), k);
        buf[k] = 0;
    }
    return s.size() + 1;
}

// first use of the HIP runtime by this process (device context, this library's code object): what a fresh process pays
// once, whichever call comes first. Returns 0, or -1 without a GPU.
__global__ void k_runtime_init(int *p) {
    if (p) *p = 1;
}

extern "C" int lwkzg_last_proof_schedule(void) { return g_last_proof_schedule.load(std::memory_order_relaxed); }

extern "C" int lwkzg_runtime_init(void) {
    if (!gpu_available()) return -1;
    if (hipSetDevice(g_default_device.load()) != hipSuccess || hipFree(nullptr) != hipSuccess) return -1;
    // the code object is loaded by asking about one of its kernels -- NOT by launching one: a launch here would have to go to
    // the NULL stream, and a process that has used the NULL stream once keeps a hardware queue for it, after which the
    // engine's sub-batch streams no longer run side by side (measured: the bucket engine's two overlapped sub-batches fell
    // from 61.8k to 52.0k ops/s behind a single one-thread launch on stream 0)
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, (const void *)k_runtime_init) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return 0;
}

// What shader clock does this box hold under a dense multiply-add stream? One wave per SIMD runs ~0.4 ms of dependent v_mad_u64_u32
// and reads the shader clock (clock64) and the 100 MHz wall clock (wall_clock64) around it; MHz = the ratio, averaged over the waves.
// bench.py prints it next to a hash of the GPU's uuid so that a profiles/ summary can be matched to the box a line came from (boxes
// of this pool differ by several per cent). Launched on a stream of its own -- never the NULL stream (see lwkzg_runtime_init).
__global__ __launch_bounds__(256) void k_clock_probe(unsigned long long *out, uint32_t iters, uint32_t seed) {
    uint64_t acc = seed + threadIdx.x;
    const uint32_t a = (seed * 2654435761u) | 1u, b = seed ^ 0x9e3779b9u;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) acc = (uint64_t)(uint32_t)acc * a + (acc >> 32) + b;
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out[3 * w] = (unsigned long long)(c1 - c0);
        out[3 * w + 1] = (unsigned long long)(w1 - w0);
        out[3 * w + 2] = acc;
    }
}

extern "C" C_KZG_RET lwkzg_clock_probe_mhz(double *mhz) {
    if (!mhz) return C_KZG_BADARGS;
    *mhz = 0;
    if (!gpu_available()) {
        set_error("no GPU: lambdaworks_kzg_amd has no CPU fallback");
        return C_KZG_ERROR;
    }
    LWK_HIP(hipSetDevice(g_default_device.load()));
    const unsigned wgs = 256, waves = wgs * 4;
    unsigned long long *d = nullptr;
    hipStream_t st = nullptr;
    LWK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (hipMalloc((void **)&d, waves * 3 * sizeof(unsigned long long)) != hipSuccess) {
        hipStreamDestroy(st);
        return C_KZG_MALLOC;
    }
    std::vector<unsigned long long> h(waves * 3);
    hipError_t e = hipSuccess;
    for (int rep = 0; rep < 3 && e == hipSuccess; rep++) {   // (the last of three back-to-back launches is the one read)
        hipLaunchKernelGGL(k_clock_probe, dim3(wgs), dim3(256), 0, st, d, 250u, 12345u + rep);
        e = hipStreamSynchronize(st);
    }
    if (e == hipSuccess) e = hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    hipFree(d);
    hipStreamDestroy(st);
    if (e != hipSuccess) {
        set_error("lwkzg_clock_probe_mhz: %s", hipGetErrorString(e));
        return C_KZG_ERROR;
    }
    double clk = 0, wall = 0;
    for (unsigned w = 0; w < waves; w++) {
        clk += (double)h[3 * w];
        wall += (double)h[3 * w + 1];
    }
    if (wall > 0) *mhz = clk / wall * 100.0;   // wall_clock64 ticks at 100 MHz
    return C_KZG_OK;
}

// JSON: where the milliseconds of this settings object's load and of its last table build went
extern "C" size_t lwkzg_timing_report(const KZGSettings *s, char *buf, size_t cap) {
    Ctx *c = ctx_of(s);
    char tmp[1024];
    int k = 0;
    if (c) {
        std::lock_guard<std::mutex> lk(c->mu);
        const LoadTiming &l = c->load_timing;
        const BuildTiming &b = c->last_build;
        k = snprintf(tmp, sizeof tmp,
                     "{\"load\": {\"context_ms\": %.3f, \"points_and_tables_ms\": %.3f, \"g2_and_fft_ms\": %.3f, \"default_table_ms\": %.3f, "
                     "\"total_ms\": %.3f}, \"last_table_build\": {\"bits\": %d, \"row_bytes\": %zu, \"table_bytes\": %zu, \"free_old_ms\": %.3f, "
                     "\"table_malloc_ms\": %.3f, \"scratch_malloc_ms\": %.3f, \"kernels_ms\": %.3f, \"scratch_free_ms\": %.3f, \"total_ms\": %.3f, \"in_place\": %d}}",
                     l.context_ms, l.points_and_tables_ms, l.g2_and_fft_ms, l.default_table_ms, l.total_ms, b.bits, b.row_bytes, b.table_bytes,
                     b.free_old_ms, b.table_malloc_ms, b.scratch_malloc_ms, b.kernels_ms, b.scratch_free_ms, b.total_ms, (int)b.in_place);
    } else {
        k = snprintf(tmp, sizeof tmp, "{}");
    }
    if (buf && cap) {
        size_t n = (size_t)k < cap - 1 ? (size_t)k : cap - 1;
        memcpy(buf, tmp, n);
        buf[n] = 0;
    }
    return (size_t)k + 1;
}

// ------------------------------------------------------------------------------------------------
// trusted setup

static C_KZG_RET setup_from_bytes(KZGSettings *out, const uint8_t *g1_bytes, const uint8_t *g2_bytes) {
    Ctx *c = nullptr;
    auto wall = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = wall();
    C_KZG_RET rc = ctx_new(&c);
    if (rc != C_KZG_OK) return rc;
    LoadTiming lt;
    lt.context_ms = wall() - t_start;
    double t_mark = wall();
    const size_t n1 = kBlobElems, n2 = TRUSTED_SETUP_NUM_G2_POINTS;
    uint8_t *d_in = nullptr;
    int32_t *d_status = nullptr;
    uint64_t *d_blst = nullptr;
    g1_t *g1v = (g1_t *)malloc(n1 * sizeof(g1_t));  // libc malloc: the reference frees these with libc::free (lib.rs:824-826)
    g2_t *g2v = (g2_t *)malloc(n2 * sizeof(g2_t));
    std::vector<int32_t> h_status(n1);
    rc = C_KZG_ERROR;
    do {
        if (!g1v || !g2v) { rc = C_KZG_MALLOC; break; }
        if (hipMalloc((void **)&d_in, n1 * 48) != hipSuccess || hipMalloc((void **)&d_status, n1 * 4) != hipSuccess ||
            hipMalloc((void **)&d_blst, n1 * 144) != hipSuccess) { rc = C_KZG_MALLOC; set_error("hipMalloc failed in setup load"); break; }
        if (hipMemcpyAsync(d_in, g1_bytes, n1 * 48, hipMemcpyHostToDevice, c->stream) != hipSuccess) { set_error("H2D of g1 bytes failed"); break; }
        // decompress_g1_point incl. the [r]P subgroup check for every point (compression.rs:62-103)
        launch_g1_decompress(d_in, c->points, d_status, n1, 1, c->stream);
        launch_g1_to_blst(c->points, d_status, d_blst, n1, c->stream);
        launch_build_table(c->points, c->table, c->stream);
        if (hipMemcpyAsync(h_status.data(), d_status, n1 * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipMemcpyAsync(g1v, d_blst, n1 * 144, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) { set_error("setup kernels failed: %s", hipGetErrorString(hipGetLastError())); break; }
        bool good = true;
        for (size_t i = 0; i < n1; i++) {
            if (h_status[i] == 2) { set_error("g1 point %zu: invalid compressed point or not in the subgroup", i); good = false; break; }
            if (h_status[i] == 1) { set_error("g1 point %zu is the point at infinity: the reference cannot read such a setup back (srs.rs:155-172)", i); good = false; break; }
        }
        if (!good) break;
        lt.points_and_tables_ms = wall() - t_mark;
        t_mark = wall();
        if (!g2_fill_values(g2v, g2_bytes, n2)) { if (!get_error()[0]) set_error("invalid g2 point in trusted setup"); break; }
        rc = ctx_finish_fft(c);
        lt.g2_and_fft_ms = wall() - t_mark;
    } while (0);
    if (d_in) hipFree(d_in);
    if (d_status) hipFree(d_status);
    if (d_blst) hipFree(d_blst);
    if (rc != C_KZG_OK) {
        free(g1v);
        free(g2v);
        ctx_destroy(c);
        return rc;
    }
    out->fs = &c->fs;
    out->g1_values = g1v;
    out->g2_values = g2v;
    t_mark = wall();
    direct_from_env(out);
    lt.default_table_ms = wall() - t_mark;
    lt.total_ms = wall() - t_start;
    c->load_timing = lt;
    return C_KZG_OK;
}

C_KZG_RET load_trusted_setup(KZGSettings *out, const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2) {
    if (!out || !g1_bytes || !g2_bytes) return C_KZG_BADARGS;
    if (n1 != TRUSTED_SETUP_NUM_G1_POINTS || n2 != TRUSTED_SETUP_NUM_G2_POINTS) return C_KZG_BADARGS;  // lib.rs:716-718
    return setup_from_bytes(out, g1_bytes, g2_bytes);
}

static int hexv(int ch) {
    if (ch >= '0' && ch <= '9') return ch - '0';
    if (ch >= 'a' && ch <= 'f') return ch - 'a' + 10;
    if (ch >= 'A' && ch <= 'F') return ch - 'A' + 10;
    return -1;
}

// srs.rs:25-82: line 1 = n1, line 2 = n2 (decimal), then exactly one hex point per line.
C_KZG_RET load_trusted_setup_file(KZGSettings *out, FILE *in) {
    if (!out || !in) return C_KZG_BADARGS;
    std::string text;
    char buf[64 * 1024];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, in)) > 0) text.append(buf, got);  // lib.rs:780-789
    std::vector<std::pair<size_t, size_t>> lines;  // (offset, length), str::lines semantics
    size_t pos = 0;
    while (pos < text.size()) {
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        size_t len = e - pos;
        if (len && text[pos + len - 1] == '\r') len--;
        lines.push_back({pos, len});
        pos = e + 1;
    }
    auto parse_count = [&](size_t li, size_t *v) -> bool {
        if (li >= lines.size()) return false;
        size_t o = lines[li].first, l = lines[li].second, k = 0;
        if (l && text[o] == '+') k = 1;
        if (k == l) return false;
        size_t acc = 0;
        for (; k < l; k++) {
            char ch = text[o + k];
            if (ch < '0' || ch > '9') return false;
            acc = acc * 10 + (size_t)(ch - '0');
            if (acc > (1u << 24)) return false;
        }
        *v = acc;
        return true;
    };
    size_t n1 = 0, n2 = 0;
    if (!parse_count(0, &n1) || !parse_count(1, &n2)) {
        set_error("trusted setup file: bad header");
        return C_KZG_ERROR;
    }
    // The reference does not check n1 here and later reads 4096 entries regardless (UB for other
    // sizes); this engine is built for 4096/65 and says so.
    if (n1 != TRUSTED_SETUP_NUM_G1_POINTS || n2 != TRUSTED_SETUP_NUM_G2_POINTS) {
        set_error("trusted setup file announces %zu/%zu points; this engine needs 4096/65", n1, n2);
        return C_KZG_BADARGS;
    }
    if (lines.size() < 2 + n1 + n2) {
        set_error("trusted setup file: %zu point lines, expected %zu", lines.size() - 2, n1 + n2);
        return C_KZG_ERROR;
    }
    std::vector<uint8_t> g1(n1 * 48), g2(n2 * 96);
    for (size_t i = 0; i < n1 + n2; i++) {
        size_t nb = i < n1 ? 48 : 96;
        uint8_t *dst = i < n1 ? &g1[i * 48] : &g2[(i - n1) * 96];
        size_t o = lines[2 + i].first, l = lines[2 + i].second;
        if (l != 2 * nb) {
            set_error("trusted setup file: line %zu has %zu characters, expected %zu", i + 3, l, 2 * nb);
            return C_KZG_ERROR;
        }
        for (size_t k = 0; k < nb; k++) {
            int h = hexv(text[o + 2 * k]), lo = hexv(text[o + 2 * k + 1]);
            if (h < 0 || lo < 0) {
                set_error("trusted setup file: line %zu is not hex", i + 3);
                return C_KZG_ERROR;
            }
            dst[k] = (uint8_t)(h * 16 + lo);
        }
    }
    return setup_from_bytes(out, g1.data(), g2.data());
}

C_KZG_RET free_trusted_setup(KZGSettings *s) {
    if (!s) return C_KZG_OK;
    Ctx *c = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (s->fs && g_live_fs.count((const void *)s->fs)) {
            c = (Ctx *)s->fs;
        } else {
            auto it = g_registry.find(s->g1_values);
            if (it != g_registry.end()) {
                c = it->second.ctx;
                g_registry.erase(it);
            }
        }
    }
    ctx_destroy(c);
    free(s->g1_values);  // lib.rs:824-826
    free(s->g2_values);
    s->fs = nullptr;
    s->g1_values = nullptr;
    s->g2_values = nullptr;
    return C_KZG_OK;
}

// A KZGSettings filled in by hand (fs == NULL, caller-owned arrays: the reference's own layout) gets a device context
// on first use, cached by its g1_values pointer. free_trusted_setup would free() the caller's arrays; this drops only
// the cached context (tables, workspace, streams). The settings stay usable: the next call builds a new one.
C_KZG_RET lwkzg_release_context(const KZGSettings *s) {
    if (!s) return C_KZG_BADARGS;
    Ctx *c = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (s->fs && g_live_fs.count((const void *)s->fs)) return C_KZG_BADARGS;  // a loaded setup: free_trusted_setup owns it
        auto it = g_registry.find(s->g1_values);
        if (it == g_registry.end()) return C_KZG_OK;
        c = it->second.ctx;
        g_registry.erase(it);
    }
    ctx_destroy(c);
    return C_KZG_OK;
}

// ------------------------------------------------------------------------------------------------
// host-pointer entry points

// maps per-blob status words to one return code; first_bad gets the first offender
static C_KZG_RET collect_status(Ctx *c, const int32_t *d_status, size_t n, size_t base, size_t *first_bad) {
    std::vector<int32_t> h(n);
    LWK_HIP(hipMemcpyAsync(h.data(), d_status, n * 4, hipMemcpyDeviceToHost, c->stream));
    LWK_HIP(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; i++)
        if (h[i] != 0) {
            if (first_bad) *first_bad = base + i;
            set_error("blob %zu rejected (status %d)", base + i, h[i]);
            return (C_KZG_RET)h[i];
        }
    return C_KZG_OK;
}

// In reference mode every failure is C_KZG_ERROR (lib.rs:263,267,272,...).
static C_KZG_RET map_rc(C_KZG_RET rc, int mode) {
    if (rc == C_KZG_OK) return rc;
    if (mode == LWKZG_MODE_REFERENCE) return C_KZG_ERROR;
    return rc;
}

// Slice schedule of the long host batches: 512 blobs at a time (one half of the workspace), but the first two slices are
// 128 + 384, so that the GPU is at work after a quarter of the first upload (512 commitments 6.4 instead of 7.1 ms; 1024
// on a direct table 11.9 instead of 12.8 ms, 85.7k instead of 80.2k ops/s through the host ABI). On the bucket engine a
// batch of a whole chunk or more keeps whole slices: its small launches cost what the earlier start gains.
static size_t slice_len(size_t k, size_t remaining, size_t n, bool direct) {
    const size_t first_env = knobs().slice0;  // experiment: length of the first slice
    const size_t first = first_env ? first_env : kMaxChunk / 8;
    size_t want = kMaxChunk / 2;
    if ((n < kMaxChunk || direct || first_env) && k < 2 && first < kMaxChunk / 2) want = k == 0 ? first : kMaxChunk / 2 - first;
    return remaining < want ? remaining : want;
}

// ------------------------------------------------------------------------------------------------
// Coalescing front of blob_to_kzg_commitment (engine.h: Combiner). Contract matched: concurrent callers on one
// KZGSettings, /root/reference/src/lib.rs:253-283 + SURVEY 8b "Threading".

// LWKZG_COALESCE=0: single-blob calls are not merged with concurrent ones
static bool coalesce_singles() {
    return knobs().coalesce;
}

static bool combiner_init(Ctx *c) {
    Combiner &cb = c->comb;
    std::lock_guard<std::mutex> lk(cb.init_m);
    if (cb.ready || cb.failed) return cb.ready;
    hipSetDevice(c->device);
    bool ok = hipHostMalloc((void **)&cb.pinned_blobs, kCombineSlots * (size_t)kBlobBytes, hipHostMallocDefault) == hipSuccess;
    for (int k = 0; k < kCombineLanes && ok; k++)
        ok = hipHostMalloc((void **)&cb.pinned_out[k], kCombineMaxBatch * sizeof(G1Xyzz29), hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void **)&cb.pinned_status[k], kCombineMaxBatch * 4, hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        cb.failed = true;  // callers fall back to one launch set each
        return false;
    }
    {
        std::lock_guard<std::mutex> fl(cb.front.m);
        cb.front.add_slots((int)kCombineSlots);
    }
    cb.ready = true;
    return true;
}

// one launch set for `batch` (all of one mode) on lane `lane`; fills every request's rc and output
static void combine_run(Ctx *c, int lane, const std::vector<CombineReq *> &batch, bool plain = false) {
    Combiner &cb = c->comb;
    const size_t n = batch.size();
    const int mode = batch[0]->mode;
    const size_t lo = (size_t)lane * kCombineMaxBatch;  // this lane's slice of the workspace
    hipStream_t sk = c->aux[lane];
    const bool host_finish = n <= host_finish_limit();
    bool zero_copy = false;
    uint32_t *redo_flag = nullptr;
    C_KZG_RET rc = C_KZG_OK;
    {
        std::lock_guard<std::mutex> lk(c->mu);  // enqueue only: the wait below happens outside
        bool ok = hipSetDevice(c->device) == hipSuccess;
        if (ok) {
            rc = ctx_reserve(c, kCombineLanes * kCombineMaxBatch);
            ok = rc == C_KZG_OK;
        }
        if (ok) {
            WsLaneUse use(c, lane);
            Workspace &w = c->ws;
            uint8_t *d_blobs = w.blobs + lo * (size_t)kBlobBytes;
            // ONE blob -- the reference's call shape (src/lib.rs:253-283) -- moves no buffer at all (r06): the parse kernel reads the blob from
            // its pinned staging slot across the link, the cooperative kernel's last wave stores the sum into pinned memory, and in
            // reference mode, where a blob cannot be rejected, no verdict is cleared or fetched. A kernel trace of r05's call showed the
            // 118 us kernel among 57 us of copies, fills and the gaps between them (profiles/r06_experiments.md section 7).
            zero_copy = n == 1 && host_finish && !plain && knobs().zero_copy && (c->direct_table != nullptr || c->lag.direct_table != nullptr);
            if (zero_copy) d_blobs = cb.pinned_blobs + (size_t)batch[0]->slot * kBlobBytes;
            // ... and in reference mode on the cooperative kernel two more launches go: the fill of the hand-off counters (the parse kernel
            // clears them on its way) and the second pass that exits at once on honest data (the redo flag is a pinned word this thread
            // looks at after its one synchronisation; a flagged call -- P = +-Q inside a quad: chosen scalars only -- is repeated the long way)
            uint32_t ctr_words = 0;
            if (zero_copy && mode == LWKZG_MODE_REFERENCE && c->direct_table) ctr_words = direct_one_blob_counter_words(c->direct_bits);
            if (ctr_words) {
                redo_flag = (uint32_t *)&cb.pinned_status[lane][1];
                *redo_flag = 0;
            }
            for (size_t i = 0; i < n && ok && !zero_copy; i++)
                ok = hipMemcpyAsync(d_blobs + i * (size_t)kBlobBytes, cb.pinned_blobs + (size_t)batch[i]->slot * kBlobBytes, kBlobBytes,
                                    hipMemcpyHostToDevice, sk) == hipSuccess;
            const bool verdicts = !(zero_copy && mode == LWKZG_MODE_REFERENCE);
            if (verdicts) ok = ok && hipMemsetAsync(w.status + lo, 0, n * 4, sk) == hipSuccess;
            else cb.pinned_status[lane][0] = 0;
            if (ok) {
                uint32_t *ctr0 = w.bucket_start + lo * (size_t)(kNumBuckets + 1) + 1;   // (msm_sums_stage: redo flags at bstart, the counters behind them)
                const bool lg = coefficients_stage(c, d_blobs, n, mode, w.status + lo, sk, lo, true, redo_flag ? ctr0 : nullptr, ctr_words);
                if (zero_copy) {
                    (void)msm_sums_stage(c, w.scalars + lo * (size_t)kBlobElems * 8, n, sk, lo, false, lg, (G1Xyzz29 *)cb.pinned_out[lane], redo_flag);
                } else if (host_finish) {  // the sums come back as they are; inversion and compression below, on this thread
                    const G1Xyzz29 *sums = msm_sums_stage(c, w.scalars + lo * (size_t)kBlobElems * 8, n, sk, lo, false, lg);
                    ok = hipMemcpyAsync(cb.pinned_out[lane], sums, n * sizeof(G1Xyzz29), hipMemcpyDeviceToHost, sk) == hipSuccess;
                } else {
                    msm_stages(c, w.scalars + lo * (size_t)kBlobElems * 8, w.out48 + 48 * lo, n, sk, lo, false, lg);
                    ok = hipMemcpyAsync(cb.pinned_out[lane], w.out48 + 48 * lo, n * 48, hipMemcpyDeviceToHost, sk) == hipSuccess;
                }
                if (verdicts) ok = ok && hipMemcpyAsync(cb.pinned_status[lane], w.status + lo, n * 4, hipMemcpyDeviceToHost, sk) == hipSuccess;
            }
        }
        if (!ok && rc == C_KZG_OK) {
            set_error("blob_to_kzg_commitment (coalesced): %s", hipGetErrorString(hipGetLastError()));
            rc = C_KZG_ERROR;
        }
    }
    if (rc == C_KZG_OK && hipStreamSynchronize(sk) != hipSuccess) {
        set_error("blob_to_kzg_commitment (coalesced): %s", hipGetErrorString(hipGetLastError()));
        rc = C_KZG_ERROR;
    }
    if (rc == C_KZG_OK && redo_flag && *redo_flag != 0) return combine_run(c, lane, batch, true);   // the complete-branches pass, the long way
    for (size_t i = 0; i < n; i++) {
        CombineReq *r = batch[i];
        if (rc != C_KZG_OK) {
            r->rc = rc;
        } else if (cb.pinned_status[lane][i] != 0) {
            r->rc = (int)map_rc((C_KZG_RET)cb.pinned_status[lane][i], mode);
        } else {
            if (host_finish) host_finish_compress(r->out48, ((const G1Xyzz29 *)cb.pinned_out[lane])[i]);
            else memcpy(r->out48, cb.pinned_out[lane] + 48 * i, 48);
            r->rc = C_KZG_OK;
        }
    }
}

// blob_to_kzg_commitment for one blob, merged with whatever other callers are waiting. Returns false when the front is
// unavailable (no pinned memory): the caller then takes the plain path.
static bool combine_commit(Ctx *c, uint8_t *out48, const uint8_t *blob, int mode, C_KZG_RET *rc_out) {
    Combiner &cb = c->comb;
    if (!combiner_init(c)) return false;
    CombineReq req;
    req.mode = mode;
    req.out48 = out48;
    // front.h: every caller stages its own blob (in parallel), the first to find a free lane leads everything queued
    const int rc = cb.front.submit(
        req, kCombineMaxBatch, (int)C_KZG_MALLOC,
        [&](int slot) { memcpy(cb.pinned_blobs + (size_t)slot * kBlobBytes, blob, kBlobBytes); },
        [&](int lane, const std::vector<CombineReq *> &batch) { combine_run(c, lane, batch); });
    *rc_out = (C_KZG_RET)rc;
    return true;
}

// (the host-pointer entry points allocate host vectors: nothing may unwind across the C ABI -- the exported symbols wrap these)
static C_KZG_RET commitment_batch_impl(KZGCommitment *out, const Blob *blobs, size_t n, const KZGSettings *s, size_t *first_bad) {
    const int mode = mode_of(s);
    if (!out || !blobs) return map_rc(C_KZG_BADARGS, mode);
    Ctx *c = ctx_of(s);
    if (!c) return C_KZG_ERROR;
    ensure_lagrange(c, mode);
    if (n == 1) {  // the reference's symbol: merged with the other callers of the moment
        C_KZG_RET rc1;
        uint8_t tmp[48];
        if (coalesce_singles() && combine_commit(c, tmp, (const uint8_t *)blobs, mode, &rc1)) {
            if (rc1 == C_KZG_OK) memcpy(out, tmp, 48);
            else if (first_bad) *first_bad = 0;
            if (rc1 != C_KZG_OK && !get_error()[0]) set_error("blob 0 rejected");
            return rc1;
        }
    }
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    WsUse wsu(c, c->stream);
    // (the bucket engine overlaps the tails of its own sub-batches inside commit_batch_device, which a lone 512-blob slice
    // would forgo: it takes the single pass up to a whole chunk)
    if (n < 512 || (!c->direct_table && n <= kMaxChunk)) {  // one launch set, results and verdicts back in one go
        C_KZG_RET rc = ctx_reserve(c, n);
        if (rc != C_KZG_OK) return rc;
        Workspace &w = c->ws;
        LWK_HIP(hipMemcpyAsync(w.blobs, blobs, n * (size_t)kBlobBytes, hipMemcpyHostToDevice, c->stream));
        if (n <= host_finish_limit()) {   // a handful of results: the sums come back as they are, inversion and compression on this thread
            LWK_HIP(hipMemsetAsync(w.status, 0, n * 4, c->stream));
            const bool lg = coefficients_stage(c, w.blobs, n, mode, w.status, c->stream, 0, true);
            const G1Xyzz29 *d_sums = msm_sums_stage(c, w.scalars, n, c->stream, 0, false, lg);
            std::vector<G1Xyzz29> h_sums(n);
            LWK_HIP(hipMemcpyAsync(h_sums.data(), d_sums, n * sizeof(G1Xyzz29), hipMemcpyDeviceToHost, c->stream));
            rc = collect_status(c, w.status, n, 0, first_bad);
            if (rc != C_KZG_OK) return map_rc(rc, mode);
            for (size_t i = 0; i < n; i++) host_finish_compress(out[i].bytes, h_sums[i]);
            return C_KZG_OK;
        }
        rc = commit_batch_device(c, w.out48, w.blobs, n, mode, c->stream, w.status);
        if (rc != C_KZG_OK) return rc;
        std::vector<uint8_t> h_out(n * 48);
        LWK_HIP(hipMemcpyAsync(h_out.data(), w.out48, n * 48, hipMemcpyDeviceToHost, c->stream));
        rc = collect_status(c, w.status, n, 0, first_bad);
        if (rc != C_KZG_OK) return map_rc(rc, mode);
        memcpy(out, h_out.data(), n * 48);
        return C_KZG_OK;
    }
    // Long batches stream through in slices of 512 blobs: slice k uses half k mod 2 of the workspace and stream
    // k mod 2, so the pageable H2D copy of a slice (which blocks this thread while it is staged) runs beside the GPU's
    // work on the previous one, a half is only reused by the stream that used it last (stream order is the only
    // synchronisation needed), and nothing waits for the host until every slice has been submitted.
    constexpr size_t kSlice = kMaxChunk / 2;
    C_KZG_RET rc = ctx_reserve(c, kMaxChunk);
    if (rc != C_KZG_OK) return rc;
    Workspace &w = c->ws;
    std::vector<int32_t> h_status(n);
    // results and verdicts of all slices stay on the device until the end: a D2H copy into pageable memory would make
    // this thread wait for the slice it belongs to
    uint8_t *d_out_all = host_res_block(c, ((n * 48 + 255) & ~(size_t)255) + n * 4);
    if (!d_out_all) {
        set_error("lwkzg_blob_to_kzg_commitment_batch: out of device memory for %zu results", n);
        return C_KZG_MALLOC;
    }
    int32_t *d_status_all = (int32_t *)(d_out_all + ((n * 48 + 255) & ~(size_t)255));
    LWK_HIP(hipEventRecord(c->ev_fork, c->stream));
    if (n > kMaxChunk && dev_stage_ready(c)) {   // (up to one chunk r05's 128 + 384 + 512 slices measure better: 81.9k against 79.9k ops/s at 1024 blobs)
        // r06 (engine.h: DevStage): the slices are uploaded into a device-side double buffer on a copy stream and go through the
        // device-resident pipeline itself -- whole chunks, one compute stream -- while the next one is on its way
        hipStream_t st = c->stream;
        LWK_HIP(hipStreamWaitEvent(upload_stream(c), c->ev_fork, 0));
        size_t k = 0;
        for (size_t off = 0, cnt = 0; off < n; off += cnt, k++) {
            cnt = stage_slice_len(k, n - off);
            uint8_t *d_blobs = nullptr;
            const auto tu0 = std::chrono::steady_clock::now();
            rc = stage_upload(c, k, (const uint8_t *)(blobs + off), cnt, st, &d_blobs);
            if (knobs().timing)
                fprintf(stderr, "[lambdaworks_kzg_amd] staged commitments: slice %zu (%zu blobs) upload call %.2f ms\n", k, cnt,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tu0).count());
            if (rc == C_KZG_OK) rc = commit_batch_device(c, d_out_all + 48 * off, d_blobs, cnt, mode, st, d_status_all + off);
            if (rc == C_KZG_OK) rc = stage_parsed(c, k, st);   // (behind the whole slice: the bucket engine's sub-batches parse on streams of their own)
            if (rc != C_KZG_OK) {
                hipStreamSynchronize(upload_stream(c));
                hipStreamSynchronize(st);
                return rc;
            }
        }
        LWK_HIP(hipMemcpyAsync((uint8_t *)out, d_out_all, n * 48, hipMemcpyDeviceToHost, st));
        LWK_HIP(hipMemcpyAsync(h_status.data(), d_status_all, n * 4, hipMemcpyDeviceToHost, st));
        LWK_HIP(hipStreamSynchronize(st));
        for (size_t i = 0; i < n; i++)
            if (h_status[i] != 0) {
                if (first_bad) *first_bad = i;
                set_error("blob %zu rejected (status %d)", i, h_status[i]);
                return map_rc((C_KZG_RET)h_status[i], mode);
            }
        return C_KZG_OK;
    }
    LWK_HIP(hipStreamWaitEvent(c->aux[0], c->ev_fork, 0));
    LWK_HIP(hipStreamWaitEvent(c->aux[1], c->ev_fork, 0));
    size_t k = 0;
    for (size_t off = 0, cnt = 0; off < n; off += cnt, k++) {
        cnt = slice_len(k, n - off, n, c->direct_table != nullptr || c->lag.direct_table != nullptr);
        const size_t lo = (k % 2) * kSlice;
        hipStream_t sk = c->aux[k & 1];
        uint8_t *d_blobs = w.blobs + lo * (size_t)kBlobBytes;
        LWK_HIP(hipMemcpyAsync(d_blobs, (const uint8_t *)(blobs + off), cnt * (size_t)kBlobBytes, hipMemcpyHostToDevice, sk));
        LWK_HIP(hipMemsetAsync(d_status_all + off, 0, cnt * 4, sk));
        const bool lg = coefficients_stage(c, d_blobs, cnt, mode, d_status_all + off, sk, lo, true);
        msm_stages(c, w.scalars + lo * (size_t)kBlobElems * 8, d_out_all + 48 * off, cnt, sk, lo, false, lg);
    }
    for (int j = 0; j < 2; j++) {
        LWK_HIP(hipEventRecord(c->ev_join[j], c->aux[j]));
        LWK_HIP(hipStreamWaitEvent(c->stream, c->ev_join[j], 0));
    }
    LWK_HIP(hipMemcpyAsync((uint8_t *)out, d_out_all, n * 48, hipMemcpyDeviceToHost, c->stream));
    LWK_HIP(hipMemcpyAsync(h_status.data(), d_status_all, n * 4, hipMemcpyDeviceToHost, c->stream));
    LWK_HIP(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; i++)
        if (h_status[i] != 0) {
            if (first_bad) *first_bad = i;
            set_error("blob %zu rejected (status %d)", i, h_status[i]);
            return map_rc((C_KZG_RET)h_status[i], mode);
        }
    return C_KZG_OK;
}

// ------------------------------------------------------------------------------------------------
// Long host batches of proofs (512 blobs and up) stream through in slices of 512 blobs, like the commitments above:
// slice k uses half k mod 2 of the workspace on stream k mod 2, so the pageable H2D copy of slice k + 1 (which blocks
// this thread while it is staged) runs beside the GPU's work on slice k. Results and verdicts of all slices stay on
// the device until the end. The caller holds c->mu.

namespace {

struct DevBlock {  // a piece of the context's result block (host_res_block) carved into 256-byte aligned pieces
    uint8_t *base = nullptr;
    size_t used = 0, cap = 0;
    static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
    bool alloc(Ctx *c, size_t bytes) {
        cap = bytes;
        base = host_res_block(c, bytes);
        return base != nullptr;
    }
    uint8_t *take(size_t bytes) {
        uint8_t *p = base + used;
        used += pad(bytes);
        return p;
    }
};

C_KZG_RET scan_status(const std::vector<int32_t> &h_status, size_t *first_bad, int mode) {
    for (size_t i = 0; i < h_status.size(); i++)
        if (h_status[i] != 0) {
            if (first_bad) *first_bad = i;
            set_error("blob %zu rejected (status %d)", i, h_status[i]);
            return map_rc((C_KZG_RET)h_status[i], mode);
        }
    return C_KZG_OK;
}

C_KZG_RET point_proofs_sliced(Ctx *c, uint8_t *proofs_out, uint8_t *ys_out, const uint8_t *blobs, const uint8_t *zs, size_t n,
                              int mode, size_t *first_bad) {
    constexpr size_t kSlice = kMaxChunk / 2;
    const int le = mode == LWKZG_MODE_CKZG;
    C_KZG_RET rc = ctx_reserve(c, kMaxChunk);
    if (rc != C_KZG_OK) return rc;
    Workspace &w = c->ws;
    DevBlock blk;
    if (!blk.alloc(c, DevBlock::pad(n * 48) + 2 * DevBlock::pad(n * 32) + DevBlock::pad(n * 4))) {
        set_error("lwkzg_compute_kzg_proof_batch: out of device memory for %zu results", n);
        return C_KZG_MALLOC;
    }
    uint8_t *d_out = blk.take(n * 48), *d_y = blk.take(n * 32), *d_z = blk.take(n * 32);
    int32_t *d_status = (int32_t *)blk.take(n * 4);
    std::vector<int32_t> h_status(n);
    hipStream_t st = c->stream;
    LWK_HIP(hipMemcpyAsync(d_z, zs, n * 32, hipMemcpyHostToDevice, st));
    LWK_HIP(hipMemsetAsync(d_status, 0, n * 4, st));
    LWK_HIP(hipEventRecord(c->ev_fork, st));
    // (r06: the staged whole-chunk schedule of the commitments -- engine.h: DevStage -- measured 3 % SLOWER here, 88.6k against 91.2k proofs/s at
    // 4096 blobs: on two streams the evaluation, fold and finalize of one slice run beside the other slice's MSM; profiles/r06_experiments.md section 6)
    LWK_HIP(hipStreamWaitEvent(c->aux[0], c->ev_fork, 0));
    LWK_HIP(hipStreamWaitEvent(c->aux[1], c->ev_fork, 0));
    size_t k = 0;
    for (size_t off = 0, cnt = 0; off < n; off += cnt, k++) {
        cnt = slice_len(k, n - off, n, c->direct_table != nullptr || c->lag.direct_table != nullptr);
        const size_t lo = (k % 2) * kSlice, so = lo * (size_t)kBlobElems * 8;
        hipStream_t sk = c->aux[k & 1];
        uint8_t *d_blobs = w.blobs + lo * (size_t)kBlobBytes;
        LWK_HIP(hipMemcpyAsync(d_blobs, blobs + off * (size_t)kBlobBytes, cnt * (size_t)kBlobBytes, hipMemcpyHostToDevice, sk));
        coefficients_stage(c, d_blobs, cnt, mode, d_status + off, sk, lo);
        launch_z_from_bytes(d_z + 32 * off, w.z + lo, d_status + off, le, cnt, sk);
        quotient_stage(c, mode, w.scalars + so, w.z + lo, w.scalars2 + so, d_y + 32 * off, le, cnt, sk);
        msm_stages(c, w.scalars2 + so, d_out + 48 * off, cnt, sk, lo, false, quotient_to_msm_form(c, mode, cnt, sk, lo));
    }
    for (int j = 0; j < 2; j++) {
        LWK_HIP(hipEventRecord(c->ev_join[j], c->aux[j]));
        LWK_HIP(hipStreamWaitEvent(st, c->ev_join[j], 0));
    }
    std::vector<uint8_t> h_out(n * 48), h_y(n * 32);  // the caller's buffers are only written on success
    LWK_HIP(hipMemcpyAsync(h_out.data(), d_out, n * 48, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipMemcpyAsync(h_y.data(), d_y, n * 32, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipMemcpyAsync(h_status.data(), d_status, n * 4, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipStreamSynchronize(st));
    rc = scan_status(h_status, first_bad, mode);
    if (rc != C_KZG_OK) return rc;
    memcpy(proofs_out, h_out.data(), n * 48);
    memcpy(ys_out, h_y.data(), n * 32);
    return C_KZG_OK;
}

// compute_blob_kzg_proof for a long batch: every commitment is validated once up front on the validation stream (the
// kernel is a ~2 ms latency chain whatever n is); the host threads hash slice k (the digests assume the caller's
// commitment bytes are canonical) while it is copied; the canonical bytes decide per slice between those digests and
// the GPU hash over the canonical encoding.
C_KZG_RET blob_proofs_sliced(Ctx *c, uint8_t *out, const uint8_t *blobs, const uint8_t *comm48, size_t n, int mode,
                             size_t *first_bad) {
    constexpr size_t kSlice = kMaxChunk / 2;
    const int le = mode == LWKZG_MODE_CKZG;
    C_KZG_RET rc = ctx_reserve(c, kMaxChunk);
    if (rc != C_KZG_OK) return rc;
    Workspace &w = c->ws;
    DevBlock blk;
    if (!blk.alloc(c, 3 * DevBlock::pad(n * 48) + DevBlock::pad(n * 32) + DevBlock::pad(n * 4))) {
        set_error("lwkzg_compute_blob_kzg_proof_batch: out of device memory for %zu results", n);
        return C_KZG_MALLOC;
    }
    const bool timing = knobs().timing;  // phase wall-clock to stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    uint8_t *d_out = blk.take(n * 48), *d_comm = blk.take(n * 48), *d_canon = blk.take(n * 48);
    uint8_t *d_dig = blk.take(n * 32);
    int32_t *d_status = (int32_t *)blk.take(n * 4);
    std::vector<int32_t> h_status(n);
    std::vector<uint8_t> h_canon(n * 48), h_dig(n * 32);
    hipStream_t st = c->stream, sv = c->vstream;
    LWK_HIP(hipMemsetAsync(d_status, 0, n * 4, st));
    LWK_HIP(hipEventRecord(c->ev_fork, st));
    LWK_HIP(hipStreamWaitEvent(sv, c->ev_fork, 0));
    LWK_HIP(hipStreamWaitEvent(c->aux[0], c->ev_fork, 0));
    LWK_HIP(hipStreamWaitEvent(c->aux[1], c->ev_fork, 0));
    LWK_HIP(hipMemcpyAsync(d_comm, comm48, n * 48, hipMemcpyHostToDevice, sv));
    launch_validate_commitments(d_comm, d_canon, d_status, le ? kStatusBadArgs : kStatusError, n, sv);  // lib.rs:372-375
    LWK_HIP(hipEventRecord(c->ev_join[kMaxSplit - 1], sv));
    bool validated = false;
    // (r06: slices on two streams stay -- the staged whole-chunk schedule of the commitments measured 84.9k against 88.1k proofs/s here)
    size_t k = 0;
    for (size_t off = 0, cnt = 0; off < n; off += cnt, k++) {
        cnt = slice_len(k, n - off, n, c->direct_table != nullptr || c->lag.direct_table != nullptr);
        const size_t lo = (k % 2) * kSlice, so = lo * (size_t)kBlobElems * 8;
        hipStream_t sk = c->aux[k & 1];
        uint8_t *d_blobs = w.blobs + lo * (size_t)kBlobBytes;
        const uint8_t *hb = blobs + off * (size_t)kBlobBytes, *hc = comm48 + 48 * off;
        uint8_t *dig = h_dig.data() + 32 * off;
        // the host threads hash the slice (the digests assume canonical commitment bytes) beside its upload, which blocks this thread while
        // the runtime stages it
        SideTask hasher([=]() { challenge_digests_host(dig, hb, hc, cnt); });
        const auto ta = now();
        LWK_HIP(hipMemcpyAsync(d_blobs, hb, cnt * (size_t)kBlobBytes, hipMemcpyHostToDevice, sk));
        const auto tb = now();
        coefficients_stage(c, d_blobs, cnt, mode, d_status + off, sk, lo);
        hasher.join();
        const auto tc = now();
        if (timing) fprintf(stderr, "[blob_proofs_sliced] slice %zu at %.2f ms: h2d %.2f ms, hash wait %.2f ms\n", k, ms(t0, ta), ms(ta, tb), ms(tb, tc));
        if (!validated) {
            // A device-to-host copy into pageable memory blocks this thread until the stream has reached it, so the
            // canonical bytes are only fetched here, after the first slice has been submitted. From here on they are
            // on the host, and the slice streams may read d_canon.
            LWK_HIP(hipMemcpyAsync(h_canon.data(), d_canon, n * 48, hipMemcpyDeviceToHost, sv));
            LWK_HIP(hipStreamSynchronize(sv));
            validated = true;
        }
        if (memcmp(h_canon.data() + 48 * off, hc, cnt * 48) == 0) {
            LWK_HIP(hipMemcpyAsync(d_dig + 32 * off, dig, cnt * 32, hipMemcpyHostToDevice, sk));
            launch_z_from_bytes(d_dig + 32 * off, w.z + lo, nullptr, le, cnt, sk);  // digest -> Fr, reduced (utils.rs:148-154)
        } else {  // a non-canonical (or invalid) encoding in this slice: hash the canonical bytes on the GPU
            launch_challenge(d_blobs, d_canon + 48 * off, w.z + lo, le, cnt, sk);
        }
        quotient_stage(c, mode, w.scalars + so, w.z + lo, w.scalars2 + so, nullptr, le, cnt, sk);
        msm_stages(c, w.scalars2 + so, d_out + 48 * off, cnt, sk, lo, false, quotient_to_msm_form(c, mode, cnt, sk, lo));
    }
    for (int j = 0; j < 2; j++) {
        LWK_HIP(hipEventRecord(c->ev_join[j], c->aux[j]));
        LWK_HIP(hipStreamWaitEvent(st, c->ev_join[j], 0));
    }
    LWK_HIP(hipStreamWaitEvent(st, c->ev_join[kMaxSplit - 1], 0));
    std::vector<uint8_t> h_out(n * 48);
    LWK_HIP(hipMemcpyAsync(h_out.data(), d_out, n * 48, hipMemcpyDeviceToHost, st));
    LWK_HIP(hipMemcpyAsync(h_status.data(), d_status, n * 4, hipMemcpyDeviceToHost, st));
    const auto td = now();
    LWK_HIP(hipStreamSynchronize(st));
    if (timing) fprintf(stderr, "[blob_proofs_sliced] n=%zu: submitted at %.2f ms, drained at %.2f ms\n", n, ms(t0, td), ms(t0, now()));
    rc = scan_status(h_status, first_bad, mode);
    if (rc != C_KZG_OK) return rc;
    memcpy(out, h_out.data(), n * 48);
    return C_KZG_OK;
}

}  // namespace

static C_KZG_RET blob_proof_batch_host(Ctx *c, KZGProof *out, const Blob *blobs, const Bytes48 *commitments, size_t n, int mode,
                                       size_t *first_bad);

static C_KZG_RET point_proof_batch_host(Ctx *c, KZGProof *proofs_out, Bytes32 *ys_out, const Blob *blobs, const Bytes32 *zs, size_t n,
                                        int mode, size_t *first_bad);

// Concurrent callers of compute_blob_kzg_proof / compute_kzg_proof (one blob per call, as a block builder issues them):
// whoever arrives while no batch is being run becomes the leader of everything queued in its mode (<= 64) and hands it to
// `run`, which answers every member; the others wait for their bytes (front.h: LeaderFront). A `run` that throws (the
// leader's host vectors: std::bad_alloc) answers every member with C_KZG_MALLOC instead of unwinding across the C ABI.
static C_KZG_RET front_run(ProofFront &pf, ProofReq &req, const std::function<void(const std::vector<ProofReq *> &)> &run) {
    return (C_KZG_RET)pf.submit(req, kCombineMaxBatch, (int)C_KZG_MALLOC, run);
}

// The leader copies the blobs and commitments into contiguous host arrays and runs them as ONE host-pointer batch (host
// threads hash, one launch set). If the batch fails (an invalid commitment somewhere in it), every member is redone on
// its own, so each caller gets exactly the return code a lone call would have given.
static C_KZG_RET combine_blob_proof(Ctx *c, KZGProof *out, const Blob *blob, const Bytes48 *commitment, int mode) {
    ProofReq req;
    req.blob = (const uint8_t *)blob;
    req.second = (const uint8_t *)commitment;
    req.out = (uint8_t *)out;
    req.mode = mode;
    return front_run(c->blob_proof_front, req, [c, mode](const std::vector<ProofReq *> &batch) {
        auto alone = [c, mode](ProofReq *r) {
            r->rc = blob_proof_batch_host(c, (KZGProof *)r->out, (const Blob *)r->blob, (const Bytes48 *)r->second, 1, mode, nullptr);
        };
        const size_t m = batch.size();
        if (m == 1) return alone(batch[0]);
        std::vector<uint8_t> hb(m * (size_t)kBlobBytes), hc(m * 48), ho(m * 48);
        for (size_t i = 0; i < m; i++) {
            memcpy(&hb[i * (size_t)kBlobBytes], batch[i]->blob, kBlobBytes);
            memcpy(&hc[48 * i], batch[i]->second, 48);
        }
        C_KZG_RET rc = blob_proof_batch_host(c, (KZGProof *)ho.data(), (const Blob *)hb.data(), (const Bytes48 *)hc.data(), m, mode, nullptr);
        for (size_t i = 0; i < m; i++) {
            if (rc != C_KZG_OK) {  // somebody's input was rejected: everyone gets the verdict of a call of their own
                alone(batch[i]);
                continue;
            }
            memcpy(batch[i]->out, &ho[48 * i], 48);
            batch[i]->rc = C_KZG_OK;
        }
    });
}

// compute_kzg_proof the same way: blobs and evaluation points side by side, proofs and y values back.
static C_KZG_RET combine_point_proof(Ctx *c, KZGProof *proof_out, Bytes32 *y_out, const Blob *blob, const Bytes32 *z, int mode) {
    ProofReq req;
    req.blob = (const uint8_t *)blob;
    req.second = (const uint8_t *)z;
    req.out = (uint8_t *)proof_out;
    req.y_out = (uint8_t *)y_out;
    req.mode = mode;
    return front_run(c->point_proof_front, req, [c, mode](const std::vector<ProofReq *> &batch) {
        auto alone = [c, mode](ProofReq *r) {
            r->rc = point_proof_batch_host(c, (KZGProof *)r->out, (Bytes32 *)r->y_out, (const Blob *)r->blob, (const Bytes32 *)r->second, 1,
                                           mode, nullptr);
        };
        const size_t m = batch.size();
        if (m == 1) return alone(batch[0]);
        std::vector<uint8_t> hb(m * (size_t)kBlobBytes), hz(m * 32), ho(m * 48), hy(m * 32);
        for (size_t i = 0; i < m; i++) {
            memcpy(&hb[i * (size_t)kBlobBytes], batch[i]->blob, kBlobBytes);
            memcpy(&hz[32 * i], batch[i]->second, 32);
        }
        C_KZG_RET rc = point_proof_batch_host(c, (KZGProof *)ho.data(), (Bytes32 *)hy.data(), (const Blob *)hb.data(),
                                              (const Bytes32 *)hz.data(), m, mode, nullptr);
        for (size_t i = 0; i < m; i++) {
            if (rc != C_KZG_OK) {
                alone(batch[i]);
                continue;
            }
            memcpy(batch[i]->out, &ho[48 * i], 48);
            memcpy(batch[i]->y_out, &hy[32 * i], 32);
            batch[i]->rc = C_KZG_OK;
        }
    });
}

static C_KZG_RET blob_proof_batch_impl(KZGProof *out, const Blob *blobs, const Bytes48 *commitments, size_t n, const KZGSettings *s,
                                       size_t *first_bad) {
    const int mode = mode_of(s);
    if (!out || !blobs || !commitments) return map_rc(C_KZG_BADARGS, mode);
    Ctx *c = ctx_of(s);
    if (!c) return C_KZG_ERROR;
    ensure_lagrange(c, mode);
    if (n == 1) {  // the reference's symbol: merged with the other callers of the moment
        if (coalesce_singles()) {
            C_KZG_RET rc1 = combine_blob_proof(c, out, blobs, commitments, mode);
            if (rc1 != C_KZG_OK) {
                if (first_bad) *first_bad = 0;
                if (!get_error()[0]) set_error("blob 0 rejected");
            }
            return rc1;
        }
    }
    return blob_proof_batch_host(c, out, blobs, commitments, n, mode, first_bad);
}

// compute_blob_kzg_proof of ONE blob in reference mode on a direct table -- the reference's own call shape (src/lib.rs:361-404) -- with
// nothing on its critical path that need not be there (r06; r05: 0.32-0.35 ms, of which the host's validation of the commitment, 0.2 ms
// of one thread, sat between the enqueue and the wait, LONGER than the GPU's whole chain behind the digest). Here the commitment is
// validated on a thread of its own from the first instruction on; the parse kernel runs beside this thread's hashing (and clears the
// cooperative kernel's hand-off counters on its way); the digest is read by k_z_from_bytes from pinned memory, the sum is stored into
// pinned memory by the MSM's last wave, the redo flag is a pinned word: no copy of a result, no fill, no second-pass launch, no verdict
// word (a reference-mode parse cannot fail; the validation's verdict is this process's own). Anything irregular -- an invalid or
// non-canonically encoded commitment, P = +-Q inside a quad, no pinned memory -- returns kOneBlobFallback and the caller takes the
// general path, which owns the error codes. (An int, not a C_KZG_RET: 100 is not a value of that enumeration, and loading it into one was
// undefined behaviour that the host-UBSan run of the GPU suite caught.) Caller holds c->mu and the workspace.
static const int kOneBlobFallback = 100;   // (not a value of the ABI: internal)
static int blob_proof_one_host(Ctx *c, uint8_t *out48, const uint8_t *blob, const uint8_t *comm48, int mode) {
    if (mode != LWKZG_MODE_REFERENCE || !c->direct_table || !knobs().zero_copy || peer_busy(c)) return kOneBlobFallback;
    const uint32_t ctr_words = direct_one_blob_counter_words(c->direct_bits);
    if (!ctr_words) return kOneBlobFallback;
    if (!c->one_pin && hipHostMalloc((void **)&c->one_pin, 4096, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->one_pin = nullptr;
        return kOneBlobFallback;
    }
    if (ctx_reserve(c, 1) != C_KZG_OK) return kOneBlobFallback;
    Workspace &w = c->ws;
    hipStream_t st = c->stream;
    G1Xyzz29 *p_sum = (G1Xyzz29 *)c->one_pin;
    uint32_t *p_redo = (uint32_t *)(c->one_pin + 256);
    uint8_t *p_dig = c->one_pin + 320;
    *p_redo = 0;
    uint8_t canon[48];
    int vrc = 2;
    SideTask validator([&]() { vrc = host_validate_commitment(comm48, canon); });   // lib.rs:372-375, beside everything below
    LWK_HIP(hipMemcpyAsync(w.blobs, blob, kBlobBytes, hipMemcpyHostToDevice, st));
    (void)coefficients_stage(c, w.blobs, 1, mode, w.status, st, 0, false, w.bucket_start + 1, ctr_words);
    challenge_digests_host(p_dig, blob, comm48, 1);   // on this thread, beside the upload and the parse (assumes canonical commitment bytes)
    launch_z_from_bytes(p_dig, w.z, nullptr, 0, 1, st);   // digest -> Fr, reduced (utils.rs:148-154); read across the link
    quotient_stage(c, mode, w.scalars, w.z, w.scalars2, nullptr, 0, 1, st);
    (void)msm_sums_stage(c, w.scalars2, 1, st, 0, false, quotient_to_msm_form(c, mode, 1, st), p_sum, p_redo);
    validator.join();
    LWK_HIP(hipStreamSynchronize(st));
    if (vrc != 0 || memcmp(canon, comm48, 48) != 0 || *p_redo != 0) return kOneBlobFallback;   // (vrc 1: infinity -- canonical c0 00.. only; keep it simple)
    host_finish_compress(out48, *p_sum);
    return C_KZG_OK;
}

static C_KZG_RET blob_proof_batch_host(Ctx *c, KZGProof *out, const Blob *blobs, const Bytes48 *commitments, size_t n, int mode,
                                       size_t *first_bad) {
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    WsUse wsu(c, c->stream);
    if (n >= kMaxChunk / 2)
        return blob_proofs_sliced(c, (uint8_t *)out, (const uint8_t *)blobs, (const uint8_t *)commitments, n, mode, first_bad);
    if (n == 1) {
        const int r1 = blob_proof_one_host(c, (uint8_t *)out, (const uint8_t *)blobs, (const uint8_t *)commitments, mode);
        if (r1 != kOneBlobFallback) return (C_KZG_RET)r1;
    }
    for (size_t off = 0; off < n; off += kMaxChunk) {
        size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        C_KZG_RET rc = ctx_reserve(c, m);
        if (rc != C_KZG_OK) return rc;
        Workspace &w = c->ws;
        hipStream_t st = c->stream;
        const int le = mode == LWKZG_MODE_CKZG;
        const uint8_t *h_blobs = (const uint8_t *)(blobs + off), *h_comm = (const uint8_t *)(commitments + off);
        // host threads start hashing at once (the blobs are in host memory here; one GPU lane would need ~7 ms per
        // 131 KB message, a core with SHA extensions ~0.1 ms) and run beside the pageable H2D copy, which blocks
        // this thread for a few milliseconds. The digests assume the caller's commitment bytes are the canonical
        // encoding; the validation's re-compression confirms or refutes that below.
        std::vector<uint8_t> h_canon(m * 48), h_dig(m * 32);
        SideTask hasher([&]() { challenge_digests_host(h_dig.data(), h_blobs, h_comm, m); });
        LWK_HIP(hipMemcpyAsync(w.blobs, h_blobs, m * (size_t)kBlobBytes, hipMemcpyHostToDevice, st));
        LWK_HIP(hipMemcpyAsync(w.comm48, h_comm, m * 48, hipMemcpyHostToDevice, st));
        LWK_HIP(hipMemsetAsync(w.status, 0, m * 4, st));
        // Validate the commitments (lib.rs:372-375): a long serial scalar multiplication per point whose verdict and
        // canonical bytes are only needed at the very end. Up to 64 points: on the host threads while the GPU works
        // (~0.2 ms each on the 64-bit host field, against a 2 ms latency-shaped kernel). A batch: on the GPU, on an
        // auxiliary stream beside everything else.
        const bool host_validate = m <= host_small_batch_limit();
        std::vector<int32_t> h_code(m, le ? kStatusBadArgs : kStatusError);
        if (!host_validate) {
            LWK_HIP(hipEventRecord(c->ev_fork, st));
            LWK_HIP(hipStreamWaitEvent(c->vstream, c->ev_fork, 0));
            launch_validate_commitments(w.comm48, w.canon48, w.status, le ? kStatusBadArgs : kStatusError, m, c->vstream, w.val_pts, w.val_kind, w.val_verdict);
            LWK_HIP(hipEventRecord(c->ev_join[0], c->vstream));
        }
        // GPU, main stream: parse the blobs, then the digests as soon as the host threads have them
        coefficients_stage(c, w.blobs, m, mode, w.status, st);
        hasher.join();
        LWK_HIP(hipMemcpyAsync(w.zbytes, h_dig.data(), m * 32, hipMemcpyHostToDevice, st));
        launch_z_from_bytes(w.zbytes, w.z, nullptr, le, m, st);  // digest -> Fr, reduced (utils.rs:148-154)
        quotient_stage(c, mode, w.scalars, w.z, w.scalars2, nullptr, le, m, st);
        const bool hf = m <= host_finish_limit();  // a small call: inversion and compression on this thread, at the end
        const G1Xyzz29 *d_sums = nullptr;
        auto quotient_msm = [&]() {
            const bool lg = quotient_to_msm_form(c, mode, m, st);
            if (hf) d_sums = msm_sums_stage(c, w.scalars2, m, st, 0, false, lg);
            else msm_stages(c, w.scalars2, w.out48, m, st, 0, false, lg);
        };
        quotient_msm();
        if (host_validate) {
            // after the hashing (both want every host thread) and after the GPU has been given everything that needs only the
            // digests: the verdicts are read at the very end, and 0.2 ms of host work per point now runs beside the quotient's MSM
            std::vector<int> vrc(m);
            host_validate_commitments(h_comm, h_canon.data(), vrc.data(), m);
            for (size_t i = 0; i < m; i++)
                if (vrc[i] == 2) LWK_HIP(hipMemcpyAsync(w.status + i, &h_code[i], 4, hipMemcpyHostToDevice, st));
        }
        if (!host_validate) {
            // (a device-to-host copy into pageable memory blocks this thread until the stream has reached it: the
            // canonical bytes are fetched only now that everything else has been submitted)
            LWK_HIP(hipStreamWaitEvent(st, c->ev_join[0], 0));
            LWK_HIP(hipMemcpyAsync(h_canon.data(), w.canon48, m * 48, hipMemcpyDeviceToHost, st));
        }
        LWK_HIP(hipStreamSynchronize(st));
        if (memcmp(h_canon.data(), h_comm, m * 48) != 0) {
            // a non-canonical but valid encoding somewhere in the chunk (or an invalid point, reported through
            // status): redo the chunk with the hash taken over the canonical bytes on the GPU
            if (host_validate) LWK_HIP(hipMemcpyAsync(w.canon48, h_canon.data(), m * 48, hipMemcpyHostToDevice, st));
            launch_challenge(w.blobs, w.canon48, w.z, le, m, st);
            coefficients_stage(c, w.blobs, m, mode, w.status, st);  // (the first attempt's forward transform may have used them as scratch)
            quotient_stage(c, mode, w.scalars, w.z, w.scalars2, nullptr, le, m, st);
            quotient_msm();
        }
        std::vector<uint8_t> h_out(m * 48);
        std::vector<G1Xyzz29> h_sums(hf ? m : 0);
        if (hf) LWK_HIP(hipMemcpyAsync(h_sums.data(), d_sums, m * sizeof(G1Xyzz29), hipMemcpyDeviceToHost, c->stream));
        else LWK_HIP(hipMemcpyAsync(h_out.data(), w.out48, m * 48, hipMemcpyDeviceToHost, c->stream));
        rc = collect_status(c, w.status, m, off, first_bad);
        if (rc != C_KZG_OK) return map_rc(rc, mode);
        for (size_t i = 0; i < h_sums.size(); i++) host_finish_compress(h_out.data() + 48 * i, h_sums[i]);
        memcpy(out + off, h_out.data(), m * 48);
    }
    return C_KZG_OK;
}

static C_KZG_RET point_proof_batch_impl(KZGProof *proofs_out, Bytes32 *ys_out, const Blob *blobs, const Bytes32 *zs, size_t n,
                                        const KZGSettings *s, size_t *first_bad) {
    const int mode = mode_of(s);
    if (!proofs_out || !ys_out || !blobs || !zs) return map_rc(C_KZG_BADARGS, mode);
    Ctx *c = ctx_of(s);
    if (!c) return C_KZG_ERROR;
    ensure_lagrange(c, mode);
    if (n == 1 && coalesce_singles()) {  // the reference's symbol: merged with the other callers of the moment
        C_KZG_RET rc1 = combine_point_proof(c, proofs_out, ys_out, blobs, zs, mode);
        if (rc1 != C_KZG_OK) {
            if (first_bad) *first_bad = 0;
            if (!get_error()[0]) set_error("blob 0 rejected");
        }
        return rc1;
    }
    return point_proof_batch_host(c, proofs_out, ys_out, blobs, zs, n, mode, first_bad);
}

static C_KZG_RET point_proof_batch_host(Ctx *c, KZGProof *proofs_out, Bytes32 *ys_out, const Blob *blobs, const Bytes32 *zs, size_t n,
                                        int mode, size_t *first_bad) {
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    WsUse wsu(c, c->stream);
    if (n >= kMaxChunk / 2)
        return point_proofs_sliced(c, (uint8_t *)proofs_out, (uint8_t *)ys_out, (const uint8_t *)blobs, (const uint8_t *)zs, n, mode,
                                   first_bad);
    for (size_t off = 0; off < n; off += kMaxChunk) {
        size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        C_KZG_RET rc = ctx_reserve(c, m);
        if (rc != C_KZG_OK) return rc;
        Workspace &w = c->ws;
        LWK_HIP(hipMemcpyAsync(w.blobs, blobs + off, m * (size_t)kBlobBytes, hipMemcpyHostToDevice, c->stream));
        LWK_HIP(hipMemcpyAsync(w.zbytes, zs + off, m * 32, hipMemcpyHostToDevice, c->stream));
        const bool hf = m <= host_finish_limit();  // a small call: inversion and compression on this thread
        const G1Xyzz29 *d_sums = nullptr;
        rc = point_proof_batch_device(c, w.out48, w.ybytes, w.blobs, w.zbytes, m, mode, c->stream, w.status, hf ? &d_sums : nullptr);
        if (rc != C_KZG_OK) return rc;
        std::vector<uint8_t> h_out(m * 48), h_y(m * 32);
        std::vector<G1Xyzz29> h_sums(hf ? m : 0);
        if (hf) LWK_HIP(hipMemcpyAsync(h_sums.data(), d_sums, m * sizeof(G1Xyzz29), hipMemcpyDeviceToHost, c->stream));
        else LWK_HIP(hipMemcpyAsync(h_out.data(), w.out48, m * 48, hipMemcpyDeviceToHost, c->stream));
        LWK_HIP(hipMemcpyAsync(h_y.data(), w.ybytes, m * 32, hipMemcpyDeviceToHost, c->stream));
        rc = collect_status(c, w.status, m, off, first_bad);
        if (rc != C_KZG_OK) return map_rc(rc, mode);
        for (size_t i = 0; i < h_sums.size(); i++) host_finish_compress(h_out.data() + 48 * i, h_sums[i]);
        memcpy(proofs_out + off, h_out.data(), m * 48);
        memcpy(ys_out + off, h_y.data(), m * 32);
    }
    return C_KZG_OK;
}

}  // extern "C" (a template cannot have C linkage)
namespace {
template <class F>
C_KZG_RET guarded(const char *what, F &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        lwk::set_error("%s: out of host memory", what);
        return C_KZG_MALLOC;
    } catch (...) {
        lwk::set_error("%s: unexpected exception", what);
        return C_KZG_ERROR;
    }
}
}  // namespace
extern "C" {

C_KZG_RET lwkzg_blob_to_kzg_commitment_batch(KZGCommitment *out, const Blob *blobs, size_t n, const KZGSettings *s, size_t *first_bad) {
    return guarded("lwkzg_blob_to_kzg_commitment_batch", [&] { return commitment_batch_impl(out, blobs, n, s, first_bad); });
}
C_KZG_RET lwkzg_compute_blob_kzg_proof_batch(KZGProof *out, const Blob *blobs, const Bytes48 *commitments, size_t n,
                                             const KZGSettings *s, size_t *first_bad) {
    return guarded("lwkzg_compute_blob_kzg_proof_batch", [&] { return blob_proof_batch_impl(out, blobs, commitments, n, s, first_bad); });
}
C_KZG_RET lwkzg_compute_kzg_proof_batch(KZGProof *proofs_out, Bytes32 *ys_out, const Blob *blobs, const Bytes32 *zs, size_t n,
                                        const KZGSettings *s, size_t *first_bad) {
    return guarded("lwkzg_compute_kzg_proof_batch", [&] { return point_proof_batch_impl(proofs_out, ys_out, blobs, zs, n, s, first_bad); });
}

C_KZG_RET blob_to_kzg_commitment(KZGCommitment *out, const Blob *blob, const KZGSettings *s) {
    return lwkzg_blob_to_kzg_commitment_batch(out, blob, 1, s, nullptr);
}

C_KZG_RET compute_kzg_proof(KZGProof *proof_out, Bytes32 *y_out, const Blob *blob, const Bytes32 *z_bytes,
                            const KZGSettings *s) {
    return lwkzg_compute_kzg_proof_batch(proof_out, y_out, blob, z_bytes, 1, s, nullptr);
}

C_KZG_RET compute_blob_kzg_proof(KZGProof *out, const Blob *blob, const Bytes48 *commitment_bytes, const KZGSettings *s) {
    return lwkzg_compute_blob_kzg_proof_batch(out, blob, commitment_bytes, 1, s, nullptr);
}

// ------------------------------------------------------------------------------------------------
// device-resident entry points

// everything a device-resident call of up to max_batch blobs would otherwise allocate or synchronise for on first use: the workspace,
// the pinned staging of the host-assisted challenge paths (once, at its final size), and -- for settings that answer in c-kzg mode --
// the Lagrange form of the setup
static void host_noop_fn(void *) {}
static C_KZG_RET reserve_ctx(Ctx *c, size_t max_batch) {
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    C_KZG_RET rc = ctx_reserve(c, max_batch);
    if (rc != C_KZG_OK) return rc;
    const size_t host_n = max_batch < mid_proof_host_limit() ? max_batch : mid_proof_host_limit();
    const size_t small_n = max_batch < small_proof_host_limit() ? max_batch : small_proof_host_limit();
    if (host_n || small_n) {
        if (sph_reserve(c, host_n > small_n ? host_n : small_n, true)) {   // (no pinned memory: the calls take the GPU hash)
            SmallProofHost &h = c->sph;   // first touches of the staging happen here, not in the first call
            memset(h.blobs, 0, h.cap * (size_t)kBlobBytes);
            memset(h.dig, 0, h.cap * 32);
        }
        host_pool_warm();   // the host threads exist and have run once
        // the runtime's own first-use costs of a host function on each helper stream (its callback machinery: the first host-assisted
        // call of a process took ~6 ms longer than the second, gpurun_out r05/gpu23) are paid here too
        for (hipStream_t hs : {c->aux[0], c->aux[1], c->vstream})
            if (hipLaunchHostFunc(hs, host_noop_fn, nullptr) != hipSuccess) (void)hipGetLastError();
        for (hipStream_t hs : {c->aux[0], c->aux[1], c->vstream}) (void)hipStreamSynchronize(hs);
    }
    // a caller that announces batches of more than a chunk gets the device-side double buffer of the long host-pointer batches now (256 MiB)
    // instead of inside its first long call; the twin context never runs host-pointer batches
    if (max_batch > kMaxChunk && !c->is_twin) {
        (void)dev_stage_ready(c);
        (void)upload_stream(c);   // (the high-priority stream their uploads run on)
    }
    return C_KZG_OK;
}

C_KZG_RET lwkzg_reserve(const KZGSettings *s, size_t max_batch) {
    Ctx *c = ctx_of(s);
    if (!c) return C_KZG_ERROR;
    ensure_lagrange(c, mode_of(s));
    C_KZG_RET rc = reserve_ctx(c, max_batch);
    if (rc == C_KZG_OK) c->reserved.store(true, std::memory_order_release);
    return rc;
}

// the same for a caller that will issue device-resident calls on `caller_streams` streams at once: with two or more the
// settings' second context (own streams and workspace over the same tables, pick_ctx) is created and reserved HERE -- workspace and
// pinned staging both --, so that the first overlapped call neither allocates nor synchronises the device
C_KZG_RET lwkzg_reserve_streams(const KZGSettings *s, size_t max_batch, int caller_streams) {
    C_KZG_RET rc = lwkzg_reserve(s, max_batch);
    if (rc != C_KZG_OK || caller_streams < 2 || twin_off()) return rc;  // LWKZG_TWIN=0: pick_ctx never uses a twin
    Ctx *c = ctx_of(s);
    if (!c) return C_KZG_ERROR;
    Ctx *t = nullptr;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        t = c->twin.load(std::memory_order_acquire);
        if (!t) {
            if ((rc = ctx_new(&t, c)) != C_KZG_OK) return rc;
            c->twin.store(t, std::memory_order_release);
        }
    }
    return reserve_ctx(t, max_batch);
}

// ---- the two forms of the setup and their direct tables --------------------------------------------------------------------------
//
// A settings object has its MSM tables in up to two forms: monomial ([tau^i]G: what reference mode commits over, and what a
// quotient in coefficient form needs) and Lagrange ([l_i(tau)]G: what a c-kzg blob's evaluations commit over with no transform).
// The 9 MB bucket tables exist in both forms as soon as c-kzg mode is first used; the DIRECT table (7 .. 275 GB) is built
//   * in the form of the mode the settings answer in when it is built (lwkzg_enable_direct_table, the load's own choice) or are
//     switched to (lwkzg_settings_set_mode) -- the primary form --,
//   * and in the other form too whenever that fits beside it with kDirectAlignedHeadroom to spare (always up to 14 bits on an empty
//     MI355X; never at 15 / 16 bits) and is of use (the Lagrange form only once c-kzg mode has been used).
// A call never depends on which exist: a c-kzg commitment takes the Lagrange table when there is one, else the transform and the
// monomial table, else Lagrange buckets; a proof's quotient (coefficient form) takes the monomial table when there is one, else one
// forward transform and the Lagrange table; reference mode on a Lagrange-only table falls back to the monomial buckets.

// the twin context holds copies of every table address: refresh them (caller holds both contexts' locks)
static void sync_twin_tables(Ctx *c) {
    Ctx *t = c->twin.load(std::memory_order_acquire);
    if (!t) return;
    t->direct_table = c->direct_table;
    t->direct_tab.win_dev = c->direct_tab.win_dev;
    t->direct_bits = c->direct_bits;
    t->direct_row_bytes = c->direct_row_bytes;
    t->lag = c->lag;
    t->lag.direct_tab = DirectTable();
    t->lag.direct_tab.win_dev = c->lag.direct_tab.win_dev;
    t->lag_ready.store(c->lag.ready, std::memory_order_release);
}

struct FormRef {  // one form's direct-table fields
    const G1Affine *points;
    DirectTable &tab;
    G1Affine29 *&table0;
    int &bits;
    size_t &row;
};
static FormRef form_ref(Ctx *c, bool lagrange) {
    if (lagrange) return {c->lag.points, c->lag.direct_tab, c->lag.direct_table, c->lag.direct_bits, c->lag.direct_row_bytes};
    return {c->points, c->direct_tab, c->direct_table, c->direct_bits, c->direct_row_bytes};
}
static void form_drop(Ctx *c, bool lagrange) {
    FormRef f = form_ref(c, lagrange);
    free_direct_table(f.tab);
    f.table0 = nullptr;
    f.bits = 0;
}

// one form's direct table of `bits`: rows aligned to 128-byte lines when that table leaves kDirectAlignedHeadroom of HBM free
// (workspaces, the caller's own buffers), packed otherwise; LWKZG_DIRECT_ROW=112|128 or row_pref forces one (A/B runs).
// need_headroom: 1 = build only if even the packed table leaves the headroom (a secondary form somebody asked for must not crowd the
// device); 2 = only if it takes at most a quarter of what is free now (the rule the load picks its own table by: a second table
// nobody asked for -- the first c-kzg call of a settings object that follows the process default -- gets no more than the first did).
static hipError_t form_build(Ctx *c, bool lagrange, int bits, size_t row_pref, int need_headroom, BuildTiming &bt) {
    FormRef f = form_ref(c, lagrange);
    auto build_rows = [&](size_t row) -> hipError_t {
        double ms[4] = {0, 0, 0, 0};
        const hipError_t e = build_direct_table(bits, f.points, f.tab, row, c->stream, ms);
        bt.scratch_malloc_ms += ms[0];
        bt.table_malloc_ms += ms[1];
        bt.kernels_ms += ms[2];
        bt.scratch_free_ms += ms[3];
        if (e != hipSuccess) {
            (void)hipGetLastError();  // an out-of-memory here is an answer, not a sticky failure
            return e;
        }
        f.table0 = (G1Affine29 *)f.tab.win[0];
        f.bits = bits;
        f.row = row;
        bt.bits = bits;
        bt.row_bytes = row;
        bt.table_bytes += direct_table_entries(bits) * row;
        return hipSuccess;
    };
    const int forced_env = knobs().direct_row;
    const int forced = forced_env ? forced_env : (int)row_pref;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) (void)hipGetLastError();
    if (need_headroom && direct_table_entries(bits) * kDirectRowPacked + kDirectAlignedHeadroom > free_b) return hipErrorOutOfMemory;
    if (need_headroom == 2 && direct_table_entries(bits) * kDirectRowPacked > free_b / 4) return hipErrorOutOfMemory;
    const bool aligned_fits = direct_table_entries(bits) * kDirectRowAligned + kDirectAlignedHeadroom <= free_b;
    if (forced == (int)kDirectRowAligned || (forced != (int)kDirectRowPacked && aligned_fits))
        if (build_rows(kDirectRowAligned) == hipSuccess) return hipSuccess;
    return build_rows(kDirectRowPacked);
}

// The Lagrange form of the setup, derived on the device: the 4096 rows of inverse-DFT coefficients (k_idft_columns) committed by
// the monomial engine as it stands, the results decompressed into affine points, the bucket engine's table built over them.
// About 50 ms on the default engine. Caller holds the primary context's lock and its twin's.
static C_KZG_RET lagrange_prepare(Ctx *c) {
    if (c->lag.ready) return C_KZG_OK;
    if (c->lag_failed) return C_KZG_MALLOC;
    LWK_HIP(hipSetDevice(c->device));
    C_KZG_RET rc = ctx_reserve(c, kMaxChunk);
    if (rc != C_KZG_OK) return rc;
    Workspace &w = c->ws;
    hipStream_t st = c->stream;
    WsUse wsu(c, st);
    uint8_t *d_comp = nullptr;
    int32_t *d_status = nullptr;
    bool ok = hipMalloc((void **)&c->lag.points, (size_t)kBlobElems * sizeof(G1Affine)) == hipSuccess &&
              hipMalloc((void **)&c->lag.table, (size_t)kTablePoints * sizeof(G1Affine29)) == hipSuccess &&
              hipMalloc((void **)&d_comp, (size_t)kBlobElems * 48) == hipSuccess && hipMalloc((void **)&d_status, (size_t)kBlobElems * 4) == hipSuccess;
    std::vector<int32_t> h_status(kBlobElems, 0);
    if (ok) {
        for (size_t off = 0; off < (size_t)kBlobElems; off += kMaxChunk) {
            launch_idft_columns(w.scalars, c->tw_inv, (uint32_t)off, kMaxChunk, st);
            msm_stages(c, w.scalars, d_comp + 48 * off, kMaxChunk, st);   // over the monomial form, on whatever engine it has
        }
        launch_g1_decompress(d_comp, c->lag.points, d_status, kBlobElems, 0, st);  // (our own sums: in the subgroup by construction)
        launch_build_table(c->lag.points, c->lag.table, st);
        ok = hipMemcpyAsync(h_status.data(), d_status, (size_t)kBlobElems * 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    }
    if (d_comp) hipFree(d_comp);
    if (d_status) hipFree(d_status);
    for (int i = 0; ok && i < kBlobElems; i++)
        if (h_status[i] != 0) ok = false;  // (an l_i(tau) G at infinity: tau would have to be a root of l_i, i.e. a domain point)
    if (!ok) {
        (void)hipGetLastError();
        dev_free(c->lag.points);
        dev_free(c->lag.table);
        c->lag_failed = true;
        set_error("the Lagrange form of the setup could not be derived (out of device memory, or tau is a 4096th root of unity): c-kzg mode stays on the transform");
        return C_KZG_MALLOC;
    }
    c->lag.ready = true;
    c->lag_ready.store(true, std::memory_order_release);
    return C_KZG_OK;
}

// Bring the tables to what mode `mode` wants (see the block comment above). `may_swap`: when the mode's form has no direct table,
// the other has one and the two do not fit side by side, move the table over (explicit requests: lwkzg_settings_set_mode,
// lwkzg_enable_direct_table); lazily (first c-kzg call of a settings object that follows the process-wide default) only the
// free-of-charge secondary build is tried. Caller holds the primary context's lock and its twin's; the device is idle.
static void tables_follow_mode(Ctx *c, int mode, bool may_swap) {
    const bool want_lag = mode == LWKZG_MODE_CKZG;
    if (want_lag && lagrange_prepare(c) != C_KZG_OK) return;
    const int bits = c->direct_bits ? c->direct_bits : c->lag.direct_bits;
    if (!bits) return;                                     // bucket engine: both forms have their 9 MB tables
    FormRef mine = form_ref(c, want_lag);
    if (mine.bits == bits) return;
    BuildTiming bt;
    if (form_build(c, want_lag, bits, 0, may_swap ? 1 : 2, bt) == hipSuccess) return;  // fits beside the other: both forms live
    if (!may_swap) return;
    // The two forms' tables have the SAME geometry (width, windows, row size): the other form's allocations move over as they are and
    // the build kernels run again over them -- no hipFree / hipMalloc of 275 GB, no wait for the driver's scrub (7-8.5 s ->
    // the kernels' half second). LWKZG_SET_MODE_IN_PLACE=0: free and allocate, the A/B arm.
    const bool in_place_on = knobs().set_mode_in_place;
    FormRef other = form_ref(c, !want_lag);
    if (in_place_on && other.bits == bits && other.table0) {
        auto wall = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = wall();
        const size_t row = other.row;
        mine.tab = other.tab;
        other.tab = DirectTable();
        other.table0 = nullptr;
        other.bits = 0;
        double ms[4] = {0, 0, 0, 0};
        if (build_direct_table(bits, mine.points, mine.tab, row, c->stream, ms, true) == hipSuccess) {
            mine.table0 = (G1Affine29 *)mine.tab.win[0];
            mine.bits = bits;
            mine.row = row;
            bt.bits = bits;
            bt.row_bytes = row;
            bt.table_bytes = direct_table_entries(bits) * row;
            bt.scratch_malloc_ms = ms[0];
            bt.kernels_ms = ms[2];
            bt.scratch_free_ms = ms[3];
            bt.total_ms = wall() - t0;
            bt.in_place = true;
            c->last_build = bt;
            return;
        }
        (void)hipGetLastError();   // (the failed build freed what it held: fall through to a build from nothing)
        mine.table0 = nullptr;
        mine.bits = 0;
    } else {
        form_drop(c, !want_lag);
    }
    if (form_build(c, want_lag, bits, 0, 0, bt) != hipSuccess) (void)form_build(c, !want_lag, bits, 0, 0, bt);  // (cannot happen: the other form's table just left)
    c->last_build = bt;
}

// Called by every compute entry point before it takes its context: a settings object answering in c-kzg mode gets the Lagrange form
// on first use (and a Lagrange direct table if one fits beside the monomial one). One atomic load afterwards.
static void ensure_lagrange(Ctx *c, int mode) {
    if (mode != LWKZG_MODE_CKZG || c->lag_ready.load(std::memory_order_acquire)) return;
    c = c->primary;
    std::lock_guard<std::mutex> lk(c->mu);
    std::unique_lock<std::mutex> lk_twin;
    Ctx *const twin = c->twin.load(std::memory_order_acquire);
    if (twin) lk_twin = std::unique_lock<std::mutex>(twin->mu);
    if (c->lag.ready || c->lag_failed) return;
    if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return;
    tables_follow_mode(c, mode, false);
    hipDeviceSynchronize();
    sync_twin_tables(c);
}

static void settings_follow_mode(Ctx *c, int mode) {
    c = c->primary;
    std::lock_guard<std::mutex> lk(c->mu);
    std::unique_lock<std::mutex> lk_twin;
    Ctx *const twin = c->twin.load(std::memory_order_acquire);
    if (twin) lk_twin = std::unique_lock<std::mutex>(twin->mu);
    if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return;
    c->lag_failed = false;   // an explicit request retries a derivation that once ran out of memory (only the lazy path stays off: ADVICE r04)
    tables_follow_mode(c, mode, true);
    hipDeviceSynchronize();
    sync_twin_tables(c);
}

// row_pref: 0 = rows aligned to 128-byte lines when that leaves headroom on the device, else packed; or one of the two
// forms: 0 = by the settings' mode (the rule above); 1 / 2 / 3 = exactly the monomial / the Lagrange / both forms (lwkzg_enable_direct_table_forms)
static C_KZG_RET enable_direct_table(const KZGSettings *s, int window_bits, size_t row_pref, int forms = 0) {
    Ctx *c = ctx_of(s);
    if (!c) return C_KZG_ERROR;
    if (forms < 0 || forms > 3) {
        set_error("lwkzg_enable_direct_table_forms: forms must be 1 (monomial), 2 (Lagrange) or 3 (both)");
        return C_KZG_BADARGS;
    }
    const int mode = forms == 2 || forms == 3 ? LWKZG_MODE_CKZG : forms == 1 ? LWKZG_MODE_REFERENCE : mode_of(s);
    std::lock_guard<std::mutex> lk(c->mu);
    // the twin context launches against the same table under its own lock: keep it out as well (lock order: main, twin)
    std::unique_lock<std::mutex> lk_twin;
    Ctx *const twin = c->twin.load(std::memory_order_acquire);
    if (twin) lk_twin = std::unique_lock<std::mutex>(twin->mu);
    LWK_HIP(hipSetDevice(c->device));
    LWK_HIP(hipDeviceSynchronize());  // the table may be in use on any stream, the callers' included
    const int old_bits = c->direct_bits ? c->direct_bits : c->lag.direct_bits;
    const int old_forms = (c->direct_table ? 1 : 0) | (c->lag.direct_table ? 2 : 0);
    if (window_bits == old_bits && (forms == 0 || forms == old_forms || window_bits == 0)) return C_KZG_OK;
    if (window_bits != 0 && direct_table_entries(window_bits) == 0) {
        set_error("lwkzg_enable_direct_table: window_bits must be 0 or 10 .. 16 (got %d)", window_bits);
        return C_KZG_BADARGS;
    }
    auto wall = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = wall();
    // the primary form: the one the settings' mode commits over (Lagrange only if it can be had)
    c->lag_failed = false;   // (an explicit request retries a derivation that once ran out of memory)
    const bool primary_lag = mode == LWKZG_MODE_CKZG && lagrange_prepare(c) == C_KZG_OK;
    if (forms >= 2 && !primary_lag) return C_KZG_MALLOC;  // (lagrange_prepare has said why)
    form_drop(c, false);  // (the old and the new tables need not fit side by side)
    form_drop(c, true);
    BuildTiming bt;
    bt.free_old_ms = wall() - t_start;
    sync_twin_tables(c);
    if (window_bits == 0) return C_KZG_OK;
    hipError_t e = form_build(c, primary_lag, window_bits, row_pref, 0, bt);
    if (e == hipSuccess && forms == 3) e = form_build(c, false, window_bits, row_pref, 0, bt);
    if (e == hipSuccess && forms == 0 && (primary_lag || c->lag.ready)) (void)form_build(c, !primary_lag, window_bits, row_pref, 1, bt);  // the other form, if it fits beside
    if (e != hipSuccess) {
        form_drop(c, false);
        form_drop(c, true);
    }
    bt.total_ms = wall() - t_start;
    c->last_build = bt;
    if (e != hipSuccess) {
        if (old_bits) {  // the engine the settings had stays in place (the form it was in first)
            BuildTiming scratch;
            const bool first_lag = old_forms == 2 || (old_forms == 3 && primary_lag);
            if (form_build(c, first_lag, old_bits, row_pref, 0, scratch) == hipSuccess && old_forms == 3)
                (void)form_build(c, !first_lag, old_bits, row_pref, 1, scratch);
        }
        sync_twin_tables(c);
        set_error("lwkzg_enable_direct_table(%d): %zu bytes: %s", window_bits, direct_table_entries(window_bits) * kDirectRowPacked,
                  hipGetErrorString(e));
        return C_KZG_MALLOC;
    }
    sync_twin_tables(c);
    return C_KZG_OK;
}

C_KZG_RET lwkzg_enable_direct_table(const KZGSettings *s, int window_bits) { return enable_direct_table(s, window_bits, 0); }
C_KZG_RET lwkzg_enable_direct_table_forms(const KZGSettings *s, int window_bits, int forms) {
    if (forms < 1 || forms > 3) {
        set_error("lwkzg_enable_direct_table_forms: forms must be 1 (monomial), 2 (Lagrange) or 3 (both)");
        return C_KZG_BADARGS;
    }
    return enable_direct_table(s, window_bits, 0, forms);
}

int lwkzg_direct_table_bits(const KZGSettings *s) {
    Ctx *c = ctx_of(s);
    return !c ? -1 : c->direct_bits ? c->direct_bits : c->lag.direct_bits;
}

// bit 0: a direct table over the monomial form is live; bit 1: one over the Lagrange form (c-kzg mode without the transform)
int lwkzg_direct_table_forms(const KZGSettings *s) {
    Ctx *c = ctx_of(s);
    return !c ? -1 : (c->direct_table ? 1 : 0) | (c->lag.direct_table ? 2 : 0);
}

int lwkzg_direct_num_windows(int window_bits) { return direct_num_windows(window_bits); }

int lwkzg_direct_row_bytes(const KZGSettings *s) {
    Ctx *c = ctx_of(s);
    return !c ? -1 : c->direct_table ? (int)c->direct_row_bytes : c->lag.direct_table ? (int)c->lag.direct_row_bytes : 0;
}

C_KZG_RET lwkzg_blob_to_kzg_commitment_batch_device(void *out48_dev, const void *blobs_dev, size_t n, const KZGSettings *s,
                                                    void *stream, int32_t *status_dev) {
    Ctx *c = ctx_of(s);
    if (!c) return C_KZG_ERROR;
    const int mode = mode_of(s);
    ensure_lagrange(c, mode);
    c = pick_ctx(c, (hipStream_t)stream);
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    WsUse wsu(c, st);
    return commit_batch_device(c, (uint8_t *)out48_dev, (const uint8_t *)blobs_dev, n, mode, st, status_dev);
}

C_KZG_RET lwkzg_compute_blob_kzg_proof_batch_device(void *out48_dev, const void *blobs_dev, const void *commitments48_dev,
                                                    size_t n, const KZGSettings *s, void *stream, int32_t *status_dev) {
    Ctx *c = ctx_of(s);
    if (!c) return C_KZG_ERROR;
    const int mode = mode_of(s);
    ensure_lagrange(c, mode);
    c = pick_ctx(c, (hipStream_t)stream);
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    WsUse wsu(c, st);
    return blob_proof_batch_device(c, (uint8_t *)out48_dev, (const uint8_t *)blobs_dev, (const uint8_t *)commitments48_dev,
                                   n, mode, st, status_dev);
}

C_KZG_RET lwkzg_commit_and_prove_batch_device(void *commitments48_dev, void *proofs48_dev, const void *blobs_dev, size_t n,
                                              const KZGSettings *s, void *stream, int32_t *status_dev) {
    if (!commitments48_dev || !proofs48_dev || !blobs_dev) return map_rc(C_KZG_BADARGS, mode_of(s));
    Ctx *c = ctx_of(s);
    if (!c) return C_KZG_ERROR;
    if (n == 0) return C_KZG_OK;
    const int mode = mode_of(s);
    ensure_lagrange(c, mode);
    c = pick_ctx(c, (hipStream_t)stream);
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    WsUse wsu(c, st);
    return commit_and_prove_batch_device(c, (uint8_t *)commitments48_dev, (uint8_t *)proofs48_dev, (const uint8_t *)blobs_dev, n,
                                         mode, st, status_dev);
}

// z_i = compute_challenge(blob_i, commitment_i) (src/utils.rs:120-154) for device-resident blobs, as 32 bytes in the
// mode's byte order (canonical, reduced mod r): the Fiat-Shamir kernel of the proof path, exposed so that a test can
// put its output next to hashlib's at any batch size. The commitment bytes are hashed as given (the proof path
// hashes the canonical re-compression; for canonical inputs the two are the same bytes).
C_KZG_RET lwkzg_compute_challenges_device(void *z32_dev, const void *blobs_dev, const void *commitments48_dev, size_t n,
                                          const KZGSettings *s, void *stream) {
    if (!z32_dev || !blobs_dev || !commitments48_dev) return C_KZG_BADARGS;
    Ctx *c = ctx_of(s);
    if (!c) return C_KZG_ERROR;
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    WsUse wsu(c, st);
    C_KZG_RET rc = ctx_reserve(c, n);
    if (rc != C_KZG_OK) return rc;
    if (n > kMaxChunk && (rc = ws_long_reserve(c, n)) != C_KZG_OK) return rc;
    const int le = mode_of(s) == LWKZG_MODE_CKZG;
    Fr *z = n > kMaxChunk ? c->ws.z_long : c->ws.z;
    launch_challenge((const uint8_t *)blobs_dev, (const uint8_t *)commitments48_dev, z, le, n, st);
    launch_fr_mont_to_bytes(z, (uint8_t *)z32_dev, le, n, st);
    return C_KZG_OK;
}

C_KZG_RET lwkzg_g1_lincomb_setup_device(void *out48_dev, const void *scalars_be_dev, size_t n_msm, const KZGSettings *s,
                                        void *stream) {
    Ctx *c = ctx_of(s);
    if (!c) return C_KZG_ERROR;
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    WsUse wsu(c, st);
    C_KZG_RET rc = ctx_reserve(c, n_msm);
    if (rc != C_KZG_OK) return rc;
    for (size_t off = 0; off < n_msm; off += kMaxChunk) {
        size_t m = n_msm - off < kMaxChunk ? n_msm - off : kMaxChunk;
        launch_parse_be_reduce((const uint8_t *)scalars_be_dev + off * (size_t)kBlobBytes, c->ws.scalars, m * kBlobElems, st);
        msm_scalars_raw_device(c, (uint8_t *)out48_dev + 48 * off, c->ws.scalars, m, st);
    }
    return C_KZG_OK;
}

// sum_k s_k * g1[k mod 4096] for n_terms = tiles * 4096 scalars (BASELINE config "2^20-point MSM, tiled
// trusted setup"): one 4096-term fixed-base MSM per tile, then one sum of the tile results.
C_KZG_RET lwkzg_g1_msm_tiled_device(void *out48_dev, const void *scalars_be_dev, size_t n_terms, const KZGSettings *s,
                                    void *stream) {
    if (n_terms == 0 || n_terms % kBlobElems != 0) {
        set_error("lwkzg_g1_msm_tiled_device: n_terms must be a positive multiple of 4096");
        return C_KZG_BADARGS;
    }
    Ctx *c = ctx_of(s);
    if (!c) return C_KZG_ERROR;
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    WsUse wsu(c, st);
    const size_t tiles = n_terms / kBlobElems;
    C_KZG_RET rc = ctx_reserve(c, tiles);
    if (rc != C_KZG_OK) return rc;
    Workspace &w = c->ws;
    G1Xyzz29 *total = w.sums + w.cap;  // the extra slot behind the per-tile sums
    for (size_t off = 0; off < tiles; off += kMaxChunk) {
        size_t m = tiles - off < kMaxChunk ? tiles - off : kMaxChunk;
        launch_parse_be_reduce((const uint8_t *)scalars_be_dev + off * (size_t)kBlobBytes, w.scalars, m * kBlobElems, st);
        launch_sum_points(msm_sums_stage(c, w.scalars, m, st), m, total, off != 0, st);
    }
    launch_finalize_compress(total, (uint8_t *)out48_dev, 1, st);
    return C_KZG_OK;
}

C_KZG_RET lwkzg_fr_ntt4096_device(void *out_dev, const void *in_dev, size_t n, int inverse, const KZGSettings *s,
                                  void *stream) {
    Ctx *c = ctx_of(s);
    if (!c) return C_KZG_ERROR;
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    WsUse wsu(c, st);
    C_KZG_RET rc = ctx_reserve(c, n);
    if (rc != C_KZG_OK) return rc;
    Workspace &w = c->ws;
    for (size_t off = 0; off < n; off += kMaxChunk) {
        size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        const uint8_t *src = (const uint8_t *)in_dev + off * (size_t)kBlobBytes;
        uint8_t *dst = (uint8_t *)out_dev + off * (size_t)kBlobBytes;
        launch_fr_be_to_mont(src, (Fr *)w.scalars2, m * kBlobElems, st);
        launch_bitrev_permute((const Fr *)w.scalars2, w.fr, m, st);  // natural order in -> DIT wants bit-reversed
        if (inverse) {
            launch_ntt4096(w.fr, (Fr *)w.scalars, c->tw28_inv, 1, m, st);  // scaled by 4096^-1, canonical limbs out
            launch_raw_to_be(w.scalars, dst, m * kBlobElems, st);
        } else {
            launch_ntt4096(w.fr, (Fr *)w.scalars2, c->tw28_fwd, 0, m, st);
            launch_fr_mont_to_be((const Fr *)w.scalars2, dst, m * kBlobElems, st);
        }
    }
    return C_KZG_OK;
}

// ------------------------------------------------------------------------------------------------
// multi-GPU setup hand-off: [hdr 64][g1_values 589,824][g2_values 18,720 -> padded][table][tw_fwd][tw_inv][points]

static constexpr size_t kImgHdr = 64;
static constexpr size_t kImgG1 = (size_t)kBlobElems * 144;
static constexpr size_t kImgG2 = ((size_t)TRUSTED_SETUP_NUM_G2_POINTS * 288 + 63) / 64 * 64;
static constexpr size_t kImgTable = (size_t)kTablePoints * sizeof(G1Affine29);
static constexpr size_t kImgTw = (size_t)(kBlobElems / 2) * sizeof(Fr);
static constexpr size_t kImgPoints = (size_t)kBlobElems * sizeof(G1Affine);
static constexpr size_t kImgBytes = kImgHdr + kImgG1 + kImgG2 + kImgTable + 2 * kImgTw + kImgPoints;

size_t lwkzg_setup_image_bytes(void) { return kImgBytes; }

C_KZG_RET lwkzg_setup_export_device(const KZGSettings *s, void *image_dev, void *stream) {
    Ctx *c = ctx_of(s);
    if (!c || !image_dev) return C_KZG_ERROR;
    std::lock_guard<std::mutex> lk(c->mu);
    LWK_HIP(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    uint8_t *img = (uint8_t *)image_dev;
    uint64_t hdr[8] = {kCtxMagic, kImgBytes, (uint64_t)kWindowBits, (uint64_t)kNumWindows, (uint64_t)P29::W, 0, 0, 0};
    LWK_HIP(hipMemcpyAsync(img, hdr, sizeof hdr, hipMemcpyHostToDevice, st));
    LWK_HIP(hipMemcpyAsync(img + kImgHdr, s->g1_values, kImgG1, hipMemcpyHostToDevice, st));
    LWK_HIP(hipMemsetAsync(img + kImgHdr + kImgG1, 0, kImgG2, st));
    LWK_HIP(hipMemcpyAsync(img + kImgHdr + kImgG1, s->g2_values, (size_t)TRUSTED_SETUP_NUM_G2_POINTS * 288, hipMemcpyHostToDevice, st));
    LWK_HIP(hipMemcpyAsync(img + kImgHdr + kImgG1 + kImgG2, c->table, kImgTable, hipMemcpyDeviceToDevice, st));
    LWK_HIP(hipMemcpyAsync(img + kImgHdr + kImgG1 + kImgG2 + kImgTable, c->tw_fwd, kImgTw, hipMemcpyDeviceToDevice, st));
    LWK_HIP(hipMemcpyAsync(img + kImgHdr + kImgG1 + kImgG2 + kImgTable + kImgTw, c->tw_inv, kImgTw, hipMemcpyDeviceToDevice, st));
    LWK_HIP(hipMemcpyAsync(img + kImgHdr + kImgG1 + kImgG2 + kImgTable + 2 * kImgTw, c->points, kImgPoints, hipMemcpyDeviceToDevice, st));
    LWK_HIP(hipStreamSynchronize(st));  // the host sources above must stay valid until the copies ran
    return C_KZG_OK;
}

C_KZG_RET lwkzg_setup_import_device(KZGSettings *out, const void *image_dev) {
    if (!out || !image_dev) return C_KZG_BADARGS;
    Ctx *c = nullptr;
    C_KZG_RET rc = ctx_new(&c);
    if (rc != C_KZG_OK) return rc;
    const uint8_t *img = (const uint8_t *)image_dev;
    uint64_t hdr[8];
    g1_t *g1v = (g1_t *)malloc(kImgG1);
    g2_t *g2v = (g2_t *)malloc((size_t)TRUSTED_SETUP_NUM_G2_POINTS * 288);
    bool ok = g1v && g2v && hipMemcpy(hdr, img, sizeof hdr, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && (hdr[0] != kCtxMagic || hdr[1] != kImgBytes || hdr[2] != (uint64_t)kWindowBits || hdr[3] != (uint64_t)kNumWindows ||
               hdr[4] != (uint64_t)P29::W)) {  // the table's limb width is part of the format
        set_error("setup image header mismatch (different build or not an image)");
        ok = false;
    }
    ok = ok && hipMemcpy(g1v, img + kImgHdr, kImgG1, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(g2v, img + kImgHdr + kImgG1, (size_t)TRUSTED_SETUP_NUM_G2_POINTS * 288, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(c->table, img + kImgHdr + kImgG1 + kImgG2, kImgTable, hipMemcpyDeviceToDevice) == hipSuccess &&
         hipMemcpy(c->points, img + kImgHdr + kImgG1 + kImgG2 + kImgTable + 2 * kImgTw, kImgPoints, hipMemcpyDeviceToDevice) == hipSuccess;
    if (ok) ok = ctx_finish_fft(c) == C_KZG_OK;
    if (!ok) {
        if (!get_error()[0]) set_error("setup image import failed");
        free(g1v);
        free(g2v);
        ctx_destroy(c);
        return C_KZG_ERROR;
    }
    out->fs = &c->fs;
    out->g1_values = g1v;
    out->g2_values = g2v;
    direct_from_env(out);
    return C_KZG_OK;
}

}  // extern "C"

namespace lwk {
// Which MSM engine a freshly loaded setup gets. A consumer that only knows the reference's nine symbols never calls
// lwkzg_enable_direct_table, so the choice is made here, at the end of every load:
//   LWKZG_DIRECT_BITS unset : the DEFAULT engine = the widest direct table of 13 .. 10 bit windows that takes at most a
//                             quarter of the device memory that is free right now (13 bits = 41 GB on an empty MI355X:
//                             20 additions per scalar, no sort, no buckets, no reduction); the bucket engine when
//                             even the 10-bit table (6 GB) does not pass that test, e.g. beside another process's table
//   LWKZG_DIRECT_BITS=0     : the bucket engine (9 MB table), whatever is free
//   LWKZG_DIRECT_BITS=10..16: that width, or the bucket engine when it does not fit
//   LWKZG_DIRECT_BITS=auto  : the widest of 16 .. 10 that fits at all (what bench.py's headline asks for explicitly)
void direct_from_env(const KZGSettings *s) {
    if (knobs().has_direct_bits) {
        if (knobs().direct_bits < 0) {   // "auto"
            for (int bits = kDirectMaxBits; bits >= kDirectMinBits; bits--)
                if (lwkzg_enable_direct_table(s, bits) == C_KZG_OK) return;
            return;
        }
        if (knobs().direct_bits != 0) (void)lwkzg_enable_direct_table(s, knobs().direct_bits);
        return;
    }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    // the widest of 13 .. 10 bits within a quarter of the free memory: with its rows in 128-byte lines if that fits the
    // quarter too, packed if only that does
    for (int bits = 13; bits >= kDirectMinBits; bits--) {
        const size_t rows = direct_table_entries(bits);
        const size_t pref = rows * kDirectRowAligned <= free_b / 4 ? kDirectRowAligned : rows * kDirectRowPacked <= free_b / 4 ? kDirectRowPacked : 0;
        if (pref && enable_direct_table(s, bits, pref) == C_KZG_OK) return;
    }
}
}  // namespace lwk
