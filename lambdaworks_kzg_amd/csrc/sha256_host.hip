// sha256_host.hip -- Fiat-Shamir hashing on the host for the host-pointer proof entry points.
//
// compute_challenge (/root/reference/src/utils.rs:120-154) hashes 131,152 bytes per blob. SHA-256 is
// strictly sequential per message: one GPU lane needs ~7 ms for it (k_challenge, issue-bound), a CPU
// core with SHA extensions ~0.1 ms. When the blobs are in host memory anyway (the reference's C ABI),
// the digests are computed here, one std::thread per slice of the batch, while the GPU validates the
// commitments and parses the blobs; the device-resident entry points keep using k_challenge.
#include <chrono>
#include <atomic>
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "plan.h"
#include "knobs.h"

namespace lwk {

void sha256_host(uint8_t out[32], const uint8_t *msg, size_t len);  // portable, sha256.hip
void sha256_blocks_portable(uint32_t h[8], const uint8_t *blocks, size_t n_blocks);  // portable compression of whole blocks, sha256.hip

namespace {

const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// SHA extensions: state kept as (ABEF, CDGH), four rounds per _mm_sha256rnds2_epu32 pair
__attribute__((target("sha,sse4.1,ssse3"))) void compress_shani(uint32_t state[8], const uint8_t *data, size_t nblocks) {
    const __m128i shuf = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i tmp = _mm_loadu_si128((const __m128i *)&state[0]);
    __m128i st1 = _mm_loadu_si128((const __m128i *)&state[4]);
    tmp = _mm_shuffle_epi32(tmp, 0xB1);       // CDAB
    st1 = _mm_shuffle_epi32(st1, 0x1B);       // EFGH
    __m128i st0 = _mm_alignr_epi8(tmp, st1, 8);   // ABEF
    st1 = _mm_blend_epi16(st1, tmp, 0xF0);        // CDGH
    while (nblocks--) {
        __m128i save0 = st0, save1 = st1;
        __m128i m0 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(data + 0)), shuf);
        __m128i m1 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(data + 16)), shuf);
        __m128i m2 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(data + 32)), shuf);
        __m128i m3 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(data + 48)), shuf);
        __m128i msg;
#define RND4(mk, k)                                                              \
    msg = _mm_add_epi32(mk, _mm_loadu_si128((const __m128i *)&K256[k]));         \
    st1 = _mm_sha256rnds2_epu32(st1, st0, msg);                                  \
    msg = _mm_shuffle_epi32(msg, 0x0E);                                          \
    st0 = _mm_sha256rnds2_epu32(st0, st1, msg);
#define SCHED(a, b, c, d) /* a = next four words from a, b, c, d (oldest .. newest) */ \
    a = _mm_sha256msg1_epu32(a, b);                                              \
    a = _mm_add_epi32(a, _mm_alignr_epi8(d, c, 4));                              \
    a = _mm_sha256msg2_epu32(a, d);
        RND4(m0, 0)
        RND4(m1, 4)
        RND4(m2, 8)
        RND4(m3, 12)
        for (int k = 16; k < 64; k += 16) {
            SCHED(m0, m1, m2, m3)
            RND4(m0, k)
            SCHED(m1, m2, m3, m0)
            RND4(m1, k + 4)
            SCHED(m2, m3, m0, m1)
            RND4(m2, k + 8)
            SCHED(m3, m0, m1, m2)
            RND4(m3, k + 12)
        }
#undef RND4
#undef SCHED
        st0 = _mm_add_epi32(st0, save0);
        st1 = _mm_add_epi32(st1, save1);
        data += 64;
    }
    tmp = _mm_shuffle_epi32(st0, 0x1B);       // FEBA
    st1 = _mm_shuffle_epi32(st1, 0xB1);       // DCHG
    st0 = _mm_blend_epi16(tmp, st1, 0xF0);    // DCBA
    st1 = _mm_alignr_epi8(st1, tmp, 8);       // HGFE
    _mm_storeu_si128((__m128i *)&state[0], st0);
    _mm_storeu_si128((__m128i *)&state[4], st1);
}

bool have_shani() {
    static const bool v = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3");
    return v;
}

// digest of header(32) | blob(131072) | commitment(48), the compute_challenge message
void challenge_digest(uint8_t out[32], const uint8_t *blob, const uint8_t *comm48) {
    static const uint8_t header[32] = {'F', 'S', 'B', 'L', 'O', 'B', 'V', 'E', 'R', 'I', 'F', 'Y', '_', 'V', '1', '_',
                                       0x00, 0x10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!have_shani()) {
        std::vector<uint8_t> m(32 + kBlobBytes + 48);
        memcpy(m.data(), header, 32);
        memcpy(m.data() + 32, blob, kBlobBytes);
        memcpy(m.data() + 32 + kBlobBytes, comm48, 48);
        sha256_host(out, m.data(), m.size());
        return;
    }
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint8_t blk[64];
    memcpy(blk, header, 32);
    memcpy(blk + 32, blob, 32);
    compress_shani(h, blk, 1);
    compress_shani(h, blob + 32, (kBlobBytes - 64) / 64);       // 2047 whole blocks straight from the blob
    memcpy(blk, blob + kBlobBytes - 32, 32);
    memcpy(blk + 32, comm48, 32);
    compress_shani(h, blk, 1);
    memset(blk, 0, 64);
    memcpy(blk, comm48 + 32, 16);
    blk[16] = 0x80;
    const uint64_t bits = (uint64_t)(32 + kBlobBytes + 48) * 8;
    for (int k = 0; k < 8; k++) blk[63 - k] = (uint8_t)(bits >> (8 * k));
    compress_shani(h, blk, 1);
    for (int k = 0; k < 8; k++) {
        out[4 * k] = (uint8_t)(h[k] >> 24);
        out[4 * k + 1] = (uint8_t)(h[k] >> 16);
        out[4 * k + 2] = (uint8_t)(h[k] >> 8);
        out[4 * k + 3] = (uint8_t)h[k];
    }
}

}  // namespace

// SHA-256 of an arbitrary message with the SHA extensions when the CPU has them (the batch-level Fiat-Shamir hash of
// verify_blob_kzg_proof_batch: 160 bytes per blob), the portable routine otherwise
void sha256_fast(uint8_t out[32], const uint8_t *msg, size_t len) {
    if (!have_shani()) {
        sha256_host(out, msg, len);
        return;
    }
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    const size_t whole = len / 64;
    compress_shani(h, msg, whole);
    uint8_t tail[128];
    memset(tail, 0, sizeof tail);
    const size_t rem = len - 64 * whole;
    memcpy(tail, msg + 64 * whole, rem);
    tail[rem] = 0x80;
    const size_t tl = rem + 9 <= 64 ? 64 : 128;
    const uint64_t bits = (uint64_t)len * 8;
    for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    compress_shani(h, tail, tl / 64);
    for (int k = 0; k < 8; k++) {
        out[4 * k] = (uint8_t)(h[k] >> 24);
        out[4 * k + 1] = (uint8_t)(h[k] >> 16);
        out[4 * k + 2] = (uint8_t)(h[k] >> 8);
        out[4 * k + 3] = (uint8_t)h[k];
    }
}

// SHA-256 of prefix | msg without building the concatenation (the batch challenge: a 32-byte header in front of 160 bytes per blob that
// already lie in one buffer, verify.hip); prefix_len < 64
void sha256_fast_prefixed(uint8_t out[32], const uint8_t *prefix, size_t prefix_len, const uint8_t *msg, size_t len) {
    if (!have_shani() || prefix_len >= 64) {
        std::vector<uint8_t> m(prefix_len + len);
        memcpy(m.data(), prefix, prefix_len);
        if (len) memcpy(m.data() + prefix_len, msg, len);
        sha256_fast(out, m.data(), m.size());
        return;
    }
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint8_t tail[192];
    memset(tail, 0, sizeof tail);
    memcpy(tail, prefix, prefix_len);
    size_t used = 0, fill = prefix_len;   // bytes of msg consumed; bytes in tail
    if (prefix_len + len >= 64) {
        used = 64 - prefix_len;
        memcpy(tail + prefix_len, msg, used);
        compress_shani(h, tail, 1);
        const size_t whole = (len - used) / 64;
        compress_shani(h, msg + used, whole);
        used += 64 * whole;
        memset(tail, 0, sizeof tail);
        fill = 0;
    }
    memcpy(tail + fill, msg + used, len - used);
    fill += len - used;
    tail[fill] = 0x80;
    const size_t tl = fill + 9 <= 64 ? 64 : 128;
    const uint64_t bits = (uint64_t)(prefix_len + len) * 8;
    for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    compress_shani(h, tail, tl / 64);
    for (int k = 0; k < 8; k++) {
        out[4 * k] = (uint8_t)(h[k] >> 24);
        out[4 * k + 1] = (uint8_t)(h[k] >> 16);
        out[4 * k + 2] = (uint8_t)(h[k] >> 8);
        out[4 * k + 3] = (uint8_t)h[k];
    }
}

// hardware threads this process may really use (shared with the host-side validation of verify.hip)
static unsigned probe_host_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    // respect a cgroup CPU quota (containers expose every hardware thread but allow far fewer)
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32];
        long long period = 0;
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            long long quota = atoll(q);
            unsigned lim = (unsigned)((quota + period - 1) / period);
            if (lim >= 1 && lim < n) n = lim;
        }
        fclose(f);
    }
    // and a bound of the library's own: 32 unless LWKZG_HOST_THREADS says otherwise (64 and 128 measured the same or worse on a
    // 256-thread box whose container does not get them all: profiles/r06_experiments.md section 9)
    const unsigned cap = knobs().host_threads > 0 ? (unsigned)knobs().host_threads : 32u;
    return n > cap ? cap : n;
}

unsigned host_threads() {
    static const unsigned n = probe_host_threads();
    return n;
}

size_t host_small_batch_limit() {
    const size_t lim = 4 * (size_t)host_threads();
    return lim > 64 ? 64 : lim;
}


// ---- the host threads ----------------------------------------------------------------------------
// Hashing, point validation and the small linear combinations are spread over host_threads() - 1 persistent workers
// plus the calling thread: starting sixteen std::threads costs ~0.4 ms, several times the work of a small batch.
// The workers sleep on a condition variable between calls and are never joined (the pool lives as long as the
// process; a forked child gets a fresh one, since threads do not survive fork).
namespace {

class HostPool {
  public:
    explicit HostPool(unsigned workers) : pid_(getpid()) {
        for (unsigned k = 0; k < workers; k++) {
            try {
                std::thread([this]() { worker(); }).detach();
                workers_++;
            } catch (...) {
                break;  // fewer workers (none: the caller does everything itself); nothing unwinds across the C ABI
            }
        }
    }
    pid_t pid() const { return pid_; }

    // `grain`: indices a woken worker should find for itself. Waking is not free -- 255 parked threads woken at once re-acquire one mutex
    // in turn, and a 512-blob slice of a long verification (70 us of SHA-256 per blob) spent 1.7 ms per job that way, more than its
    // 1.2 ms upload (profiles/r06_experiments.md section 8) -- so a job wakes ceil(n / grain) workers, one notification each, and the
    // others sleep through it (they skip finished generations when they next wake).
    void run(size_t n, const std::function<void(size_t)> &fn, size_t grain) {
        std::lock_guard<std::mutex> one_at_a_time(run_mu_);
        size_t wake = grain <= 1 ? n : (n + grain - 1) / grain;
        if (wake > workers_) wake = workers_;
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            n_ = n;
            next_.store(0, std::memory_order_relaxed);
            gen_++;
        }
        if (wake >= workers_) cv_work_.notify_all();
        else
            for (size_t k = 0; k < wake; k++) cv_work_.notify_one();
        claim_loop(fn, n);
        std::unique_lock<std::mutex> lk(mu_);
        done_gen_ = gen_;  // workers that wake up from here on skip this round
        cv_done_.wait(lk, [this]() { return active_ == 0; });
        fn_ = nullptr;
    }

  private:
    void claim_loop(const std::function<void(size_t)> &fn, size_t n) {
        for (;;) {
            const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n) break;
            fn(i);
        }
    }
    void worker() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_work_.wait(lk, [&]() { return gen_ != seen; });
            seen = gen_;
            if (done_gen_ == gen_) continue;
            const std::function<void(size_t)> *fn = fn_;
            const size_t n = n_;
            active_++;
            lk.unlock();
            claim_loop(*fn, n);
            lk.lock();
            if (--active_ == 0) cv_done_.notify_all();
        }
    }

    const pid_t pid_;
    size_t workers_ = 0;
    std::mutex run_mu_, mu_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0, done_gen_ = 0;
    unsigned active_ = 0;
};

}  // namespace

// ---- side workers (engine.h: SideTask) ----------------------------------------------------------------------------------------------
// One job at a time per worker, handed over under the worker's own mutex; the workers are detached and the table is leaked (a thread
// parked on a condition variable must never see it destroyed), a forked child starts a table of its own.
struct SideWorker {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::function<void()> fn;
    int state = 0;                    // 0 parked, 1 job handed over, 2 job done
    std::atomic<bool> taken{false};
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_job.wait(lk, [this]() { return state == 1; });
            lk.unlock();
            fn();
            lk.lock();
            fn = nullptr;
            state = 2;
            cv_done.notify_one();
        }
    }
};
namespace {
constexpr unsigned kSideWorkersMax = 64;
struct SideTable {
    pid_t pid;
    std::mutex mu;                    // growing the table
    std::atomic<unsigned> n{0};
    SideWorker *w[kSideWorkersMax];
};
std::atomic<SideTable *> g_side{nullptr};
std::mutex g_side_mu;
}  // namespace

bool side_workers_on() { return knobs().side_workers != 0; }

SideWorker *side_worker_acquire() {
    SideTable *t = g_side.load(std::memory_order_acquire);
    if (!t || t->pid != getpid()) {
        std::lock_guard<std::mutex> lk(g_side_mu);
        t = g_side.load(std::memory_order_acquire);
        if (!t || t->pid != getpid()) {
            t = new (std::nothrow) SideTable;
            if (!t) return nullptr;
            t->pid = getpid();
            g_side.store(t, std::memory_order_release);
        }
    }
    const unsigned n = t->n.load(std::memory_order_acquire);
    for (unsigned i = 0; i < n; i++)
        if (!t->w[i]->taken.exchange(true, std::memory_order_acquire)) return t->w[i];
    std::lock_guard<std::mutex> lk(t->mu);
    const unsigned m = t->n.load(std::memory_order_acquire);
    if (m >= kSideWorkersMax) return nullptr;
    SideWorker *w = new (std::nothrow) SideWorker;
    if (!w) return nullptr;
    w->taken.store(true, std::memory_order_relaxed);
    try {
        std::thread([w]() { w->loop(); }).detach();
    } catch (...) {
        delete w;
        return nullptr;
    }
    t->w[m] = w;
    t->n.store(m + 1, std::memory_order_release);
    return w;
}

void side_worker_run(SideWorker *w, std::function<void()> fn) {
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->fn = std::move(fn);
        w->state = 1;
    }
    w->cv_job.notify_one();
}

void side_worker_wait(SideWorker *w) {
    {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv_done.wait(lk, [w]() { return w->state == 2; });
        w->state = 0;
    }
    w->taken.store(false, std::memory_order_release);
}

namespace {
std::atomic<double> g_host_hash_rate{0.0};
std::atomic<int64_t> g_host_last_active_ns{0};
int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

// fn(0) .. fn(n - 1), each exactly once, on the host threads (the caller works too); returns when all are done.
// Calls from different threads take turns. fn must not call host_parallel_for itself.
void host_parallel_for_grain(size_t n, const std::function<void(size_t)> &fn, size_t grain);
void host_parallel_for(size_t n, const std::function<void(size_t)> &fn) { host_parallel_for_grain(n, fn, 1); }
// blobs a thread woken for a hashing job should find (70 us each): LWKZG_HOST_HASH_GRAIN (experiment; 1 = every parked thread, as before r06)
static size_t host_hash_grain() { return knobs().host_hash_grain < 1 ? 1 : (size_t)knobs().host_hash_grain; }

// the same with `grain` indices per woken thread (HostPool::run): for jobs of many short pieces
void host_parallel_for_grain(size_t n, const std::function<void(size_t)> &fn, size_t grain) {
    const unsigned nt = host_threads();
    if (n <= 1 || nt <= 1) {
        for (size_t i = 0; i < n; i++) fn(i);
        return;
    }
    static std::mutex pool_mu;
    static HostPool *pool = nullptr;  // deliberately leaked: its workers are detached
    HostPool *p;
    {
        std::lock_guard<std::mutex> lk(pool_mu);
        if (!pool || pool->pid() != getpid()) pool = new HostPool(nt - 1);
        p = pool;
    }
    p->run(n, fn, grain);
    g_host_last_active_ns.store(now_ns(), std::memory_order_release);
}

// when did the host threads last finish a job (0: never in this process)? The host-assisted challenge paths of the device-resident
// proof calls ask: threads that have been idle for long wake slowly (first touches, parked cores), and the first job of a process
// also creates the pool (engine.hip: host_assist_warm)
int64_t host_last_active_ns() { return g_host_last_active_ns.load(std::memory_order_acquire); }
int64_t host_now_ns() { return now_ns(); }

// wake every worker with a job of a few microseconds each (creates the pool on first use): lwkzg_reserve* calls it, and so does a
// proof call that finds the threads cold and takes the GPU's hash kernel this once
void host_pool_warm() {
    const unsigned nt = host_threads();
    std::atomic<unsigned> sink{0};
    host_parallel_for(4 * (size_t)nt, [&](size_t i) {
        uint8_t buf[64] = {(uint8_t)i}, dg[32];
        sha256_host(dg, buf, sizeof buf);
        sink.fetch_add(dg[0], std::memory_order_relaxed);
    });
}

// mid[8 i .. 8 i + 7] = the SHA-256 chaining value after the 2048 blocks of the compute_challenge message that do not contain a byte of
// the commitment (header | blob without its last 32 bytes): what k_challenge_pairs<true> leaves for k_challenge_finish (sha256.hip), computed
// on the host threads for the one-pass commit-and-prove of mid-size batches (engine.hip)
void challenge_midstates_host(uint32_t *mid, const uint8_t *blobs, size_t n) {
    static const uint8_t header[32] = {'F', 'S', 'B', 'L', 'O', 'B', 'V', 'E', 'R', 'I', 'F', 'Y', '_', 'V', '1', '_',
                                       0x00, 0x10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    host_parallel_for_grain(n, [=](size_t i) {
        const uint8_t *blob = blobs + (size_t)kBlobBytes * i;
        uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
        uint8_t blk[64];
        memcpy(blk, header, 32);
        memcpy(blk + 32, blob, 32);
        if (have_shani()) {
            compress_shani(h, blk, 1);
            compress_shani(h, blob + 32, (kBlobBytes - 64) / 64);   // 2047 whole blocks straight from the blob
        } else {
            sha256_blocks_portable(h, blk, 1);
            sha256_blocks_portable(h, blob + 32, (kBlobBytes - 64) / 64);
        }
        for (int k = 0; k < 8; k++) mid[8 * i + k] = h[k];
    }, host_hash_grain());
}

// digests[i] = SHA-256(header | blobs[i] | comms[i]) for i < n, spread over the host threads
void challenge_digests_host(uint8_t *digests32, const uint8_t *blobs, const uint8_t *comms48, size_t n) {
    const int64_t t0 = now_ns();
    host_parallel_for_grain(n, [=](size_t i) { challenge_digest(digests32 + 32 * i, blobs + (size_t)kBlobBytes * i, comms48 + 48 * i); }, host_hash_grain());
    const int64_t dt = now_ns() - t0;
    if (n >= 256 && dt > 0) {   // what this host's threads really hash per second (a cgroup quota, SMT siblings and the caller's memory all show here)
        const double rate = (double)n * kBlobBytes / ((double)dt * 1e-9);
        const double old = g_host_hash_rate.load(std::memory_order_relaxed);
        g_host_hash_rate.store(old > 0 ? 0.5 * old + 0.5 * rate : rate, std::memory_order_relaxed);
    }
}

// bytes per second the host threads hashed in their recent jobs of 256 blobs and more; before the first such job an estimate from the thread
// count (1.1 GB/s per thread beside its SMT sibling, on at most 32 of them: what an EPYC 9575F box measured, profiles/r06_experiments.md section 9)
double host_hash_rate() {
    const double r = g_host_hash_rate.load(std::memory_order_relaxed);
    if (r > 0) return r;
    const unsigned t = host_threads();
    return 1.1e9 * (t > 32 ? 32 : t);
}

}  // namespace lwk
