// The coalescing fronts of the single-blob symbols (lambdaworks_kzg_amd/csrc/front.h) against a STUB device, for
// -fsanitize=thread on a machine without a GPU (tests/test_front_tsan_cpu.py builds and runs this):
//   g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -I lambdaworks_kzg_amd/csrc tests/front_tsan.cpp -o front_tsan && ./front_tsan
// Same scenario as the GPU stress test (many threads on one settings object, both modes, every kind of caller at once), with
// "kernels" that sleep: every request must get exactly the answer a call of its own would have had; a lane never runs two
// batches at once, a staging slot never holds two blobs, a batch never mixes modes or exceeds its size, one leader at a
// time on the proof front; a `run` that throws answers its whole batch with the error code and leaves the front usable.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <thread>
#include <vector>

#include "front.h"

using namespace lwk;

static constexpr int kOK = 0, kErrThrow = 3, kRejected = 2;
static constexpr int kLanes = 2, kSlots = 24, kMaxBatch = 8;

static uint64_t expected(uint64_t payload, int mode) { return payload * 0x9E3779B97F4A7C15ull + (uint64_t)mode * 77; }

struct LReq {
    enum State { QUEUED, TAKEN, DONE };
    int mode = 0, rc = -1, slot = -1;
    uint64_t payload = 0, out = 0;
    State state = QUEUED;
};

struct PReq {
    enum State { QUEUED, TAKEN, DONE };
    int mode = 0, rc = -1;
    uint64_t payload = 0, out = 0;
    State state = QUEUED;
};

static std::atomic<int> failures{0};
#define CHECK(x)                                                        \
    do {                                                                \
        if (!(x)) {                                                     \
            fprintf(stderr, "CHECK failed: %s (line %d)\n", #x, __LINE__); \
            failures++;                                                 \
        }                                                               \
    } while (0)

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 12, per_thread = argc > 2 ? atoi(argv[2]) : 150;
    // ---- lane front (blob_to_kzg_commitment)
    LaneFront<LReq, kLanes> lf;
    lf.add_slots(kSlots);
    std::vector<uint64_t> staging(kSlots, 0);                 // the "pinned host memory"
    std::vector<std::atomic<int>> slot_owner(kSlots);
    for (auto &s : slot_owner) s = 0;
    std::atomic<int> lane_running[kLanes];
    for (auto &l : lane_running) l = 0;
    std::atomic<long> batches{0}, merged{0}, thrown{0}, answered_throw{0}, rejected{0};
    auto lane_run = [&](int lane, const std::vector<LReq *> &batch) {
        CHECK(lane >= 0 && lane < kLanes);
        CHECK(lane_running[lane].fetch_add(1) == 0);          // a lane runs one batch at a time
        CHECK(!batch.empty() && (int)batch.size() <= kMaxBatch);
        std::this_thread::sleep_for(std::chrono::microseconds(30 + 10 * batch.size()));   // the "launch set"
        const long b = batches.fetch_add(1);
        merged += (long)batch.size();
        const bool boom = b % 37 == 36;
        for (LReq *r : batch) {
            CHECK(r->mode == batch[0]->mode);                  // one mode per batch
            CHECK(r->state == LReq::TAKEN);
            const uint64_t in = staging[r->slot];              // what the caller staged
            CHECK(in == r->payload);
            CHECK(slot_owner[r->slot].fetch_sub(1) == 1);      // read: the slot may be handed on once the batch is answered
            if (in % 101 == 0) {                               // an input the device rejects: only its own caller hears of it
                r->rc = kRejected;
            } else {
                r->out = expected(in, r->mode);
                r->rc = kOK;
            }
        }
        CHECK(lane_running[lane].fetch_sub(1) == 1);
        if (boom) {
            thrown++;
            throw std::bad_alloc();
        }
    };
    // ---- leader front (compute_blob_kzg_proof / compute_kzg_proof)
    LeaderFront<PReq> pf;
    std::atomic<int> leader_running{0};
    std::atomic<long> pbatches{0};
    auto leader_run = [&](const std::vector<PReq *> &batch) {
        CHECK(leader_running.fetch_add(1) == 0);               // one leader at a time
        CHECK(!batch.empty() && (int)batch.size() <= kMaxBatch);
        std::this_thread::sleep_for(std::chrono::microseconds(60));
        const long b = pbatches.fetch_add(1);
        for (PReq *r : batch) {
            CHECK(r->mode == batch[0]->mode);
            r->out = expected(r->payload, r->mode) ^ 0xABCD;
            r->rc = kOK;
        }
        CHECK(leader_running.fetch_sub(1) == 1);
        if (b % 29 == 28) throw std::runtime_error("leader's host vectors");
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t] {
            uint64_t x = 0x1234567 + 7919 * (uint64_t)t;
            for (int i = 0; i < per_thread; i++) {
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                const int mode = (int)((x >> 33) & 1);
                if ((x >> 40) % 3 != 0) {
                    LReq r;
                    r.mode = mode;
                    r.payload = x >> 8;
                    const int rc = lf.submit(
                        r, kMaxBatch, kErrThrow,
                        [&](int slot) {
                            CHECK(slot_owner[slot].fetch_add(1) == 0);   // a staging slot holds one blob at a time
                            staging[slot] = r.payload;
                        },
                        lane_run);
                    CHECK(rc == r.rc || rc == kErrThrow);
                    if (rc == kOK) CHECK(r.out == expected(r.payload, mode));
                    else if (rc == kRejected) { CHECK((r.payload % 101) == 0); rejected++; }
                    else { CHECK(rc == kErrThrow); answered_throw++; }
                } else {
                    PReq r;
                    r.mode = mode;
                    r.payload = x >> 8;
                    const int rc = pf.submit(r, kMaxBatch, kErrThrow, leader_run);
                    if (rc == kOK) CHECK(r.out == (expected(r.payload, mode) ^ 0xABCD));
                    else { CHECK(rc == kErrThrow); answered_throw++; }
                }
            }
        });
    for (auto &th : pool) th.join();
    CHECK(lf.queue.empty() && lf.leaders == 0 && (int)lf.free_slots.size() == kSlots);
    CHECK(pf.queue.empty() && !pf.leader_active);
    printf("front_tsan: %d threads x %d calls: %ld lane batches (%.2f requests each), %ld proof batches, %ld runs threw, %ld requests answered "
           "with the error code, %ld rejected inputs, %d check failures\n",
           threads, per_thread, batches.load(), batches ? (double)merged / batches : 0.0, pbatches.load(), thrown.load(), answered_throw.load(),
           rejected.load(), failures.load());
    return failures ? 1 : 0;
}
