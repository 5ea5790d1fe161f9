// front.h -- the coalescing fronts of the single-blob symbols, free of any device code.
//
// The reference's KZGSettings is read-only after load, so any number of threads may call blob_to_kzg_commitment /
// compute_blob_kzg_proof / compute_kzg_proof on one settings object at once (/root/reference/src/lib.rs:253-283,
// SURVEY 8b "Threading"); a GPU launch set per blob would serialise them. Callers that arrive while a launch set is in
// flight are merged into the next one. This header holds the THREADING of that -- queues, leaders, lanes, staging slots --
// as templates over what a leader does with its batch, so that the same code that engine.hip drives the GPU with runs on
// a CPU under -fsanitize=thread against a stub device (tests/front_tsan.cpp, tests/test_front_tsan_cpu.py).
//
// Both fronts give every caller the return code a call of its own would have had, survive a `run` that throws (nothing
// may unwind across the C ABI: every member of the batch gets `rc_on_throw`, the front's state is restored and the
// waiters are woken), and never hold the front's mutex while a batch runs.
#pragma once
#include <condition_variable>
#include <deque>
#include <mutex>
#include <vector>

namespace lwk {

// ---- one leader at a time: compute_blob_kzg_proof / compute_kzg_proof -------------------------------------------------
// Req needs: int mode; int rc; enum State { QUEUED, TAKEN, DONE } state.
template <class Req>
struct LeaderFront {
    std::mutex m;
    std::condition_variable cv;
    std::deque<Req *> queue;
    bool leader_active = false;

    // Whoever arrives while no batch is being run becomes the leader of everything queued in its mode (<= max_batch) and
    // hands it to `run`, which answers every member (sets rc and writes the outputs); the others wait for their bytes.
    template <class Run>
    int submit(Req &req, size_t max_batch, int rc_on_throw, Run &&run) {
        std::unique_lock<std::mutex> lk(m);
        queue.push_back(&req);
        for (;;) {
            if (req.state == Req::DONE) break;
            if (req.state == Req::QUEUED && !leader_active) {
                leader_active = true;
                std::vector<Req *> batch;
                bool threw = false, collecting = true;
                try {
                    for (auto it = queue.begin(); it != queue.end() && batch.size() < max_batch;) {
                        if ((*it)->mode == req.mode) {
                            batch.push_back(*it);      // (may throw: nothing has been taken off the queue's books yet)
                            (*it)->state = Req::TAKEN;
                            it = queue.erase(it);
                        } else {
                            ++it;
                        }
                    }
                    collecting = false;
                    lk.unlock();
                    try {
                        run(batch);
                    } catch (...) {
                        threw = true;
                    }
                    lk.lock();
                } catch (...) {  // out of memory while the batch was being collected
                    threw = true;
                    if (!lk.owns_lock()) lk.lock();
                }
                for (Req *r : batch) {
                    if (threw) r->rc = rc_on_throw;
                    r->state = Req::DONE;
                }
                // (a leader that is not in its own batch -- 64 requests were ahead of it -- stays queued and leads again)
                if (threw && collecting && req.state != Req::DONE) {  // the collection itself failed: this request is answered too
                    for (auto it = queue.begin(); it != queue.end(); ++it)
                        if (*it == &req) {
                            queue.erase(it);
                            break;
                        }
                    req.rc = rc_on_throw;
                    req.state = Req::DONE;
                }
                leader_active = false;
                cv.notify_all();
                continue;  // req is DONE now: it was part of its own batch
            }
            cv.wait(lk);
        }
        return req.rc;
    }
};

// ---- up to `Lanes` leaders, staging slots: blob_to_kzg_commitment ----------------------------------------------------
// Every caller copies its blob into a staging slot (in parallel, outside the lock); the first to find a free lane leads
// everything queued in its mode on that lane, so that one batch uploads while the other computes.
// Req needs: int mode; int rc; int slot; enum State { QUEUED, TAKEN, DONE } state.
template <class Req, int Lanes>
struct LaneFront {
    std::mutex m;
    std::condition_variable cv;
    std::deque<Req *> queue;
    int leaders = 0;
    bool lane_busy[Lanes] = {};
    std::vector<int> free_slots;

    void add_slots(int n) {  // caller holds m (or nobody else knows the object yet)
        for (int k = n - 1; k >= 0; k--) free_slots.push_back(k);
    }

    // stage(slot): copy the caller's input into staging slot `slot` (called WITHOUT the lock).
    // run(lane, batch): one launch set for the batch on that lane; answers every member.
    template <class Stage, class Run>
    int submit(Req &req, size_t max_batch, int rc_on_throw, Stage &&stage, Run &&run) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !free_slots.empty(); });
        req.slot = free_slots.back();
        free_slots.pop_back();
        lk.unlock();
        try {
            stage(req.slot);
        } catch (...) {
            lk.lock();
            free_slots.push_back(req.slot);
            cv.notify_all();
            return rc_on_throw;
        }
        lk.lock();
        try {
            queue.push_back(&req);
        } catch (...) {
            free_slots.push_back(req.slot);
            cv.notify_all();
            return rc_on_throw;
        }
        for (;;) {
            if (req.state == Req::DONE) break;
            if (req.state == Req::QUEUED && leaders < Lanes) {
                int lane = 0;
                while (lane_busy[lane]) lane++;
                lane_busy[lane] = true;
                leaders++;
                std::vector<Req *> batch;
                bool threw = false, collecting = true;
                try {
                    for (auto it = queue.begin(); it != queue.end() && batch.size() < max_batch;) {
                        if ((*it)->mode == req.mode) {
                            batch.push_back(*it);
                            (*it)->state = Req::TAKEN;
                            it = queue.erase(it);
                        } else {
                            ++it;
                        }
                    }
                    collecting = false;
                    lk.unlock();
                    try {
                        run(lane, batch);
                    } catch (...) {
                        threw = true;
                    }
                    lk.lock();
                } catch (...) {
                    threw = true;
                    if (!lk.owns_lock()) lk.lock();
                }
                for (Req *r : batch) {
                    if (threw) r->rc = rc_on_throw;
                    r->state = Req::DONE;
                    free_slots.push_back(r->slot);   // (capacity was reserved by add_slots: no allocation here)
                }
                if (threw && collecting && req.state != Req::DONE) {
                    for (auto it = queue.begin(); it != queue.end(); ++it)
                        if (*it == &req) {
                            queue.erase(it);
                            break;
                        }
                    req.rc = rc_on_throw;
                    req.state = Req::DONE;
                    free_slots.push_back(req.slot);
                }
                lane_busy[lane] = false;
                leaders--;
                cv.notify_all();
                continue;
            }
            cv.wait(lk);
        }
        return req.rc;
    }
};

}  // namespace lwk
