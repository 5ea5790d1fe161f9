// Where does a host-pointer batch's time go between the caller's pageable memory and HBM? (r06; profiles/r06_experiments.md section 5)
// 64 MiB slices (512 blobs), as the long host-pointer batches move them:
//   (a) hipMemcpy from pageable memory (what r05 does; the runtime stages through its own pinned buffers on the calling thread)
//   (b) hipMemcpyAsync from pinned memory (the DMA alone)
//   (c) memcpy pageable -> pinned on 1 / 2 / 4 / 8 / 16 host threads (what a pinned ring adds in front of (b))
//   (d) the same source read once by N threads without writing (the memory system's read side alone)
// hipcc --offload-arch=gfx950 -O3 -o tools/h2d_bench_bin tools/h2d_bench.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

// a full-chip compute kernel (dense multiply-adds, every SIMD busy for ~`iters` x 64 products) to copy beside
__global__ __launch_bounds__(256) void k_busy(unsigned long long *out, uint32_t iters, uint32_t seed) {
    uint64_t acc = seed + threadIdx.x;
    const uint32_t a = (seed * 2654435761u) | 1u, b = seed ^ 0x9e3779b9u;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) acc = (uint64_t)(uint32_t)acc * a + (acc >> 32) + b;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t slice = 64u << 20, total = 8 * slice;
    uint8_t *src = (uint8_t *)malloc(total);
    for (size_t i = 0; i < total; i += 4096) src[i] = (uint8_t)i;   // resident
    memset(src, 7, total);
    uint8_t *pin = nullptr, *dev = nullptr;
    hipHostMalloc((void **)&pin, 2 * slice, hipHostMallocDefault);
    memset(pin, 1, 2 * slice);
    hipMalloc((void **)&dev, 2 * slice);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    printf("host threads available: %u\n", std::thread::hardware_concurrency());
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now_ms();
        for (int k = 0; k < 8; k++) hipMemcpy(dev + (k & 1) * slice, src + k * slice, slice, hipMemcpyHostToDevice);
        double t1 = now_ms();
        printf("(a) hipMemcpy pageable -> device, 8 x 64 MiB: %.2f ms  %.1f GB/s\n", t1 - t0, total / (t1 - t0) / 1e6);
    }
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now_ms();
        for (int k = 0; k < 8; k++) hipMemcpyAsync(dev + (k & 1) * slice, pin + (k & 1) * slice, slice, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
        double t1 = now_ms();
        printf("(b) hipMemcpyAsync pinned -> device, 8 x 64 MiB: %.2f ms  %.1f GB/s\n", t1 - t0, total / (t1 - t0) / 1e6);
    }
    for (unsigned nt : {1u, 2u, 4u, 8u, 16u}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            double t0 = now_ms();
            for (int k = 0; k < 8; k++) {
                std::vector<std::thread> th;
                const size_t per = slice / nt;
                for (unsigned t = 0; t < nt; t++)
                    th.emplace_back([=]() { memcpy(pin + (k & 1) * slice + t * per, src + k * slice + t * per, per); });
                for (auto &x : th) x.join();
            }
            double t1 = now_ms();
            if (t1 - t0 < best) best = t1 - t0;
        }
        printf("(c) memcpy pageable -> pinned on %2u threads, 8 x 64 MiB: %.2f ms  %.1f GB/s\n", nt, best, total / best / 1e6);
    }
    uint8_t *dst2 = (uint8_t *)malloc(2 * slice);
    memset(dst2, 1, 2 * slice);
    for (unsigned nt : {1u, 4u, 16u}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            double t0 = now_ms();
            for (int k = 0; k < 8; k++) {
                std::vector<std::thread> th;
                const size_t per = slice / nt;
                for (unsigned t = 0; t < nt; t++)
                    th.emplace_back([=]() { memcpy(dst2 + (k & 1) * slice + t * per, src + k * slice + t * per, per); });
                for (auto &x : th) x.join();
            }
            double t1 = now_ms();
            if (t1 - t0 < best) best = t1 - t0;
        }
        printf("(c') memcpy pageable -> pageable on %2u threads, 8 x 64 MiB: %.2f ms  %.1f GB/s\n", nt, best, total / best / 1e6);
    }
    for (unsigned nt : {1u, 4u, 16u}) {
        std::vector<uint64_t> sink(nt);
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            double t0 = now_ms();
            std::vector<std::thread> th;
            const size_t per = total / nt;
            for (unsigned t = 0; t < nt; t++)
                th.emplace_back([&, t]() {
                    const uint64_t *p = (const uint64_t *)(src + t * per);
                    uint64_t s = 0;
                    for (size_t i = 0; i < per / 8; i++) s += p[i];
                    sink[t] = s;
                });
            for (auto &x : th) x.join();
            double t1 = now_ms();
            if (t1 - t0 < best) best = t1 - t0;
        }
        printf("(d) read 512 MiB on %2u threads: %.2f ms  %.1f GB/s (%llu)\n", nt, best, total / best / 1e6, (unsigned long long)sink[0]);
    }
    // (f) the same uploads BESIDE a kernel that keeps every SIMD busy (what a staged slice meets: the previous slice's MSM)
    {
        hipStream_t cs;
        hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
        unsigned long long *dbusy;
        hipMalloc((void **)&dbusy, 4096 * 8);
        auto busy = [&](uint32_t iters) { hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, cs, dbusy, iters, 7u); };
        busy(2000); hipStreamSynchronize(cs);
        double t0 = now_ms(); busy(40000); hipStreamSynchronize(cs); double tb = now_ms() - t0;
        printf("(f) busy kernel alone: %.2f ms\n", tb);
        for (int pinned = 0; pinned < 2; pinned++) {
            busy(40000);
            double c0 = now_ms();
            for (int k = 0; k < 4; k++) {
                if (pinned) hipMemcpyAsync(dev + (k & 1) * slice, pin + (k & 1) * slice, slice, hipMemcpyHostToDevice, st);
                else hipMemcpyAsync(dev + (k & 1) * slice, src + k * slice, slice, hipMemcpyHostToDevice, st);
            }
            hipStreamSynchronize(st);
            double c1 = now_ms();
            hipStreamSynchronize(cs);
            double c2 = now_ms();
            printf("(f) 4 x 64 MiB from %s memory beside the busy kernel: copies %.2f ms (%.1f GB/s), kernel done at %.2f ms (alone: %.2f)\n",
                   pinned ? "pinned" : "pageable", c1 - c0, 4.0 * slice / (c1 - c0) / 1e6, c2 - c0, tb);
        }
    }
    // (e) pipelined: host threads fill slot k+1 while the DMA of slot k runs
    for (unsigned nt : {4u, 16u}) {
        hipEvent_t ev[2];
        hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
        bool used[2] = {false, false};
        double t0 = now_ms();
        for (int k = 0; k < 8; k++) {
            const int s = k & 1;
            if (used[s]) hipEventSynchronize(ev[s]);
            std::vector<std::thread> th;
            const size_t per = slice / nt;
            for (unsigned t = 0; t < nt; t++) th.emplace_back([=]() { memcpy(pin + s * slice + t * per, src + k * slice + t * per, per); });
            for (auto &x : th) x.join();
            hipMemcpyAsync(dev + s * slice, pin + s * slice, slice, hipMemcpyHostToDevice, st);
            hipEventRecord(ev[s], st);
            used[s] = true;
        }
        hipStreamSynchronize(st);
        double t1 = now_ms();
        printf("(e) ring: %2u threads fill slot k+1 beside the DMA of slot k, 8 x 64 MiB: %.2f ms  %.1f GB/s\n", nt, t1 - t0, total / (t1 - t0) / 1e6);
    }
    return 0;
}
