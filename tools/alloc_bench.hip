// How long does it take to get a 240 GB table's worth of device memory, and can it overlap with kernels?
//   hipcc --offload-arch=gfx950 -O2 -o tools/alloc_bench_bin tools/alloc_bench.hip && tools/alloc_bench_bin
// (1) one hipMalloc / hipFree; (2) virtual-memory API: one address reservation, chunks created and mapped one at a time.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void touch(char *p, size_t n, size_t stride) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (i < n) p[i] = 1;
}

int main(int argc, char **argv) {
    const size_t total = (argc > 1 ? (size_t)atoll(argv[1]) : 240) << 30;
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int phase = argc > 2 ? atoi(argv[2]) : 3;   // 1 = hipMalloc, 2 = virtual-memory API, 3 = both
    hipSetDevice(0);
    hipFree(0);
    size_t fr, tot;
    hipMemGetInfo(&fr, &tot);
    printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
    for (int rep = 0; rep < 2 && (phase & 1); rep++) {
        char *p = nullptr;
        double t0 = now();
        hipError_t e = hipMalloc((void **)&p, total);
        double t1 = now();
        printf("hipMalloc(%zu GB): %s, %.3f s\n", total >> 30, hipGetErrorString(e), t1 - t0);
        if (e != hipSuccess) return 1;
        touch<<<(unsigned)((total / (2 << 20) + 255) / 256), 256>>>(p, total, 2 << 20);
        hipDeviceSynchronize();
        double t2 = now();
        printf("  first touch of every 2 MB page: %.3f s\n", t2 - t1);
        hipFree(p);
        printf("  hipFree: %.3f s\n", now() - t2);
    }
    if (phase & 4) {   // the same bytes as N allocations made by N host threads at once, and one after the other by one thread
        for (int nthreads : {16, 4}) {
            std::vector<char *> ps(nthreads, nullptr);
            std::vector<double> ts(nthreads, 0);
            double t0 = now();
            std::vector<std::thread> th;
            for (int k = 0; k < nthreads; k++)
                th.emplace_back([&, k] {
                    hipSetDevice(0);
                    double a = now();
                    hipMalloc((void **)&ps[k], total / nthreads);
                    ts[k] = now() - a;
                });
            for (auto &t : th) t.join();
            double t1 = now();
            double mx = 0;
            for (double x : ts) mx = x > mx ? x : mx;
            printf("%d threads x hipMalloc(%zu GB) at once: %.3f s wall (slowest call %.3f s)\n", nthreads, (total / nthreads) >> 30, t1 - t0, mx);
            for (auto p : ps) hipFree(p);
            printf("  frees: %.3f s\n", now() - t1);
            t0 = now();
            for (int k = 0; k < nthreads; k++) hipMalloc((void **)&ps[k], total / nthreads);
            printf("%d x hipMalloc(%zu GB) one after the other: %.3f s\n", nthreads, (total / nthreads) >> 30, now() - t0);
            for (auto p : ps) hipFree(p);
        }
        return 0;
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    printf("VMM granularity: %s, %zu bytes\n", hipGetErrorString(e), gran);
    if (e != hipSuccess || !(phase & 2)) return 0;
    for (size_t chunk_gb : {1, 4, 16}) {
        const size_t chunk = chunk_gb << 30;
        void *va = nullptr;
        double t0 = now();
        e = hipMemAddressReserve(&va, total, 0, nullptr, 0);
        if (e != hipSuccess) { printf("reserve: %s\n", hipGetErrorString(e)); return 0; }
        double t_res = now() - t0;
        std::vector<hipMemGenericAllocationHandle_t> hs;
        double t_create = 0, t_map = 0, t_acc = 0;
        hipMemAccessDesc ad = {};
        ad.location = prop.location;
        ad.flags = hipMemAccessFlagsProtReadWrite;
        bool ok = true;
        for (size_t off = 0; off < total && ok; off += chunk) {
            hipMemGenericAllocationHandle_t h;
            double a = now();
            ok = hipMemCreate(&h, chunk, &prop, 0) == hipSuccess;
            double b = now();
            ok = ok && hipMemMap((char *)va + off, chunk, 0, h, 0) == hipSuccess;
            double c = now();
            ok = ok && hipMemSetAccess((char *)va + off, chunk, &ad, 1) == hipSuccess;
            double d = now();
            t_create += b - a; t_map += c - b; t_acc += d - c;
            if (ok) hs.push_back(h);
        }
        printf("VMM chunks of %zu GB: %s; reserve %.3f s, create %.3f s, map %.3f s, set access %.3f s (total %.3f s)\n", chunk_gb,
               ok ? "ok" : "FAILED", t_res, t_create, t_map, t_acc, t_res + t_create + t_map + t_acc);
        if (ok) {
            double a = now();
            touch<<<(unsigned)((total / (2 << 20) + 255) / 256), 256>>>((char *)va, total, 2 << 20);
            hipError_t se = hipDeviceSynchronize();
            printf("  touch: %s, %.3f s\n", hipGetErrorString(se), now() - a);
        }
        double a = now();
        for (size_t k = 0; k < hs.size(); k++) {
            hipMemUnmap((char *)va + k * chunk, chunk);
            hipMemRelease(hs[k]);
        }
        hipMemAddressFree(va, total);
        printf("  unmap + release: %.3f s\n", now() - a);
    }
    return 0;
}
