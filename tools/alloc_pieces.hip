// tools/alloc_pieces.hip -- does the cost of hipMalloc per GB depend on the SIZE of the call? (bench.py: the default table's twenty 2 GB
// windows cost 0.4-4 ms in all, the 16-bit table's sixteen 17 GB windows 2-5 s.) Allocates `total` GB in pieces of each size in turn,
// times the calls, touches every 2 MB page from a kernel, frees.   Build: hipcc --offload-arch=gfx950 -O2 tools/alloc_pieces.hip -o /tmp/alloc_pieces
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_touch(uint8_t *p, size_t bytes) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) << 21;
    if (i < bytes) p[i] = 1;
}

int main(int argc, char **argv) {
    const size_t total = (argc > 1 ? (size_t)atoll(argv[1]) : 256) << 30;
    size_t fr, tot;
    hipMemGetInfo(&fr, &tot);
    printf("free %.1f GB of %.1f GB; allocating %zu GB per round\n", fr / 1e9, tot / 1e9, total >> 30);
    // one size per process: a freed piece is not necessarily back with the driver when hipFree returns
    const double pg_arg = argc > 2 ? atof(argv[2]) : 2.0;
    for (double pg : {pg_arg}) {
        size_t piece = (size_t)(pg * (1 << 30));
        size_t n = total / piece;
        std::vector<void *> ptrs;
        double t0 = now(), slowest = 0;
        for (size_t i = 0; i < n; i++) {
            void *p = nullptr;
            double a = now();
            hipError_t e = hipMalloc(&p, piece);
            double d = now() - a;
            if (d > slowest) slowest = d;
            if (e != hipSuccess) { printf("  hipMalloc %zu failed: %s\n", i, hipGetErrorString(e)); break; }
            ptrs.push_back(p);
        }
        double t1 = now();
        for (void *p : ptrs) hipLaunchKernelGGL(k_touch, dim3((unsigned)((piece >> 21) / 256 + 1)), dim3(256), 0, 0, (uint8_t *)p, piece);
        hipError_t se = hipDeviceSynchronize();
        double t2 = now();
        for (void *p : ptrs) { hipError_t fe = hipFree(p); if (fe != hipSuccess) printf("  hipFree: %s\n", hipGetErrorString(fe)); }
        (void)hipGetLastError();
        double t3 = now();
        printf("pieces of %5.1f GB x %4zu: hipMalloc %.3f s (%.2f ms per GB, slowest call %.3f s), touch %.3f s (%s), hipFree %.3f s\n", pg, ptrs.size(),
               t1 - t0, (t1 - t0) * 1e3 / (ptrs.size() * pg), slowest, t2 - t1, hipGetErrorString(se), t3 - t2);
    }
    return 0;
}
