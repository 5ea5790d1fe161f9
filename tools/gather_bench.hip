// tools/gather_bench.hip -- can HBM serve the gathers of a giant fixed-base table?
// Each lane reads 112 contiguous bytes (7 x 16 B) at a pseudo-random row of a table of `gib` GiB, `iters` times,
// with the next row's loads issued before the current row is consumed. Reports rows/s and GB/s.
// hipcc --offload-arch=gfx950 -O3 [-DROW_PAD=1] tools/gather_bench.hip -o tools/gather_bench_bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#ifndef ROW_PAD   // -DROW_PAD=1: rows 128 bytes apart (one line each) instead of packed at 112
#define ROW_PAD 0
#endif
struct Row { uint4 v[7 + ROW_PAD]; };

__global__ __launch_bounds__(256) void k_gather(const Row *__restrict__ table, uint64_t nrows, int iters, int alu, uint32_t *out) {
    uint64_t s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s % nrows; };
    Row cur = table[next()];
    for (int it = 0; it < iters; it++) {
        Row nxt = table[next()];
        uint32_t x = cur.v[0].x ^ cur.v[1].y ^ cur.v[2].z ^ cur.v[3].w ^ cur.v[4].x ^ cur.v[5].y ^ cur.v[6].z;
        for (int k = 0; k < alu; k++) x = x * 2654435761u + (x >> 7);   // stand-in for the mixed addition
        acc ^= x;
        cur = nxt;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc ^ cur.v[0].x;
}

int main(int argc, char **argv) {
    double gib = argc > 1 ? atof(argv[1]) : 128.0;
    int alu = argc > 2 ? atoi(argv[2]) : 0;
    size_t bytes = (size_t)(gib * (1ull << 30));
    uint64_t nrows = bytes / sizeof(Row);
    Row *table;
    uint32_t *out;
    CHECK(hipMalloc(&table, nrows * sizeof(Row)));
    CHECK(hipMemset(table, 1, nrows * sizeof(Row)));
    const int blocks = 256 * 8, iters = 256;
    CHECK(hipMalloc(&out, blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, table, nrows, iters, alu, out);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, table, nrows, iters, alu, out);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double rows = (double)blocks * 256 * iters;
    printf("{\"table_gib\": %.1f, \"alu_per_row\": %d, \"ms\": %.3f, \"rows_per_s\": %.3e, \"GB_per_s\": %.1f}\n", gib, alu, best,
           rows / (best * 1e-3), rows * 112 / (best * 1e-3) / 1e9);
    return 0;
}
