// tools/gather_sectors.hip -- does the memory side fetch a gathered table row by 32-byte sector, by 64-byte half line or by whole 128-byte
// line? (would 96-byte canonical rows in 128-byte slots cost less than the 112-byte rows k_direct_accumulate_asm reads?)
// Rows sit at a stride of 128 bytes in a table far larger than every cache; each lane reads the FIRST N x 16 bytes of a pseudo-random row,
// the next row's loads issued before the current row is consumed (as the kernel does). One kernel per N, so that a rocprofv3 --pmc pass
// (FETCH_SIZE, TCC_EA0_RDREQ_sum, TCC_EA0_RDREQ_32B_sum) attributes requests per N.
// hipcc --offload-arch=gfx950 -O3 tools/gather_sectors.hip -o tools/gather_sectors_bin ; ./tools/gather_sectors_bin [GiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int N>
__global__ __launch_bounds__(256) void k_gather_first(const uint4 *__restrict__ table, uint64_t nrows, int iters, uint32_t *out) {
    uint64_t s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s % nrows; };
    uint4 cur[N], nxt[N];
    const uint4 *r = table + 8 * next();
#pragma unroll
    for (int k = 0; k < N; k++) cur[k] = r[k];
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        r = table + 8 * next();
#pragma unroll
        for (int k = 0; k < N; k++) nxt[k] = r[k];
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < N; k++) x ^= cur[k].x + cur[k].w;
        for (int k = 0; k < 64; k++) x = x * 2654435761u + (x >> 7);   // a little ALU between rows
        acc ^= x;
#pragma unroll
        for (int k = 0; k < N; k++) cur[k] = nxt[k];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc ^ cur[0].x;
}

template <int N>
static int run(const uint4 *table, uint64_t nrows, uint32_t *out) {
    const int blocks = 256 * 8, iters = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather_first<N>, dim3(blocks), dim3(256), 0, 0, table, nrows, iters, out);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_gather_first<N>, dim3(blocks), dim3(256), 0, 0, table, nrows, iters, out);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double rows = (double)blocks * 256 * iters;
    printf("{\"bytes_read_per_row\": %d, \"ms\": %.3f, \"rows_per_s\": %.3e, \"payload_GB_per_s\": %.1f, \"lines_GB_per_s\": %.1f}\n", 16 * N, best,
           rows / (best * 1e-3), rows * 16 * N / (best * 1e-3) / 1e9, rows * 128 / (best * 1e-3) / 1e9);
    return 0;
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 96.0;
    const size_t bytes = (size_t)(gib * (1ull << 30));
    const uint64_t nrows = bytes / 128;
    uint4 *table;
    uint32_t *out;
    CHECK(hipMalloc(&table, nrows * 128));
    CHECK(hipMemset(table, 1, nrows * 128));
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    if (run<8>(table, nrows, out) || run<7>(table, nrows, out) || run<6>(table, nrows, out) || run<4>(table, nrows, out) || run<2>(table, nrows, out)) return 1;
    return 0;
}
