// What shader clock does a LATENCY-shaped launch run at? (r06, profiles/r06_experiments.md section 4.) The verification's critical kernels are
// a few hundred waves on a chip of 1024 SIMDs: k_challenge_pairs is 64 workgroups x 4 waves for 4096 blobs. This measures clock64 (shader
// cycles) against wall_clock64 (100 MHz) inside such a launch, for grids of 64 / 128 / 256 workgroups, back to back and after idle gaps,
// and with a filler kernel on the other compute units.   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_clock_latency_bin tools/ubench_clock_latency.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <unistd.h>
#include <vector>

__global__ __launch_bounds__(256) void k_chain(unsigned long long *out, uint32_t iters, uint32_t seed, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(2);
    uint32_t a = seed + threadIdx.x, b = seed * 2654435761u;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) {   // dependent 32-bit chain: rotate, add, xor (SHA-like)
            a = __builtin_rotateright32(a, 7) + b;
            b = b ^ a;
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out[3 * w] = (unsigned long long)(c1 - c0);
        out[3 * w + 1] = (unsigned long long)(w1 - w0);
        out[3 * w + 2] = a ^ b;
    }
}

// a filler: dense multiply-adds at priority 0 on as many workgroups as asked, LDS-padded so that it cannot share a compute unit with k_chain's padded workgroups
__global__ __launch_bounds__(256) void k_filler(unsigned long long *out, uint32_t iters, uint32_t seed) {
    uint64_t acc = seed + threadIdx.x;
    const uint32_t a = (seed * 2654435761u) | 1u, b = seed ^ 0x9e3779b9u;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) acc = (uint64_t)(uint32_t)acc * a + (acc >> 32) + b;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

static double run(hipStream_t st, unsigned long long *d, unsigned wgs, uint32_t iters, unsigned lds, int reps, int gap_us, hipStream_t fill_st, unsigned fill_wgs,
                  unsigned long long *d_fill, double *ms_out) {
    std::vector<unsigned long long> h(wgs * 4 * 3);
    double mhz = 0, ms = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < reps; r++) {
        if (gap_us) usleep(gap_us);
        if (fill_wgs) hipLaunchKernelGGL(k_filler, dim3(fill_wgs), dim3(256), 100 * 1024, fill_st, d_fill, 3000u, 99u + r);
        hipEventRecord(e0, st);
        hipLaunchKernelGGL(k_chain, dim3(wgs), dim3(256), lds, st, d, iters, 12345u + r, 1);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        if (fill_wgs) hipStreamSynchronize(fill_st);
        float t = 0;
        hipEventElapsedTime(&t, e0, e1);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double clk = 0, wall = 0;
        for (unsigned w = 0; w < wgs * 4; w++) { clk += (double)h[3 * w]; wall += (double)h[3 * w + 1]; }
        if (r >= reps / 2) { mhz += clk / wall * 100.0; ms += t; }
    }
    const int cnt = reps - reps / 2;
    *ms_out = ms / cnt;
    return mhz / cnt;
}

int main() {
    hipStream_t st, fs;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&fs, hipStreamNonBlocking);
    unsigned long long *d, *df;
    hipMalloc((void **)&d, 1024 * 4 * 3 * 8);
    hipMalloc((void **)&df, 4096 * 8);
    hipFuncSetAttribute((const void *)k_filler, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void *)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const uint32_t iters = 12000;   // ~3 ms of a lone wave
    printf("%-44s %10s %10s\n", "launch", "MHz", "ms");
    for (unsigned wgs : {64u, 128u, 256u, 1024u}) {
        double ms;
        double m = run(st, d, wgs, iters, 0, 8, 0, fs, 0, df, &ms);
        printf("%4u workgroups, back to back %16s %10.0f %10.3f\n", wgs, "", m, ms);
        m = run(st, d, wgs, iters, 0, 8, 3000, fs, 0, df, &ms);
        printf("%4u workgroups, 3 ms idle before each %8s %10.0f %10.3f\n", wgs, "", m, ms);
        m = run(st, d, wgs, iters, 0, 6, 20000, fs, 0, df, &ms);
        printf("%4u workgroups, 20 ms idle before each %7s %10.0f %10.3f\n", wgs, "", m, ms);
    }
    for (unsigned fill : {64u, 128u, 192u}) {
        double ms;
        double m = run(st, d, 64, iters, 100 * 1024, 8, 3000, fs, fill, df, &ms);
        printf("  64 workgroups + %3u filler workgroups, 3 ms idle %10.0f %10.3f\n", fill, m, ms);
    }
    return 0;
}
