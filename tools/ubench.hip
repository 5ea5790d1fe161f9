// tools/ubench.hip -- instruction-rate micro-benchmarks for the integer roofline of the MSM kernel
// ("peak = CUs x 64 lanes x clock x issue rate of v_mad_u64_u32, measured by a
// micro-benchmark on the box rather than assumed"). Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "../lambdaworks_kzg_amd/csrc/g1.cuh"

using namespace lwk;
typedef unsigned long long u64;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int UNROLL = 8;  // independent chains per lane

#define DEF_KERNEL(NAME, DECL, BODY, SINK)                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) {      \
        uint32_t a = seed ^ threadIdx.x, b = (seed * 2654435761u) | 1u;             \
        DECL;                                                                        \
        for (int it = 0; it < ITERS; it++) {                                         \
            BODY;                                                                    \
        }                                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = SINK;                           \
    }

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define D64(i) u64 c##i = a + i;
#define MAD(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c##i) : "v"(a), "v"(b) : "vcc");
#define S64(i) ^ (uint32_t)c##i ^ (uint32_t)(c##i >> 32)
DEF_KERNEL(k_mad_u64_u32, R8(D64), R8(MAD), 0 R8(S64))

#define D32(i) uint32_t d##i = a + i;
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(d##i) : "v"(b));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(d##i) : "v"(b));
#define ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(d##i) : "v"(b));
#define MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(d##i) : "v"(b));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(d##i) : "v"(b));
#define S32(i) ^ d##i
DEF_KERNEL(k_mul_lo_u32, R8(D32), R8(MULLO), 0 R8(S32))
DEF_KERNEL(k_mul_hi_u32, R8(D32), R8(MULHI), 0 R8(S32))
DEF_KERNEL(k_add_u32, R8(D32), R8(ADD32), 0 R8(S32))
DEF_KERNEL(k_mad_u32_u24, R8(D32), R8(MAD24), 0 R8(S32))
DEF_KERNEL(k_add3_u32, R8(D32), R8(ADD3), 0 R8(S32))

#define DF64(i) double f##i = (double)(a + i);
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(f##i) : "v"(fb));
#define SF64(i) ^ (uint32_t)f##i
DEF_KERNEL(k_fma_f64, double fb = 1.0000001; R8(DF64), R8(FMA64), 0 R8(SF64))

#define LSHLADD(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c##i) : "v"(cb));
DEF_KERNEL(k_lshl_add_u64, u64 cb = b; R8(D64), R8(LSHLADD), 0 R8(S64))

// carry pair with the gfx950 hazard padding the compiler emits (v_add_co ; s_nop 1 ; v_addc_co)
#define ADDC(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\ts_nop 1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(d##i) : "v"(b) : "vcc");
DEF_KERNEL(k_addco_nop_addc, R8(D32), R8(ADDC), 0 R8(S32))

// the product's Montgomery multiplication (function call), dependent chain per lane
__global__ __launch_bounds__(256) void k_fp_mul(uint32_t *out, uint32_t seed) {
    Fp x, y;
    for (int i = 0; i < 12; i++) { x.l[i] = seed + threadIdx.x * 7 + i; y.l[i] = seed * 3 + i; }
    x.l[11] &= 0x0fffffff; y.l[11] &= 0x0fffffff;
    for (int it = 0; it < 256; it++) x = x * y;
    uint32_t s = 0;
    for (int i = 0; i < 12; i++) s ^= x.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fp_mul_inl(uint32_t *out, uint32_t seed) {
    Fp x, y;
    for (int i = 0; i < 12; i++) { x.l[i] = seed + threadIdx.x * 7 + i; y.l[i] = seed * 3 + i; }
    x.l[11] &= 0x0fffffff; y.l[11] &= 0x0fffffff;
    for (int it = 0; it < 256; it++) x = fe_mul_inl<FpParams>(x, y);
    uint32_t s = 0;
    for (int i = 0; i < 12; i++) s ^= x.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_xyzz_madd(uint32_t *out, uint32_t seed) {
    G1Affine q;
    for (int i = 0; i < 12; i++) { q.x.l[i] = seed + threadIdx.x * 7 + i; q.y.l[i] = seed * 3 + i; }
    q.x.l[11] &= 0x0fffffff; q.y.l[11] &= 0x0fffffff;
    G1Xyzz acc = G1Xyzz::from_affine(q.x, q.y);
    acc.x.l[0] ^= 5;
    for (int it = 0; it < 64; it++) acc = xyzz_madd(acc, q);   // not on the curve: only the arithmetic cost matters
    uint32_t s = 0;
    for (int i = 0; i < 12; i++) s ^= acc.x.l[i] ^ acc.y.l[i] ^ acc.zz.l[i] ^ acc.zzz.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_f29_mul(uint32_t *out, uint32_t seed) {
    F29<2> x, y;
    for (int i = 0; i < 14; i++) { x.l[i] = (seed + threadIdx.x * 7 + i) & P29::MASK; y.l[i] = (seed * 3 + i) & P29::MASK; }
    x.l[13] = 5; y.l[13] = 7;
    for (int it = 0; it < 256; it++) x = x * y;
    uint32_t s = 0;
    for (int i = 0; i < 14; i++) s ^= x.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_xyzz_madd29(uint32_t *out, uint32_t seed) {
    G1Affine29 q;
    for (int i = 0; i < 14; i++) { q.x.l[i] = (seed + threadIdx.x * 7 + i) & P29::MASK; q.y.l[i] = (seed * 3 + i) & P29::MASK; }
    q.x.l[13] = 5; q.y.l[13] = 7;
    G1Xyzz29 acc = G1Xyzz29::from_affine(q.x, q.y);
    acc.x.l[0] ^= 5;
    for (int it = 0; it < 64; it++) acc = xyzz_madd(acc, q.x, cneg(q.y, (it & 1) != 0));
    uint32_t s = 0;
    for (int i = 0; i < 14; i++) s ^= acc.x.l[i] ^ acc.y.l[i] ^ acc.zz.l[i] ^ acc.zzz.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class Kern>
static int run(const char *name, Kern k, double ops_per_lane, int blocks_per_cu, uint32_t *d_out, int n_cu) {
    int grid = n_cu * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d_out, 12345u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d_out, 12345u + rep);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double lanes = (double)grid * 256;
    double rate = lanes * ops_per_lane / (best * 1e-3);
    // cycles per wave-instruction per SIMD at 2.4 GHz: SIMDs = n_cu*4
    double wave_instr = lanes / 64 * ops_per_lane;
    double cyc = (best * 1e-3) * 2.4e9 * (n_cu * 4) / wave_instr;
    printf("{\"ubench\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"lane_ops_per_s\": %.4e, \"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.2f}\n",
           name, blocks_per_cu, best, rate, cyc);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int n_cu = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.name, n_cu, prop.clockRate);
    uint32_t *d_out;
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 16 * 256 * 4));
    double per = (double)ITERS * UNROLL;
    for (int w : {8}) {
        run("v_add_u32", k_add_u32, per, w, d_out, n_cu);
        run("v_add3_u32", k_add3_u32, per, w, d_out, n_cu);
        run("v_mad_u64_u32", k_mad_u64_u32, per, w, d_out, n_cu);
        run("v_mul_lo_u32", k_mul_lo_u32, per, w, d_out, n_cu);
        run("v_mul_hi_u32", k_mul_hi_u32, per, w, d_out, n_cu);
        run("v_mad_u32_u24", k_mad_u32_u24, per, w, d_out, n_cu);
        run("v_fma_f64", k_fma_f64, per, w, d_out, n_cu);
        run("v_lshl_add_u64", k_lshl_add_u64, per, w, d_out, n_cu);
        run("v_add_co+s_nop1+v_addc_co(pair)", k_addco_nop_addc, per, w, d_out, n_cu);
    }
    for (int w : {1, 2, 4}) {
        run("fp_mul_call", k_fp_mul, 256, w, d_out, n_cu);
        run("fp_mul_inline", k_fp_mul_inl, 256, w, d_out, n_cu);
        run("xyzz_madd", k_xyzz_madd, 64, w, d_out, n_cu);
        run("f29_mul_call", k_f29_mul, 256, w, d_out, n_cu);
        run("xyzz_madd_f29", k_xyzz_madd29, 64, w, d_out, n_cu);
    }
    return 0;
}
