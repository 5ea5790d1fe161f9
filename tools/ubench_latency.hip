// tools/ubench_latency.hip -- what does ONE wave per SIMD issue? The latency kernels of this library (a single commitment,
// the folds, the final inversion) are dependent chains on lone waves; tools/ubench_mad.hip and ubench_sustain.hip measure the
// opposite regime (eight / two waves per SIMD, throughput). This program times N back-to-back instructions of one kind on a lone
// wave (one workgroup of 64 lanes, and 1024 of them: one per SIMD) with the shader clock (clock64) and the 100 MHz wall clock
// (wall_clock64) read inside the kernel, and prints cycles per instruction:
//   mad1 .. mad4     v_mad_u64_u32, 1 .. 4 interleaved dependent chains
//   madcol           the product-scanning column pattern with TWO column accumulators and a v_lshl_add_u64 merge per column
//   add1 / add4      v_add_u32, dependent / four chains
//   dpp1 / dpp4      v_mov_b32_dpp quad_perm, dependent / four independent
//   dppmad           14 quad_perm moves feeding a 14-instruction multiply-add chain (the cooperative addition's exchange)
//   bperm14          14 ds_bpermute_b32 + s_waitcnt lgkmcnt(0) (the cross-quad fetch of a point coordinate)
//   salu_add/mul     s_add_u32 / s_mul_i32 dependent chains
//   readlane         v_readlane_b32 -> s_add -> v_mov round trip
//   lds_rt           ds_write_b128 x4 + ds_read_b128 x4 + wait (a 56-byte coordinate through LDS)
//   atomic_rt        global atomic add (returning) round trip; store + fence + load of another line
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_latency.hip -o tools/bin/ubench_latency
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R32(x) R16(x) R16(x)
#define R64(x) R32(x) R32(x)
#define R128(x) R64(x) R64(x)
#define R256(x) R128(x) R128(x)

#define M(D, A, B) "v_mad_u64_u32 " D ", vcc, " A ", " B ", " D "\n"
#define VCLOB "vcc", "scc", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", \
              "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "s20", "s21", "s22", "s23"
#define INIT "v_mov_b32 v40, %1\nv_mov_b32 v41, %2\nv_mov_b32 v8, %1\nv_mov_b32 v9, 0\nv_mov_b32 v10, %2\nv_mov_b32 v11, 0\nv_mov_b32 v12, %1\nv_mov_b32 v13, 0\n" \
             "v_mov_b32 v14, %2\nv_mov_b32 v15, 0\nv_mov_b32 v16, %1\nv_mov_b32 v17, %2\nv_mov_b32 v18, %1\nv_mov_b32 v19, %2\nv_mov_b32 v20, %1\nv_mov_b32 v21, %2\n" \
             "v_mov_b32 v22, %1\nv_mov_b32 v23, %2\nv_mov_b32 v24, %1\nv_mov_b32 v25, %2\nv_mov_b32 v26, %1\nv_mov_b32 v27, %2\nv_mov_b32 v28, %1\nv_mov_b32 v29, %2\n" \
             "v_mov_b32 v30, %1\nv_mov_b32 v31, %2\nv_mov_b32 v32, %1\nv_mov_b32 v33, %2\nv_mov_b32 v34, %1\nv_mov_b32 v35, %2\nv_mov_b32 v36, %1\nv_mov_b32 v37, %2\n" \
             "v_mov_b32 v38, %1\nv_mov_b32 v39, %2\nv_mov_b32 v42, %1\nv_mov_b32 v43, %2\nv_mov_b32 v44, %1\nv_mov_b32 v45, %2\ns_mov_b32 s20, 3\ns_mov_b32 s21, 5\n"
#define FINI "v_xor_b32 %0, v8, v10\nv_xor_b32 %0, %0, v12\nv_xor_b32 %0, %0, v14\nv_xor_b32 %0, %0, v16\nv_xor_b32 %0, %0, v30\nv_xor_b32 %0, %0, v20\nv_xor_b32 %0, %0, s20\n"

struct Stamp {
    long long clk, wall;
};

// every kernel: out[wg] = {clock64 delta, wall_clock64 delta}; INSTR = instructions of the timed kind inside BODY
#define KERNEL(NAME, BODY)                                                                                         \
    __global__ __launch_bounds__(64) void NAME(Stamp *out, uint32_t *sink, uint32_t seed, uint32_t *mem) {         \
        __shared__ uint32_t lds[1024];                                                                             \
        lds[threadIdx.x] = seed;                                                                                   \
        uint32_t a = (seed ^ threadIdx.x) & 0xfffffff, b = ((seed * 2654435761u) | 1u) & 0xfffffff, r;            \
        uint32_t laddr = threadIdx.x * 16;                                                                         \
        uint32_t *gaddr = mem + blockIdx.x * 64;                                                                   \
        long long c0 = clock64(), w0 = wall_clock64();                                                             \
        asm volatile(INIT ".p2align 3\n" BODY FINI : "=&v"(r) : "v"(a), "v"(b), "v"(laddr), "v"(gaddr) : VCLOB, "memory"); \
        long long c1 = clock64(), w1 = wall_clock64();                                                             \
        sink[blockIdx.x * 64 + threadIdx.x] = r + lds[(threadIdx.x + 1) & 63];                                     \
        if (threadIdx.x == 0) out[blockIdx.x] = Stamp{c1 - c0, w1 - w0};                                           \
    }

KERNEL(k_mad1, R256(R16(M("v[8:9]", "v40", "v41"))))
KERNEL(k_mad2, R256(R8(M("v[8:9]", "v40", "v41") M("v[10:11]", "v41", "v40"))))
KERNEL(k_mad3, R256(R4(M("v[8:9]", "v40", "v41") M("v[10:11]", "v41", "v40") M("v[12:13]", "v40", "v40") M("v[8:9]", "v41", "v41")
                       M("v[10:11]", "v40", "v41") M("v[12:13]", "v41", "v40"))))   // 6 per R4 -> 24 per R16-equivalent: counted below
KERNEL(k_mad4, R256(R4(M("v[8:9]", "v40", "v41") M("v[10:11]", "v41", "v40") M("v[12:13]", "v40", "v40") M("v[14:15]", "v41", "v41"))))
// product-scanning with two column accumulators: 14 multiply-adds into one, then its shift and the merge into the other, which has
// been collecting the next column's products meanwhile
#define COLPAIR R4(M("v[8:9]", "v40", "v41") M("v[10:11]", "v41", "v40")) R2(M("v[8:9]", "v40", "v41") M("v[10:11]", "v41", "v40")) \
    M("v[8:9]", "v40", "v41") M("v[10:11]", "v41", "v40") "v_lshrrev_b64 v[8:9], 28, v[8:9]\n" M("v[10:11]", "v41", "v40") "v_lshl_add_u64 v[10:11], v[8:9], 0, v[10:11]\n" \
    "v_and_b32 v8, 0xfffffff, v10\nv_mov_b32 v9, 0\n"
KERNEL(k_madcol, R128(COLPAIR))
KERNEL(k_add1, R256(R16("v_add_u32 v8, v8, v40\n")))
KERNEL(k_add4, R256(R4("v_add_u32 v8, v8, v40\nv_add_u32 v10, v10, v40\nv_add_u32 v12, v12, v40\nv_add_u32 v14, v14, v40\n")))
KERNEL(k_dpp1, R256(R16("v_mov_b32_dpp v8, v8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n")))
KERNEL(k_dpp4, R256(R4("v_mov_b32_dpp v8, v16 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp v10, v17 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                       "v_mov_b32_dpp v12, v18 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp v14, v19 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n")))
#define D(d, s) "v_mov_b32_dpp " d ", " s " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define DPP14 D("v16", "v30") D("v17", "v31") D("v18", "v32") D("v19", "v33") D("v20", "v34") D("v21", "v35") D("v22", "v36") D("v23", "v37") D("v24", "v38") D("v25", "v39") \
    D("v26", "v42") D("v27", "v43") D("v28", "v44") D("v29", "v45")
#define MAD14 M("v[8:9]", "v16", "v29") M("v[8:9]", "v17", "v28") M("v[8:9]", "v18", "v27") M("v[8:9]", "v19", "v26") M("v[8:9]", "v20", "v25") M("v[8:9]", "v21", "v24") \
    M("v[8:9]", "v22", "v23") M("v[8:9]", "v23", "v22") M("v[8:9]", "v24", "v21") M("v[8:9]", "v25", "v20") M("v[8:9]", "v26", "v19") M("v[8:9]", "v27", "v18") \
    M("v[8:9]", "v28", "v17") M("v[8:9]", "v29", "v16")
KERNEL(k_dppmad, R128(DPP14 MAD14 "v_and_b32 v30, 0xfffffff, v8\n"))
#define BP(d, s) "ds_bpermute_b32 " d ", v46, " s "\n"
#define BPERM14 BP("v16", "v30") BP("v17", "v31") BP("v18", "v32") BP("v19", "v33") BP("v20", "v34") BP("v21", "v35") BP("v22", "v36") BP("v23", "v37") BP("v24", "v38") \
    BP("v25", "v39") BP("v26", "v42") BP("v27", "v43") BP("v28", "v44") BP("v29", "v45") "s_waitcnt lgkmcnt(0)\nv_add_u32 v30, v16, v29\n"
KERNEL(k_salu_add, R256(R16("s_add_u32 s20, s20, s21\n")))
KERNEL(k_salu_mul, R256(R16("s_mul_i32 s20, s20, s21\n")))
KERNEL(k_readlane, R256(R4("v_readlane_b32 s22, v8, 3\ns_nop 0\ns_add_u32 s22, s22, s21\nv_mov_b32 v8, s22\n")))
KERNEL(k_lds_rt, R256("ds_write_b128 %3, v[16:19]\nds_write_b128 %3, v[20:23] offset:1024\nds_write_b128 %3, v[24:27] offset:2048\nds_write_b64 %3, v[28:29] offset:3072\n"
                      "s_waitcnt lgkmcnt(0)\nds_read_b128 v[16:19], %3\nds_read_b128 v[20:23], %3 offset:1024\nds_read_b128 v[24:27], %3 offset:2048\n"
                      "ds_read_b64 v[28:29], %3 offset:3072\ns_waitcnt lgkmcnt(0)\nv_add_u32 v16, v16, v29\n"))
KERNEL(k_atomic_rt, R256("global_atomic_add v8, %4, v40, off sc0\ns_waitcnt vmcnt(0)\n"))
KERNEL(k_store_fence_load, R256("global_store_dword %4, v8, off sc0 sc1\ns_waitcnt vmcnt(0)\nbuffer_wbl2 sc1\ns_waitcnt vmcnt(0)\nglobal_load_dword v8, %4, off offset:256 sc0 sc1\ns_waitcnt vmcnt(0)\n"))

__global__ __launch_bounds__(64) void k_bperm14(Stamp *out, uint32_t *sink, uint32_t seed, uint32_t *mem) {
    uint32_t a = (seed ^ threadIdx.x) & 0xfffffff, b = ((seed * 2654435761u) | 1u) & 0xfffffff, r;
    uint32_t baddr = ((threadIdx.x + 4) & 63) * 4;
    long long c0 = clock64(), w0 = wall_clock64();
    asm volatile(INIT "v_mov_b32 v46, %3\n.p2align 3\n" R256(BPERM14) FINI : "=&v"(r) : "v"(a), "v"(b), "v"(baddr) : VCLOB, "v46", "memory");
    long long c1 = clock64(), w1 = wall_clock64();
    sink[blockIdx.x * 64 + threadIdx.x] = r;
    if (threadIdx.x == 0) out[blockIdx.x] = Stamp{c1 - c0, w1 - w0};
}

typedef void (*Kern)(Stamp *, uint32_t *, uint32_t, uint32_t *);
struct Case {
    const char *name;
    Kern k;
    int instr;        // timed instructions (or round trips) in the body
    const char *unit;
};

int main(int argc, char **argv) {
    Case cases[] = {
        {"mad1 (one dependent chain of v_mad_u64_u32)", k_mad1, 4096, "instruction"},
        {"mad2 (two interleaved chains)", k_mad2, 4096, "instruction"},
        {"mad3 (three interleaved chains)", k_mad3, 6144, "instruction"},
        {"mad4 (four interleaved chains)", k_mad4, 4096, "instruction"},
        {"madcol (two column accumulators, 30 multiply-adds + shift + 64-bit merge + mask + mov per column pair)", k_madcol, 128 * 35, "instruction"},
        {"add1 (dependent v_add_u32)", k_add1, 4096, "instruction"},
        {"add4 (four chains of v_add_u32)", k_add4, 4096, "instruction"},
        {"dpp1 (dependent v_mov_b32_dpp quad_perm)", k_dpp1, 4096, "instruction"},
        {"dpp4 (independent v_mov_b32_dpp quad_perm)", k_dpp4, 4096, "instruction"},
        {"dppmad (14 quad_perm moves + a 14-long multiply-add chain on them + mask)", k_dppmad, 128 * 29, "instruction"},
        {"bperm14 (14 ds_bpermute_b32 + wait + add)", k_bperm14, 256, "batch of 14"},
        {"salu_add (dependent s_add_u32)", k_salu_add, 4096, "instruction"},
        {"salu_mul (dependent s_mul_i32)", k_salu_mul, 4096, "instruction"},
        {"readlane (v_readlane_b32 -> s_add_u32 -> v_mov_b32)", k_readlane, 1024, "round trip"},
        {"lds_rt (56 bytes per lane through LDS: 4 writes, wait, 4 reads, wait)", k_lds_rt, 256, "round trip"},
        {"atomic_rt (returning global atomic add, device scope)", k_atomic_rt, 256, "round trip"},
        {"store_fence_load (store, write-back, load of another line, all waited for)", k_store_fence_load, 256, "round trip"},
    };
    Stamp *out;
    uint32_t *sink, *mem;
    CHECK(hipMalloc(&out, 4096 * sizeof(Stamp)));
    CHECK(hipMalloc(&sink, 4096 * 64 * 4));
    CHECK(hipMalloc(&mem, 4096 * 64 * 4 + 4096));
    CHECK(hipMemset(mem, 0, 4096 * 64 * 4 + 4096));
    Stamp *h = (Stamp *)malloc(4096 * sizeof(Stamp));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grids[] = {1, 1024, 2048};
    for (const Case &c : cases) {
        for (int grid : grids) {
            float best_ms = 1e9f;
            double clk = 0, wall = 0;
            for (int rep = 0; rep < 5; rep++) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(c.k, dim3(grid), dim3(64), 0, 0, out, sink, 12345u + rep, mem);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best_ms) {
                    best_ms = ms;
                    CHECK(hipMemcpy(h, out, grid * sizeof(Stamp), hipMemcpyDeviceToHost));
                    clk = wall = 0;
                    for (int i = 0; i < grid; i++) {
                        clk += (double)h[i].clk;
                        wall += (double)h[i].wall;
                    }
                    clk /= grid;
                    wall /= grid;
                }
            }
            // wall_clock64 ticks at 100 MHz
            const double ns = wall * 10.0;
            printf("{\"case\": \"%s\", \"waves\": %d, \"per\": \"%s\", \"n\": %d, \"clock64_per\": %.2f, \"ns_per\": %.3f, \"clock64_per_ns\": %.3f, \"event_ms\": %.4f}\n",
                   c.name, grid, c.unit, c.instr, clk / c.instr, ns / c.instr, ns > 0 ? clk / ns : 0.0, best_ms);
        }
    }
    return 0;
}
