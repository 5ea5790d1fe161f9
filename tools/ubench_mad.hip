// tools/ubench_mad.hip -- does the issue rate of v_mad_u64_u32 depend on WHERE its operands live?
// Whole loops in assembly with physical registers: VGPR bank (index mod 4) of the two factors and of the 64-bit
// accumulator, vcc or an SGPR pair as the carry sink, a scalar register as one factor (VOP3 takes no literal on gfx950), one dependent chain
// or eight independent ones. Prints cycles per wave-instruction per SIMD at the 2.4 GHz nominal clock.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mad.hip -o /tmp/ubench_mad
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 16384;  // x 8 instructions

#define LOOP_HEAD "s_movk_i32 s20, 0x400\n1:\n"  // x 16 copies of the 8-instruction body: the taken branch must not be what is measured
#define LOOP_TAIL "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\n"
#define CLOB "s20", "s22", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "scc", "vcc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", \
             "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
             "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39"

#define INIT "v_mov_b32 v40, %1\nv_mov_b32 v41, %2\nv_mov_b32 v42, %1\nv_mov_b32 v43, %2\nv_mov_b32 v44, %1\nv_mov_b32 v45, %2\nv_mov_b32 v46, %1\nv_mov_b32 v47, %2\nv_mov_b32 v48, %1\nv_mov_b32 v49, %2\nv_mov_b32 v50, %1\nv_mov_b32 v51, %2\nv_mov_b32 v52, %1\nv_mov_b32 v53, %2\nv_mov_b32 v54, %1\nv_mov_b32 v55, %2\nv_mov_b32 v56, %1\nv_mov_b32 v57, %2\nv_mov_b32 v58, %1\nv_mov_b32 v59, %2\nv_mov_b32 v60, %1\nv_mov_b32 v61, %2\nv_mov_b32 v62, %1\nv_mov_b32 v63, %2\nv_mov_b32 v64, %1\nv_mov_b32 v65, %2\nv_mov_b32 v66, %1\nv_mov_b32 v67, %2\nv_mov_b32 v68, %1\nv_mov_b32 v69, %2\nv_mov_b32 v70, %1\nv_mov_b32 v71, %2\ns_mov_b32 s22, 0x0abcdef1\n" \
             "v_mov_b32 v8, %1\nv_mov_b32 v9, %2\nv_mov_b32 v10, %1\nv_mov_b32 v11, %2\nv_mov_b32 v12, %1\nv_mov_b32 v13, %2\nv_mov_b32 v14, %1\nv_mov_b32 v15, %2\n" \
             "v_mov_b32 v16, %1\nv_mov_b32 v17, %2\nv_mov_b32 v18, %1\nv_mov_b32 v19, %2\nv_mov_b32 v20, %1\nv_mov_b32 v21, %2\nv_mov_b32 v22, %1\nv_mov_b32 v23, %2\n" \
             "v_mov_b32 v24, %1\nv_mov_b32 v25, %2\nv_mov_b32 v26, %1\nv_mov_b32 v27, %2\nv_mov_b32 v28, %1\nv_mov_b32 v29, %2\nv_mov_b32 v30, %1\nv_mov_b32 v31, %2\n" \
             "v_mov_b32 v32, %1\nv_mov_b32 v33, %2\nv_mov_b32 v34, %1\nv_mov_b32 v35, %2\nv_mov_b32 v36, %1\nv_mov_b32 v37, %2\nv_mov_b32 v38, %1\nv_mov_b32 v39, %2\n"
#define FINI "v_xor_b32 %0, v8, v10\nv_xor_b32 %0, %0, v12\nv_xor_b32 %0, %0, v14\nv_xor_b32 %0, %0, v18\nv_xor_b32 %0, %0, v22\nv_xor_b32 %0, %0, v26\nv_xor_b32 %0, %0, v30\nv_xor_b32 %0, %0, v34\nv_xor_b32 %0, %0, v38\n" \
             "v_xor_b32 %0, %0, v9\nv_xor_b32 %0, %0, v11\nv_xor_b32 %0, %0, v15\nv_xor_b32 %0, %0, v19\n"

#define KERNEL(NAME, BODY)                                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) {                     \
        uint32_t a = seed ^ threadIdx.x, b = (seed * 2654435761u) | 1u, r;                          \
        asm volatile(INIT LOOP_HEAD BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY LOOP_TAIL FINI : "=&v"(r) : "v"(a), "v"(b) : CLOB);        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                             \
    }

// M(dst pair, carry sink, factor 0, factor 1, addend pair)
#define M(D, S, A, B, C) "v_mad_u64_u32 " D ", " S ", " A ", " B ", " C "\n"

// mixed banks: accumulators alternate between banks (0,1) and (2,3); factors in banks 0 and 1
KERNEL(k_mixed, M("v[8:9]", "vcc", "v40", "v41", "v[8:9]") M("v[10:11]", "vcc", "v40", "v41", "v[10:11]") M("v[12:13]", "vcc", "v40", "v41", "v[12:13]")
       M("v[14:15]", "vcc", "v40", "v41", "v[14:15]") M("v[16:17]", "vcc", "v40", "v41", "v[16:17]") M("v[18:19]", "vcc", "v40", "v41", "v[18:19]")
       M("v[20:21]", "vcc", "v40", "v41", "v[20:21]") M("v[22:23]", "vcc", "v40", "v41", "v[22:23]"))
// no two operands in one bank: factors in banks 0, 1; accumulators in banks 2, 3
KERNEL(k_distinct, M("v[10:11]", "vcc", "v40", "v41", "v[10:11]") M("v[14:15]", "vcc", "v40", "v41", "v[14:15]") M("v[18:19]", "vcc", "v40", "v41", "v[18:19]")
       M("v[22:23]", "vcc", "v40", "v41", "v[22:23]") M("v[26:27]", "vcc", "v40", "v41", "v[26:27]") M("v[30:31]", "vcc", "v40", "v41", "v[30:31]")
       M("v[34:35]", "vcc", "v40", "v41", "v[34:35]") M("v[38:39]", "vcc", "v40", "v41", "v[38:39]"))
// everything in banks 0 and 1: both factors in bank 0, accumulators in banks (0,1)
KERNEL(k_clash, M("v[8:9]", "vcc", "v40", "v44", "v[8:9]") M("v[12:13]", "vcc", "v40", "v44", "v[12:13]") M("v[16:17]", "vcc", "v40", "v44", "v[16:17]")
       M("v[20:21]", "vcc", "v40", "v44", "v[20:21]") M("v[24:25]", "vcc", "v40", "v44", "v[24:25]") M("v[28:29]", "vcc", "v40", "v44", "v[28:29]")
       M("v[32:33]", "vcc", "v40", "v44", "v[32:33]") M("v[36:37]", "vcc", "v40", "v44", "v[36:37]"))
// as k_distinct, the carry goes to a different SGPR pair each time
KERNEL(k_distinct_sgpr_sink, M("v[10:11]", "s[40:41]", "v40", "v41", "v[10:11]") M("v[14:15]", "s[42:43]", "v40", "v41", "v[14:15]") M("v[18:19]", "s[44:45]", "v40", "v41", "v[18:19]")
       M("v[22:23]", "s[46:47]", "v40", "v41", "v[22:23]") M("v[26:27]", "s[48:49]", "v40", "v41", "v[26:27]") M("v[30:31]", "s[50:51]", "v40", "v41", "v[30:31]")
       M("v[34:35]", "s[52:53]", "v40", "v41", "v[34:35]") M("v[38:39]", "s[54:55]", "v40", "v41", "v[38:39]"))
// as k_distinct, one factor is a scalar register (the modulus limbs of the reduction half could live there)
KERNEL(k_distinct_sgpr_factor, M("v[10:11]", "vcc", "v40", "s22", "v[10:11]") M("v[14:15]", "vcc", "v40", "s22", "v[14:15]") M("v[18:19]", "vcc", "v40", "s22", "v[18:19]")
       M("v[22:23]", "vcc", "v40", "s22", "v[22:23]") M("v[26:27]", "vcc", "v40", "s22", "v[26:27]") M("v[30:31]", "vcc", "v40", "s22", "v[30:31]")
       M("v[34:35]", "vcc", "v40", "s22", "v[34:35]") M("v[38:39]", "vcc", "v40", "s22", "v[38:39]"))
// ONE dependent chain (a column of the product: every multiply-add feeds the next)
KERNEL(k_chain, M("v[10:11]", "vcc", "v40", "v41", "v[10:11]") M("v[10:11]", "vcc", "v40", "v41", "v[10:11]") M("v[10:11]", "vcc", "v40", "v41", "v[10:11]")
       M("v[10:11]", "vcc", "v40", "v41", "v[10:11]") M("v[10:11]", "vcc", "v40", "v41", "v[10:11]") M("v[10:11]", "vcc", "v40", "v41", "v[10:11]")
       M("v[10:11]", "vcc", "v40", "v41", "v[10:11]") M("v[10:11]", "vcc", "v40", "v41", "v[10:11]"))
// two interleaved chains
KERNEL(k_chain2, M("v[10:11]", "vcc", "v40", "v41", "v[10:11]") M("v[14:15]", "vcc", "v40", "v41", "v[14:15]") M("v[10:11]", "vcc", "v40", "v41", "v[10:11]")
       M("v[14:15]", "vcc", "v40", "v41", "v[14:15]") M("v[10:11]", "vcc", "v40", "v41", "v[10:11]") M("v[14:15]", "vcc", "v40", "v41", "v[14:15]")
       M("v[10:11]", "vcc", "v40", "v41", "v[10:11]") M("v[14:15]", "vcc", "v40", "v41", "v[14:15]"))
// factors change with every instruction, as in a product column (a_i * b_(k-i)): eight accumulators
KERNEL(k_varied, M("v[10:11]", "vcc", "v40", "v63", "v[10:11]") M("v[14:15]", "vcc", "v41", "v62", "v[14:15]") M("v[18:19]", "vcc", "v42", "v61", "v[18:19]") M("v[22:23]", "vcc", "v43", "v60", "v[22:23]") M("v[26:27]", "vcc", "v44", "v59", "v[26:27]") M("v[30:31]", "vcc", "v45", "v58", "v[30:31]") M("v[34:35]", "vcc", "v46", "v57", "v[34:35]") M("v[38:39]", "vcc", "v47", "v56", "v[38:39]"))
// the same on ONE accumulator: a product column exactly
KERNEL(k_varied_chain, M("v[10:11]", "vcc", "v40", "v63", "v[10:11]") M("v[10:11]", "vcc", "v41", "v62", "v[10:11]") M("v[10:11]", "vcc", "v42", "v61", "v[10:11]") M("v[10:11]", "vcc", "v43", "v60", "v[10:11]") M("v[10:11]", "vcc", "v44", "v59", "v[10:11]") M("v[10:11]", "vcc", "v45", "v58", "v[10:11]") M("v[10:11]", "vcc", "v46", "v57", "v[10:11]") M("v[10:11]", "vcc", "v47", "v56", "v[10:11]"))
// one factor fixed for the whole body, the other changes (operand scanning order)
KERNEL(k_one_fixed_chain, M("v[10:11]", "vcc", "v40", "v63", "v[10:11]") M("v[10:11]", "vcc", "v40", "v62", "v[10:11]") M("v[10:11]", "vcc", "v40", "v61", "v[10:11]") M("v[10:11]", "vcc", "v40", "v60", "v[10:11]") M("v[10:11]", "vcc", "v40", "v59", "v[10:11]") M("v[10:11]", "vcc", "v40", "v58", "v[10:11]") M("v[10:11]", "vcc", "v40", "v57", "v[10:11]") M("v[10:11]", "vcc", "v40", "v56", "v[10:11]"))
// 32-bit reference points, same frame
#define A3(D) "v_add3_u32 " D ", " D ", v40, v41\n"
KERNEL(k_add3, A3("v10") A3("v14") A3("v18") A3("v22") A3("v26") A3("v30") A3("v34") A3("v38"))
#define ML(D) "v_mul_lo_u32 " D ", " D ", v41\n"
KERNEL(k_mul_lo, ML("v10") ML("v14") ML("v18") ML("v22") ML("v26") ML("v30") ML("v34") ML("v38"))
// multiply-adds and simple 32-bit operations alternating (do they share the issue slot one for one?)
#define AD(D) "v_add_u32 " D ", " D ", v41\n"
KERNEL(k_mad_add_mix, M("v[10:11]", "vcc", "v40", "v41", "v[10:11]") AD("v8") M("v[14:15]", "vcc", "v40", "v41", "v[14:15]") AD("v12")
       M("v[18:19]", "vcc", "v40", "v41", "v[18:19]") AD("v16") M("v[22:23]", "vcc", "v40", "v41", "v[22:23]") AD("v20"))

template <class Kern>
static int run(const char *name, Kern k, int blocks_per_cu, uint32_t *d_out, int n_cu) {
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
    double wave_instr = (double)grid * 4 * ITERS * 8;
    double cyc = (best * 1e-3) * 2.4e9 * (n_cu * 4) / wave_instr;
    printf("{\"ubench\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.3f}\n", name,
           blocks_per_cu, best, cyc);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int n_cu = prop.multiProcessorCount;
    uint32_t *d_out;
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4));
    for (int w : {1, 2, 4, 8}) {
        run("mad mixed banks", k_mixed, w, d_out, n_cu);
        run("mad distinct banks", k_distinct, w, d_out, n_cu);
        run("mad clashing banks", k_clash, w, d_out, n_cu);
        run("mad distinct banks, sgpr carry sinks", k_distinct_sgpr_sink, w, d_out, n_cu);
        run("mad distinct banks, sgpr factor", k_distinct_sgpr_factor, w, d_out, n_cu);
        run("mad one dependent chain", k_chain, w, d_out, n_cu);
        run("mad two interleaved chains", k_chain2, w, d_out, n_cu);
        run("mad varied factors, eight accumulators", k_varied, w, d_out, n_cu);
        run("mad varied factors, one chain", k_varied_chain, w, d_out, n_cu);
        run("mad one factor fixed, one chain", k_one_fixed_chain, w, d_out, n_cu);
        run("add3", k_add3, w, d_out, n_cu);
        run("mul_lo", k_mul_lo, w, d_out, n_cu);
        run("mad + add_u32 alternating (per instruction)", k_mad_add_mix, w, d_out, n_cu);
    }
    return 0;
}
