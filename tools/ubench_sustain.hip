// tools/ubench_sustain.hip -- what rate does a dense v_mad_u64_u32 stream hold when it is SUSTAINED, and what do random
// 128-byte gathers beside it cost? tools/ubench_mad.hip measures loops of half a millisecond (2.40-2.45 GHz by GRBM_GUI_ACTIVE);
// the accumulation kernel runs 9 ms per launch, back to back, at 1.98-2.05 GHz. This runs the same eight-accumulator loop with
// two waves per SIMD for a launch length given at run time, back to back for ~0.2 s, and prints ns per wave-instruction per SIMD:
//   plain      multiply-adds only
//   gather     the same plus one random 112-byte row (seven dwordx4 loads, one 128-byte line) per ~4256 instructions per lane,
//              out of a buffer of --gb gigabytes (default 128), waited for one body later -- the accumulation's memory pattern
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_sustain.hip -o /tmp/ubench_sustain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define M(D, A, B) "v_mad_u64_u32 " D ", vcc, " A ", " B ", " D "\n"
#define BODY8 M("v[8:9]", "v40", "v41") M("v[10:11]", "v40", "v41") M("v[12:13]", "v40", "v41") M("v[14:15]", "v40", "v41") \
              M("v[16:17]", "v40", "v41") M("v[18:19]", "v40", "v41") M("v[20:21]", "v40", "v41") M("v[22:23]", "v40", "v41")
#define BODY128 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8
#define BODY512 BODY128 BODY128 BODY128 BODY128
#define BODY4096 BODY512 BODY512 BODY512 BODY512 BODY512 BODY512 BODY512 BODY512
#define INIT "v_mov_b32 v40, %1\nv_mov_b32 v41, %2\n" \
             "v_mov_b32 v8, %1\nv_mov_b32 v9, %2\nv_mov_b32 v10, %1\nv_mov_b32 v11, %2\nv_mov_b32 v12, %1\nv_mov_b32 v13, %2\nv_mov_b32 v14, %1\nv_mov_b32 v15, %2\n" \
             "v_mov_b32 v16, %1\nv_mov_b32 v17, %2\nv_mov_b32 v18, %1\nv_mov_b32 v19, %2\nv_mov_b32 v20, %1\nv_mov_b32 v21, %2\nv_mov_b32 v22, %1\nv_mov_b32 v23, %2\n"
#define FINI "v_xor_b32 %0, v8, v10\nv_xor_b32 %0, %0, v12\nv_xor_b32 %0, %0, v14\nv_xor_b32 %0, %0, v16\nv_xor_b32 %0, %0, v18\nv_xor_b32 %0, %0, v20\nv_xor_b32 %0, %0, v22\n"
#define CLOB "s20", "scc", "vcc", "v40", "v41", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23"

// 232 registers as the accumulation kernel: two waves per SIMD and no more
#define PAD_CLOB "v231"

__global__ __launch_bounds__(256) void k_plain(uint32_t *out, uint32_t seed, uint32_t iters) {
    uint32_t a = seed ^ threadIdx.x, b = (seed * 2654435761u) | 1u, r;
    asm volatile(INIT "s_mov_b32 s20, %3\n1:\n.p2align 3\n" BODY4096 "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\n" FINI
                 : "=&v"(r) : "v"(a), "v"(b), "s"(iters) : CLOB, PAD_CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}


// ---- does the rate depend on the DATA? sixteen factor registers with random 28-bit (or 32-bit) contents
#define INIT16 "v_mov_b32 v40, %1\n" \
    "v_mul_lo_u32 v41, v40, %4\n v_add_u32 v41, 0x9e3779b9, v41\n v_mul_lo_u32 v42, v41, %4\n v_add_u32 v42, 0x9e3779b9, v42\n" \
    "v_mul_lo_u32 v43, v42, %4\n v_add_u32 v43, 0x9e3779b9, v43\n v_mul_lo_u32 v44, v43, %4\n v_add_u32 v44, 0x9e3779b9, v44\n" \
    "v_mul_lo_u32 v45, v44, %4\n v_add_u32 v45, 0x9e3779b9, v45\n v_mul_lo_u32 v46, v45, %4\n v_add_u32 v46, 0x9e3779b9, v46\n" \
    "v_mul_lo_u32 v47, v46, %4\n v_add_u32 v47, 0x9e3779b9, v47\n v_mul_lo_u32 v48, v47, %4\n v_add_u32 v48, 0x9e3779b9, v48\n" \
    "v_mul_lo_u32 v49, v48, %4\n v_add_u32 v49, 0x9e3779b9, v49\n v_mul_lo_u32 v50, v49, %4\n v_add_u32 v50, 0x9e3779b9, v50\n" \
    "v_mul_lo_u32 v51, v50, %4\n v_add_u32 v51, 0x9e3779b9, v51\n v_mul_lo_u32 v52, v51, %4\n v_add_u32 v52, 0x9e3779b9, v52\n" \
    "v_mul_lo_u32 v53, v52, %4\n v_add_u32 v53, 0x9e3779b9, v53\n v_mul_lo_u32 v54, v53, %4\n v_add_u32 v54, 0x9e3779b9, v54\n" \
    "v_mul_lo_u32 v55, v54, %4\n v_add_u32 v55, 0x9e3779b9, v55\n" \
    "v_lshrrev_b32 v40, %5, v40\n v_lshrrev_b32 v41, %5, v41\n v_lshrrev_b32 v42, %5, v42\n v_lshrrev_b32 v43, %5, v43\n" \
    "v_lshrrev_b32 v44, %5, v44\n v_lshrrev_b32 v45, %5, v45\n v_lshrrev_b32 v46, %5, v46\n v_lshrrev_b32 v47, %5, v47\n" \
    "v_lshrrev_b32 v48, %5, v48\n v_lshrrev_b32 v49, %5, v49\n v_lshrrev_b32 v50, %5, v50\n v_lshrrev_b32 v51, %5, v51\n" \
    "v_lshrrev_b32 v52, %5, v52\n v_lshrrev_b32 v53, %5, v53\n v_lshrrev_b32 v54, %5, v54\n v_lshrrev_b32 v55, %5, v55\n" \
    "v_mov_b32 v8, 0\nv_mov_b32 v9, 0\nv_mov_b32 v10, 0\nv_mov_b32 v11, 0\nv_mov_b32 v12, 0\nv_mov_b32 v13, 0\nv_mov_b32 v14, 0\nv_mov_b32 v15, 0\n" \
    "v_mov_b32 v16, 0\nv_mov_b32 v17, 0\nv_mov_b32 v18, 0\nv_mov_b32 v19, 0\nv_mov_b32 v20, 0\nv_mov_b32 v21, 0\nv_mov_b32 v22, 0\nv_mov_b32 v23, 0\n"
#define CLOB16_NOPAD CLOB, "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"
#define CLOB16 CLOB, PAD_CLOB, "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"
// both factors change with every instruction (product scanning: a_i b_(k-i))
#define V16 M("v[8:9]", "v40", "v55") M("v[10:11]", "v41", "v54") M("v[12:13]", "v42", "v53") M("v[14:15]", "v43", "v52") \
            M("v[16:17]", "v44", "v51") M("v[18:19]", "v45", "v50") M("v[20:21]", "v46", "v49") M("v[22:23]", "v47", "v48") \
            M("v[8:9]", "v48", "v47") M("v[10:11]", "v49", "v46") M("v[12:13]", "v50", "v45") M("v[14:15]", "v51", "v44") \
            M("v[16:17]", "v52", "v43") M("v[18:19]", "v53", "v42") M("v[20:21]", "v54", "v41") M("v[22:23]", "v55", "v40")
// one factor stays for eight instructions (operand scanning: a_i b_j, j = 0 .. 7)
#define F16 M("v[8:9]", "v40", "v48") M("v[10:11]", "v40", "v49") M("v[12:13]", "v40", "v50") M("v[14:15]", "v40", "v51") \
            M("v[16:17]", "v40", "v52") M("v[18:19]", "v40", "v53") M("v[20:21]", "v40", "v54") M("v[22:23]", "v40", "v55") \
            M("v[8:9]", "v41", "v48") M("v[10:11]", "v41", "v49") M("v[12:13]", "v41", "v50") M("v[14:15]", "v41", "v51") \
            M("v[16:17]", "v41", "v52") M("v[18:19]", "v41", "v53") M("v[20:21]", "v41", "v54") M("v[22:23]", "v41", "v55")
// the same two factors all the time, random contents
#define S16 M("v[8:9]", "v40", "v48") M("v[10:11]", "v40", "v48") M("v[12:13]", "v40", "v48") M("v[14:15]", "v40", "v48") \
            M("v[16:17]", "v40", "v48") M("v[18:19]", "v40", "v48") M("v[20:21]", "v40", "v48") M("v[22:23]", "v40", "v48") \
            M("v[8:9]", "v40", "v48") M("v[10:11]", "v40", "v48") M("v[12:13]", "v40", "v48") M("v[14:15]", "v40", "v48") \
            M("v[16:17]", "v40", "v48") M("v[18:19]", "v40", "v48") M("v[20:21]", "v40", "v48") M("v[22:23]", "v40", "v48")
#define X4(B) B B B B
#define X256(B) X4(X4(X4(X4(B))))
#define DATA_KERNEL(NAME, B16) DATA_KERNEL_C(NAME, B16, CLOB16, "")
#define DATA_KERNEL_C(NAME, B16, CL, MIS)                                                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed, uint32_t iters, uint32_t shift) {            \
        uint32_t a = (seed ^ (threadIdx.x * 2246822519u)) * 3266489917u, b = 0, r;                                          \
        asm volatile(INIT16 "s_mov_b32 s20, %3\n1:\n.p2align 3\n" MIS X256(B16) "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\n" FINI \
                     : "=&v"(r) : "v"(a), "v"(b), "s"(iters), "s"(747796405u), "s"(shift) : CL);                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                                     \
    }
DATA_KERNEL(k_varied, V16)
DATA_KERNEL(k_one_fixed, F16)
DATA_KERNEL(k_same, S16)
DATA_KERNEL_C(k_varied_small, V16, CLOB16_NOPAD, "")
DATA_KERNEL_C(k_varied_mis, V16, CLOB16, "s_nop 0\n")                // every multiply-add at 4 mod 8: one in eight straddles a 64-byte line
DATA_KERNEL_C(k_varied_small_mis, V16, CLOB16_NOPAD, "s_nop 0\n")   // 56 registers: up to eight waves per SIMD

// the random-operand stream WITH the accumulation's memory pattern: one random 112-byte row per 4096 instructions per lane, used one body later
__global__ __launch_bounds__(256) void k_varied_gather(uint32_t *out, uint32_t seed, uint32_t iters, uint32_t shift, const uint8_t *table, uint32_t row_mask) {
    uint32_t a = (seed ^ (threadIdx.x * 2246822519u)) * 3266489917u, b = 0, r;
    asm volatile(INIT16
                 "v_mov_b32 v56, %1\n"
                 "s_mov_b32 s20, %3\n1:\n"
                 "s_waitcnt vmcnt(0)\n"
                 "v_mul_lo_u32 v56, v56, %4\n v_add_u32 v56, 0x9e3779b9, v56\n"
                 "v_lshrrev_b32 v57, 2, v56\n v_and_b32 v57, %7, v57\n"
                 "v_mov_b32 v58, 128\n"
                 "v_mad_u64_u32 v[60:61], vcc, v57, v58, %6\n"
                 "global_load_dwordx4 v[64:67], v[60:61], off\n"
                 "global_load_dwordx4 v[68:71], v[60:61], off offset:16\n"
                 "global_load_dwordx4 v[72:75], v[60:61], off offset:32\n"
                 "global_load_dwordx4 v[76:79], v[60:61], off offset:48\n"
                 "global_load_dwordx4 v[80:83], v[60:61], off offset:64\n"
                 "global_load_dwordx4 v[84:87], v[60:61], off offset:80\n"
                 "global_load_dwordx4 v[88:91], v[60:61], off offset:96\n.p2align 3\n"
                 X256(V16) "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\ns_waitcnt vmcnt(0)\n"
                 "v_xor_b32 v8, v8, v64\n" FINI
                 : "=&v"(r) : "v"(a), "v"(b), "s"(iters), "s"(747796405u), "s"(shift), "v"((uint64_t)table), "s"(row_mask)
                 : CLOB16, "v56", "v57", "v58", "v60", "v61", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78",
                   "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91");
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// ---- what in the accumulation's stream costs more than pure multiply-adds? (all with an s_nop 15 per 4096 instructions: the fast regime)
// one factor from scalar registers (the modulus limbs of the reduction half)
#define SG16 M("v[8:9]", "v40", "s40") M("v[10:11]", "v41", "s41") M("v[12:13]", "v42", "s42") M("v[14:15]", "v43", "s43") \
             M("v[16:17]", "v44", "s44") M("v[18:19]", "v45", "s45") M("v[20:21]", "v46", "s46") M("v[22:23]", "v47", "s47") \
             M("v[8:9]", "v48", "s47") M("v[10:11]", "v49", "s46") M("v[12:13]", "v50", "s45") M("v[14:15]", "v51", "s44") \
             M("v[16:17]", "v52", "s43") M("v[18:19]", "v53", "s42") M("v[20:21]", "v54", "s41") M("v[22:23]", "v55", "s40")
#define SG16_SRC0 M("v[8:9]", "s40", "v40") M("v[10:11]", "s41", "v41") M("v[12:13]", "s42", "v42") M("v[14:15]", "s43", "v43") \
             M("v[16:17]", "s44", "v44") M("v[18:19]", "s45", "v45") M("v[20:21]", "s46", "v46") M("v[22:23]", "s47", "v47") \
             M("v[8:9]", "s47", "v48") M("v[10:11]", "s46", "v49") M("v[12:13]", "s45", "v50") M("v[14:15]", "s44", "v51") \
             M("v[16:17]", "s43", "v52") M("v[18:19]", "s42", "v53") M("v[20:21]", "s41", "v54") M("v[22:23]", "s40", "v55")
// one factor kept for eight instructions, as the FIRST source or as the second
#define F16_SRC1 M("v[8:9]", "v48", "v40") M("v[10:11]", "v49", "v40") M("v[12:13]", "v50", "v40") M("v[14:15]", "v51", "v40") \
            M("v[16:17]", "v52", "v40") M("v[18:19]", "v53", "v40") M("v[20:21]", "v54", "v40") M("v[22:23]", "v55", "v40") \
            M("v[8:9]", "v48", "v41") M("v[10:11]", "v49", "v41") M("v[12:13]", "v50", "v41") M("v[14:15]", "v51", "v41") \
            M("v[16:17]", "v52", "v41") M("v[18:19]", "v53", "v41") M("v[20:21]", "v54", "v41") M("v[22:23]", "v55", "v41")
#define SGINIT "v_readfirstlane_b32 s40, v48\n v_readfirstlane_b32 s41, v49\n v_readfirstlane_b32 s42, v50\n v_readfirstlane_b32 s43, v51\n" \
               "v_readfirstlane_b32 s44, v52\n v_readfirstlane_b32 s45, v53\n v_readfirstlane_b32 s46, v54\n v_readfirstlane_b32 s47, v55\n"
// the kernel's mix: 13 multiply-adds, one 64-bit shift, one mask, one 32-bit multiply per 16
#define MIX16 M("v[8:9]", "v40", "v55") M("v[10:11]", "v41", "v54") M("v[12:13]", "v42", "v53") M("v[14:15]", "v43", "v52") \
              M("v[16:17]", "v44", "v51") "v_lshrrev_b64 v[24:25], 28, v[8:9]\n" M("v[18:19]", "v45", "v50") M("v[20:21]", "v46", "v49") M("v[22:23]", "v47", "v48") \
              "v_and_b32 v26, 0xfffffff, v10\n" M("v[8:9]", "v48", "v47") M("v[10:11]", "v49", "v46") M("v[12:13]", "v50", "v45") "v_mul_lo_u32 v27, v12, v41\n" \
              M("v[14:15]", "v51", "v44") M("v[16:17]", "v52", "v43")
// three interleaved dependent chains (what three products in flight look like), and two
#define C3_16 M("v[8:9]", "v40", "v55") M("v[10:11]", "v41", "v54") M("v[12:13]", "v42", "v53") M("v[8:9]", "v43", "v52") M("v[10:11]", "v44", "v51") M("v[12:13]", "v45", "v50") \
              M("v[8:9]", "v46", "v49") M("v[10:11]", "v47", "v48") M("v[12:13]", "v48", "v47") M("v[8:9]", "v49", "v46") M("v[10:11]", "v50", "v45") M("v[12:13]", "v51", "v44") \
              M("v[8:9]", "v52", "v43") M("v[10:11]", "v53", "v42") M("v[12:13]", "v54", "v41") M("v[8:9]", "v55", "v40")
#define C2_16 M("v[8:9]", "v40", "v55") M("v[10:11]", "v41", "v54") M("v[8:9]", "v42", "v53") M("v[10:11]", "v43", "v52") M("v[8:9]", "v44", "v51") M("v[10:11]", "v45", "v50") \
              M("v[8:9]", "v46", "v49") M("v[10:11]", "v47", "v48") M("v[8:9]", "v48", "v47") M("v[10:11]", "v49", "v46") M("v[8:9]", "v50", "v45") M("v[10:11]", "v51", "v44") \
              M("v[8:9]", "v52", "v43") M("v[10:11]", "v53", "v42") M("v[8:9]", "v54", "v41") M("v[10:11]", "v55", "v40")
#define SCLOB "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "v24", "v25", "v26", "v27"
#define STREAM_KERNEL(NAME, PRE, B16)                                                                                       \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed, uint32_t iters, uint32_t shift) {            \
        uint32_t a = (seed ^ (threadIdx.x * 2246822519u)) * 3266489917u, b = 0, r;                                          \
        asm volatile(INIT16 PRE "s_mov_b32 s20, %3\n1:\n s_nop 15\n.p2align 3\n" X256(B16) "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\n" FINI \
                     : "=&v"(r) : "v"(a), "v"(b), "s"(iters), "s"(747796405u), "s"(shift) : CLOB16, SCLOB);              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                                     \
    }
STREAM_KERNEL(k_s_base, "", V16)
STREAM_KERNEL(k_s_sgpr, SGINIT, SG16)
STREAM_KERNEL(k_s_sgpr_src0, SGINIT, SG16_SRC0)
STREAM_KERNEL(k_s_fixed_src0, "", F16)
STREAM_KERNEL(k_s_fixed_src1, "", F16_SRC1)
STREAM_KERNEL(k_s_mix, "", MIX16)
STREAM_KERNEL(k_s_chain3, "", C3_16)
STREAM_KERNEL(k_s_chain2, "", C2_16)

// ---- which ingredient of the gather changes the rate? the random-operand stream with something else at the head of every body of PER x 16 instructions
#define XN_16(B) B B B B B B B B B B B B B B B B
#define X64(B) X4(X4(X4(B)))
#define EXTRA_KERNEL(NAME, BODYX, EXTRA)                                                                                                           \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed, uint32_t iters, uint32_t shift, const uint8_t *table, uint32_t row_mask) { \
        uint32_t a = (seed ^ (threadIdx.x * 2246822519u)) * 3266489917u, b = 0, r;                                                                 \
        asm volatile(INIT16 "v_mov_b32 v56, %1\n v_lshrrev_b64 v[60:61], 0, %6\n"                                                         \
                     "s_mov_b32 s20, %3\n1:\n" EXTRA ".p2align 3\n" BODYX "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\ns_waitcnt vmcnt(0)\n" FINI \
                     : "=&v"(r) : "v"(a), "v"(b), "s"(iters), "s"(747796405u), "s"(shift), "v"((uint64_t)table), "s"(row_mask)                      \
                     : CLOB16, "v56", "v57", "v58", "v60", "v61", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", \
                       "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91");                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                                                            \
    }
#define ADDR_ONLY "v_mul_lo_u32 v56, v56, %4\n v_add_u32 v56, 0x9e3779b9, v56\n v_lshrrev_b32 v57, 2, v56\n v_and_b32 v57, %7, v57\n v_mov_b32 v58, 128\n v_mad_u64_u32 v[64:65], vcc, v57, v58, %6\n"
#define ONE_LOAD_SAME "s_waitcnt vmcnt(0)\n global_load_dword v64, v[60:61], off\n"
#define SEVEN_LOADS_SAME "s_waitcnt vmcnt(0)\n global_load_dwordx4 v[64:67], v[60:61], off\n global_load_dwordx4 v[68:71], v[60:61], off offset:16\n global_load_dwordx4 v[72:75], v[60:61], off offset:32\n" \
                         "global_load_dwordx4 v[76:79], v[60:61], off offset:48\n global_load_dwordx4 v[80:83], v[60:61], off offset:64\n global_load_dwordx4 v[84:87], v[60:61], off offset:80\n global_load_dwordx4 v[88:91], v[60:61], off offset:96\n"
#define SLEEP1 "s_sleep 1\n"
#define NOP16 "s_nop 15\n"
#define SETPRIO "s_setprio 1\n s_setprio 0\n"
EXTRA_KERNEL(k_x_addr, X256(V16), ADDR_ONLY)
EXTRA_KERNEL(k_x_load1, X256(V16), ONE_LOAD_SAME)
EXTRA_KERNEL(k_x_load7, X256(V16), SEVEN_LOADS_SAME)
EXTRA_KERNEL(k_x_sleep, X256(V16), SLEEP1)
EXTRA_KERNEL(k_x_nop, X256(V16), NOP16)
EXTRA_KERNEL(k_x_load1_per1024, X64(V16), ONE_LOAD_SAME)
EXTRA_KERNEL(k_x_sleep_per1024, X64(V16), SLEEP1)
EXTRA_KERNEL(k_x_none, X256(V16), "")

// the random-operand stream with the ACCUMULATION's address pattern: iteration t of every workgroup reads window (t mod 8) of a table laid out as 8 windows x 4096 points x 32768
// rows of 128 bytes; lane l of a workgroup owns point l + 256 (t / 8 mod 16), so a wave's 64 rows lie 4 MB apart and all workgroups walk the windows in lockstep
__global__ __launch_bounds__(256) void k_varied_gather_kernel_pattern(uint32_t *out, uint32_t seed, uint32_t iters, uint32_t shift, const uint8_t *table, uint32_t unused) {
    uint32_t a = (seed ^ (threadIdx.x * 2246822519u)) * 3266489917u, b = threadIdx.x, r;
    asm volatile(INIT16
                 "v_mov_b32 v56, %1\n v_mov_b32 v59, %2\n s_mov_b32 s21, 0\n"
                 "s_mov_b32 s20, %3\n1:\n"
                 "s_waitcnt vmcnt(0)\n"
                 "v_mul_lo_u32 v56, v56, %4\n v_add_u32 v56, 0x9e3779b9, v56\n"
                 "v_lshrrev_b32 v57, 17, v56\n"                                /* 15 random bits: the row within (window, point) */
                 "s_lshr_b32 s22, s21, 3\n s_and_b32 s22, s22, 15\n s_lshl_b32 s22, s22, 8\n"
                 "v_add_u32 v58, s22, v59\n"                                     /* point = lane + 256 (t / 8 mod 16) */
                 "v_lshl_add_u32 v57, v58, 15, v57\n"                            /* row index within the window */
                 "s_and_b32 s22, s21, 7\n s_lshl_b32 s23, s22, 2\n s_mov_b32 s22, 0\n"   /* window t mod 8 at 2^34 bytes: high word 4 (t mod 8) */
                 "v_mov_b32 v58, 128\n"
                 "v_mad_u64_u32 v[60:61], vcc, v57, v58, %6\n"
                 "v_add_u32 v61, s23, v61\n"
                 "s_add_u32 s21, s21, 1\n"
                 "global_load_dwordx4 v[64:67], v[60:61], off\n"
                 "global_load_dwordx4 v[68:71], v[60:61], off offset:16\n"
                 "global_load_dwordx4 v[72:75], v[60:61], off offset:32\n"
                 "global_load_dwordx4 v[76:79], v[60:61], off offset:48\n"
                 "global_load_dwordx4 v[80:83], v[60:61], off offset:64\n"
                 "global_load_dwordx4 v[84:87], v[60:61], off offset:80\n"
                 "global_load_dwordx4 v[88:91], v[60:61], off offset:96\n.p2align 3\n"
                 X256(V16) "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\ns_waitcnt vmcnt(0)\n"
                 "v_xor_b32 v8, v8, v64\n" FINI
                 : "=&v"(r) : "v"(a), "v"(b), "s"(iters), "s"(747796405u), "s"(shift), "v"((uint64_t)table), "s"(unused)
                 : CLOB16, "s21", "s22", "s23", "v56", "v57", "v58", "v59", "v60", "v61", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78",
                   "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91");
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// one random row per 4096 multiply-adds: address = base + (lcg >> shift) * 128; loads issued at the head of a body, waited at the head of the next
__global__ __launch_bounds__(256) void k_gather(uint32_t *out, uint32_t seed, uint32_t iters, const uint8_t *table, uint32_t row_mask) {
    uint32_t a = seed ^ threadIdx.x, b = (seed * 2654435761u) | 1u, r;
    uint32_t lcg = (seed + blockIdx.x * 256 + threadIdx.x) * 747796405u + 2891336453u;
    asm volatile(INIT
                 "v_mov_b32 v50, %6\n"
                 "s_mov_b32 s20, %3\n1:\n"
                 "s_waitcnt vmcnt(0)\n"
                 "v_xor_b32 v40, v40, v60\n v_or_b32 v40, 1, v40\n"          /* the loaded data is used */
                 "v_mul_lo_u32 v50, v50, %7\n v_add_u32 v50, 0x9e3779b9, v50\n"
                 "v_lshrrev_b32 v51, 2, v50\n v_and_b32 v51, %5, v51\n"
                 "v_mov_b32 v52, 128\n"
                 "v_mad_u64_u32 v[54:55], vcc, v51, v52, %4\n"
                 "global_load_dwordx4 v[60:63], v[54:55], off\n"
                 "global_load_dwordx4 v[64:67], v[54:55], off offset:16\n"
                 "global_load_dwordx4 v[68:71], v[54:55], off offset:32\n"
                 "global_load_dwordx4 v[72:75], v[54:55], off offset:48\n"
                 "global_load_dwordx4 v[76:79], v[54:55], off offset:64\n"
                 "global_load_dwordx4 v[80:83], v[54:55], off offset:80\n"
                 "global_load_dwordx4 v[84:87], v[54:55], off offset:96\n.p2align 3\n"
                 BODY4096 "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\ns_waitcnt vmcnt(0)\n" FINI
                 : "=&v"(r) : "v"(a), "v"(b), "s"(iters), "v"((uint64_t)table), "s"(row_mask), "v"(lcg), "s"(747796405u)
                 : CLOB, PAD_CLOB, "v50", "v51", "v52", "v54", "v55", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72",
                   "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87");
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void k_fill_random(uint4 *p, size_t n16) {   // what a table of field elements looks like to the memory system: no two words alike
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) {
        uint64_t x = i * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull;
        x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32;
        uint64_t y = x * 0x94d049bb133111ebull; y ^= y >> 31;
        p[i] = make_uint4((uint32_t)x & 0xfffffff, (uint32_t)(x >> 32) & 0xfffffff, (uint32_t)y & 0xfffffff, (uint32_t)(y >> 32) & 0xfffffff);
    }
}

int main(int argc, char **argv) {
    double gb = 128;
    bool random_table = false;   // --random-table: the gathered table holds random 28-bit words (as the real one does) instead of one repeated byte
    bool quick = false;   // --quick: one size of launch (8.39e6 wave-instructions per SIMD each), six launches per stream: for a rocprofv3 --pmc pass
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--gb") && i + 1 < argc) gb = atof(argv[i + 1]);
        if (!strcmp(argv[i], "--quick")) quick = true;
        if (!strcmp(argv[i], "--random-table")) random_table = true;
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int n_cu = prop.multiProcessorCount;
    uint32_t *d_out;
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 2 * 256 * 4));
    size_t rows = 1;
    while (rows * 2 * 128 <= (size_t)(gb * (1 << 30))) rows *= 2;
    uint8_t *table;
    CHECK(hipMalloc(&table, rows * 128));
    CHECK(hipMemset(table, 1, rows * 128));
    if (random_table) {
        hipLaunchKernelGGL(k_fill_random, dim3(8192), dim3(256), 0, 0, (uint4 *)table, rows * 8);
        CHECK(hipDeviceSynchronize());
    }
    printf("table contents: %s\n", random_table ? "random 28-bit words" : "the byte 0x01 throughout");
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    int grid = n_cu * 2;  // two workgroups of four waves per CU: two waves per SIMD
    printf("table %.1f GB (%zu rows of 128 bytes), %d workgroups of 256 lanes\n", rows * 128 / 1e9, rows, grid);
    const int passes = quick ? 1 : 2;
    for (int pass = 0; pass < passes; pass++)
        for (int with_gather = 0; with_gather < 2; with_gather++)
            for (uint32_t iters : {32u, 128u, 512u, 2048u}) {  // x 4096 instructions: ~0.25, 1, 4, 16 ms per launch
                if (quick && iters != 2048u) continue;
                if (quick) iters = 1024;
                int launches = quick ? 6 : (int)(200.0 / (iters * 4096 * 2 * 1.9e-6)) + 2;
                for (int rep = 0; rep < 2; rep++) {  // the second repetition is reported: the board has been under this load for 0.2 s
                    CHECK(hipEventRecord(e0, st));
                    for (int l = 0; l < launches; l++) {
                        if (with_gather)
                            hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, st, d_out, 12345u + l, iters, table, (uint32_t)(rows - 1));
                        else
                            hipLaunchKernelGGL(k_plain, dim3(grid), dim3(256), 0, st, d_out, 12345u + l, iters);
                    }
                    CHECK(hipEventRecord(e1, st));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    double per_launch = ms / launches;
                    double ns = per_launch * 1e6 / ((double)iters * 4096 * 2);  // two waves share a SIMD
                    if (rep == 1)
                        printf("{\"stream\": \"%s\", \"ms_per_launch\": %.3f, \"launches\": %d, \"ns_per_wave_instruction_per_simd\": %.4f, \"cycles_at_2.4GHz\": %.3f}\n",
                               with_gather ? "mad + gathers" : "mad only", per_launch, launches, ns, ns * 2.4);
                }
            }
    struct { const char *name; void (*k)(uint32_t *, uint32_t, uint32_t, uint32_t); uint32_t shift; } dk[] = {
        {"same two factors, random 28-bit contents", k_same, 4}, {"one factor fixed for 8 instructions, random 28-bit", k_one_fixed, 4},
        {"both factors change every instruction, random 28-bit", k_varied, 4}, {"both factors change every instruction, random 32-bit", k_varied, 0},
        {"both factors change every instruction, 8-bit contents", k_varied, 24},
        {"both factors change every instruction, random 28-bit, every multiply-add at 4 mod 8", k_varied_mis, 4}};
    for (int pass = 0; pass < passes; pass++)
        for (auto &d : dk) {
            uint32_t iters = 1024;  // ~7 ms per launch
            int launches = quick ? 6 : 28;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0, st));
                for (int l = 0; l < launches; l++) hipLaunchKernelGGL(d.k, dim3(grid), dim3(256), 0, st, d_out, 12345u + l, iters, d.shift);
                CHECK(hipEventRecord(e1, st));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                double per_launch = ms / launches;
                double ns = per_launch * 1e6 / ((double)iters * 4096 * 2);
                if (rep == 1)
                    printf("{\"stream\": \"%s\", \"ms_per_launch\": %.3f, \"ns_per_wave_instruction_per_simd\": %.4f, \"cycles_at_2.4GHz\": %.3f}\n", d.name, per_launch, ns, ns * 2.4);
            }
        }
    {   // the random-operand stream with one random row per 4096 instructions per lane out of regions of different sizes (the whole table; one 17 GB
        // window's worth; what the memory-side cache holds; what an L2 holds), against the stream with a pause instead of the loads
        struct { const char *name; uint32_t mask; } gm[] = {{"no loads (s_nop 15 per 4096)", 0}, {"rows out of the whole 137 GB", (uint32_t)(rows - 1)},
            {"rows out of 17 GB", (1u << 27) - 1}, {"the accumulation's pattern (8 windows in lockstep, a wave's rows 4 MB apart)", 0xffffffffu}, {"rows out of 268 MB", (1u << 21) - 1}, {"rows out of 1 MB", (1u << 13) - 1}};
        for (int pass = 0; pass < 2 * passes; pass++)
            for (auto &g : gm) {
                uint32_t iters = 1024;
                int launches = quick ? 6 : 20;
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipEventRecord(e0, st));
                    for (int l = 0; l < launches; l++) {
                        if (g.mask == 0xffffffffu) hipLaunchKernelGGL(k_varied_gather_kernel_pattern, dim3(grid), dim3(256), 0, st, d_out, 12345u + l, iters, 4u, table, 0u);
                        else if (g.mask) hipLaunchKernelGGL(k_varied_gather, dim3(grid), dim3(256), 0, st, d_out, 12345u + l, iters, 4u, table, g.mask);
                        else hipLaunchKernelGGL(k_x_nop, dim3(grid), dim3(256), 0, st, d_out, 12345u + l, iters, 4u, table, (uint32_t)(rows - 1));
                    }
                    CHECK(hipEventRecord(e1, st));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    double ns = ms / launches * 1e6 / ((double)iters * 4096 * 2);
                    if (rep == 1)
                        printf("{\"stream\": \"random 28-bit, both factors change, %s\", \"ms_per_launch\": %.3f, \"ns_per_wave_instruction_per_simd\": %.4f}\n", g.name, ms / launches, ns);
                }
            }
    }
    struct { const char *name; void (*k)(uint32_t *, uint32_t, uint32_t, uint32_t); } sk[] = {
        {"eight accumulators, both factors vector registers", k_s_base}, {"one factor from scalar registers (second source)", k_s_sgpr}, {"one factor from scalar registers (first source)", k_s_sgpr_src0},
        {"one factor kept for 8 instructions, first source", k_s_fixed_src0}, {"one factor kept for 8 instructions, second source", k_s_fixed_src1},
        {"13 multiply-adds + 64-bit shift + mask + 32-bit multiply per 16", k_s_mix}, {"three interleaved dependent chains", k_s_chain3},
        {"two interleaved dependent chains", k_s_chain2}, {"eight accumulators (again)", k_s_base}};
    for (int pass = 0; pass < passes; pass++)
        for (auto &x : sk) {
            int launches = quick ? 6 : 20;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0, st));
                for (int l = 0; l < launches; l++) hipLaunchKernelGGL(x.k, dim3(grid), dim3(256), 0, st, d_out, 12345u + l, 1024u, 4u);
                CHECK(hipEventRecord(e1, st));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                double ns = ms / launches * 1e6 / ((double)1024 * 4096 * 2);
                if (rep == 1)
                    printf("{\"stream\": \"random 28-bit + s_nop 15 per 4096, %s\", \"ms_per_launch\": %.3f, \"ns_per_wave_instruction_per_simd\": %.4f}\n", x.name, ms / launches, ns);
            }
        }
    struct { const char *name; void (*k)(uint32_t *, uint32_t, uint32_t, uint32_t, const uint8_t *, uint32_t); uint32_t iters; } xk[] = {
        {"nothing extra", k_x_none, 1024}, {"+ the six address instructions per 4096", k_x_addr, 1024}, {"+ one dword load (same address) per 4096", k_x_load1, 1024},
        {"+ seven dwordx4 loads (same address) per 4096", k_x_load7, 1024}, {"+ s_sleep 1 per 4096", k_x_sleep, 1024}, {"+ s_nop 15 per 4096", k_x_nop, 1024},
        {"+ one dword load per 1024", k_x_load1_per1024, 4096}, {"+ s_sleep 1 per 1024", k_x_sleep_per1024, 4096}, {"nothing extra (again)", k_x_none, 1024}};
    for (int pass = 0; pass < passes; pass++)
        for (auto &x : xk) {
            int launches = quick ? 6 : 20;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0, st));
                for (int l = 0; l < launches; l++) hipLaunchKernelGGL(x.k, dim3(grid), dim3(256), 0, st, d_out, 12345u + l, x.iters, 4u, table, (uint32_t)(rows - 1));
                CHECK(hipEventRecord(e1, st));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                double per_launch = ms / launches;
                double ns = per_launch * 1e6 / ((double)1024 * 4096 * 2);
                if (rep == 1)
                    printf("{\"stream\": \"random 28-bit, both factors change, %s\", \"ms_per_launch\": %.3f, \"ns_per_wave_instruction_per_simd\": %.4f}\n", x.name, per_launch, ns);
            }
        }
    for (int pass = 0; pass < passes; pass++)
      for (int mis = 0; mis < 2; mis++)
        for (int w : {1, 2, 4, 8}) {
            uint32_t iters = 2048 / w;   // the same 8.39e6 wave-instructions per SIMD per launch at every occupancy
            int launches = quick ? 6 : 14;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0, st));
                for (int l = 0; l < launches; l++) hipLaunchKernelGGL(mis ? k_varied_small_mis : k_varied_small, dim3(n_cu * w), dim3(256), 0, st, d_out, 12345u + l, iters, 4u);
                CHECK(hipEventRecord(e1, st));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                double per_launch = ms / launches;
                double ns = per_launch * 1e6 / ((double)iters * 4096 * w);
                if (rep == 1)
                    printf("{\"stream\": \"both factors change every instruction, random 28-bit, %s, %d waves per SIMD\", \"ms_per_launch\": %.3f, \"ns_per_wave_instruction_per_simd\": %.4f, \"cycles_at_2.4GHz\": %.3f}\n", mis ? "at 4 mod 8" : "8-byte aligned", w, per_launch, ns, ns * 2.4);
            }
        }
    return 0;
}
