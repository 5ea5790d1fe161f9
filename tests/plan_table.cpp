// CPU harness of csrc/plan.h (compiled by tests/test_plan_cpu.py with g++): prints plan_proof_call over a grid of batch sizes, host-thread
// warmth, peer state, table form, staging availability and three knob sets, one line per point:
//   n warm busy direct staging knobset -> schedule per_chunk chunks parts heavy_serial
// With the argument "staged": plan_staged_verification over batch sizes and host hashing rates, one line per point:
//   n rate_GBps -> n_gpu n_host slice every launches-after-slices...
#include <stdio.h>
#include <string.h>
#include "../lambdaworks_kzg_amd/csrc/plan.h"

static int staged_table() {
    const size_t ns[] = {1025, 1100, 1536, 2048, 2600, 3000, 4096, 8192, 16384, 65536};
    const double rates[] = {0.0, 4.0, 18.0, 30.0, 36.0, 70.0, 400.0};
    for (size_t n : ns)
        for (double r : rates) {
            const lwk::StagedSplit p = lwk::plan_staged_verification(n, r * 1e9);
            printf("%zu %.0f -> %zu %zu %zu %zu", n, r, p.n_gpu, p.n_host, p.slice, p.every);
            for (size_t landed = 1; landed * p.slice <= p.n_gpu; landed++)
                if (p.launch_after(landed)) printf(" %zu", landed);
            printf("\n");
        }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && strcmp(argv[1], "staged") == 0) return staged_table();
    const size_t ns[] = {1, 2, 8, 63, 64, 65, 100, 128, 191, 192, 193, 255, 256, 257, 300, 319, 320, 383, 384, 385, 512, 513, 1000, 1024, 1025, 2048, 4096};
    lwk::PlanKnobs sets[3];
    sets[1].mid_proof_pipe = false;                              // LWKZG_MID_PROOF_PIPE=0
    sets[2].small_proof_host = 0; sets[2].mid_proof_host = 0;    // LWKZG_SMALL_PROOF_HOST=0 LWKZG_MID_PROOF_HOST=0: the GPU chains always
    for (int ks = 0; ks < 3; ks++)
        for (size_t n : ns)
            for (int warm = 0; warm < 2; warm++)
                for (int busy = 0; busy < 2; busy++)
                    for (int direct = 0; direct < 2; direct++)
                        for (int staging = 0; staging < 2; staging++) {
                            const lwk::ProofPlan p = lwk::plan_proof_call(n, warm, busy, direct, staging, sets[ks]);
                            printf("%zu %d %d %d %d %d -> %d %zu %zu %zu %d\n", n, warm, busy, direct, staging, ks, (int)p.schedule, p.per_chunk, p.chunks,
                                   p.parts, (int)p.heavy_serial);
                        }
    return 0;
}
