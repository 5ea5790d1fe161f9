// CPU harness of csrc/plan.h (compiled by tests/test_plan_cpu.py with g++): prints plan_proof_call over a grid of batch sizes, host-thread
// warmth, peer state, table form, staging availability and three knob sets, one line per point:
//   n warm busy direct staging knobset -> schedule per_chunk chunks parts heavy_serial
#include <stdio.h>
#include "../lambdaworks_kzg_amd/csrc/plan.h"

int main() {
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
