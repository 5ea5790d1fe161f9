// knobs.hip -- the one place the library reads its environment (knobs.h). Host code only.
#include "knobs.h"

#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace lwk {
namespace {

struct Reader {
    bool experimental;
    // operational: always; experiment: only with LWKZG_EXPERIMENTAL=1 (an ignored experiment variable is reported once under LWKZG_VERBOSE)
    const char *get(const char *name, bool exp) const {
        const char *v = getenv(name);
        if (!v || !*v) return nullptr;
        if (exp && !experimental) {
            if (getenv("LWKZG_VERBOSE")) fprintf(stderr, "[lambdaworks_kzg_amd] %s ignored: experiment knobs need LWKZG_EXPERIMENTAL=1\n", name);
            return nullptr;
        }
        return v;
    }
    void flag(bool &dst, const char *name, bool exp) const {  // "0" (or anything atoi reads as 0) switches off
        if (const char *v = get(name, exp)) dst = atoi(v) != 0;
    }
    void present(bool &dst, const char *name, bool exp) const {  // set by mere presence (r01-r05 semantics of the TIMING / VERBOSE / PAIRING_* variables)
        if (get(name, exp)) dst = true;
    }
    void num(int &dst, const char *name, bool exp) const {
        if (const char *v = get(name, exp)) dst = atoi(v);
    }
    void num(size_t &dst, const char *name, bool exp) const {
        if (const char *v = get(name, exp)) {
            const long x = atol(v);
            dst = x < 0 ? 0 : (size_t)x;
        }
    }
    void num(unsigned &dst, const char *name, bool exp) const {
        if (const char *v = get(name, exp)) dst = (unsigned)atoi(v);
    }
};

Knobs read_env() {
    Knobs k;
    {
        const char *e = getenv("LWKZG_EXPERIMENTAL");
        k.experimental = e && atoi(e) != 0;
    }
    const Reader r{k.experimental};
    constexpr bool OP = false, EXP = true;
    if (const char *e = r.get("LWKZG_MODE", OP))
        k.mode = (!strcmp(e, "ckzg") || !strcmp(e, "c") || !strcmp(e, "C") || !strcmp(e, "1")) ? 1 : 0;
    if (const char *e = r.get("LWKZG_DIRECT_BITS", OP)) {
        k.has_direct_bits = true;
        k.direct_bits = !strcmp(e, "auto") ? -1 : atoi(e);
    }
    r.num(k.direct_row, "LWKZG_DIRECT_ROW", OP);
    r.flag(k.coalesce, "LWKZG_COALESCE", OP);
    r.flag(k.twin, "LWKZG_TWIN", OP);
    r.num(k.small_proof_host, "LWKZG_SMALL_PROOF_HOST", OP);
    r.num(k.mid_proof_host, "LWKZG_MID_PROOF_HOST", OP);
    r.num(k.host_threads, "LWKZG_HOST_THREADS", OP);
    r.num(k.host_warm_ms, "LWKZG_HOST_WARM_MS", OP);
    r.num(k.host_finish, "LWKZG_HOST_FINISH", OP);
    r.present(k.timing, "LWKZG_TIMING", OP);
    r.present(k.verbose, "LWKZG_VERBOSE", OP);

    r.flag(k.direct_asm, "LWKZG_DIRECT_ASM", EXP);
    r.flag(k.fold_asm, "LWKZG_FOLD_ASM", EXP);
    r.flag(k.bucket_asm, "LWKZG_BUCKET_ASM", EXP);
    r.num(k.direct_fill, "LWKZG_DIRECT_FILL", EXP);
    r.num(k.coop, "LWKZG_COOP", EXP);
    r.num(k.coop_max, "LWKZG_COOP_MAX", EXP);
    if (k.coop_max < 1) k.coop_max = 1;
    if (k.coop_max > 8) k.coop_max = 8;
    r.num(k.coop_rpq, "LWKZG_COOP_RPQ", EXP);
    r.flag(k.sort_stage, "LWKZG_SORT_STAGE", EXP);
    r.num(k.reduce_lanes, "LWKZG_REDUCE_LANES", EXP);
    r.flag(k.hash_pairs, "LWKZG_HASH_PAIRS", EXP);
    r.num(k.hash_prio, "LWKZG_HASH_PRIO", EXP);
    r.flag(k.validate_coop, "LWKZG_VALIDATE_COOP", EXP);
    r.num(k.validate_lds_pad, "LWKZG_VALIDATE_LDS_PAD", EXP);
    r.flag(k.ckzg_eval_proofs, "LWKZG_CKZG_EVAL_PROOFS", EXP);
    r.flag(k.mid_proof_pipe, "LWKZG_MID_PROOF_PIPE", EXP);
    r.num(k.mid_proof_pipe_min, "LWKZG_MID_PROOF_PIPE_MIN", EXP);
    r.num(k.mid_proof_parts, "LWKZG_MID_PROOF_PARTS", EXP);
    r.num(k.mid_proof_chunks, "LWKZG_MID_PROOF_CHUNKS", EXP);
    r.num(k.heavy_serial, "LWKZG_HEAVY_SERIAL", EXP);
    r.num(k.proof_schedule, "LWKZG_PROOF_SCHEDULE", EXP);
    r.num(k.split, "LWKZG_SPLIT", EXP);
    r.num(k.slice0, "LWKZG_SLICE0", EXP);
    r.flag(k.set_mode_in_place, "LWKZG_SET_MODE_IN_PLACE", EXP);
    r.num(k.host_fp_portable, "LWKZG_HOST_FP_PORTABLE", EXP);
    r.num(k.side_workers, "LWKZG_SIDE_WORKERS", EXP);
    r.num(k.host_hash_grain, "LWKZG_HOST_HASH_GRAIN", EXP);
    r.present(k.pairing_generic_sqr, "LWKZG_PAIRING_GENERIC_SQR", EXP);
    r.present(k.pairing_naive, "LWKZG_PAIRING_NAIVE", EXP);
    r.present(k.pairing_no_precomp, "LWKZG_PAIRING_NO_PRECOMP", EXP);
    r.present(k.pairing_one_thread, "LWKZG_PAIRING_ONE_THREAD", EXP);
    r.num(k.verify_msm, "LWKZG_VERIFY_MSM", EXP);
    r.num(k.verify_fused, "LWKZG_VERIFY_FUSED", EXP);
    if (const char *e = r.get("LWKZG_VERIFY_PAD_KB", EXP)) {
        int a = 0, b = 0, c = 0;
        const int got = sscanf(e, "%d,%d,%d", &a, &b, &c);
        if (got >= 1) k.verify_pad_kb[0] = a;
        k.verify_pad_kb[1] = got >= 2 ? b : a;
        k.verify_pad_kb[2] = got >= 3 ? c : (got >= 2 ? b : a);
        for (int &p : k.verify_pad_kb) p = p < 0 ? 0 : (p > 150 ? 150 : p);
    }
    r.num(k.verify_order, "LWKZG_VERIFY_ORDER", EXP);
    r.num(k.verify_cu_mask, "LWKZG_VERIFY_CU_MASK", EXP);
    r.num(k.vmsm_list_cap, "LWKZG_VMSM_LIST_CAP", EXP);
    r.flag(k.zero_copy, "LWKZG_ZERO_COPY", EXP);
    r.flag(k.host_stage, "LWKZG_HOST_STAGE", EXP);
    if (const char *e = r.get("LWKZG_STAGE_STREAMS", EXP)) {
        int a = 8, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 0 && a <= 8 && b >= 0 && b < 8 && a != b) {   // (8: a high-priority stream for the uploads)
            k.stage_streams[0] = a;
            k.stage_streams[1] = b;
        }
    }
    return k;
}

}  // namespace

const Knobs &knobs() {
    static const Knobs k = read_env();
    return k;
}

const char *knob_names_operational() {
    return "LWKZG_MODE LWKZG_DIRECT_BITS LWKZG_DIRECT_ROW LWKZG_COALESCE LWKZG_TWIN LWKZG_SMALL_PROOF_HOST LWKZG_MID_PROOF_HOST "
           "LWKZG_HOST_THREADS LWKZG_HOST_WARM_MS LWKZG_HOST_FINISH LWKZG_TIMING LWKZG_VERBOSE LWKZG_EXPERIMENTAL";
}

const char *knob_names_experimental() {
    return "LWKZG_DIRECT_ASM LWKZG_FOLD_ASM LWKZG_BUCKET_ASM LWKZG_DIRECT_FILL LWKZG_COOP LWKZG_COOP_MAX LWKZG_COOP_RPQ LWKZG_SORT_STAGE "
           "LWKZG_REDUCE_LANES LWKZG_HASH_PAIRS LWKZG_HASH_PRIO LWKZG_VALIDATE_COOP LWKZG_VALIDATE_LDS_PAD LWKZG_CKZG_EVAL_PROOFS "
           "LWKZG_MID_PROOF_PIPE LWKZG_MID_PROOF_PIPE_MIN LWKZG_MID_PROOF_PARTS LWKZG_MID_PROOF_CHUNKS LWKZG_HEAVY_SERIAL LWKZG_PROOF_SCHEDULE LWKZG_SPLIT "
           "LWKZG_SLICE0 LWKZG_SET_MODE_IN_PLACE LWKZG_PAIRING_GENERIC_SQR LWKZG_PAIRING_NAIVE LWKZG_PAIRING_NO_PRECOMP "
           "LWKZG_PAIRING_ONE_THREAD LWKZG_VERIFY_MSM LWKZG_VERIFY_FUSED LWKZG_VERIFY_PAD_KB LWKZG_VERIFY_ORDER LWKZG_VERIFY_CU_MASK LWKZG_VMSM_LIST_CAP "
           "LWKZG_HOST_STAGE LWKZG_ZERO_COPY LWKZG_HOST_FP_PORTABLE LWKZG_SIDE_WORKERS LWKZG_HOST_HASH_GRAIN LWKZG_STAGE_STREAMS";
}

}  // namespace lwk

extern "C" __attribute__((visibility("default"))) size_t lwkzg_knob_report(char *buf, size_t cap) {
    const lwk::Knobs &k = lwk::knobs();
    char tmp[4096];
    const int n = snprintf(
        tmp, sizeof tmp,
        "{\"experimental\": %s, \"mode\": %d, \"direct_bits\": %s%d, \"direct_row\": %d, \"coalesce\": %d, \"twin\": %d, "
        "\"small_proof_host\": %zu, \"mid_proof_host\": %zu, \"host_threads\": %d, \"host_warm_ms\": %d, \"host_finish\": %zu, "
        "\"timing\": %d, \"verbose\": %d, \"direct_asm\": %d, \"fold_asm\": %d, \"bucket_asm\": %d, \"coop\": %d, \"coop_max\": %d, "
        "\"hash_pairs\": %d, \"hash_prio\": %d, \"validate_coop\": %d, \"ckzg_eval_proofs\": %d, \"mid_proof_pipe\": %d, "
        "\"verify_msm\": %d, \"verify_fused\": %d, \"verify_pad_kb\": [%d, %d, %d], \"verify_order\": %d, \"vmsm_list_cap\": %d, "
        "\"host_stage\": %d, \"operational\": \"%s\", \"experimental_names\": \"%s\"}",
        k.experimental ? "true" : "false", k.mode, k.has_direct_bits ? "" : "null, \"direct_bits_unset_default\": ", k.direct_bits, k.direct_row,
        (int)k.coalesce, (int)k.twin, k.small_proof_host, k.mid_proof_host, k.host_threads, k.host_warm_ms, k.host_finish, (int)k.timing,
        (int)k.verbose, (int)k.direct_asm, (int)k.fold_asm, (int)k.bucket_asm, k.coop, k.coop_max, (int)k.hash_pairs, k.hash_prio,
        (int)k.validate_coop, (int)k.ckzg_eval_proofs, (int)k.mid_proof_pipe, k.verify_msm, k.verify_fused, k.verify_pad_kb[0],
        k.verify_pad_kb[1], k.verify_pad_kb[2], k.verify_order, k.vmsm_list_cap, (int)k.host_stage, lwk::knob_names_operational(),
        lwk::knob_names_experimental());
    const size_t need = (size_t)(n < 0 ? 0 : n) + 1;
    if (buf && cap) {
        const size_t c = need <= cap ? need - 1 : cap - 1;
        memcpy(buf, tmp, c);
        buf[c] = 0;
    }
    return need;
}
