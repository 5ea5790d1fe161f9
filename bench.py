#!/usr/bin/env python3
"""bench.py -- blob_to_kzg_commitment throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (blob bytes -> 48-byte commitments: parse, fixed-base MSM, compress) over one batch
of synthetic 4096-element blobs that is already resident in HBM. Each GPU works on its own shard (weak scaling: 1024
blobs per GPU per step); the only collective is the one broadcast of the prepared trusted setup before the timed region.

Rank 0 prints ONE JSON line of at most 8,000 bytes (compact_line below: the contract's fields first, numbers only, no prose --
the driver keeps a bounded tail of stdout and round 3's 21 KB line lost its head there) and writes everything else -- every
note, breakdown and per-kernel table -- to bench_detail.json beside this file (path on stderr; LWKZG_BENCH_DETAIL overrides).
`python bench.py --gpus N` without a launcher starts its N ranks itself (spawn_ranks). The detail file carries
  roofline        the dominant kernel (k_direct_accumulate / k_bucket_accumulate) priced against HBM as the north star
                  mandates AND the integer-multiply picture, because the kernel is integer-ALU bound;
                  `traffic` comes from committed PMC passes and says so in `traffic_source`;
  default_engine  the same workload on the engine a plain load_trusted_setup* selects (what a drop-in consumer gets),
  bucket_engine   and on the low-memory fallback: short regions of their own, timed like the headline's;
  host_abi        the same batch through the host-pointer C ABI, PCIe included (never `value`);
  configs         BASELINE configs[2]-[4] and the Fr transform as short legs of their own (never folded into `value`):
                  blob proofs at 256 on one and two caller streams, commit-and-prove at 256, c-kzg-mode commitments with
                  k_ntt4096's own roofline object, a 4096-blob batch verification, the 2^20-term tiled MSM;
  dist            the process group this run used (`--force-dist` takes the RCCL branch at --gpus 1 too) and the per-rank
                  HBM budget the table width was chosen against;
  cpu_baseline    the CPU oracle (a restatement of the reference's algorithm, NOT the reference binary, which cannot be
                  built here) on this box's host cores, with and without the SRS rebuild the reference pays per call.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

BLOBS_PER_GPU = 1024
ALGO_BYTES_PER_MSM = 131072 + 393216 + 48   # scalars + affine points + output = 524,336 B
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec
SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup.txt")


def _cpu_worker(args):
    """One host core: `count` calls of the oracle's blob_to_kzg_commitment, each preceded by the per-call SRS rebuild
    the reference pays (srs.rs:258-280, lib.rs:266-269) when g2_blst is given. Returns (seconds in the commitments,
    seconds in the rebuilds, outputs)."""
    lib_path, first, count, g2_blst = args
    from oracle import oracle as O
    l = O.lib(lib_path)
    s = O.Settings.from_file(SETUP, check_subgroup=False, _lib=l)
    g1_blst = s.g1_blst()
    import blobs as B
    data = [B.synthetic_blob(first + i) for i in range(count)]
    t_commit = t_rebuild = 0.0
    outs = []
    for b in data:
        t0 = time.perf_counter()
        if g2_blst is not None:
            assert O.srs_rebuild(g1_blst, g2_blst, _lib=l) == 0
        t1 = time.perf_counter()
        rc, cm = O.blob_to_kzg_commitment(b, s, O.MODE_R)
        t2 = time.perf_counter()
        assert rc == 0
        t_rebuild += t1 - t0
        t_commit += t2 - t1
        outs.append(cm)
    return t_commit, t_rebuild, outs


def usable_cores():
    """Host threads this process may really use: the affinity mask capped by the cgroup CPU quota
    (the GPU boxes expose 256 hardware threads but a quota of 16 CPUs)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(gpu_outputs, g2_blst):
    """Time the oracle on the host: all cores, one blob stream per process. Bounded sample. Every call is timed in two
    parts -- the reference's per-call SRS rebuild and the commitment itself -- so that the rate is reported with and
    without the rebuild."""
    import multiprocessing as mp
    import tempfile
    from oracle import oracle as O
    try:
        lib_path = O.build(native=True, out_dir=tempfile.mkdtemp(prefix="lwkzg_oracle_"))
    except Exception:
        lib_path = os.path.join(ROOT, "oracle", "liboracle_kzg.so")
        if not os.path.exists(lib_path):
            lib_path = O.build()
    cores = usable_cores()
    # one thread first: calibrates the per-blob cost and is itself the reference's configuration
    tc1, tr1, o1 = _cpu_worker((lib_path, 0, 4, g2_blst))
    per_blob, per_rebuild = tc1 / 4, tr1 / 4
    per_proc = max(2, min(64, int(12.0 / (per_blob + per_rebuild))))    # about 12 s of work per core
    per_proc = min(per_proc, max(2, BLOBS_PER_GPU // cores))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(lib_path, i * per_proc, per_proc, g2_blst) for i in range(cores)])
    wall = time.perf_counter() - t0
    total = cores * per_proc
    busy_commit = max(r[0] for r in res)
    busy_both = max(r[0] + r[1] for r in res)
    verified = True
    flat = [o for r in res for o in r[2]]
    for i, cm in enumerate(flat[:len(gpu_outputs)]):
        if gpu_outputs[i] != cm:
            verified = False
    return {
        "value": total / busy_both,
        "unit": "blob_to_kzg_commitment ops/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d synthetic blobs (%d per core x %d cores, one process per core), %.1f s busy, %.1f s wall incl. "
                  "process start; CPU restatement of lambdaworks_kzg's algorithm (Pippenger w=9, 29 windows, projective "
                  "adds); `value` INCLUDES the SRS rebuild the reference runs on every call (srs.rs:258-280: 4096 + 65 "
                  "point conversions with curve checks, %.2f ms per call here), `value_without_srs_rebuild` does not"
                  % (total, per_proc, cores, busy_both, wall, per_rebuild * 1e3),
        "value_without_srs_rebuild": total / busy_commit,
        "single_thread_ops_per_s": 1.0 / (per_blob + per_rebuild),
        "single_thread_ops_per_s_without_srs_rebuild": 1.0 / per_blob,
        "srs_rebuild_ms_per_call": per_rebuild * 1e3,
        "gpu_outputs_match_oracle": verified,
    }


MADS_PER_MIXED_ADD = 6 * 392 + 2 * 301 + 588    # 6 Montgomery products (392 v_mad_u64_u32 each on 14 x 28-bit limbs), 2 squares (301), one fused pair (588)
# tools/ubench_mad.hip on MI355X (profiles/r02_ubench_mad.txt): v_mad_u64_u32 issues every 4.17 cycles per SIMD with 8 waves per
# SIMD at the 2.40 GHz the chip holds in that short loop (3.77e13 lane-ops/s), every 4.63 cycles with the 2 waves per SIMD this
# kernel's 204 registers allow. (Round 1 quoted 2.93e13 from a loop of 8 instructions per taken branch, which measured the branch.)
INT_MAD_PEAK_MEASURED = 256 * 4 * 64 / 4.17 * 2.4e9
INT_MAD_CYCLES_2_WAVES = 4.63
KERNEL_CLOCK_HZ = 2.05e9                         # GRBM_GUI_ACTIVE under this kernel: 1.99-2.08 GHz (profiles/r02_issue_summary.json)
INT_MAD_PEAK_THEORETICAL = 256 * 4 * 64 / 4 * 2.4e9   # 256 CUs x 4 SIMDs x 64 lanes / 4 cycles per wave-instruction x 2.4 GHz = 3.93e13
# tools/ubench_sustain.hip (profiles/r03_ubench_sustain.txt): the same multiply-add loop SUSTAINED (7-28 ms launches back to back), every instruction 8-byte
# aligned. The rate depends on the DATA, not on the occupancy: 1.70 ns per wave-instruction per SIMD with the near-constant operands of ubench_mad, 1.83 ns with
# random 28-bit factors that change every instruction (the same at 1, 2, 4 and 8 waves per SIMD), 1.75 ns for the kernel's own mix (13 multiply-adds, a 64-bit
# shift, a mask and a 32-bit multiply per 16), +2.4 % with one random table row per 4096 instructions per lane out of HBM -- on the box where the kernel itself ran
# 1.93-1.96 ns per instruction; another box gave 2.04-2.09 ns for the random stream. Field elements ARE random limbs, so this is the kernel's ceiling.
INT_MAD_NS_RANDOM = 1.832e-9
VALU_PER_MIXED_ADD = {"k_direct_accumulate_asm": 4256, "k_direct_accumulate": 4814, "k_bucket_accumulate_asm": 4249}   # tools/gen_direct_asm.py --mix; SQ_INSTS_VALU of the compiler's schedule
GATHER_PEAK_ROWS = 1.31e10                       # tools/gather_bench.hip on MI355X: random 112-byte rows/s out of a 128-200 GiB table


def engine_picture(K, capi, direct_bits, prof, elapsed, steps, n, msms_per_launch_override=None):
    """Roofline object of the dominant MSM kernel of one engine, from the library's hipEvent timings of a timed region:
    `achieved` = algorithmic bytes per launch / average launch duration (524,336 B per 4096-term MSM)."""
    # the direct engine's accumulation runs as the hand-scheduled kernel (k_direct_accumulate_asm, tools/gen_direct_asm.py) unless
    # LWKZG_DIRECT_ASM=0 or a handful of blobs put it on the compiler-scheduled one
    dom = ("k_direct_accumulate_asm" if "k_direct_accumulate_asm" in prof else "k_direct_accumulate") if direct_bits else \
          ("k_bucket_accumulate_asm" if "k_bucket_accumulate_asm" in prof else "k_bucket_accumulate")    # (LWKZG_BUCKET_ASM=0: the compiler-scheduled arm)
    k = prof.get(dom, {"launches": 0, "total_ms": 0.0})
    avg_ms = k["total_ms"] / max(1, k["launches"])
    # a step may cut its batch into sub-batches on concurrent streams (engine.hip: commit_batch_device), so
    # the units one launch processes = blobs per step / launches per step
    launches_per_step = max(1, round(k["launches"] / max(1, steps)))
    msms_per_launch = msms_per_launch_override if msms_per_launch_override is not None else n / launches_per_step
    achieved = msms_per_launch * ALGO_BYTES_PER_MSM / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    if direct_bits:
        nwin = K.lib().lwkzg_direct_num_windows(direct_bits)
        adds_per_msm = 4096 * nwin * (1 - 2.0 ** -direct_bits)
    else:
        nwin = K.lib().lwkzg_msm_num_windows()
        adds_per_msm = 4096 * nwin * (1 - 2.0 ** -K.lib().lwkzg_msm_window_bits())
    gather = None
    if direct_bits and avg_ms > 0:
        rows = msms_per_launch * adds_per_msm
        gather = {"rows_per_launch": rows, "bytes_per_launch": rows * 112,
                  "achieved_rows_per_s": rows / (avg_ms * 1e-3), "peak_rows_per_s": GATHER_PEAK_ROWS,
                  "frac": rows / (avg_ms * 1e-3) / GATHER_PEAK_ROWS,
                  "note": "the direct path really reads one random 112-byte table row per mixed addition from HBM "
                          "(table far larger than every cache); ceiling measured by tools/gather_bench.hip"}
    mads_per_launch = msms_per_launch * adds_per_msm * MADS_PER_MIXED_ADD
    mad_rate = mads_per_launch / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
    sustained = {"peak_sustained_random_operands_Gmad_per_s": 256 * 4 * 64 / INT_MAD_NS_RANDOM / 1e9,
                 "frac_of_sustained_random_operands": mad_rate / (256 * 4 * 64 / INT_MAD_NS_RANDOM),
                 "peak_sustained_random_operands_note": "tools/ubench_sustain.hip (profiles/r03_ubench_sustain.txt): a pure v_mad_u64_u32 stream, 8-byte aligned, whose two factors "
                                                        "are random 28-bit values and change with every instruction, held for 0.3 s: 1.83 ns per wave-instruction per SIMD at any "
                                                        "occupancy (1.70 ns with near-constant operands: the board slows busy multipliers; 2.00 ns at two waves per SIMD when the "
                                                        "instructions sit at 4 mod 8), on a box where this kernel ran 8.42-8.55 ms; boxes differ by several per cent on both"}
    if dom in VALU_PER_MIXED_ADD and avg_ms > 0:
        valu_rate = msms_per_launch * adds_per_msm * VALU_PER_MIXED_ADD[dom] / (avg_ms * 1e-3)
        sustained["valu_instructions_per_mixed_addition"] = VALU_PER_MIXED_ADD[dom]
        sustained["valu_instruction_rate_frac_of_a_pure_random_mad_stream"] = valu_rate / (256 * 4 * 64 / INT_MAD_NS_RANDOM)
    return {
        "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "algorithmic_bytes_per_launch": msms_per_launch * ALGO_BYTES_PER_MSM,
        "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step,
        "concurrency_note": "the %d launches of a step run concurrently on separate streams, each on a share of "
                            "the CUs: per-launch rates are per share; multiply by %d for the rate while both run"
                            % (launches_per_step, launches_per_step) if launches_per_step > 1 else "one launch per step",
        "note": "integer-ALU bound, not HBM bound (about 440-600 32-bit multiply-adds per algorithmic byte): see int_mad",
        "gather": gather,
        "int_mad": {"mad_u64_u32_per_launch": mads_per_launch,
                    "achieved_Gmad_per_s": mad_rate / 1e9,
                    "peak_Gmad_per_s": INT_MAD_PEAK_MEASURED / 1e9,
                    "peak_note": "measured v_mad_u64_u32 issue rate (tools/ubench_mad.hip: 4.17 cycles per wave-instruction per SIMD with 8 waves per "
                                 "SIMD, at the 2.40 GHz the chip holds in that short loop; profiles/r02_ubench_mad.txt)",
                    "frac": mad_rate / INT_MAD_PEAK_MEASURED,
                    "peak_at_kernel_occupancy_and_clock_Gmad_per_s": 256 * 4 * 64 / INT_MAD_CYCLES_2_WAVES * KERNEL_CLOCK_HZ / 1e9,
                    "frac_at_kernel_occupancy_and_clock": mad_rate / (256 * 4 * 64 / INT_MAD_CYCLES_2_WAVES * KERNEL_CLOCK_HZ),
                    "peak_at_kernel_occupancy_and_clock_note": "the same micro-benchmark with 2 waves per SIMD (all that 204 VGPRs allow: 4.63 cycles) "
                                                               "at the ~2.05 GHz effective the board holds under this kernel (GRBM_GUI_ACTIVE / launch time). Round 3 found that micro-benchmark's loops were not "
                                                               "alignment-controlled (multiply-adds at 4 mod 8 make the rate occupancy-dependent): peak_sustained_random_operands is the measured ceiling",
                    "peak_theoretical_Gmad_per_s": INT_MAD_PEAK_THEORETICAL / 1e9,
                    "peak_theoretical_note": "256 CUs x 4 SIMDs x 64 lanes / 4 cycles per wave-instruction at the 2.4 GHz maximum clock; "
                                             "the kernel holds 1.99-2.08 GHz (GRBM_GUI_ACTIVE, profiles/r02_pmc_*)",
                    "frac_of_theoretical": mad_rate / INT_MAD_PEAK_THEORETICAL,
                    **sustained,
                    # all multiply-adds of a step over the step's wall time (every kernel, all streams):
                    "frac_whole_step": mads_per_launch * launches_per_step / (elapsed / steps) / INT_MAD_PEAK_MEASURED},
    }, nwin


NTT_ALGO_BYTES = 131072 + 131072               # one 4096-point Fr transform reads and writes 131,072 B
NTT_PRODUCTS_PER_BLOB = 30720                  # 22,528 butterflies + entry and exit of every element
NTT_MADS_PER_PRODUCT = 190                     # 10 lazy limbs of 28 bits (fr28.cuh)


def config_legs(K, capi, D, B, ts, dev, torch, np, direct_bits):
    """BASELINE configs[2]-[4] and the Fr transform, each a short region of its own on the engine of
    the headline, timed like it (untimed warm-up, synchronize on both sides, hipEvent kernel times from the library).
    One GPU only; never part of `value`. A leg that fails reports its error instead of taking the line down."""
    out = {"engine_direct_bits": direct_bits}
    stream = torch.cuda.current_stream(dev).cuda_stream
    ckzg_outputs = []

    def region(step, steps, warmup):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize(dev)
        capi.profile_reset()
        capi.profile_enable(True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        capi.profile_enable(False)
        pr = capi.profile_report()
        return el, {name: {"launches": v["launches"], "avg_ms": v["total_ms"] / max(1, v["launches"])} for name, v in pr.items()}

    def leg(name, fn):
        try:
            out[name] = fn()
            out[name]["leg_name"] = name
        except Exception as e:      # noqa: BLE001 -- the headline must survive a leg
            out[name] = {"error": repr(e)}

    def dev_bytes(b):
        return torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).to(dev)

    nb = 256
    d_blobs = dev_bytes(B.synthetic_batch(7000, nb))
    d_comm = torch.empty(48 * nb, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(nb, dtype=torch.int32, device=dev)
    K.blob_to_kzg_commitment_batch_device(d_comm.data_ptr(), d_blobs.data_ptr(), nb, ts, stream, d_st.data_ptr())
    torch.cuda.synchronize(dev)

    def blob_proof(n_streams):
        def run():
            streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else []
            outs = [torch.empty(48 * nb, dtype=torch.uint8, device=dev) for _ in range(max(1, n_streams))]
            stats = [torch.zeros(nb, dtype=torch.int32, device=dev) for _ in range(max(1, n_streams))]
            k = [0]

            def step():
                i = k[0] % max(1, n_streams)
                k[0] += 1
                K.compute_blob_kzg_proof_batch_device(outs[i].data_ptr(), d_blobs.data_ptr(), d_comm.data_ptr(), nb, ts,
                                                      streams[i].cuda_stream if streams else stream, stats[i].data_ptr())
            # cold: the leg's first three calls, no warm-up at all (whatever state the host threads are in: the library takes the GPU's
            # hash kernel while they are cold and wakes them on the side); steady: forty calls after three untimed ones
            cold_el, _ = region(step, 3, 0)
            steps = 40     # (r05: 20 -> 40; a process pays one call of ~10 ms somewhere among its first host-assisted calls, profiles/r05_experiments.md section 8)
            el, kern = region(step, steps, 3)
            assert all(int(x.abs().sum().item()) == 0 for x in stats) and all(torch.equal(o, outs[0]) for o in outs)
            return {"workload": "BASELINE configs[2]: compute_blob_kzg_proof, batch=%d device-resident blobs per call, %d caller stream%s"
                                % (nb, max(1, n_streams), "s (consecutive calls alternate; the library overlaps one call's hash with the other's MSM)" if n_streams > 1 else ""),
                    "value": nb * steps / el, "unit": "proofs/s", "steps": steps, "warmup": 3, "ms_per_step": el / steps * 1e3, "kernels": kern,
                    "cold_value": nb * 3 / cold_el, "cold_calls": 3, "cold_ms_per_step": cold_el / 3 * 1e3}
        return run
    def commit_two_streams():
        n = BLOBS_PER_GPU
        d_b = dev_bytes(B.synthetic_batch(0, n))
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        outs = [torch.empty(48 * n, dtype=torch.uint8, device=dev) for _ in range(2)]
        stats = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
        ts.reserve(n, caller_streams=2)
        k = [0]

        def step():
            i = k[0] % 2
            k[0] += 1
            K.blob_to_kzg_commitment_batch_device(outs[i].data_ptr(), d_b.data_ptr(), n, ts, streams[i].cuda_stream, stats[i].data_ptr())
        steps = 10
        el, kern = region(step, steps, 4)
        assert all(int(x.abs().sum().item()) == 0 for x in stats) and torch.equal(outs[0], outs[1])
        return {"workload": "BASELINE configs[1] as a pipelined producer issues it: batch=%d device-resident blobs per call, consecutive calls alternate "
                            "between two caller streams, so the latency-shaped tail of one call (lane fold, inversion) runs beside the next call's accumulation" % n,
                "value": n * steps / el, "unit": "ops/s", "steps": steps, "warmup": 4, "ms_per_step": el / steps * 1e3, "kernels": kern}
    # every single-stream leg runs before either two-stream leg: the second caller stream creates the settings' twin context
    # (the library now picks the two-stream geometry only while the other context is busy, engine.hip: peer_busy)
    leg("blob_proof_b256", blob_proof(1))

    def commit_prove():
        d_c2 = torch.empty(48 * nb, dtype=torch.uint8, device=dev)
        d_p2 = torch.empty(48 * nb, dtype=torch.uint8, device=dev)
        steps = 20
        el, kern = region(lambda: K.commit_and_prove_batch_device(d_c2.data_ptr(), d_p2.data_ptr(), d_blobs.data_ptr(), nb, ts, stream, d_st.data_ptr()), steps, 3)
        assert int(d_st.abs().sum().item()) == 0 and torch.equal(d_c2, d_comm)
        return {"workload": "commitment AND blob proof of batch=%d device-resident blobs in one pass (configs[1] then configs[2] on its output)" % nb,
                "value": nb * steps / el, "unit": "pairs/s", "steps": steps, "warmup": 3, "ms_per_step": el / steps * 1e3, "kernels": kern}
    leg("commit_prove_b256", commit_prove)

    def blob_proof_1024(ckzg):
        """a whole chunk of proofs per call on one stream: reference mode here, c-kzg mode after the settings object has moved to the Lagrange form
        (last leg but one): there the quotient is taken in evaluation form (k_eval_quotient_evalform) and no transform runs at all"""
        def run():
            n = BLOBS_PER_GPU
            d_b = dev_bytes(B.synthetic_batch(7000, n, big_endian=not ckzg))
            d_c = torch.empty(48 * n, dtype=torch.uint8, device=dev)
            d_p = torch.empty(48 * n, dtype=torch.uint8, device=dev)
            d_s = torch.zeros(n, dtype=torch.int32, device=dev)
            K.blob_to_kzg_commitment_batch_device(d_c.data_ptr(), d_b.data_ptr(), n, ts, stream, d_s.data_ptr())
            torch.cuda.synchronize(dev)
            assert int(d_s.abs().sum().item()) == 0
            steps = 10
            el, kern = region(lambda: K.compute_blob_kzg_proof_batch_device(d_p.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), n, ts, stream, d_s.data_ptr()), steps, 3)
            assert int(d_s.abs().sum().item()) == 0
            return {"workload": "BASELINE configs[2] at batch=%d device-resident blobs per call, one caller stream, %s" %
                                (n, "c-kzg-4844 semantics on the Lagrange form: the quotient in evaluation form, no transform" if ckzg else "reference semantics"),
                    "value": n * steps / el, "unit": "proofs/s", "steps": steps, "warmup": 3, "ms_per_step": el / steps * 1e3, "kernels": kern}
        return run
    leg("blob_proof_b1024", blob_proof_1024(False))

    def ckzg_commit():
        n = BLOBS_PER_GPU
        d_le = dev_bytes(B.synthetic_batch(0, n, big_endian=False))
        d_o = torch.empty(48 * n, dtype=torch.uint8, device=dev)
        d_s = torch.zeros(n, dtype=torch.int32, device=dev)
        prev_mode = K.get_mode()
        K.set_mode(K.MODE_CKZG)           # the process-wide default: it never moves a table, so on a table that leaves no room for a second one
        try:                              # (the headline's 16-bit one) this leg stays on the transform path and times k_ntt4096
            steps = 10
            el, kern = region(lambda: K.blob_to_kzg_commitment_batch_device(d_o.data_ptr(), d_le.data_ptr(), n, ts, stream, d_s.data_ptr()), steps, 3)
        finally:
            K.set_mode(prev_mode)         # (whatever the process default was, not a constant)
        assert int(d_s.abs().sum().item()) == 0
        ckzg_outputs.append(bytes(d_o.cpu().numpy().tobytes()))
        ntt_ms = kern.get("k_ntt4096", {}).get("avg_ms", 0.0)
        ntt_traffic = ntt_traffic_source = None
        try:   # committed PMC passes over the all-legs run (tools/pmc_traffic_all.py): every k_ntt4096 launch there is this one
            import glob
            src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_traffic_all_legs.json")))[-1]     # the latest round's
            pmc = json.load(open(src))
            ntt_traffic = pmc["kernels"]["k_ntt4096"]["largest_launch_traffic_bytes"]
            ntt_traffic_source = "profiles/%s: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, NOT measured in this run" % os.path.basename(src)
        except Exception:
            pass
        launches_per_step = max(1, round(kern.get("k_ntt4096", {}).get("launches", steps) / steps))
        per_launch = n / launches_per_step
        ach = per_launch * NTT_ALGO_BYTES / (ntt_ms * 1e-3) / 1e9 if ntt_ms > 0 else 0.0
        mads = per_launch * NTT_PRODUCTS_PER_BLOB * NTT_MADS_PER_PRODUCT
        return {"workload": "c-kzg-4844 semantics: batch=%d little-endian evaluation-form blobs -> commitments (k_ntt4096: parse, range check, "
                            "inverse 4096-point Fr transform, in front of the same MSM)" % n,
                "value": n * steps / el, "unit": "ops/s", "steps": steps, "warmup": 3, "ms_per_step": el / steps * 1e3, "kernels": kern,
                "ntt_roofline": {"kernel": "k_ntt4096", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_launch": per_launch * NTT_ALGO_BYTES, "avg_launch_ms": ntt_ms, "blobs_per_launch": per_launch,
                                 "traffic": ntt_traffic if per_launch == BLOBS_PER_GPU else None, "traffic_source": ntt_traffic_source,
                                 "int_mad": {"mad_u64_u32_per_launch": mads, "achieved_Gmad_per_s": mads / (ntt_ms * 1e-3) / 1e9 if ntt_ms > 0 else 0.0,
                                             "peak_theoretical_Gmad_per_s": INT_MAD_PEAK_THEORETICAL / 1e9,
                                             "frac_of_theoretical": mads / (ntt_ms * 1e-3) / INT_MAD_PEAK_THEORETICAL if ntt_ms > 0 else 0.0},
                                 "note": "SURVEY 8d: 262,144 algorithmic bytes per 4096-point transform; the whole blob stays in LDS for the twelve stages, "
                                         "so HBM sees each element once in and once out; the kernel is bound by its 30,720 Fr products per blob"}}
    leg("ckzg_commit_b1024_with_ntt", ckzg_commit)

    def ckzg_commit_lagrange():
        """the same c-kzg commitments with the settings object SWITCHED to c-kzg mode (lwkzg_settings_set_mode): its direct table moves to
        the Lagrange form -- beside the monomial one if both fit, instead of it otherwise (the seconds are reported) -- and a commitment is
        the MSM over the blob's evaluations as they stand: no k_ntt4096. Last leg: the table stays in that form."""
        n = BLOBS_PER_GPU
        d_le = dev_bytes(B.synthetic_batch(0, n, big_endian=False))
        d_o = torch.empty(48 * n, dtype=torch.uint8, device=dev)
        d_s = torch.zeros(n, dtype=torch.int32, device=dev)
        t0 = time.perf_counter()
        ts.set_mode(K.MODE_CKZG)
        t_move = time.perf_counter() - t0
        steps = 10
        el, kern = region(lambda: K.blob_to_kzg_commitment_batch_device(d_o.data_ptr(), d_le.data_ptr(), n, ts, stream, d_s.data_ptr()), steps, 3)
        assert int(d_s.abs().sum().item()) == 0
        assert not ckzg_outputs or ckzg_outputs[0] == bytes(d_o.cpu().numpy().tobytes()), "the Lagrange form and the transform path disagree"
        return {"workload": "c-kzg-4844 semantics on the Lagrange form of the setup: batch=%d little-endian evaluation-form blobs -> commitments, no transform "
                            "(range check + copy, then the same MSM kernel over [l_i(tau)]G)" % n,
                "value": n * steps / el, "unit": "ops/s", "steps": steps, "warmup": 3, "ms_per_step": el / steps * 1e3, "kernels": kern,
                "table_forms": ts.direct_table_forms(), "settings_set_mode_s": t_move, "equal_to_transform_path": bool(ckzg_outputs)}

    def verify_batch():
        n = 4096
        h_blobs = B.synthetic_batch(9000, n)
        h_comms = b"".join(K.blob_to_kzg_commitment_batch(h_blobs, ts))
        h_proofs = b"".join(K.compute_blob_kzg_proof_batch(h_blobs, h_comms, ts))
        steps = 3

        def step():
            assert D.verify_blob_kzg_proof_batch_sharded(h_blobs, h_comms, h_proofs, n, ts)
        el, kern = region(step, steps, 1)
        return {"workload": "BASELINE configs[3] on one GPU: verify_blob_kzg_proof_batch of %d blobs (one transcript, one r, one pairing check), "
                            "host-pointer ABI: 512 MiB of pageable host blobs uploaded inside the clock" % n,
                "value": n * steps / el, "unit": "blobs/s", "steps": steps, "warmup": 1, "ms_per_step": el / steps * 1e3, "kernels": kern}
    leg("verify_batch_b4096", verify_batch)

    def verify_batch_device():
        n = 4096
        h_blobs = B.synthetic_batch(9000, n)
        h_comms = b"".join(K.blob_to_kzg_commitment_batch(h_blobs, ts))
        h_proofs = b"".join(K.compute_blob_kzg_proof_batch(h_blobs, h_comms, ts))
        d_b, d_c, d_p = dev_bytes(h_blobs), dev_bytes(h_comms), dev_bytes(h_proofs)
        steps = 5

        def step():
            assert K.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, ts, stream)
        el, kern = region(step, steps, 1)
        return {"workload": "BASELINE configs[3] on one GPU, the same %d blobs already in HBM: lwkzg_verify_blob_kzg_proof_batch_device (device pointers in, "
                            "the verdict out; the 160-byte records and the pairing check are all the host sees)" % n,
                "value": n * steps / el, "unit": "blobs/s", "steps": steps, "warmup": 1, "ms_per_step": el / steps * 1e3, "kernels": kern}
    leg("verify_batch_b4096_device", verify_batch_device)

    def tiled_msm():
        tiles = 256
        d_t = dev_bytes(B.synthetic_batch(5000, tiles))
        steps = 10
        el, kern = region(lambda: D.msm_tiled_sharded(d_t, tiles * 4096, ts, dev), steps, 3)
        dom = ("k_direct_accumulate_asm" if "k_direct_accumulate_asm" in kern else "k_direct_accumulate") if direct_bits else \
              ("k_bucket_accumulate_asm" if "k_bucket_accumulate_asm" in kern else "k_bucket_accumulate")
        ms = kern.get(dom, {}).get("avg_ms", 0.0)
        lps = max(1, round(kern.get(dom, {}).get("launches", steps) / steps))
        algo = (tiles * 4096 * 128 + 48) / lps  # 2^20 x (32 B scalar + 96 B affine point) + 48 B, per launch
        return {"workload": "BASELINE configs[4] on one GPU: one 2^20-term G1 MSM over the setup tiled 256 times",
                "value": tiles * 4096 * steps / el, "unit": "terms/s", "steps": steps, "warmup": 3, "ms_per_step": el / steps * 1e3, "kernels": kern,
                "roofline": {"kernel": dom, "bound": "hbm", "achieved": algo / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0, "algorithmic_bytes_per_launch": algo, "avg_launch_ms": ms,
                             "launches_per_step": lps, "traffic": None}}
    leg("tiled_msm_2_pow_20", tiled_msm)
    leg("blob_proof_b256_two_streams", blob_proof(2))
    leg("commit_b1024_two_streams", commit_two_streams)

    def commit_after_two_streams():
        # the headline's call again, on ONE stream, after the twin context exists: must be the single-stream geometry again
        n = BLOBS_PER_GPU
        d_b = dev_bytes(B.synthetic_batch(0, n))
        d_o = torch.empty(48 * n, dtype=torch.uint8, device=dev)
        d_s = torch.zeros(n, dtype=torch.int32, device=dev)
        steps = 10
        el, kern = region(lambda: K.blob_to_kzg_commitment_batch_device(d_o.data_ptr(), d_b.data_ptr(), n, ts, stream, d_s.data_ptr()), steps, 3)
        assert int(d_s.abs().sum().item()) == 0
        return {"workload": "BASELINE configs[1] once more on ONE caller stream after the two-stream legs (the twin context exists and is idle)",
                "value": n * steps / el, "unit": "ops/s", "steps": steps, "warmup": 3, "ms_per_step": el / steps * 1e3, "kernels": kern}
    leg("commit_b1024_one_stream_after_twin", commit_after_two_streams)
    leg("ckzg_commit_b1024_lagrange", ckzg_commit_lagrange)
    if "error" not in out["ckzg_commit_b1024_lagrange"]:
        leg("ckzg_blob_proof_b1024_lagrange", blob_proof_1024(True))
    return out

LINE_LIMIT = 8000            # bytes of the one stdout line (the driver keeps a bounded tail of stdout)


def _round(x, sig=6):
    """floats to `sig` significant digits, recursively (the line is read by a parser, not by a person)"""
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if x == x and x not in (float("inf"), float("-inf")) else None
    if isinstance(x, dict):
        return {k: _round(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _dominant(kernels):
    """(name, avg ms) of the kernel a leg spent most of its time in"""
    if not kernels:
        return None, None
    name = max(kernels, key=lambda k: kernels[k].get("avg_ms", 0.0) * kernels[k].get("launches", 1))
    return name, kernels[name].get("avg_ms")


def _roof_compact(r):
    out = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "launches_per_step"))
    out.setdefault("traffic", None)
    if r.get("traffic_source"):
        out["traffic_source"] = r["traffic_source"].split(":")[0][:80]
    if isinstance(r.get("int_mad"), dict):
        out["int_mad"] = _pick(r["int_mad"], ("achieved_Gmad_per_s", "peak_theoretical_Gmad_per_s", "frac_of_theoretical", "frac_whole_step"))
    return out


def _leg_compact(l):
    if not isinstance(l, dict):
        return l
    if "error" in l:
        return {"error": str(l["error"])[:120]}
    out = _pick(l, ("value", "unit", "ms_per_step", "steps", "cold_value"))
    name, ms = _dominant(l.get("kernels"))
    if name:
        # the dominant kernel as the share of the leg's wall clock its launches add up to (sum of launch durations / wall): a leg whose
        # launches overlap on two caller streams shows a share above what one stream could have, where a per-launch average longer than
        # the step looked like an accounting error; one-stream legs carry the per-launch average as well
        k = l["kernels"][name]
        wall_ms = (l.get("ms_per_step") or 0) * (l.get("steps") or 0)
        out["kernel"] = name
        if wall_ms > 0:
            out["kernel_sum_over_wall"] = k.get("avg_ms", 0.0) * k.get("launches", 0) / wall_ms
        if "two_streams" not in str(l.get("leg_name", "")) and ms is not None and ms <= (l.get("ms_per_step") or ms):
            out["kernel_ms"] = ms
    for sub in ("ntt_roofline", "roofline"):
        if isinstance(l.get(sub), dict):
            out[sub] = _pick(l[sub], ("kernel", "achieved", "peak", "frac", "avg_launch_ms", "traffic"))
            if isinstance(l[sub].get("int_mad"), dict):
                out[sub]["int_mad_frac_of_theoretical"] = l[sub]["int_mad"].get("frac_of_theoretical")
    return out


def api_latency_leg(K, B, ts, bits):
    """The reference's own call shape (/root/reference/src/lib.rs:253-283, 300-404, 456-505): ONE blob per synchronous call through the
    nine symbols, host pointers in and out, one thread, on the engine a plain load selected. Wall clock per call (PCIe, launches, the
    host's share of the work all inside); never `value`."""
    blob = B.synthetic_blob(9001)
    c = K.blob_to_kzg_commitment(blob, ts)
    pr = K.compute_blob_kzg_proof(blob, c, ts)
    z = blob[32:64]
    ops = {"blob_to_kzg_commitment": lambda: K.blob_to_kzg_commitment(blob, ts),
           "compute_blob_kzg_proof": lambda: K.compute_blob_kzg_proof(blob, c, ts),
           "compute_kzg_proof": lambda: K.compute_kzg_proof(blob, z, ts),
           "verify_blob_kzg_proof": lambda: K.verify_blob_kzg_proof(blob, c, pr, ts)}
    out = {"engine_direct_bits": bits, "calls_timed": 20, "unit": "ms per call, one blob, one thread (median; best)",
           "note": "the nine reference symbols are one-blob synchronous calls: a handful of blobs runs on the cooperative kernel "
                   "(k_coop_msm_asm, four lanes per group addition) and the inversion + compression of the result on the calling thread"}
    for name, fn in ops.items():
        for _ in range(3):
            fn()
        t = []
        for _ in range(20):
            t0 = time.perf_counter()
            fn()
            t.append((time.perf_counter() - t0) * 1e3)
        t.sort()
        out[name] = {"median_ms": t[10], "best_ms": t[0], "calls_per_s_one_thread": 1e3 / t[10]}
    # the reference's threading contract (KZGSettings is read-only after load, src/lib.rs:253-283): sixteen threads, one blob per call; the
    # library merges whoever is waiting into one launch set (csrc/front.h)
    import threading
    blobs = [B.synthetic_blob(9100 + i) for i in range(16)]
    per_thread = 40

    def worker(i):
        for _ in range(per_thread):
            K.blob_to_kzg_commitment(blobs[i], ts)
    for rep in range(2):      # the second round is the timed one (threads and staging warm)
        th = [threading.Thread(target=worker, args=(i,)) for i in range(16)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        el = time.perf_counter() - t0
    out["blob_to_kzg_commitment_16_threads"] = {"calls_per_s": 16 * per_thread / el, "threads": 16, "calls_per_thread": per_thread}
    return out


def compact_line(res, detail_path=None):
    """The ONE stdout line: the contract's fields first, then one small object per engine and per leg. Numbers only; every note,
    breakdown and per-kernel table stays in the detail file. Pure function of the detail dictionary (tests/test_bench_accounting_cpu.py
    runs it on committed detail files and on a synthetic worst case and asserts the size and the keys)."""
    line = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    if isinstance(res.get("step_ms"), dict):     # r06: the spread of the K timed steps (per-step events on the caller's stream)
        line["step_ms"] = _pick(res["step_ms"], ("p50_ms", "min_ms", "max_ms"))
    line["config"] = _pick(res.get("config", {}), ("workload", "blobs_per_gpu_per_step", "caller_streams", "scalars", "direct_bits",
                                                    "direct_bits_min_over_ranks", "mode", "op", "parallelism"))
    line["roofline"] = _roof_compact(res.get("roofline", {}))
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "value_without_srs_rebuild", "single_thread_ops_per_s",
                                          "single_thread_ops_per_s_without_srs_rebuild", "srs_rebuild_ms_per_call", "gpu_outputs_match_oracle"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
    line["kernels_avg_ms"] = {k: v.get("avg_ms") for k, v in (res.get("kernels") or {}).items()}
    for eng in ("default_engine", "bucket_engine"):
        e = res.get(eng)
        if isinstance(e, dict):
            line[eng] = _pick(e, ("value", "unit", "ms_per_step", "table_bytes"))
            line[eng]["bits"] = e.get("direct_bits")
            r = e.get("roofline") or {}
            line[eng]["kernel"], line[eng]["kernel_ms"] = r.get("kernel"), r.get("avg_launch_ms")
            line[eng]["int_mad_frac_of_theoretical"] = (r.get("int_mad") or {}).get("frac_of_theoretical")
            line[eng]["frac"] = r.get("frac")
    if isinstance(res.get("host_abi"), dict):
        line["host_abi"] = _pick(res["host_abi"], ("value", "unit", "ms_per_call_median", "blobs_per_call"))
    if isinstance(res.get("api_latency"), dict):
        line["api_latency_ms"] = {k: v.get("median_ms") for k, v in res["api_latency"].items() if isinstance(v, dict) and "median_ms" in v}
        mt = res["api_latency"].get("blob_to_kzg_commitment_16_threads")
        if isinstance(mt, dict):
            line["api_latency_ms"]["commit_calls_per_s_16_threads"] = mt.get("calls_per_s")
    if isinstance(res.get("configs"), dict):
        line["configs"] = {k: _leg_compact(v) for k, v in res["configs"].items()}
    line["dist"] = _pick(res.get("dist") or {}, ("initialised", "backend", "ranks", "nccl_version", "devices"))
    pr = (res.get("dist") or {}).get("per_rank")
    if isinstance(pr, list) and len(pr) > 1:
        line["dist"]["per_rank_value"] = [r.get("value") for r in pr]
    if isinstance(res.get("box"), dict):
        line["box"] = res["box"]
    line.update(_pick(res, ("msm_path", "hip_first_use_init_s", "setup_load_s", "direct_table_build_s", "direct_table_build_s_max_over_ranks")))
    if detail_path:
        line["detail"] = detail_path
    line = _round(line)
    text = json.dumps(line, separators=(",", ":"))
    # belt and braces: whatever a future field does, the contract's head of the line survives
    for drop in ("kernels_avg_ms", "host_abi", "api_latency_ms", "configs", "bucket_engine", "default_engine", "dist"):
        if len(text) < LINE_LIMIT:
            break
        line.pop(drop, None)
        line["dropped_for_size"] = line.get("dropped_for_size", []) + [drop]
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    return text


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes through torch.distributed.run --
    BEFORE this process has touched the GPU (nothing has imported torch yet) -- relay their stdout (rank 0's one line) and exit with
    their return code. The launcher form (WORLD_SIZE / RANK in the environment) never comes here."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("bench.py: no launcher in the environment, starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr)
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BLOBS_PER_GPU, help="blobs per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the short untimed-region legs of the commit bench: default engine, bucket engine, host-pointer ABI")
    ap.add_argument("--op", default="commit", choices=["commit", "blob_proof", "commit_prove", "verify_batch", "tiled_msm"],
                    help="commit = the headline (BASELINE configs[1]); blob_proof = configs[2]; commit_prove = commitment AND blob proof of every blob in one pass (lwkzg_commit_and_prove_batch_device); verify_batch = configs[3], host-pointer ABI; tiled_msm = configs[4], one 2^20-term MSM split over the GPUs (strong scaling)")
    ap.add_argument("--mode", default="reference", choices=["reference", "ckzg"],
                    help="reference = lambdaworks_kzg semantics (default, the headline); ckzg = c-kzg-4844 semantics (adds the inverse NTT)")
    ap.add_argument("--direct-bits", default="auto",
                    help="MSM engine of the timed region (lwkzg_enable_direct_table): auto = the widest direct table of 16 .. 10 bits "
                         "that fits in HBM, else the bucket engine; default = what a plain load selected; 0 = bucket engine; "
                         "10 .. 16 = that width or fail")
    ap.add_argument("--scalars", default="31byte", choices=["31byte", "full"],
                    help="31byte = the BASELINE workload (31 random bytes per element, SURVEY 8d); full = 32 random bytes per element, "
                         "reduced mod r by the parse kernel (reference mode only): every window of every scalar is busy")
    ap.add_argument("--caller-streams", type=int, default=1,
                    help="commit / blob_proof: consecutive steps alternate between this many caller streams (each with its own output "
                         "buffer), as a pipelined producer would issue them; the library then overlaps the latency-shaped head of one "
                         "call with the MSM of the other (engine.hip: pick_ctx). 1 = every step on one stream, strictly in series")
    ap.add_argument("--idle-ms", type=float, default=0.0,
                    help="experiment: the host synchronizes and sleeps this long before every step (is a kernel slower after an idle gap?)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU plumbing tests)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and take every collective branch (setup broadcast, barriers, max over ranks) "
                         "at --gpus 1 too: the RCCL code path on a one-GPU box")
    ap.add_argument("--no-config-legs", action="store_true",
                    help="skip the short legs for BASELINE configs[2]-[4] and the Fr transform (they run at --gpus 1 only)")
    ap.add_argument("--hbm-headroom-gib", type=float, default=6.0,
                    help="--direct-bits auto: device memory the chosen table must leave free on this rank (workspaces of the other "
                         "legs, RCCL's buffers, the caller's own tensors)")
    args = ap.parse_args()
    if args.op == "commit_prove" and args.caller_streams > 1:
        ap.error("--op commit_prove runs on one caller stream")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    # stdout carries exactly ONE line, the JSON result: everything else that writes to file descriptor 1 meanwhile (RCCL
    # prints a version banner there when its first communicator comes up) goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    # r06: the headline runs FIRST. The load is told to build no direct table (LWKZG_DIRECT_BITS=0, set here before the
    # library reads its environment), so that the timed region's table is this process's first large allocation and `direct_table_build_s`
    # reports a build that does not follow the bench's own free of a 41 GB default table (r05: 6.5 s of which 5.5 were hipMalloc waiting for the
    # driver's scrub). The default engine's leg -- the table a plain load selects -- and the bucket engine's follow the headline and the config legs.
    headline_first = (args.op == "commit" and not args.no_extra_legs and args.direct_bits == "auto" and "LWKZG_DIRECT_BITS" not in os.environ)
    if headline_first:
        os.environ["LWKZG_DIRECT_BITS"] = "0"

    import numpy as np
    import torch
    import torch.distributed as dist

    import blobs as B
    import lambdaworks_kzg_amd as K
    from lambdaworks_kzg_amd import capi
    from lambdaworks_kzg_amd import dist as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    dev_index = local_rank % max(1, torch.cuda.device_count())   # == local_rank on a real N-GPU node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    K.set_device(dev_index)
    K.set_mode(K.MODE_REFERENCE if args.mode == "reference" else K.MODE_CKZG)
    distributed = world > 1 or args.force_dist
    dist_info = {"initialised": False, "backend": None, "ranks": world}
    if distributed:
        if world == 1:     # --force-dist without a launcher: a rendezvous with ourselves
            import socket
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                sk = socket.socket()
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
                sk.close()
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        dist_info = {"initialised": True, "backend": dist.get_backend(), "ranks": dist.get_world_size(),
                     "nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else None,
                     "note": ("backend nccl IS RCCL on ROCm" if dist.get_backend() == "nccl" else "backend %s: a plumbing rehearsal, not the RCCL path" % dist.get_backend()) + "; collectives of this run: one broadcast of the setup image, barriers, "
                             "the max-over-ranks of the timings"}

    # first use of the HIP runtime by this process, timed apart from the load (a fresh process on a fresh box pays about a
    # second here, whichever call comes first)
    t_init0 = time.perf_counter()
    capi.runtime_init()
    t_hip_init = time.perf_counter() - t_init0

    # trusted setup: rank 0 parses + validates + builds the fixed-base table, one RCCL broadcast delivers it; every rank
    # then holds the engine a plain load selects (the default engine: engine.hip, direct_from_env)
    t_load0 = time.perf_counter()
    ts = K.TrustedSetup.from_file(SETUP) if rank == 0 else None
    if distributed:
        ts = D.broadcast_trusted_setup(ts, dev, src=0)
    t_load = time.perf_counter() - t_load0
    default_bits = ts.direct_table_bits()
    if headline_first:   # what a plain load would have selected on this device (engine.hip: direct_from_env): the widest of 13 .. 10 bits within a quarter of the free HBM
        free_b0 = torch.cuda.mem_get_info(dev)[0]
        default_bits = next((b for b in (13, 12, 11, 10) if capi.direct_table_bytes(b, 112) <= free_b0 // 4), 0)
    load_breakdown = ts.timing_report()

    n = args.batch
    first = rank * n                       # shard: blob k of the job lives on GPU floor(k / n)
    host = np.frombuffer(B.synthetic_batch(first, n, big_endian=(args.mode == "reference"), full_range=(args.scalars == "full")), dtype=np.uint8)
    d_blobs = torch.from_numpy(host.copy()).to(dev)
    d_out = torch.empty(48 * n, dtype=torch.uint8, device=dev)
    d_status = torch.zeros(n, dtype=torch.int32, device=dev)
    d_comm = None
    ts.reserve(n)
    stream = torch.cuda.current_stream(dev).cuda_stream
    n_cs = max(1, args.caller_streams)
    cstreams = [torch.cuda.Stream(device=dev) for _ in range(n_cs)] if n_cs > 1 else []
    couts = [torch.empty(48 * n, dtype=torch.uint8, device=dev) for _ in range(n_cs)] if n_cs > 1 else []
    cstats = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(n_cs)] if n_cs > 1 else []
    step_no = [0]

    h_blobs = h_comms = h_proofs = None
    d_tiles = None
    tiles_total = 256                      # 2^20 terms over the setup tiled 256 times

    def step():
        if args.op == "tiled_msm":      # each rank sums its share of the tiles, one all_gather of 48-byte partial sums
            D.msm_tiled_sharded(d_tiles, int(d_tiles.numel()) // 32, ts, dev)
        elif args.op == "verify_batch":   # ONE batch over all ranks: one transcript, one r, one pairing check (dist.py)
            assert D.verify_blob_kzg_proof_batch_sharded(h_blobs, h_comms, h_proofs, n, ts)
        elif n_cs > 1:                    # step k on caller stream k mod N, into that stream's own output buffer
            k = step_no[0] % n_cs
            step_no[0] += 1
            if args.op == "commit":
                K.blob_to_kzg_commitment_batch_device(couts[k].data_ptr(), d_blobs.data_ptr(), n, ts, cstreams[k].cuda_stream, cstats[k].data_ptr())
            else:
                K.compute_blob_kzg_proof_batch_device(couts[k].data_ptr(), d_blobs.data_ptr(), d_comm.data_ptr(), n, ts,
                                                      cstreams[k].cuda_stream, cstats[k].data_ptr())
        elif args.op == "commit":
            K.blob_to_kzg_commitment_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), n, ts, stream, d_status.data_ptr())
        elif args.op == "commit_prove":
            K.commit_and_prove_batch_device(d_comm.data_ptr(), d_out.data_ptr(), d_blobs.data_ptr(), n, ts, stream, d_status.data_ptr())
        else:
            K.compute_blob_kzg_proof_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), d_comm.data_ptr(), n, ts, stream,
                                                  d_status.data_ptr())

    local_elapsed = [0.0]     # this rank's own clock over the last timed region (the contract's number is the max over ranks)
    step_spread = [None]      # p50 / min / max of the last timed region's steps (r06)

    def timed_region(steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        for _ in range(warmup):
            step()
        torch.cuda.synchronize(dev)
        capi.profile_reset()
        capi.profile_enable(True)             # hipEvent pairs around every kernel, on the launch stream
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)
        # one event per step on the caller's stream (the library joins its side streams back into it): the spread of the K steps, not only their sum
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if n_cs == 1 and args.op in ("commit", "blob_proof", "commit_prove") else []
        t0 = time.perf_counter()
        if marks:
            marks[0].record(torch.cuda.current_stream(dev))
        for i in range(steps):
            if args.idle_ms > 0:
                torch.cuda.synchronize(dev)
                time.sleep(args.idle_ms * 1e-3)
            step()
            if marks:
                marks[i + 1].record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        el = time.perf_counter() - t0
        capi.profile_enable(False)
        pr = capi.profile_report()
        local_elapsed[0] = el
        if marks:
            per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
            step_spread[0] = {"p50_ms": per[len(per) // 2], "min_ms": per[0], "max_ms": per[-1], "steps": steps,
                              "note": "GPU time between consecutive per-step events on the caller's stream, this rank"}
        if distributed:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        assert int(d_status.abs().sum().item()) == 0, "a blob was rejected"
        for k in range(len(cstats)):
            assert int(cstats[k].abs().sum().item()) == 0, "a blob was rejected"
            if k > 0:
                assert torch.equal(couts[k], couts[0]), "the caller streams disagree"
        return el, pr

    def engine_leg(bits, label):
        """the same workload on another engine, a short region of its own before the headline's (same timing rules)"""
        el, pr = timed_region(5, 2)
        roof, nwin = engine_picture(K, capi, bits, pr, el, 5, n)
        return {"engine": label, "direct_bits": bits, "value": n * world * 5 / el, "unit": "ops/s", "steps": 5, "warmup": 2,
                "ms_per_step": el / 5 * 1e3,
                "table_bytes": capi.direct_table_bytes(bits, ts.direct_row_bytes() or 112) if bits else 20 * 4096 * 112,
                "table_row_bytes": ts.direct_row_bytes() if bits else 112,
                "kernels": {name: {"launches": v["launches"], "avg_ms": v["total_ms"] / max(1, v["launches"])} for name, v in pr.items()},
                "roofline": roof}

    extra = {}
    DEFAULT_LABEL = ("what load_trusted_setup* selects by itself (LWKZG_DIRECT_BITS unset): the widest direct table of 13..10 bits within a quarter "
                     "of the free HBM, else buckets")
    if args.op == "commit" and not args.no_extra_legs and not headline_first:
        # (a) what a consumer of the nine reference symbols gets: the engine the load selected, untouched
        extra["default_engine"] = engine_leg(default_bits, DEFAULT_LABEL)
        extra["default_engine"]["setup_load_s_incl_table_build"] = t_load
        if rank == 0:
            extra["api_latency"] = api_latency_leg(K, B, ts, default_bits)
        # (b) the low-memory fallback runs LAST (below, after the config legs): every free of a table is memory the driver scrubs before
        # the next large hipMalloc returns, and the bucket leg in this place cost the timed region's table two of them (7.8 s of
        # direct_table_build_s in r05's first collection, 6.5 s of it hipMalloc waiting)

    # the engine of the timed region. Every rank builds its own table from the (broadcast) setup points, once, outside the
    # timed region; the seconds are reported per rank below.
    t_tab0 = time.perf_counter()
    direct_bits = 0
    hbm_budget = None
    if args.direct_bits == "auto":
        # per-rank HBM budget: what is free on THIS device now (the table in place is freed before the new one is built),
        # shared with every other rank of this node that was mapped to the same device, minus the headroom the rest of the
        # run needs -- so that 8 ranks on 8 devices (or N ranks on fewer) pick a width that fits instead of finding out
        # in the middle of a launch
        free_b, total_b = torch.cuda.mem_get_info(dev)
        in_place = capi.direct_table_bytes(default_bits, ts.direct_row_bytes() or 112) if default_bits else 0
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        sharing = sum(1 for lr in range(local_world) if lr % max(1, torch.cuda.device_count()) == dev_index)
        headroom = int(args.hbm_headroom_gib * (1 << 30))
        budget = (free_b + in_place) // max(1, sharing) - headroom
        fits = [b for b in (16, 15, 14, 13, 12, 11, 10) if capi.direct_table_bytes(b, 112) <= budget]
        hbm_budget = {"device_total_bytes": total_b, "free_bytes_before": free_b, "table_in_place_bytes": in_place,
                      "ranks_sharing_device": sharing, "headroom_bytes": headroom, "budget_bytes": budget,
                      "widths_within_budget": fits}
        for bits in fits + [0]:
            try:
                ts.enable_direct_table(bits)
                direct_bits = bits
                break
            except capi.KzgError as e:
                if e.rc != capi.C_KZG_MALLOC:
                    raise
        hbm_budget["chosen_bits"] = direct_bits
        hbm_budget["free_bytes_after"] = torch.cuda.mem_get_info(dev)[0]
    elif args.direct_bits == "default":     # whatever a plain load selected (engine.hip: direct_from_env)
        direct_bits = default_bits
    else:
        ts.enable_direct_table(int(args.direct_bits))
        direct_bits = int(args.direct_bits)
    t_table = time.perf_counter() - t_tab0
    table_breakdown = ts.timing_report().get("last_table_build") if direct_bits else None
    direct_bits_min, t_table_max = direct_bits, t_table
    if distributed:   # every rank should have got the same width; report it if one did not
        tb = torch.tensor([direct_bits], dtype=torch.int32, device=dev)
        dist.all_reduce(tb, op=dist.ReduceOp.MIN)
        direct_bits_min = int(tb.item())
        tt = torch.tensor([t_table], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_table_max = float(tt.item())

    if args.op == "commit_prove":
        d_comm = torch.empty(48 * n, dtype=torch.uint8, device=dev)
    if args.op == "blob_proof":
        d_comm = torch.empty(48 * n, dtype=torch.uint8, device=dev)
        K.blob_to_kzg_commitment_batch_device(d_comm.data_ptr(), d_blobs.data_ptr(), n, ts, stream, d_status.data_ptr())
        torch.cuda.synchronize(dev)

    if args.op == "tiled_msm":
        t_first, t_cnt = D.shard_range(tiles_total, world, rank)
        tiles = np.frombuffer(B.synthetic_batch(5000 + t_first, t_cnt), dtype=np.uint8)   # canonical 248-bit scalars, big-endian
        d_tiles = torch.from_numpy(tiles.copy()).to(dev)
    if args.op == "verify_batch":      # inputs of the host-pointer ABI: blobs, commitments and proofs in host memory
        h_blobs = host.tobytes()
        h_comms = b"".join(K.blob_to_kzg_commitment_batch(h_blobs, ts))
        h_proofs = b"".join(K.compute_blob_kzg_proof_batch(h_blobs, h_comms, ts))

    elapsed, prof = timed_region(args.steps, args.warmup)
    headline_spread = step_spread[0]

    # who ran where, and how fast each rank was by its own clock (the contract's `value` uses the slowest): rank, device ordinal, a hash
    # of the device's uuid, the shader clock it holds under a short multiply-add stream, its own ops/s -- so that a SCALE record shows
    # that N ranks on N devices took part, and a profiles/ summary can be matched to the box a line came from
    import hashlib
    props = torch.cuda.get_device_properties(dev)
    uuid = str(getattr(props, "uuid", "")) or "%s/%d" % (props.name, dev_index)
    try:
        clock_mhz = capi.clock_probe_mhz()
    except Exception:       # noqa: BLE001 -- a measurement aid must not take the run down
        clock_mhz = 0.0
    me = {"rank": rank, "device": dev_index, "gpu": props.name, "gpu_uuid_sha256_12": hashlib.sha256(uuid.encode()).hexdigest()[:12],
          "clock_mhz_under_mad_probe": round(clock_mhz, 1), "elapsed_s": local_elapsed[0],
          "value": (n * args.steps / local_elapsed[0]) if local_elapsed[0] > 0 and args.op != "tiled_msm" else None}
    per_rank = [me]
    if distributed:
        gathered = [None] * world
        dist.all_gather_object(gathered, me)
        per_rank = gathered

    if args.op == "commit" and not args.no_extra_legs and rank == 0:
        # (c) the same batch through the host-pointer C ABI (blobs in pageable host memory: H2D of 128 KiB per blob and D2H
        # of the 48-byte results inside the clock). Never `value`.
        hb = host.tobytes()
        K.blob_to_kzg_commitment_batch(hb, ts)
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            got = K.blob_to_kzg_commitment_batch(hb, ts)
            times.append(time.perf_counter() - t0)
        times.sort()
        assert b"".join(got) == bytes(d_out.cpu().numpy().tobytes())
        extra["host_abi"] = {"symbol": "lwkzg_blob_to_kzg_commitment_batch", "blobs_per_call": n, "direct_bits": direct_bits,
                             "value": n / times[len(times) // 2], "unit": "ops/s", "ms_per_call_median": times[len(times) // 2] * 1e3,
                             "ms_per_call_best": times[0] * 1e3, "calls": len(times),
                             "note": "PCIe-inclusive wall clock on rank 0: pageable host blobs in, 48-byte commitments out, one GPU; "
                                     "results equal to the device-resident path's"}

    if args.op == "commit" and world == 1 and not args.no_config_legs and not args.no_extra_legs and args.mode == "reference":
        extra["configs"] = config_legs(K, capi, D, B, ts, dev, torch, np, direct_bits)

    timed_row_bytes = ts.direct_row_bytes() if direct_bits else None
    if args.op == "commit" and not args.no_extra_legs:
        # the table of the timed region is not rebuilt afterwards (nothing below needs it)
        ts.enable_direct_table(0)
        ts.set_mode(-1)      # (the config legs' last ones left the settings object in c-kzg mode; no table is left to move)
        if headline_first:
            # (a) what a consumer of the nine reference symbols gets: the engine a plain load selects. Built HERE, behind the free of the timed
            # region's table: its build time includes the driver's scrub of what it needs of those 275 GB and is reported as such
            t_d0 = time.perf_counter()
            if default_bits:
                ts.enable_direct_table(default_bits)
            t_default_build = time.perf_counter() - t_d0
            extra["default_engine"] = engine_leg(default_bits, DEFAULT_LABEL)
            extra["default_engine"]["table_build_s_after_freeing_the_headline_table"] = t_default_build
            extra["default_engine"]["table_build_breakdown_ms"] = ts.timing_report().get("last_table_build") if default_bits else None
            if rank == 0:
                extra["api_latency"] = api_latency_leg(K, B, ts, default_bits)
            ts.enable_direct_table(0)
        # (b) the low-memory fallback, last
        extra["bucket_engine"] = engine_leg(0, "Pippenger buckets over the 9 MB fixed-base table (LWKZG_DIRECT_BITS=0, or no memory for a table)")

    if rank == 0:
        total_blobs = n * world * args.steps
        value = total_blobs / elapsed
        if args.op == "tiled_msm":
            value = tiles_total * 4096 * args.steps / elapsed      # terms per second, whole job
        override = None
        if args.op == "tiled_msm":    # a tile is one 4096-term MSM; rank 0's share of the tiles per launch
            k0 = prof.get(("k_direct_accumulate_asm" if "k_direct_accumulate_asm" in prof else "k_direct_accumulate") if direct_bits else
                          ("k_bucket_accumulate_asm" if "k_bucket_accumulate_asm" in prof else "k_bucket_accumulate"), {"launches": 0})
            override = D.shard_range(tiles_total, world, 0)[1] / max(1, round(k0["launches"] / max(1, args.steps)))
        roofline, nwin = engine_picture(K, capi, direct_bits, prof, elapsed, args.steps, n, override)
        dom = roofline["kernel"]
        if args.op == "verify_batch":   # no MSM here: the longest kernel of the per-blob pass is the one priced
            dom = max(prof, key=lambda kk: prof[kk]["total_ms"]) if prof else dom
            roofline["kernel"] = dom
        kernels = {name: {"launches": v["launches"], "avg_ms": v["total_ms"] / max(1, v["launches"])} for name, v in prof.items()}
        traffic, traffic_source = None, None
        try:   # PMC passes are separate rocprofv3 runs (tools/pmc_summary.py); valid for the same batch size and engine only
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pmc.get("batch_blobs_per_launch") == n / roofline["launches_per_step"] and args.op == "commit" and pmc.get("direct_bits", 0) == direct_bits:
                traffic = pmc["kernels"][dom]["traffic_bytes"]
                traffic_source = "profiles/pmc_traffic.json (round %s): committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, NOT measured in this run" % pmc.get("round")
        except Exception:
            traffic = None
        roofline["traffic"] = traffic
        roofline["traffic_source"] = traffic_source
        roofline["traffic_note"] = ("bytes per launch at the L2's memory side from rocprofv3 2 x FETCH_SIZE (gfx950: 128-byte requests are tallied at 64 bytes) "
                                    "+ WRITE_SIZE, separate passes; " +
                                    ("one 128-byte line per gathered table row (1.75 lines per row with packed 112-byte rows), see DESIGN.md section 4" if direct_bits else
                                     "mostly Infinity-Cache-served re-reads of the 9.2 MB fixed-base table, see DESIGN.md section 4"))
        res = {
            "metric": {"commit": "blob_to_kzg_commitment ops/sec (4096-elem blobs)",
                       "blob_proof": "compute_blob_kzg_proof ops/sec (4096-elem blobs)",
                       "commit_prove": "blob_to_kzg_commitment + compute_blob_kzg_proof pairs/sec (4096-elem blobs, one pass)",
                       "verify_batch": "verify_blob_kzg_proof_batch blobs/sec (4096-elem blobs, host-pointer ABI, PCIe included)",
                       "tiled_msm": "G1 MSM terms/sec (one 2^20-term MSM over the tiled setup)"}[args.op],
            "value": value,
            "unit": "terms/s" if args.op == "tiled_msm" else "ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "step_ms": headline_spread,
            "higher_is_better": True,
            "scaling": "strong" if args.op == "tiled_msm" else "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "dtype_note": "28-bit limbs in 32-bit registers, 64-bit column sums (v_mad_u64_u32): 381-bit Fp / 255-bit Fr Montgomery arithmetic, all integer",
            "data": "synthetic (SplitMix64 blobs, seed 0x4B5A47 + blob index; tau=1337 testing trusted setup)",
            "config": {"workload": {"commit": "BASELINE configs[1]: single-GPU 4096-scalar G1 MSM (blob -> commitment), batch=%d synthetic blobs "
                                             "per GPU per step, device-resident, bit-exact vs CPU",
                                   "blob_proof": "BASELINE configs[2]: compute_blob_kzg_proof (Fiat-Shamir hash, quotient, MSM), batch=%d synthetic "
                                                 "blobs per GPU per step, device-resident, one call per step",
                                   "commit_prove": "a blob producer's pair of calls (BASELINE configs[1] followed by configs[2] on its output) as ONE pass: "
                                                   "commitment and blob proof of batch=%d synthetic blobs per GPU per step, device-resident",
                                   "verify_batch": "BASELINE configs[3]: verify_blob_kzg_proof_batch, %d synthetic blobs per GPU per step; all ranks' blobs "
                                                   "form ONE batch (one transcript, one r, one pairing check; records and partial sums all-gathered), "
                                                   "blobs in host memory (H2D inside the timed region)",
                                   "tiled_msm": "BASELINE configs[4]: one 2^20-term G1 MSM over the setup tiled 256 times, tiles split over "
                                                "the GPUs, partial sums gathered and added on the host (%d is unused here)"}[args.op] % n,
                       "blobs_per_gpu_per_step": n, "caller_streams": n_cs, "scalars": args.scalars, "direct_bits": direct_bits, "direct_bits_min_over_ranks": direct_bits_min, "mode": "reference (big-endian monomial)" if args.mode == "reference" else "ckzg (little-endian evaluations, inverse NTT)", "op": args.op,
                       "parallelism": ("blob-sharded x%d, setup broadcast once (%s), no data-path collective"
                                       % (world, "RCCL" if dist_info.get("backend") == "nccl" else dist_info["backend"]))
                                      if dist_info["initialised"] else "one GPU, no process group"},
            "roofline": roofline,
            "kernels": kernels,
            "hip_first_use_init_s": t_hip_init,
            "setup_load_s": t_load,
            "setup_load_breakdown_ms": load_breakdown.get("load"),
            "default_table_build_breakdown_ms": load_breakdown.get("last_table_build"),
            "msm_path": ("direct table, %d-bit windows, %d windows, %d-byte rows, %.0f GB resident" % (
                direct_bits, nwin, timed_row_bytes, capi.direct_table_bytes(direct_bits, timed_row_bytes) / 1e9)) if direct_bits else "bucket (Pippenger, 13-bit signed windows, 9 MB table)",
            "engine_note": "the timed region runs on the widest direct table that fits (--direct-bits auto, an explicit lwkzg_enable_direct_table "
                           "call); `default_engine` is the same workload on the engine a plain load selects, `bucket_engine` on the low-memory fallback",
            "leg_order": ("load (no direct table: bench.py sets LWKZG_DIRECT_BITS=0) -> the timed region's table, this process's first large allocation -> "
                          "timed region -> host_abi -> config legs -> default engine (its table built behind the free of the headline's) + api_latency -> bucket engine"
                          if headline_first else "load -> default engine -> the timed region's table -> timed region -> config legs -> bucket engine"),
            "direct_table_build_s": t_table if direct_bits else None,
            "direct_table_build_s_max_over_ranks": t_table_max if direct_bits else None,
            "direct_table_build_breakdown_ms": table_breakdown,
            "direct_table_build_breakdown_note": "free_old = hipFree of the table in place; table_malloc = the hipMallocs of the new table, one per window, summed "
                                                 "(on idle memory they return in a millisecond; what they wait for is the driver's background scrub, ~25 ms per GB, of memory released "
                                                 "shortly before -- by the previous process of this box, or by this process freeing the default table just above: "
                                                 "tools/alloc_pieces.hip, profiles/r03_alloc_pieces.txt); the GPU builds window j while the host "
                                                 "allocates window j + 1, and kernels = what was left of k_direct_build after the last allocation returned" if direct_bits else None,
            "direct_table_build_note": "every rank builds its own table from the broadcast setup points, once, outside the timed region" if direct_bits else None,
        }
        dist_info["per_rank"] = per_rank
        dist_info["devices"] = [r["device"] for r in per_rank]
        res["dist"] = dist_info
        res["box"] = {"gpu": per_rank[0]["gpu"], "gpu_uuid_sha256_12": per_rank[0]["gpu_uuid_sha256_12"],
                      "clock_mhz_under_mad_probe": per_rank[0]["clock_mhz_under_mad_probe"]}
        res["hbm_budget"] = hbm_budget
        res.update(extra)
        if world > 1:
            res["scaling_note"] = "per-GPU work is fixed (weak scaling); multi-GPU throughput is unmeasured on hardware by the builder (one-GPU boxes only)"
        if world == 1 and not args.no_cpu_baseline and args.mode == "reference" and args.scalars == "31byte":
            outs = bytes(d_out.cpu().numpy().tobytes()) if args.op == "commit" else b""
            res["cpu_baseline"] = cpu_baseline([outs[48 * i:48 * i + 48] for i in range(len(outs) // 48)], ts.g2_values_bytes())
        detail_path = os.environ.get("LWKZG_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")
        try:
            with open(detail_path, "w") as f:
                json.dump(res, f, indent=1)
            print("bench.py: detail (notes, breakdowns, per-kernel tables) written to %s" % detail_path, file=sys.stderr)
        except OSError as e:
            print("bench.py: could not write %s: %s" % (detail_path, e), file=sys.stderr)
            detail_path = None
        sys.stdout.flush()
        os.write(result_fd, (compact_line(res, os.path.relpath(detail_path, ROOT) if detail_path else None) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    ts.free() if ts is not None else None


if __name__ == "__main__":
    main()
