#!/usr/bin/env python3
"""bench.py -- blob_to_kzg_commitment throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (blob bytes -> 48-byte commitments: parse, fixed-base MSM, compress;
the MSM is the direct-table kernel when the table fits in HBM, else digit sort + bucket accumulate + bucket
reduce) over one batch of synthetic 4096-element blobs that is already
resident in HBM. Each GPU works on its own shard (weak scaling: 1024 blobs per GPU per step); the only
collective is the one broadcast of the prepared trusted setup before the timed region.

Rank 0 prints ONE JSON line. `roofline` prices the dominant kernel (k_direct_accumulate / k_bucket_accumulate) against HBM as
the north star mandates AND gives the integer-multiply picture, because the kernel is integer-ALU bound
(DESIGN.md section 5). `cpu_baseline` times the CPU oracle (a restatement of the reference's algorithm,
NOT the reference binary, which cannot be built here) on this box's host cores.
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
    lib_path, first, count = args
    from oracle import oracle as O
    l = O.lib(lib_path)
    s = O.Settings.from_file(SETUP, check_subgroup=False, _lib=l)
    import blobs as B
    data = [B.synthetic_blob(first + i) for i in range(count)]
    t0 = time.perf_counter()
    outs = []
    for b in data:
        rc, cm = O.blob_to_kzg_commitment(b, s, O.MODE_R)
        assert rc == 0
        outs.append(cm)
    return time.perf_counter() - t0, outs


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


def cpu_baseline(gpu_outputs):
    """Time the oracle on the host: all cores, one blob stream per process. Bounded sample."""
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
    t1, o1 = _cpu_worker((lib_path, 0, 4))
    per_blob = t1 / 4
    per_proc = max(2, min(64, int(12.0 / per_blob)))    # about 12 s of work per core
    per_proc = min(per_proc, max(2, BLOBS_PER_GPU // cores))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(lib_path, i * per_proc, per_proc) for i in range(cores)])
    wall = time.perf_counter() - t0
    total = cores * per_proc
    busy = max(r[0] for r in res)
    verified = True
    flat = [o for r in res for o in r[1]]
    for i, cm in enumerate(flat[:len(gpu_outputs)]):
        if gpu_outputs[i] != cm:
            verified = False
    return {
        "value": total / busy,
        "unit": "blob_to_kzg_commitment ops/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d synthetic blobs (%d per core x %d cores, one process per core), %.1f s busy, %.1f s wall incl. "
                  "process start; single-thread rate %.2f ops/s; CPU restatement of lambdaworks_kzg's algorithm "
                  "(Pippenger w=9, 29 windows, projective adds), SRS rebuild per call NOT included"
                  % (total, per_proc, cores, busy, wall, 1.0 / per_blob),
        "single_thread_ops_per_s": 1.0 / per_blob,
        "gpu_outputs_match_oracle": verified,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BLOBS_PER_GPU, help="blobs per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--op", default="commit", choices=["commit", "blob_proof", "verify_batch", "tiled_msm"],
                    help="commit = the headline (BASELINE configs[1]); blob_proof = configs[2]; verify_batch = configs[3], host-pointer ABI; tiled_msm = configs[4], one 2^20-term MSM split over the GPUs (strong scaling)")
    ap.add_argument("--mode", default="reference", choices=["reference", "ckzg"],
                    help="reference = lambdaworks_kzg semantics (default, the headline); ckzg = c-kzg-4844 semantics (adds the inverse NTT)")
    ap.add_argument("--direct-bits", default="auto",
                    help="direct fixed-base table (lwkzg_enable_direct_table): auto = widest of 16/15/14 that fits in HBM, "
                         "else the bucket path; 0 = bucket path; 14/15/16 = that width or fail")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU plumbing tests)")
    args = ap.parse_args()

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
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    dev_index = local_rank % max(1, torch.cuda.device_count())   # == local_rank on a real N-GPU node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    K.set_device(dev_index)
    K.set_mode(K.MODE_REFERENCE if args.mode == "reference" else K.MODE_CKZG)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    # trusted setup: rank 0 parses + validates + builds the fixed-base table, one RCCL broadcast delivers it
    t_load0 = time.perf_counter()
    ts = K.TrustedSetup.from_file(SETUP) if rank == 0 else None
    if world > 1:
        ts = D.broadcast_trusted_setup(ts, dev, src=0)
    t_load = time.perf_counter() - t_load0

    n = args.batch
    first = rank * n                       # shard: blob k of the job lives on GPU floor(k / n)
    host = np.frombuffer(B.synthetic_batch(first, n, big_endian=(args.mode == "reference")), dtype=np.uint8)
    d_blobs = torch.from_numpy(host.copy()).to(dev)
    d_out = torch.empty(48 * n, dtype=torch.uint8, device=dev)
    d_status = torch.zeros(n, dtype=torch.int32, device=dev)
    d_comm = None
    ts.reserve(n)
    # every rank builds its own direct table from the (broadcast) setup points: a few seconds, once, outside the timed region
    t_tab0 = time.perf_counter()
    direct_bits = 0
    if args.direct_bits == "auto":
        for bits in (16, 15, 14, 13, 12, 11, 10, 0):
            try:
                ts.enable_direct_table(bits)
                direct_bits = bits
                break
            except capi.KzgError as e:
                if e.rc != capi.C_KZG_MALLOC:
                    raise
    elif args.direct_bits == "default":     # whatever a plain load selected (engine.hip: direct_from_env)
        direct_bits = ts.direct_table_bits()
    else:
        ts.enable_direct_table(int(args.direct_bits))
        direct_bits = int(args.direct_bits)
    t_table = time.perf_counter() - t_tab0
    direct_bits_min = direct_bits
    if world > 1:   # every rank should have got the same width; report it if one did not
        tb = torch.tensor([direct_bits], dtype=torch.int32, device=dev)
        dist.all_reduce(tb, op=dist.ReduceOp.MIN)
        direct_bits_min = int(tb.item())
    stream = torch.cuda.current_stream(dev).cuda_stream

    h_blobs = h_comms = h_proofs = None
    d_tiles = None
    tiles_total = 256                      # 2^20 terms over the setup tiled 256 times

    def step():
        if args.op == "tiled_msm":      # each rank sums its share of the tiles, one all_gather of 48-byte partial sums
            D.msm_tiled_sharded(d_tiles, int(d_tiles.numel()) // 32, ts, dev)
        elif args.op == "verify_batch":   # each rank verifies its shard as an independent batch, one all_reduce of the verdict
            assert D.verify_blob_kzg_proof_batch_sharded(h_blobs, h_comms, h_proofs, n, ts)
        elif args.op == "commit":
            K.blob_to_kzg_commitment_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), n, ts, stream, d_status.data_ptr())
        else:
            K.compute_blob_kzg_proof_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), d_comm.data_ptr(), n, ts, stream,
                                                  d_status.data_ptr())

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

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)

    capi.profile_reset()
    capi.profile_enable(True)             # hipEvent pairs around every kernel, on the launch stream
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    capi.profile_enable(False)
    prof = capi.profile_report()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert int(d_status.abs().sum().item()) == 0, "a blob was rejected"

    if rank == 0:
        total_blobs = n * world * args.steps
        value = total_blobs / elapsed
        if args.op == "tiled_msm":
            value = tiles_total * 4096 * args.steps / elapsed      # terms per second, whole job
        dom = "k_direct_accumulate" if direct_bits else "k_bucket_accumulate"
        if args.op == "verify_batch":   # no MSM here: the longest kernel of the per-blob pass is the one priced
            dom = max(prof, key=lambda kk: prof[kk]["total_ms"]) if prof else dom
        k = prof.get(dom, {"launches": 0, "total_ms": 0.0})
        avg_ms = k["total_ms"] / max(1, k["launches"])
        # a step may cut its batch into sub-batches on concurrent streams (engine.hip: commit_batch_device), so
        # the units one launch processes = blobs per step / launches per step
        launches_per_step = max(1, round(k["launches"] / max(1, args.steps)))
        msms_per_launch = n / launches_per_step
        if args.op == "tiled_msm":    # a tile is one 4096-term MSM; rank 0's share of the tiles per launch
            msms_per_launch = D.shard_range(tiles_total, world, 0)[1] / launches_per_step
        achieved = msms_per_launch * ALGO_BYTES_PER_MSM / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        kernels = {name: {"launches": v["launches"], "avg_ms": v["total_ms"] / max(1, v["launches"])} for name, v in prof.items()}
        # integer picture: a mixed add is 6 Montgomery products (392 v_mad_u64_u32 each on 14 limbs of 28 bits), 2 squares
        # (301 each) and one fused product pair a*b - c*d with a single reduction (588); peak = v_mad_u64_u32 issue rate measured by tools/ubench.hip on MI355X
        # (profiles/r01_ubench_instruction_rates.jsonl: 2.93e13 lane-mads/s at 8 waves/SIMD)
        if direct_bits:
            nwin = K.lib().lwkzg_direct_num_windows(direct_bits)
            adds_per_msm = 4096 * nwin * (1 - 2.0 ** -direct_bits)
        else:
            nwin = K.lib().lwkzg_msm_num_windows()
            adds_per_msm = 4096 * nwin * (1 - 2.0 ** -K.lib().lwkzg_msm_window_bits())
        GATHER_PEAK_ROWS = 1.31e10     # tools/gather_bench.hip on MI355X: random 112-byte rows/s out of a 128-200 GiB table
        gather = None
        if direct_bits and avg_ms > 0:
            rows = msms_per_launch * adds_per_msm
            gather = {"rows_per_launch": rows, "bytes_per_launch": rows * 112,
                      "achieved_rows_per_s": rows / (avg_ms * 1e-3), "peak_rows_per_s": GATHER_PEAK_ROWS,
                      "frac": rows / (avg_ms * 1e-3) / GATHER_PEAK_ROWS,
                      "note": "the direct path really reads one random 112-byte table row per mixed addition from HBM "
                              "(table far larger than every cache); ceiling measured by tools/gather_bench.hip"}
        mads_per_launch = msms_per_launch * adds_per_msm * (6 * 392 + 2 * 301 + 588)
        INT_MAD_PEAK = 2.93e13
        traffic = None
        try:   # PMC passes are separate rocprofv3 runs (tools/pmc_summary.py); valid for the same batch size only
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pmc.get("batch_blobs_per_launch") == msms_per_launch and args.op == "commit" and pmc.get("direct_bits", 0) == direct_bits:
                traffic = pmc["kernels"][dom]["traffic_bytes"]
        except Exception:
            traffic = None
        res = {
            "metric": {"commit": "blob_to_kzg_commitment ops/sec (4096-elem blobs)",
                       "blob_proof": "compute_blob_kzg_proof ops/sec (4096-elem blobs)",
                       "verify_batch": "verify_blob_kzg_proof_batch blobs/sec (4096-elem blobs, host-pointer ABI, PCIe included)",
                       "tiled_msm": "G1 MSM terms/sec (one 2^20-term MSM over the tiled setup)"}[args.op],
            "value": value,
            "unit": "terms/s" if args.op == "tiled_msm" else "ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.op == "tiled_msm" else "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (381-bit Fp / 255-bit Fr Montgomery, integer)",
            "data": "synthetic (SplitMix64 blobs, seed 0x4B5A47 + blob index; tau=1337 testing trusted setup)",
            "config": {"workload": {"commit": "BASELINE configs[1]: single-GPU 4096-scalar G1 MSM (blob -> commitment), batch=%d synthetic blobs "
                                             "per GPU per step, device-resident, bit-exact vs CPU",
                                   "blob_proof": "BASELINE configs[2]: compute_blob_kzg_proof (Fiat-Shamir hash, quotient, MSM), batch=%d synthetic "
                                                 "blobs per GPU per step, device-resident",
                                   "verify_batch": "BASELINE configs[3]: verify_blob_kzg_proof_batch, %d synthetic blobs per GPU per step verified as "
                                                   "one batch per GPU, blobs in host memory (H2D inside the timed region)",
                                   "tiled_msm": "BASELINE configs[4]: one 2^20-term G1 MSM over the setup tiled 256 times, tiles split over "
                                                "the GPUs, partial sums gathered and added on the host (%d is unused here)"}[args.op] % n,
                       "blobs_per_gpu_per_step": n, "direct_bits": direct_bits, "direct_bits_min_over_ranks": direct_bits_min, "mode": "reference (big-endian monomial)" if args.mode == "reference" else "ckzg (little-endian evaluations, inverse NTT)", "op": args.op,
                       "parallelism": "blob-sharded x%d, setup broadcast once (RCCL), no data-path collective" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": "bytes per launch at the L2's memory side from rocprofv3 FETCH_SIZE (raw) + WRITE_SIZE, "
                                         "separate passes (profiles/pmc_traffic.json); " +
                                         ("random 112-byte row gathers out of the direct table, see DESIGN.md section 4" if direct_bits else
                                          "mostly Infinity-Cache-served re-reads of the 9.2 MB fixed-base table, see DESIGN.md section 4"),
                         "gather": gather,
                         "algorithmic_bytes_per_launch": msms_per_launch * ALGO_BYTES_PER_MSM,
                         "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step,
                         "concurrency_note": "the %d launches of a step run concurrently on separate streams, each on a share of "
                                             "the CUs: per-launch rates are per share; multiply by %d for the rate while both run"
                                             % (launches_per_step, launches_per_step) if launches_per_step > 1 else "one launch per step",
                         "note": "integer-ALU bound, not HBM bound (about 600 int-ops per algorithmic byte): see int_mad",
                         "int_mad": {"mad_u64_u32_per_launch": mads_per_launch,
                                     "achieved_Gmad_per_s": mads_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
                                     "peak_Gmad_per_s": INT_MAD_PEAK / 1e9,
                                     "frac": (mads_per_launch / (avg_ms * 1e-3) / INT_MAD_PEAK) if avg_ms > 0 else 0.0,
                                     # all multiply-adds of a step over the step's wall time (every kernel, all streams):
                                     "frac_whole_step": mads_per_launch * launches_per_step / (elapsed / args.steps) / INT_MAD_PEAK}},
            "kernels": kernels,
            "setup_load_s": t_load,
            "msm_path": ("direct table, %d-bit windows, %d windows, %.0f GB resident" % (
                direct_bits, nwin, capi.direct_table_bytes(direct_bits) / 1e9)) if direct_bits else "bucket (Pippenger, 13-bit signed windows, 9 MB table)",
            "direct_table_build_s": t_table if direct_bits else None,
        }
        if world == 1 and not args.no_cpu_baseline and args.mode == "reference":
            outs = bytes(d_out.cpu().numpy().tobytes()) if args.op == "commit" else b""
            res["cpu_baseline"] = cpu_baseline([outs[48 * i:48 * i + 48] for i in range(len(outs) // 48)])
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ts.free() if ts is not None else None


if __name__ == "__main__":
    main()
