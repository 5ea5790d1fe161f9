# Collects the round's measurement artifacts on the GPU box (one gpurun call); tools/publish_profiles.sh copies the
# summaries into profiles/. Counters are collected in their own rocprofv3 runs (--pmc only), never with trace flags.
export LWKZG_EXPERIMENTAL=1   # the A/B arms below are experiment knobs (csrc/knobs.h, r06)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
LWKZG_BENCH_DETAIL=$O/bench_detail.json python bench.py > $O/bench_line.json 2> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_force_dist.json python bench.py --force-dist --steps 10 --no-cpu-baseline --no-extra-legs > $O/bench_line_force_dist.json 2>> $O/bench_err.txt
LWKZG_DIRECT_ASM=0 LWKZG_BENCH_DETAIL=$O/bench_detail_compiler_scheduled_arm.json python bench.py --no-cpu-baseline --no-config-legs > $O/bench_line_compiler_scheduled_arm.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_full_range_scalars.json python bench.py --scalars full --no-cpu-baseline --no-extra-legs > $O/bench_line_full_range_scalars.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_ckzg_mode.json python bench.py --mode ckzg --no-cpu-baseline --no-extra-legs > $O/bench_line_ckzg_mode.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b256.json python bench.py --op blob_proof --batch 256 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_line_blob_proof_b256.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b1024.json python bench.py --op blob_proof --batch 1024 --no-cpu-baseline > $O/bench_line_blob_proof_b1024.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b4096.json python bench.py --op blob_proof --batch 4096 --steps 5 --no-cpu-baseline > $O/bench_line_blob_proof_b4096.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b256_two_streams.json python bench.py --op blob_proof --batch 256 --caller-streams 2 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_line_blob_proof_b256_two_streams.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b1024_two_streams.json python bench.py --op blob_proof --batch 1024 --caller-streams 2 --no-cpu-baseline > $O/bench_line_blob_proof_b1024_two_streams.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_commit_prove_b256.json python bench.py --op commit_prove --batch 256 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_line_commit_prove_b256.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_commit_prove_b1024.json python bench.py --op commit_prove --batch 1024 --no-cpu-baseline > $O/bench_line_commit_prove_b1024.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_verify_batch_b4096.json python bench.py --op verify_batch --batch 4096 --steps 5 --no-cpu-baseline > $O/bench_line_verify_batch_b4096.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_tiled_msm.json python bench.py --op tiled_msm --no-cpu-baseline > $O/bench_line_tiled_msm.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_bucket_compiler_arm.json LWKZG_BUCKET_ASM=0 python bench.py --direct-bits 0 --no-cpu-baseline --no-config-legs > $O/bench_line_bucket_compiler_arm.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_bucket.json python bench.py --direct-bits 0 --no-cpu-baseline --no-config-legs > $O/bench_line_bucket.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_default_engine.json python bench.py --direct-bits default --no-cpu-baseline --no-config-legs > $O/bench_line_default_engine.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_gpus2_gloo.json python bench.py --gpus 2 --backend gloo --steps 5 --no-cpu-baseline --no-extra-legs --direct-bits 13 > $O/bench_line_gpus2_gloo_one_device.json 2>> $O/bench_err.txt
# every profiled run below writes its detail file HERE (they used to overwrite ROOT/bench_detail.json, which then no longer
# described the headline run)
export LWKZG_BENCH_DETAIL=$O/bench_detail_profiled_runs.json
# per-kernel time of the headline command (20 timed steps: the kernel's average over >= 20 launches), of the default engine and of the bucket engine
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/kt_line.json 2> $O/kt_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_all -o kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/kt_all_line.json 2> $O/kt_all_err.txt
rm -f $O/kt_all/*kernel_trace.csv
LWKZG_DIRECT_ASM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cpp -o kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/kt_cpp_line.json 2> $O/kt_cpp_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_default -o kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --direct-bits default > $O/kt_default_line.json 2> $O/kt_default_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bucket -o kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --direct-bits 0 > $O/kt_bucket_line.json 2> $O/kt_bucket_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_proof -o kt -- python3 bench.py --op blob_proof --batch 1024 --steps 5 --warmup 2 --no-cpu-baseline > $O/kt_proof_line.json 2> $O/kt_proof_err.txt
rocprofv3 --kernel-trace --output-format csv -d $O/kt_two_streams -o kt -- python3 bench.py --op blob_proof --batch 256 --caller-streams 2 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/kt_two_streams_line.json 2> $O/kt_two_streams_err.txt
LWKZG_HEAVY_SERIAL=0 LWKZG_HASH_PRIO=0 LWKZG_DIRECT_FILL=512 rocprofv3 --kernel-trace --output-format csv -d $O/kt_two_streams_untuned -o kt -- python3 bench.py --op blob_proof --batch 256 --caller-streams 2 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/kt_two_streams_untuned_line.json 2> $O/kt_two_streams_untuned_err.txt
# HBM-side traffic: FETCH_SIZE and WRITE_SIZE in separate passes, headline engine and bucket engine
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- $P > $O/fetch_line.json 2> $O/fetch_err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- $P > $O/write_line.json 2> $O/write_err.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_bucket -o fetch -- $P --direct-bits 0 > $O/fetch_bucket_line.json 2> $O/fetch_bucket_err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_bucket -o write -- $P --direct-bits 0 > $O/write_bucket_line.json 2> $O/write_bucket_err.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_default -o fetch -- $P --direct-bits default > $O/fetch_default_line.json 2> $O/fetch_default_err.txt
# the same two passes over the run with every leg (k_ntt4096, k_challenge_pairs, k_eval_quotient, the verification kernels)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_all -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch_all_line.json 2> $O/fetch_all_err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_all -o write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write_all_line.json 2> $O/write_all_err.txt
# issue-side counters of the headline kernel (SQ: 8 slots per pass; GRBM apart)
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_sq1 -o sq -- $P > $O/pmc_sq1_line.json 2> $O/pmc_sq1_err.txt
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq2 -o sq -- $P > $O/pmc_sq2_line.json 2> $O/pmc_sq2_err.txt
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $O/pmc_sq3 -o sq -- $P > $O/pmc_sq3_line.json 2> $O/pmc_sq3_err.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/pmc_grbm -o grbm -- $P > $O/pmc_grbm_line.json 2> $O/pmc_grbm_err.txt
# the compiler-scheduled arm of the direct kernel (LWKZG_DIRECT_ASM=0): the A/B of the hand-scheduled loop, same passes
export LWKZG_DIRECT_ASM=0
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_cpp_sq1 -o sq -- $P > $O/pmc_cpp_sq1_line.json 2> $O/pmc_cpp_sq1_err.txt
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_cpp_sq2 -o sq -- $P > $O/pmc_cpp_sq2_line.json 2> $O/pmc_cpp_sq2_err.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/pmc_cpp_grbm -o grbm -- $P > $O/pmc_cpp_grbm_line.json 2> $O/pmc_cpp_grbm_err.txt
unset LWKZG_DIRECT_ASM
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq_bucket -o sq -- $P --direct-bits 0 > $O/pmc_sq_bucket_line.json 2> $O/pmc_sq_bucket_err.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grbm_bucket -o grbm -- $P --direct-bits 0 > $O/pmc_grbm_bucket_line.json 2> $O/pmc_grbm_bucket_err.txt
# the cooperative kernel (one blob per call through the reference's symbols): per-kernel time over 60 calls of each, issue-side counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_single -o kt -- python3 tools/single_blob_loop.py 60 > $O/kt_single_out.txt 2> $O/kt_single_err.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_single_sq -o sq -- python3 tools/single_blob_loop.py 20 > $O/pmc_single_sq_out.txt 2> $O/pmc_single_sq_err.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/pmc_single_grbm -o grbm -- python3 tools/single_blob_loop.py 20 > $O/pmc_single_grbm_out.txt 2> $O/pmc_single_grbm_err.txt
python tools/small_batch_timing.py > $O/small_batch_timing.txt 2>&1
LWKZG_COOP=0 python tools/small_batch_timing.py > $O/small_batch_timing_coop_off.txt 2>&1
LWKZG_COOP=0 LWKZG_HOST_FINISH=0 python tools/single_blob_timing.py > $O/single_blob_timing_r04_arm.txt 2>&1
LWKZG_MID_PROOF_PIPE=0 LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b256_unpiped.json python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_line_blob_proof_b256_unpiped.json 2>> $O/bench_err.txt
# c-kzg proofs: the quotient in evaluation form against r04's coefficient-form arm; the validation on quads against r04's kernel; a fresh process loading straight into the 16-bit table
LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b1024_ckzg.json python bench.py --op blob_proof --batch 1024 --mode ckzg --no-cpu-baseline > $O/bench_line_blob_proof_b1024_ckzg.json 2>> $O/bench_err.txt
LWKZG_CKZG_EVAL_PROOFS=0 LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b1024_ckzg_coefficient_arm.json python bench.py --op blob_proof --batch 1024 --mode ckzg --no-cpu-baseline > $O/bench_line_blob_proof_b1024_ckzg_coefficient_arm.json 2>> $O/bench_err.txt
LWKZG_VALIDATE_COOP=0 LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b256_validate_r04_arm.json python bench.py --op blob_proof --batch 256 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_line_blob_proof_b256_validate_r04_arm.json 2>> $O/bench_err.txt
LWKZG_DIRECT_BITS=16 python tools/setup_load_timing.py > $O/setup_load_timing_16bit.txt 2>&1
python tools/setup_load_timing.py > $O/setup_load_timing.txt 2>&1
PK="python3 bench.py --op blob_proof --mode ckzg --batch 1024 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_proof_ckzg -o kt -- $PK > $O/kt_proof_ckzg_line.json 2> $O/kt_proof_ckzg_err.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_ckzg_sq1 -o sq -- $PK > $O/pmc_ckzg_sq1_line.json 2> $O/pmc_ckzg_sq1_err.txt
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_ckzg_sq2 -o sq -- $PK > $O/pmc_ckzg_sq2_line.json 2> $O/pmc_ckzg_sq2_err.txt
# r06: BASELINE configs[3] in its device-resident form in a loop of its own (and its kernel timeline), the host-pointer form beside it, 16384 blobs,
# the arms of the verification (r05's rows, r05's placement), the host field tower on this box's cores, the scaling projection
python tools/verify_device_loop.py --n 4096 --calls 10 --tag "configs[3] device form" 2>/dev/null | tail -1 > $O/verify_loops.jsonl
python tools/verify_device_loop.py --n 4096 --calls 10 --host --tag "configs[3] host form" 2>/dev/null | tail -1 >> $O/verify_loops.jsonl
python tools/verify_device_loop.py --n 16384 --calls 4 --tag "16384 blobs, device form" 2>/dev/null | tail -1 >> $O/verify_loops.jsonl
LWKZG_VERIFY_MSM=0 LWKZG_VERIFY_FUSED=0 LWKZG_VERIFY_PAD_KB=0,0,0 python tools/verify_device_loop.py --n 4096 --calls 6 --tag "r05 arm: per-point multiples + Straus, two side streams, no footprints" 2>/dev/null | tail -1 >> $O/verify_loops.jsonl
LWKZG_VERIFY_PAD_KB=0,0,0 python tools/verify_device_loop.py --n 4096 --calls 6 --tag "bucket MSM, no footprints" 2>/dev/null | tail -1 >> $O/verify_loops.jsonl
LWKZG_HOST_FP_PORTABLE=1 python tools/verify_device_loop.py --n 4096 --calls 6 --tag "host field products in C" 2>/dev/null | tail -1 >> $O/verify_loops.jsonl
rocprofv3 --kernel-trace --output-format csv -d $O/kt_verify_dev -o kt -- python3 tools/verify_device_loop.py --n 4096 --calls 4 --no-profile > $O/kt_verify_dev_out.txt 2> $O/kt_verify_dev_err.txt
python tools/timeline.py $(ls $O/kt_verify_dev/*/kt_kernel_trace.csv $O/kt_verify_dev/kt_kernel_trace.csv 2>/dev/null | head -1) 40 > $O/verify_b4096_device_timeline.txt
rm -f $O/kt_verify_dev/*/kt_kernel_trace.csv $O/kt_verify_dev/kt_kernel_trace.csv
bash tools/host_field_bench.sh > $O/host_field_bench.txt 2>&1
LWKZG_HOST_FP_PORTABLE=2 /tmp/host_field_bench > $O/host_field_bench_fp2_in_c.txt 2>&1
LWKZG_HOST_FP_PORTABLE=1 /tmp/host_field_bench > $O/host_field_bench_all_c.txt 2>&1
lscpu | grep -E "Model name|^CPU\(s\)|MHz" > $O/lscpu.txt
LWKZG_TIMING=1 LWKZG_DIRECT=0 python tools/single_blob_timing.py 2>&1 | grep -E "verification:|Miller" | tail -8 > $O/single_blob_phases.txt
LWKZG_SIDE_WORKERS=0 LWKZG_DIRECT=0 python tools/single_blob_timing.py > $O/single_blob_timing_thread_per_job_arm.txt 2>&1
LWKZG_ZERO_COPY=0 LWKZG_DIRECT=0 python tools/single_blob_timing.py > $O/single_blob_timing_r05_arm.txt 2>&1
python tools/scaling_projection.py --out $O/scaling_projection.json 2> $O/scaling_projection_err.txt | grep '^|' > $O/scaling_projection_table.md
python tools/experiments/r05_host_cold.py > $O/host_cold.txt 2>&1
tools/ubench_latency_bin > $O/ubench_latency.txt 2>&1
python tools/host_api_timing.py > $O/host_api_timing.txt 2>&1
python tools/single_blob_timing.py > $O/single_blob_timing.txt 2>&1
python tools/config_sweep.py --direct-bits 16 > $O/config_sweep_direct16.json 2> $O/sweep_err.txt
python tools/config_sweep.py --direct-bits default > $O/config_sweep_default.json 2>> $O/sweep_err.txt
python tools/config_sweep.py --direct-bits 0 --max-verify 1024 > $O/config_sweep_bucket.json 2>> $O/sweep_err.txt
find $O -name "*.csv" | wc -l
du -sh $O
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $O/gpu_test_log.txt 2>&1
echo "pytest rc=$?" >> $O/gpu_test_log.txt
timeout 1200 python tools/soak.py --batches 300 --direct-bits 16 > $O/soak.json 2> $O/soak_err.txt
echo "{\"soak_rc\": $?}" >> $O/soak.json
timeout 600 python tools/soak_ckzg.py --batches 60 > $O/soak_ckzg.json 2> $O/soak_ckzg_err.txt
timeout 900 python tools/soak_small.py --rounds 600 2> $O/soak_small_err.txt | tail -1 > $O/soak_small.json
LWKZG_DIRECT=16 timeout 400 python tools/soak_verify.py 180 2> $O/soak_verify_err.txt | tail -1 > $O/soak_verify_direct16.json
timeout 400 python tools/soak_verify.py 180 2>> $O/soak_verify_err.txt | tail -1 > $O/soak_verify_default.json
du -sh $O
