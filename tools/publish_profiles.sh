#!/bin/bash
# Copies what tools/collect_profiles.sh left under gpurun_out/final into profiles/ (tracked), keeping only our kernels' rows.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/final; P=profiles; TAG=${1:-r06}
for f in bench_line bench_line_force_dist bench_line_compiler_scheduled_arm bench_line_full_range_scalars bench_line_ckzg_mode bench_line_blob_proof_b256 bench_line_blob_proof_b1024 bench_line_blob_proof_b4096 \
         bench_line_blob_proof_b256_two_streams bench_line_blob_proof_b1024_two_streams bench_line_commit_prove_b256 bench_line_commit_prove_b1024 bench_line_verify_batch_b4096 bench_line_tiled_msm \
         bench_line_bucket bench_line_bucket_compiler_arm bench_line_default_engine bench_line_gpus2_gloo_one_device; do
  [ -s $O/$f.json ] && tail -1 $O/$f.json > $P/${TAG}_$f.json
done
# the detail files (every note, breakdown and per-kernel table of the runs above)
for f in bench_detail bench_detail_compiler_scheduled_arm bench_detail_bucket bench_detail_bucket_compiler_arm bench_detail_default_engine bench_detail_ckzg_mode; do
  [ -s $O/$f.json ] && cp $O/$f.json $P/${TAG}_$f.json
done
[ -s $O/single_blob_timing.txt ] && grep -v "amdgpu.ids" $O/single_blob_timing.txt > $P/${TAG}_single_blob_timing.txt
for f in small_batch_timing small_batch_timing_coop_off single_blob_timing_r04_arm host_cold ubench_latency; do
  [ -s $O/$f.txt ] && grep -v "amdgpu.ids" $O/$f.txt > $P/${TAG}_$f.txt
done
for f in bench_line_blob_proof_b256_unpiped bench_line_blob_proof_b1024_ckzg bench_line_blob_proof_b1024_ckzg_coefficient_arm bench_line_blob_proof_b256_validate_r04_arm; do
  [ -s $O/$f.json ] && tail -1 $O/$f.json > $P/${TAG}_$f.json
done
for f in setup_load_timing setup_load_timing_16bit; do [ -s $O/$f.txt ] && grep -v "amdgpu.ids" $O/$f.txt > $P/${TAG}_$f.txt; done
[ -s $O/kt_line.json ] && tail -1 $O/kt_line.json > $P/${TAG}_bench_kernel_stats_run_line.json   # the line the profiled run itself printed
for f in config_sweep_direct16 config_sweep_default config_sweep_bucket; do [ -s $O/$f.json ] && cp $O/$f.json $P/${TAG}_$f.json; done
grep -v "amdgpu.ids" $O/host_api_timing.txt > $P/${TAG}_host_api_timing.txt || true
# r06: the verification in loops of its own, its kernel timeline, the host field tower, one-blob phases and arms, the scaling projection
for f in verify_loops; do [ -s $O/$f.jsonl ] && cp $O/$f.jsonl $P/${TAG}_$f.jsonl; done
for f in verify_b4096_device_timeline host_field_bench host_field_bench_fp2_in_c host_field_bench_all_c lscpu single_blob_phases single_blob_timing_thread_per_job_arm single_blob_timing_r05_arm; do
  [ -s $O/$f.txt ] && grep -v "amdgpu.ids" $O/$f.txt > $P/${TAG}_$f.txt
done
[ -s $O/scaling_projection.json ] && cp $O/scaling_projection.json $P/${TAG}_scaling_projection.json
[ -s $O/scaling_projection_table.md ] && cp $O/scaling_projection_table.md $P/${TAG}_scaling_projection_table.md
python3 - "$TAG" <<'PY'
import csv, glob, os, sys
tag = sys.argv[1]
O = 'gpurun_out/final'
def ours(rows, col):
    return [rows[0]] + [r for r in rows[1:] if 'lwk::' in r[col] or 'rocclr' in r[col]]
for d, name in (('kt_single', 'single_blob_calls_kernel_stats'), ('kt', 'bench_kernel_stats'), ('kt_all', 'bench_all_legs_kernel_stats'), ('kt_cpp', 'compiler_scheduled_arm_kernel_stats'), ('kt_default', 'default_engine_kernel_stats'), ('kt_bucket', 'bucket_engine_kernel_stats'), ('kt_proof', 'blob_proof_b1024_kernel_stats'), ('kt_proof_ckzg', 'blob_proof_b1024_ckzg_kernel_stats')):
    f = '%s/%s/kt_kernel_stats.csv' % (O, d)
    if os.path.exists(f):
        rows = list(csv.reader(open(f)))
        csv.writer(open('profiles/%s_%s.csv' % (tag, name), 'w'), quoting=csv.QUOTE_ALL).writerows(ours(rows, 0))
# PMC passes: one file per engine, all counters of our kernels
def pmc(dirs, out):
    hdr, keep = None, []
    for d in dirs:
        for f in glob.glob('%s/%s/*_counter_collection.csv' % (O, d)):
            rows = list(csv.reader(open(f)))
            hdr = rows[0]
            ki = hdr.index('Kernel_Name')
            keep += [r for r in rows[1:] if 'lwk::' in r[ki]]
    if hdr:
        csv.writer(open(out, 'w'), quoting=csv.QUOTE_ALL).writerows([hdr] + keep)
pmc(['pmc_sq1', 'pmc_sq2', 'pmc_sq3', 'pmc_grbm'], 'profiles/%s_pmc_sq_counters.csv' % tag)
pmc(['pmc_cpp_sq1', 'pmc_cpp_sq2', 'pmc_cpp_grbm'], 'profiles/%s_pmc_compiler_scheduled_arm_sq_counters.csv' % tag)
pmc(['fetch_all'], 'profiles/%s_pmc_all_legs_fetch_size.csv' % tag)
pmc(['write_all'], 'profiles/%s_pmc_all_legs_write_size.csv' % tag)
pmc(['pmc_sq_bucket', 'pmc_grbm_bucket'], 'profiles/%s_pmc_bucket_sq_counters.csv' % tag)
pmc(['fetch'], 'profiles/%s_pmc_fetch_size.csv' % tag)
pmc(['write'], 'profiles/%s_pmc_write_size.csv' % tag)
pmc(['fetch_bucket'], 'profiles/%s_pmc_bucket_fetch_size.csv' % tag)
pmc(['write_bucket'], 'profiles/%s_pmc_bucket_write_size.csv' % tag)
pmc(['fetch_default'], 'profiles/%s_pmc_default_engine_fetch_size.csv' % tag)
pmc(['pmc_single_sq', 'pmc_single_grbm'], 'profiles/%s_pmc_single_blob_calls_sq_counters.csv' % tag)
pmc(['pmc_ckzg_sq1', 'pmc_ckzg_sq2'], 'profiles/%s_pmc_blob_proof_b1024_ckzg_sq_counters.csv' % tag)
PY
python3 tools/pmc_summary.py $O/fetch/fetch_counter_collection.csv $O/write/write_counter_collection.csv $TAG 1024 16 | grep -E "direct_acc|wrote"
python3 tools/pmc_issue_summary.py k_direct_accumulate $P/${TAG}_issue_summary_compiler_scheduled_arm.json $O/pmc_cpp_sq1/sq_counter_collection.csv $O/pmc_cpp_sq2/sq_counter_collection.csv $O/pmc_cpp_grbm/grbm_counter_collection.csv > /dev/null
python3 tools/pmc_traffic_all.py $O/fetch_all/fetch_counter_collection.csv $O/write_all/write_counter_collection.csv $TAG > /dev/null
python3 tools/pmc_issue_summary.py k_direct_accumulate_asm $P/${TAG}_issue_summary.json $O/pmc_sq1/sq_counter_collection.csv $O/pmc_sq2/sq_counter_collection.csv $O/pmc_sq3/sq_counter_collection.csv $O/pmc_grbm/grbm_counter_collection.csv > /dev/null
python3 tools/pmc_issue_summary.py k_bucket_accumulate_asm $P/${TAG}_issue_summary_bucket.json $O/pmc_sq_bucket/sq_counter_collection.csv $O/pmc_grbm_bucket/grbm_counter_collection.csv > /dev/null
ls $P | grep $TAG | wc -l
[ -s $O/kt_two_streams/kt_kernel_trace.csv ] && python3 tools/timeline.py $O/kt_two_streams/kt_kernel_trace.csv 66 > $P/${TAG}_proof_two_streams_timeline.txt
[ -s $O/kt_two_streams_untuned/kt_kernel_trace.csv ] && python3 tools/timeline.py $O/kt_two_streams_untuned/kt_kernel_trace.csv 66 > $P/${TAG}_proof_two_streams_untuned_timeline.txt
true
[ -s $O/gpu_test_log.txt ] && grep -v "amdgpu.ids" $O/gpu_test_log.txt > $P/${TAG}_gpu_test_log.txt
[ -s $O/soak.json ] && cat $O/soak.json $O/soak_ckzg.json $O/soak_small.json $O/soak_verify_direct16.json $O/soak_verify_default.json > $P/${TAG}_soak_final.jsonl
true
