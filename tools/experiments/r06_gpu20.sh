#!/bin/bash
# r06 call 20: host field tower on the box's cores (three arms), parked side workers against a thread per job, host decompression on 64-bit limbs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
bash tools/host_field_bench.sh > $O/g20_host_field_bench.txt 2>&1
LWKZG_EXPERIMENTAL=1 LWKZG_HOST_FP_PORTABLE=2 /tmp/host_field_bench > $O/g20_host_field_bench_fp2_in_c.txt 2>&1
LWKZG_EXPERIMENTAL=1 LWKZG_HOST_FP_PORTABLE=1 /tmp/host_field_bench > $O/g20_host_field_bench_all_c.txt 2>&1
paste $O/g20_host_field_bench.txt $O/g20_host_field_bench_fp2_in_c.txt $O/g20_host_field_bench_all_c.txt | head -16
timeout 600 python -m pytest tests/test_gpu_verify_msm.py tests/test_gpu_verify_device.py tests/test_gpu_fuzz_seeds.py -x -q -m gpu 2>&1 | tail -2
for arm in workers threads workers threads; do
  if [ $arm = threads ]; then export LWKZG_EXPERIMENTAL=1 LWKZG_SIDE_WORKERS=0; else unset LWKZG_SIDE_WORKERS; fi
  echo "== $arm"
  LWKZG_DIRECT=0 python tools/single_blob_timing.py 2>&1 | tail -4 | cut -c1-70 | tee -a $O/g20_single_blob_timing_$arm.txt
  python tools/verify_device_loop.py --n 4096 --calls 8 --host --tag "host $arm" 2>/dev/null | tail -1 | cut -c1-120 | tee -a $O/g20_verify.jsonl
done
unset LWKZG_SIDE_WORKERS
LWKZG_TIMING=1 LWKZG_DIRECT=0 python tools/single_blob_timing.py 2>&1 | grep -E "verification:|Miller" | tail -6 | tee $O/g20_single_blob_phases.txt
python tools/host_api_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/g20_host_api_timing.txt
