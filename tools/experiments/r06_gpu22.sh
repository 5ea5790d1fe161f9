#!/bin/bash
# r06 call 22: how many parked host threads a hashing job wakes (LWKZG_HOST_HASH_GRAIN = blobs per woken thread; 1 = all, as before)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export LWKZG_EXPERIMENTAL=1
for g in 4 1 8 2 4 1; do
  export LWKZG_HOST_HASH_GRAIN=$g
  echo "== grain $g"
  python tools/verify_device_loop.py --n 4096 --calls 8 --host --tag "host form, grain $g" 2>/dev/null | tail -1 | cut -c1-140 | tee -a $O/g22_verify.jsonl
  python bench.py --op blob_proof --batch 256 --steps 40 --warmup 10 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | cut -c1-160 | tee -a $O/g22_proof256.jsonl
done
unset LWKZG_HOST_HASH_GRAIN
LWKZG_TIMING=1 python tools/verify_device_loop.py --n 4096 --calls 4 --host --tag "host" 2>&1 | grep "verify batch" | tail -3
python tools/host_api_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/g22_host_api_timing.txt
LWKZG_HOST_HASH_GRAIN=1 python tools/host_api_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/g22_host_api_timing_grain1.txt
timeout 900 python -m pytest tests/test_gpu_proof_parity.py tests/test_gpu_verify_device.py tests/test_gpu_plan.py -x -q -m gpu 2>&1 | tail -2
