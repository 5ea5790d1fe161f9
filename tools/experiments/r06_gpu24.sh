#!/bin/bash
# r06 call 24: the library's bound on its host threads (32 in r01-r05; 64 shipped): host-pointer verification, host-assisted proof calls, the host ABI
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
for t in 64 32 128 64 32 128; do
  export LWKZG_HOST_THREADS=$t
  echo "== host threads $t"
  python tools/verify_device_loop.py --n 4096 --calls 8 --host --tag "host form, $t host threads" 2>/dev/null | tail -1 | cut -c1-150 | tee -a $O/g24_verify.jsonl
  python bench.py --op blob_proof --batch 256 --steps 40 --warmup 10 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | cut -c1-160 | tee -a $O/g24_proof256.jsonl
done
for t in 64 32 128; do
  export LWKZG_HOST_THREADS=$t
  python tools/host_api_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/g24_host_api_timing_$t.txt
  LWKZG_DIRECT=0 python tools/single_blob_timing.py 2>&1 | tail -4 | cut -c1-75
done
unset LWKZG_HOST_THREADS
LWKZG_TIMING=1 python tools/verify_device_loop.py --n 4096 --calls 4 --host --tag "host" 2>&1 | grep "verify batch" | tail -3
timeout 900 python -m pytest tests/test_gpu_proof_parity.py tests/test_gpu_verify_device.py tests/test_gpu_verify_msm.py -x -q -m gpu 2>&1 | tail -2
