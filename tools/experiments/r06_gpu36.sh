#!/bin/bash
# r06 call 36: the priority copy stream as shipped: the arms test, the long-batch tests, 2048 ... 16384 blobs, the host ABI, the verification soak
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_verify_msm.py tests/test_gpu_parity.py tests/test_gpu_verify_device.py tests/test_gpu_multi.py tests/test_gpu_dist.py -x -q -m gpu -k "arms or long_batch or evaluate_straight or verify or shard" 2>&1 | tail -3
for n in 2048 4096 8192 16384; do
  python tools/verify_device_loop.py --n $n --calls 6 --host --tag "host form, $n blobs" 2>/dev/null | tail -1 | cut -c1-150 | tee -a $O/g36_verify.jsonl
done
python tools/host_api_timing.py 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python tools/soak_verify.py 120 2>/dev/null | tail -1
timeout 300 python tools/leak_check.py 2>&1 | tail -3
