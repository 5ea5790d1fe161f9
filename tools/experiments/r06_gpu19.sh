#!/bin/bash
# r06 call 19: host side of a verification after the Fp2 product, the windowed scalar products and the persistent side workers
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_verify_msm.py tests/test_gpu_verify_device.py tests/test_gpu_fuzz_seeds.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_setups.py -x -q -m gpu -k "verify or lib_test or vectors or threads" 2>&1 | tail -3
LWKZG_DIRECT=0 python tools/single_blob_timing.py 2>&1 | tail -4 | tee $O/g19_single_blob_timing.txt
LWKZG_TIMING=1 LWKZG_DIRECT=0 python tools/single_blob_timing.py 2>&1 | grep -E "verification:|Miller" | tail -8 | tee $O/g19_single_blob_phases.txt
python tools/verify_device_loop.py --n 4096 --calls 8 --tag "device" 2>/dev/null | tail -1 | tee -a $O/g19_verify.jsonl
python tools/verify_device_loop.py --n 4096 --calls 8 --host --tag "host" 2>/dev/null | tail -1 | tee -a $O/g19_verify.jsonl
python tools/host_api_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/g19_host_api_timing.txt
