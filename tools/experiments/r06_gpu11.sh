#!/bin/bash
# r06 call 11: long host-pointer batches without per-call hipMalloc / hipFree, staged above one chunk; the r05-slices arm beside it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
python tools/host_api_timing.py 2>$O/g11_err.txt | tee $O/g11_host_api_timing.txt
LWKZG_EXPERIMENTAL=1 LWKZG_HOST_STAGE=0 python tools/host_api_timing.py 2>>$O/g11_err.txt | tee $O/g11_host_api_timing_r05_slices_arm.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_proof_parity.py tests/test_gpu_host_api_extras.py -x -q -m gpu -k "2100 or long or host or slice or 4096 or 2300 or batch" 2>&1 | tail -3
python tools/leak_check.py 2>&1 | tail -3
