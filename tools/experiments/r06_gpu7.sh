#!/bin/bash
# r06 call 7: long host-pointer batches as whole chunks from a device-side double buffer against r05's 512-blob slices; parity of the long host paths
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
python tools/host_api_timing.py 2>$O/g7_err.txt | tee $O/g7_host_api_timing.txt
LWKZG_EXPERIMENTAL=1 LWKZG_HOST_STAGE=0 python tools/host_api_timing.py 2>>$O/g7_err.txt | tee $O/g7_host_api_timing_r05_slices_arm.txt
LWKZG_DIRECT_BITS=0 python tools/host_api_timing.py 2>>$O/g7_err.txt | tail -3 | tee $O/g7_host_api_timing_bucket.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_proof_parity.py tests/test_gpu_host_api_extras.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -4
python tools/leak_check.py 2>&1 | tail -5
