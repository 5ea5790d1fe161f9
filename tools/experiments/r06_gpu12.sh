#!/bin/bash
# r06 call 12: host-pointer batches again with the Python binding's quadratic slicing of its result buffer gone (capi.py); both arms
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
python tools/host_api_timing.py 2>$O/g12_err.txt | tee $O/g12_host_api_timing.txt
LWKZG_EXPERIMENTAL=1 LWKZG_HOST_STAGE=0 python tools/host_api_timing.py 2>>$O/g12_err.txt | tee $O/g12_host_api_timing_r05_slices_arm.txt
python tools/verify_device_loop.py --host --tag "host form 4096" 2>>$O/g12_err.txt | tee -a $O/g12_verify.jsonl
python tools/verify_device_loop.py --tag "device form 4096" 2>>$O/g12_err.txt | tee -a $O/g12_verify.jsonl
timeout 900 python -m pytest tests/test_gpu_verify_device.py tests/test_gpu_verify_msm.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "verify" 2>&1 | tail -3
