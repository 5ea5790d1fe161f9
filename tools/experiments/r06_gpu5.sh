#!/bin/bash
# r06 call 5: the pinned ring of the long host-pointer batches against r05's pageable copies; the submission order at 8192 blobs; the projection
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
python tools/host_api_timing.py 2>$O/g5_err.txt | tee $O/g5_host_api_timing.txt
LWKZG_EXPERIMENTAL=1 LWKZG_STAGE_PINNED=0 python tools/host_api_timing.py 2>>$O/g5_err.txt | tee $O/g5_host_api_timing_pageable_arm.txt
for n in 4096 6144 8192 12288; do python tools/verify_device_loop.py --n $n --calls 3 --tag "n=$n" 2>>$O/g5_err.txt | tee -a $O/g5_large.jsonl; done
python tools/verify_device_loop.py --host --tag "host form 4096" 2>>$O/g5_err.txt | tee -a $O/g5_large.jsonl
python tools/scaling_projection.py --out $O/scaling_projection.json 2>$O/g5_proj_err.txt | tee $O/g5_projection_table.md; tail -3 $O/g5_proj_err.txt
timeout 1200 python -m pytest tests/test_gpu_host_api_extras.py tests/test_gpu_verify_device.py tests/test_gpu_verify_msm.py -x -q -m gpu 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_proof_parity.py -x -q -m gpu -k "2100 or long or host or slice or 4096 or 2300" 2>&1 | tail -4
tail -3 $O/g5_err.txt
