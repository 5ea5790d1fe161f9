#!/bin/bash
# r06 call 40: after moving the staged split into plan.h: the tests of the long verification, both forms at 2048 ... 16384 blobs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_verify_msm.py tests/test_gpu_parity.py tests/test_gpu_verify_device.py -x -q -m gpu -k "arms or long_batch or evaluate_straight" 2>&1 | tail -2
for n in 2048 4096 16384; do python tools/verify_device_loop.py --n $n --calls 6 --host --tag "host form, $n blobs" 2>/dev/null | tail -1 | cut -c1-140; done
