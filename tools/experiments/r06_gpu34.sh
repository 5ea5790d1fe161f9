#!/bin/bash
# r06 call 34: the staged long verification's outcomes with the defect in the head of the batch; an element >= r through the host-pointer form in c-kzg mode
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_verify_device.py -x -q -m gpu -k "long_batch or evaluate_straight" 2>&1 | tail -4
