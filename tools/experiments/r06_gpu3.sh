#!/bin/bash
# r06 call 3: records assembled on the device + one pinned copy, prefixed SHA, Barrett split, inlined products in the reduce kernels: clock; then the whole GPU suite after the knob migration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_verify_msm.py -x -q -m gpu 2>&1 | tail -3
python tools/verify_device_loop.py --tag "vmsm v2" 2>$O/g3_err.txt | tee $O/g3_vmsm.json
LWKZG_TIMING=1 python tools/verify_device_loop.py --calls 3 --tag timing 2>&1 | grep -a "verify batch\|pairing" | tail -4
python tools/verify_device_loop.py --n 512 --tag "n=512" 2>>$O/g3_err.txt | tee -a $O/g3_arms.jsonl
python tools/verify_device_loop.py --n 1024 --tag "n=1024" 2>>$O/g3_err.txt | tee -a $O/g3_arms.jsonl
python tools/verify_device_loop.py --n 16384 --calls 3 --tag "n=16384" 2>>$O/g3_err.txt | tee -a $O/g3_arms.jsonl
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/g3_suite_tail.txt
