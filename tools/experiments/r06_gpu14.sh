#!/bin/bash
# r06 call 14: y = p(z) as a tree on compacted lanes; identical / opposite points in one bucket; a soak of the verification through both forms
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_verify_msm.py tests/test_gpu_verify_device.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lagrange.py tests/test_gpu_dist.py -x -q -m gpu -k "verify or eval" 2>&1 | tail -3
python tools/verify_device_loop.py --tag "eval tree" 2>$O/g14_err.txt | tee $O/g14_verify.json
python tools/verify_device_loop.py --n 16384 --calls 3 --tag "eval tree 16384" 2>>$O/g14_err.txt | tee -a $O/g14_verify.json
python tools/verify_device_loop.py --host --tag "host form" 2>>$O/g14_err.txt | tee -a $O/g14_verify.json
timeout 420 python tools/soak_verify.py 360 > $O/g14_soak_verify.jsonl 2>>$O/g14_err.txt; tail -1 $O/g14_soak_verify.jsonl; grep -c '"mismatch": false' $O/g14_soak_verify.jsonl
