#!/bin/bash
# r06 call 2: vmsm.hip + fused, padded validation: parity of every arm, then the clock
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_verify_msm.py -x -q -m gpu 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_gpu_verify_device.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -5
python tools/verify_device_loop.py --tag vmsm 2>$O/g2_err.txt | tee $O/g2_vmsm.json
LWKZG_TIMING=1 python tools/verify_device_loop.py --calls 3 --tag timing 2>&1 | grep -a "verify batch" | tail -2
export LWKZG_EXPERIMENTAL=1
LWKZG_VERIFY_MSM=0 python tools/verify_device_loop.py --tag "r05 lincomb, fused+padded validation" 2>>$O/g2_err.txt | tee -a $O/g2_arms.jsonl
LWKZG_VERIFY_PAD_KB=0,0,0 python tools/verify_device_loop.py --tag "vmsm, no pads" 2>>$O/g2_err.txt | tee -a $O/g2_arms.jsonl
LWKZG_VERIFY_PAD_KB=60,116,56 python tools/verify_device_loop.py --tag "vmsm, pads 60,116,56" 2>>$O/g2_err.txt | tee -a $O/g2_arms.jsonl
python tools/verify_device_loop.py --host --tag "host form, vmsm" 2>>$O/g2_err.txt | tee -a $O/g2_arms.jsonl
python tools/verify_device_loop.py --n 512 --tag "n=512 vmsm" 2>>$O/g2_err.txt | tee -a $O/g2_arms.jsonl
LWKZG_VERIFY_MSM=0 LWKZG_VERIFY_FUSED=0 LWKZG_VERIFY_PAD_KB=0,0,0 python tools/verify_device_loop.py --n 512 --tag "n=512 r05" 2>>$O/g2_err.txt | tee -a $O/g2_arms.jsonl
unset LWKZG_EXPERIMENTAL
rocprofv3 --kernel-trace --output-format csv -d $O/kt_verify_dev2 -o kt -- python3 tools/verify_device_loop.py --no-profile --calls 4 > $O/g2_kt_line.json 2> $O/g2_kt_err.txt
python tools/timeline.py $(ls $O/kt_verify_dev2/*/kt_kernel_trace.csv $O/kt_verify_dev2/kt_kernel_trace.csv 2>/dev/null | head -1) 40 > $O/verify_b4096_device_timeline.txt
tail -22 $O/verify_b4096_device_timeline.txt
tail -3 $O/g2_err.txt
