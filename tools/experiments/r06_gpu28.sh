#!/bin/bash
# r06 call 28: the head's hash launches round three streams (they overlap), the host takes the last 3.4 ms of upload: 2048 / 4096 / 16384 blobs against the sliced form, the arms test, a timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
for n in 2048 4096 8192 16384; do
  for arm in staged sliced staged sliced; do
    if [ $arm = sliced ]; then export LWKZG_EXPERIMENTAL=1 LWKZG_HOST_STAGE=0; else unset LWKZG_HOST_STAGE; fi
    python tools/verify_device_loop.py --n $n --calls 6 --host --tag "host form, $n blobs, $arm" 2>/dev/null | tail -1 | cut -c1-150 | tee -a $O/g28_verify.jsonl
  done
done
unset LWKZG_HOST_STAGE
LWKZG_TIMING=1 python tools/verify_device_loop.py --n 4096 --calls 4 --host --tag "host" 2>&1 | grep "verify batch" | tail -3
python tools/host_api_timing.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/g28_host_api_timing.txt
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt_host28 -o kt -- python3 tools/verify_device_loop.py --n 4096 --calls 3 --host --no-profile > $O/g28_kt_out.txt 2> $O/g28_kt_err.txt
python3 tools/experiments/trace_timeline.py $O/kt_host28 $O/g28_host_timeline.txt
rm -rf $O/kt_host28
grep -v "copyBuffer\|fillBuffer\|mont_to_bytes" $O/g28_host_timeline.txt | tail -45
timeout 1200 python -m pytest tests/test_gpu_verify_msm.py -x -q -m gpu 2>&1 | tail -2
