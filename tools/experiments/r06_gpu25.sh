#!/bin/bash
# r06 call 25: long host-pointer verifications through one device buffer, the hashing shared between the GPU (head) and the host threads (tail)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_verify_msm.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_verify_device.py tests/test_gpu_coop.py -x -q -m gpu -k "verify or long" 2>&1 | tail -3
for arm in staged sliced staged sliced; do
  if [ $arm = sliced ]; then export LWKZG_EXPERIMENTAL=1 LWKZG_HOST_STAGE=0; else unset LWKZG_HOST_STAGE; fi
  python tools/verify_device_loop.py --n 4096 --calls 8 --host --tag "host form, $arm" 2>/dev/null | tail -1 | cut -c1-150 | tee -a $O/g25_verify.jsonl
done
unset LWKZG_HOST_STAGE
python tools/verify_device_loop.py --n 2048 --calls 8 --host --tag "host form 2048" 2>/dev/null | tail -1 | cut -c1-150 | tee -a $O/g25_verify.jsonl
python tools/verify_device_loop.py --n 16384 --calls 3 --host --tag "host form 16384" 2>/dev/null | tail -1 | cut -c1-150 | tee -a $O/g25_verify.jsonl
LWKZG_EXPERIMENTAL=1 LWKZG_HOST_STAGE=0 python tools/verify_device_loop.py --n 16384 --calls 3 --host --tag "host form 16384 sliced" 2>/dev/null | tail -1 | cut -c1-150 | tee -a $O/g25_verify.jsonl
LWKZG_TIMING=1 python tools/verify_device_loop.py --n 4096 --calls 4 --host --tag "host" 2>&1 | grep "verify batch" | tail -3
python tools/host_api_timing.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/g25_host_api_timing.txt
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt_host25 -o kt -- python3 tools/verify_device_loop.py --n 4096 --calls 3 --host --no-profile > $O/g25_kt_out.txt 2> $O/g25_kt_err.txt
python3 tools/experiments/trace_timeline.py $O/kt_host25 $O/g25_host_timeline.txt
rm -rf $O/kt_host25
grep -v "copyBuffer\|fillBuffer\|mont_to_bytes" $O/g25_host_timeline.txt | tail -60
