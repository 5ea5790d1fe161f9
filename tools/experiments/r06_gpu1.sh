#!/bin/bash
# r06 call 1: the device-resident batch verification as shipped in r05 -- kernel timeline (rocprofv3 --kernel-trace), the same kernels solo
# (host-pointer form: the hash runs on host threads), and the first placement experiments (LDS footprints, submission order)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export LWKZG_EXPERIMENTAL=1
python tools/verify_device_loop.py --tag baseline 2>$O/g1_err.txt | tee $O/g1_baseline.json
python tools/verify_device_loop.py --host --tag host_form_solo_kernels 2>>$O/g1_err.txt | tee $O/g1_host.json
LWKZG_TIMING=1 python tools/verify_device_loop.py --calls 3 --tag timing 2>&1 | grep -a "verify batch" | tail -3
rocprofv3 --kernel-trace --output-format csv -d $O/kt_verify_dev -o kt -- python3 tools/verify_device_loop.py --no-profile --calls 4 > $O/g1_kt_line.json 2> $O/g1_kt_err.txt
python tools/timeline.py $(ls $O/kt_verify_dev/*/kt_kernel_trace.csv $O/kt_verify_dev/kt_kernel_trace.csv 2>/dev/null | head -1) 60 > $O/verify_b4096_device_timeline_r05_code.txt
rocprofv3 --kernel-trace --output-format csv -d $O/kt_verify_host -o kt -- python3 tools/verify_device_loop.py --host --no-profile --calls 3 > $O/g1_kt_host_line.json 2>> $O/g1_kt_err.txt
python tools/timeline.py $(ls $O/kt_verify_host/*/kt_kernel_trace.csv $O/kt_verify_host/kt_kernel_trace.csv 2>/dev/null | head -1) 40 > $O/verify_b4096_host_timeline_r05_code.txt
for pad in "60,116,0" "60,116,56" "104,150,0" "104,150,100" "0,116,0" "60,0,0"; do
  LWKZG_VERIFY_PAD_KB=$pad python tools/verify_device_loop.py --tag "pad=$pad" 2>>$O/g1_err.txt | tee -a $O/g1_pads.jsonl
done
LWKZG_VERIFY_ORDER=1 python tools/verify_device_loop.py --tag "hash_first" 2>>$O/g1_err.txt | tee -a $O/g1_pads.jsonl
LWKZG_VERIFY_ORDER=1 LWKZG_VERIFY_PAD_KB=60,116,56 python tools/verify_device_loop.py --tag "hash_first pad=60,116,56" 2>>$O/g1_err.txt | tee -a $O/g1_pads.jsonl
LWKZG_HASH_PRIO=0 python tools/verify_device_loop.py --tag "hash_prio=0" 2>>$O/g1_err.txt | tee -a $O/g1_pads.jsonl
tail -5 $O/g1_err.txt
