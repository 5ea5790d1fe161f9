#!/bin/bash
# r06 call 18: the host's Fp product on MULX/ADX (fp_x86.S) against the C product on the box's own cores: pairing phases, one-blob verification,
# configs[3] in both forms
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
lscpu | grep -E "Model name|^CPU\(s\)|MHz|Flags" | cut -c1-600 > $O/g18_lscpu.txt
timeout 600 python -m pytest tests/test_gpu_verify_msm.py tests/test_gpu_verify.py -x -q -m gpu 2>&1 | tail -3
for arm in asm portable; do
  if [ $arm = portable ]; then export LWKZG_EXPERIMENTAL=1 LWKZG_HOST_FP_PORTABLE=1; fi
  LWKZG_DIRECT=0 python tools/single_blob_timing.py 2>&1 | tail -4 | tee $O/g18_single_blob_timing_$arm.txt
  LWKZG_TIMING=1 python tools/verify_device_loop.py --n 4096 --calls 8 --tag "device $arm" 2> $O/g18_timing_$arm.txt | tail -1 | tee -a $O/g18_verify.jsonl
  grep "Miller" $O/g18_timing_$arm.txt | tail -6
  python tools/verify_device_loop.py --n 4096 --calls 8 --host --tag "host $arm" 2>/dev/null | tail -1 | tee -a $O/g18_verify.jsonl
done
