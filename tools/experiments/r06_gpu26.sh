#!/bin/bash
# r06 call 26: which pair of side streams lets the uploads of a staged verification run beside the head's 3.1 ms hash launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export LWKZG_EXPERIMENTAL=1
for p in 3,0 3,1 3,2 2,0 1,0 4,0 5,0 6,0 7,0 3,4 3,5 3,6 3,7 2,5; do
  LWKZG_STAGE_STREAMS=$p python tools/verify_device_loop.py --n 4096 --calls 6 --host --tag "streams $p" 2>/dev/null | tail -1 | cut -c1-110 | tee -a $O/g26_streams.jsonl
done
env | grep -i "GPU_MAX_HW\|HIP_\|HSA_" | head
