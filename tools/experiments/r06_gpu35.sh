#!/bin/bash
# r06 call 35: the uploads of a staged verification on a HIGH-PRIORITY stream of their own (LWKZG_STAGE_STREAMS=8,h) against the shipped pair, for every hash stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export LWKZG_EXPERIMENTAL=1
for p in 1,0 8,0 8,1 8,2 8,3 8,4 8,5 8,6 8,7 3,0 1,0 8,0; do
  LWKZG_STAGE_STREAMS=$p python tools/verify_device_loop.py --n 4096 --calls 6 --host --tag "streams $p" 2>/dev/null | tail -1 | cut -c1-110 | tee -a $O/g35_streams.jsonl
done
