#!/bin/bash
# r06 call 31: the GPU suite through the host-ASan + UBSan build on the final code (the eight-rank rehearsal left out like the two-rank one)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
bash tools/host_asan_gpu.sh 2>&1 | tail -8
cp gpurun_out/host_asan/log.txt $O/g31_host_asan_gpu_log.txt
