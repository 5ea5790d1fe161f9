#!/bin/bash
# r06 call 23: the host threads' own hashing rate (no GPU work beside it), by grain; the box's NUMA layout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
ls /sys/devices/system/node/ | grep node; cat /sys/devices/system/node/node*/cpulist; cat /proc/sys/kernel/numa_balancing
export LWKZG_EXPERIMENTAL=1
for g in 4 1 16 64; do LWKZG_HOST_HASH_GRAIN=$g python tools/experiments/r06_gpu23.py 2>&1 | grep grain; done | tee $O/g23_host_hash.txt
