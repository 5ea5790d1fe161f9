#!/bin/bash
# r06 call 30: on the final code -- __graft_entry__.smoke(), the GPU suite through the host-ASan + UBSan build, the load / free memory plateau with a long host-pointer verification inside the cycle
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/host_asan_gpu.sh 2>&1 | tail -8
cp gpurun_out/host_asan/log.txt $O/g30_host_asan_gpu_log.txt
timeout 900 python tools/leak_check.py 2>&1 | tail -12 | tee $O/g30_leak_check.txt
