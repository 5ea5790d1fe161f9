#!/bin/bash
# r06 call 37: the priority upload stream under the staged commitments as well: the long host-pointer tests, the host ABI, both forms of a long verification
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_verify_msm.py tests/test_gpu_host_api_extras.py -x -q -m gpu -k "long or host or sliced or arms or staged" 2>&1 | tail -3
python tools/host_api_timing.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/g37_host_api_timing.txt
LWKZG_EXPERIMENTAL=1 LWKZG_STAGE_STREAMS=3,0 python tools/host_api_timing.py 2>&1 | grep -v amdgpu.ids | tail -1
python tools/verify_device_loop.py --n 4096 --calls 6 --host --tag "host form" 2>/dev/null | tail -1 | cut -c1-120
