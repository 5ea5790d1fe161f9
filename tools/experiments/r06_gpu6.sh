#!/bin/bash
# r06 call 6: where a host-pointer batch's time goes between pageable memory and HBM; every proof schedule forced once; proof parity after the plan refactor
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
./tools/h2d_bench_bin | tee $O/g6_h2d_bench.txt
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|^CPU\(s\)" 
timeout 900 python -m pytest tests/test_gpu_plan.py -x -q -m gpu 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_proof_parity.py -x -q -m gpu 2>&1 | tail -3
python bench.py --op blob_proof --batch 256 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('blob_proof 256:', l['value'], l['ms_per_step'], l.get('cold_value'))"
