#!/bin/bash
# bucket engine with its fixed-base table rows in 128-byte lines: bench on the bucket engine, MSM parity + multi-process import tests
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/p31; mkdir -p $O
for rep in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --direct-bits 0 2>/dev/null | python3 -c "import json,sys; j=json.load(sys.stdin); print('bucket', round(j['value']), 'ops/s', j['kernels'])" >> $O/ab.txt 2>&1
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -2 >> $O/ab.txt
cat $O/ab.txt
