#!/bin/bash
# the fused commit-and-prove entry point: parity tests, then pairs/s against the two separate calls (same box)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/p36; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_proof_parity.py -x -q -m gpu -k "commit_and_prove" 2>&1 | tail -3 > $O/tests.txt
for b in 256 1024 4096; do
  st=10; [ $b = 4096 ] && st=5
  for op in commit blob_proof commit_prove; do
    timeout 300 python bench.py --op $op --batch $b --steps $st --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "import json,sys; j=json.load(sys.stdin); print('$op batch=$b', round(j['value']), j['unit'], round(j['ms_per_step'],3), 'ms/step', {k: round(v['avg_ms'],3) for k,v in j.get('kernels',{}).items()})" >> $O/tests.txt 2>&1 || echo "$op $b failed" >> $O/tests.txt
  done
done
cat $O/tests.txt
