#!/bin/bash
# table rows packed (112 B) against rows aligned to 128 B: same box, 13-, 15- and 16-bit tables
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/p27; mkdir -p $O
for rep in 1 2; do
for v in row112 row128; do
  if [ $v = row128 ]; then export LWKZG_LIBRARY=$GRAFT_REPO_ROOT/lambdaworks_kzg_amd/lib_row128/liblambdaworks_kzg.so; else unset LWKZG_LIBRARY; fi
  for bits in default 15 16; do
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --direct-bits $bits 2>/dev/null | python3 -c "import json,sys; j=json.load(sys.stdin); print('$v bits=$bits', round(j['value']), 'ops/s', j['roofline']['avg_launch_ms'], 'ms', j.get('msm_path'))" >> $O/ab.txt 2>&1 || echo "$v bits=$bits failed" >> $O/ab.txt
  done
done
done
export LWKZG_LIBRARY=$GRAFT_REPO_ROOT/lambdaworks_kzg_amd/lib_row128/liblambdaworks_kzg.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "direct" 2>&1 | tail -1 >> $O/ab.txt
cat $O/ab.txt
