#!/bin/bash
# row gathers with the nt (non-temporal) cache hint against plain loads, aligned rows, same box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/p29; mkdir -p $O
for rep in 1 2; do
for v in plain nt; do
  if [ $v = nt ]; then export LWKZG_LIBRARY=$GRAFT_REPO_ROOT/lambdaworks_kzg_amd/lib_nt/liblambdaworks_kzg.so; else unset LWKZG_LIBRARY; fi
  for bits in default 16; do
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --direct-bits $bits 2>/dev/null | python3 -c "import json,sys; j=json.load(sys.stdin); print('$v bits=$bits', round(j['value']), 'ops/s', j['roofline']['avg_launch_ms'], 'ms')" >> $O/ab.txt 2>&1 || echo "$v bits=$bits failed" >> $O/ab.txt
  done
done
done
cat $O/ab.txt
