#!/bin/bash
# workgroup size of the direct kernel: 128 / 256 (shipped) / 512 lanes, same box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/p38; mkdir -p $O
for rep in 1 2; do
for v in t256 t128 t512; do
  if [ $v = t256 ]; then unset LWKZG_LIBRARY; else export LWKZG_LIBRARY=$GRAFT_REPO_ROOT/lambdaworks_kzg_amd/lib_$v/liblambdaworks_kzg.so; fi
  for bits in default 16; do
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --direct-bits $bits 2>/dev/null | python3 -c "import json,sys; j=json.load(sys.stdin); print('$v bits=$bits', round(j['value']), 'ops/s', j['roofline']['avg_launch_ms'], 'ms')" >> $O/ab.txt 2>&1 || echo "$v bits=$bits failed" >> $O/ab.txt
  done
done
done
cat $O/ab.txt
