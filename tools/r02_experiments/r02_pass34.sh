#!/bin/bash
# the next scalar fetched one scalar ahead of its use (8 more live registers) against a load at the point of use
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/p34; mkdir -p $O
for rep in 1 2 3; do
for v in plain spf; do
  if [ $v = spf ]; then export LWKZG_LIBRARY=$GRAFT_REPO_ROOT/lambdaworks_kzg_amd/lib_spf/liblambdaworks_kzg.so; else unset LWKZG_LIBRARY; fi
  for bits in default 16; do
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --direct-bits $bits 2>/dev/null | python3 -c "import json,sys; j=json.load(sys.stdin); print('$v bits=$bits', round(j['value']), 'ops/s', j['roofline']['avg_launch_ms'], 'ms')" >> $O/ab.txt 2>&1 || echo "$v bits=$bits failed" >> $O/ab.txt
  done
done
done
export LWKZG_LIBRARY=$GRAFT_REPO_ROOT/lambdaworks_kzg_amd/lib_spf/liblambdaworks_kzg.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "direct or msm or commit" 2>&1 | tail -1 >> $O/ab.txt
cat $O/ab.txt
