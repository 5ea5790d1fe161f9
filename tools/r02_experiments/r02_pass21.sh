#!/bin/bash
# A/B/C of k_direct_accumulate on one box: the next row prefetched into registers (shipping) or into LDS by DMA;
# products in the compiler's form or as chained multiply-adds. Bench line of each twice, then the MSM parity tests on the LDS builds.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/p21; mkdir -p $O
for rep in 1 2; do
for v in vgprrow_nochain ldsrow_nochain ldsrow_chain; do
  export LWKZG_LIBRARY=$GRAFT_REPO_ROOT/lambdaworks_kzg_amd/lib_$v/liblambdaworks_kzg.so
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "import json,sys; j=json.load(sys.stdin); print('$v', round(j['value']), 'ops/s', j['roofline']['avg_launch_ms'], 'ms')" >> $O/ab.txt
done
done
for v in ldsrow_nochain ldsrow_chain; do
  export LWKZG_LIBRARY=$GRAFT_REPO_ROOT/lambdaworks_kzg_amd/lib_$v/liblambdaworks_kzg.so
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --direct-bits default 2>/dev/null | python3 -c "import json,sys; j=json.load(sys.stdin); print('$v default engine', round(j['value']), 'ops/s', j['roofline']['avg_launch_ms'], 'ms')" >> $O/ab.txt
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1 >> $O/ab.txt
done
cat $O/ab.txt
