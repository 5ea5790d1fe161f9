#!/bin/bash
# host finishing for host-pointer commitment batches of 2..8; the validation's own clock (GPU-hash arm: little beside it); sub-batch count again
timeout 1500 python -m pytest tests/test_gpu_coop.py tests/test_gpu_host_api_extras.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
python tools/small_batch_timing.py 2>&1 | grep -v amdgpu.ids
LWKZG_MID_PROOF_HOST=0 python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('blob_proof 256 (GPU hash):', l['value'], l['ms_per_step'], l['kernels_avg_ms'])"
for parts in 2 3 4 5 6; do
  LWKZG_MID_PROOF_PARTS=$parts python bench.py --op blob_proof --batch 256 --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('parts $parts blob_proof 256:', l['value'], l['ms_per_step'], l.get('cold_value'))"
done
for b in 64 128 512; do
  python bench.py --op blob_proof --batch $b --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('blob_proof $b:', l['value'], l['ms_per_step'], l.get('cold_value'))"
done
python bench.py --steps 10 --warmup 3 > gpurun_out/r05/gpu13_bench.json 2>/dev/null; python -c "
import json
l=json.loads(open('gpurun_out/r05/gpu13_bench.json').read().strip().splitlines()[-1]); print(l['value'], l.get('api_latency_ms')); print({k:(v.get('value'),v.get('cold_value')) for k,v in l.get('legs',{}).items()} if 'legs' in l else list(l.keys()))"
