#!/bin/bash
# the square root's chain with its window table in LDS and inlined products: parity of everything that validates, its clock; the default bench with the bucket leg last
timeout 2000 python -m pytest tests/test_gpu_coop.py tests/test_gpu_verify_device.py tests/test_gpu_proof_parity.py tests/test_gpu_fuzz_seeds.py tests/test_gpu_setups.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "verify or proof or bad or vectors" 2>&1 | tail -3
LWKZG_MID_PROOF_HOST=0 python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('blob_proof 256 (GPU hash):', l['value'], l['ms_per_step'], {a:round(b,3) for a,b in l['kernels_avg_ms'].items() if 'subgroup' in a or 'decompress' in a})"
python bench.py --op blob_proof --batch 256 --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('blob_proof 256:', l['value'], l['ms_per_step'], l.get('cold_value'))"
python bench.py --op verify_batch --batch 4096 --steps 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('verify 4096 host:', l['value'], l['ms_per_step'], {a:round(b,3) for a,b in l['kernels_avg_ms'].items() if 'subgroup' in a or 'decompress' in a})"
python bench.py > gpurun_out/r05/gpu19_bench.json 2> gpurun_out/r05/gpu19_bench.err; tail -c 600 gpurun_out/r05/gpu19_bench.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r05/gpu19_bench.json').read().strip().splitlines()[-1])
print(l['value'], l['ms_per_step'], 'build_s', l.get('direct_table_build_s'), 'load_s', l.get('setup_load_s'))
print('default', l['default_engine'].get('value'), 'bucket', l['bucket_engine'].get('value'), l.get('api_latency_ms'))
print({k:(v.get('value'), v.get('cold_value')) for k,v in l['configs'].items()} if isinstance(l.get('configs'),dict) else l.get('configs'))
print(l['roofline']); print(l['cpu_baseline']); print(len(open('gpurun_out/r05/gpu19_bench.json').read()))
PY
