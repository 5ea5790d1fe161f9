#!/bin/bash
# verification wants y = p(z) only: the quotient kernels without their back-substitution / quotient loops; parity of everything that verifies, the verify legs
timeout 2000 python -m pytest tests/test_gpu_verify_device.py tests/test_gpu_lagrange.py tests/test_gpu_fuzz_seeds.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api_extras.py -x -q -m gpu -k "verify or vectors" 2>&1 | tail -2
for op in "verify_batch --batch 4096 --steps 5" ; do
python bench.py --op $op --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('verify host 4096:', l['value'], l['ms_per_step'], {a:round(b,3) for a,b in l['kernels_avg_ms'].items() if 'quot' in a})"
done
python tools/single_blob_timing.py 2>/dev/null | grep -v amdgpu.ids | tail -4
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print(l['value'], l['api_latency_ms']); print({k:(v.get('value'), v.get('cold_value')) for k,v in l['configs'].items() if isinstance(v,dict)})"
