#!/bin/bash
# the default bench.py with the bucket leg last and the two 1024-blob proof legs
python bench.py > gpurun_out/r05/gpu20_bench.json 2> gpurun_out/r05/gpu20_bench.err; tail -c 800 gpurun_out/r05/gpu20_bench.err | grep -v amdgpu.ids
python - <<'PY'
import json
raw=open('gpurun_out/r05/gpu20_bench.json').read()
l=json.loads(raw.strip().splitlines()[-1])
print(l['value'], l['ms_per_step'], 'build_s', l.get('direct_table_build_s'), 'load_s', l.get('setup_load_s'))
print('default', l['default_engine'].get('value'), 'bucket', l['bucket_engine'].get('value'), l.get('api_latency_ms'))
print({k:(v.get('value'), v.get('cold_value'), v.get('error')) for k,v in l['configs'].items() if isinstance(v,dict)})
print(l['roofline']); print(l['cpu_baseline']); print('line bytes', len(raw))
PY
