#!/bin/bash
# device-resident batch verification in reference mode evaluates all blobs in one launch straight from their bytes: parity, the leg
timeout 1500 python -m pytest tests/test_gpu_verify_device.py tests/test_gpu_dist.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "verify" 2>&1 | tail -2
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print(l['value'], l['box']); print({k:(v.get('value'), v.get('ms_per_step')) for k,v in l['configs'].items() if isinstance(v,dict) and 'verify' in k})"
