#!/bin/bash
# k_eval_quotient on fr28 (and 1024 lanes for a handful of blobs): parity + timing
timeout 1200 python -m pytest tests/test_gpu_proof_parity.py tests/test_gpu_lagrange.py tests/test_gpu_verify_device.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "proof or verify or lib_test" 2>&1 | tail -2
python tools/single_blob_timing.py 2>&1
LWKZG_EVAL_WIDE=0 python tools/single_blob_timing.py 2>&1 | sed -n 2,3p
python bench.py --op blob_proof --batch 1024 --steps 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('blob_proof 1024:', l['value'], l['ms_per_step'], {k:v for k,v in l['kernels_avg_ms'].items() if 'eval' in k or 'ntt' in k})"
python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('blob_proof 256:', l['value'], l['ms_per_step'])"
