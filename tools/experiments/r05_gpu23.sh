#!/bin/bash
# first calls of a burst of proof calls, one by one: before (LWKZG_HOST_WARM_MS=0 forces... no: the shipped library) -- the wake-up of the host threads moved off the call's critical path
python tools/experiments/r05_proof_cold.py 2>/dev/null | tail -1
python tools/experiments/r05_proof_cold.py 2>/dev/null | tail -1
timeout 900 python -m pytest tests/test_gpu_proof_parity.py -x -q -m gpu 2>&1 | tail -2
python bench.py --op blob_proof --batch 256 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('blob_proof 256:', l['value'], l['ms_per_step'], l.get('cold_value'))"
