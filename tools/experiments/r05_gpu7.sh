#!/bin/bash
# pipelined mid-size proof calls (hash of the second half beside the first half's MSM, no wait for the validation): A/B + parity
for pipe in 1 0; do
  echo "== LWKZG_MID_PROOF_PIPE=$pipe"
  LWKZG_MID_PROOF_PIPE=$pipe python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline --direct-bits default 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('default engine 256:', l['value'], l['ms_per_step'], l['kernels_avg_ms'])"
  LWKZG_MID_PROOF_PIPE=$pipe python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('16-bit 256:', l['value'], l['ms_per_step'])"
  LWKZG_MID_PROOF_PIPE=$pipe python bench.py --op blob_proof --batch 384 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('16-bit 384:', l['value'], l['ms_per_step'])"
done
timeout 1200 python -m pytest tests/test_gpu_proof_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "proof or noncanonical" 2>&1 | tail -3
