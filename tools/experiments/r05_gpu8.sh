#!/bin/bash
for pipe in 1 2; do
  echo "== LWKZG_MID_PROOF_PIPE=$pipe (1: second half on its own stream, 2: one stream)"
  LWKZG_MID_PROOF_PIPE=$pipe python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline --direct-bits default 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('default engine 256:', l['value'], l['ms_per_step'])"
  LWKZG_MID_PROOF_PIPE=$pipe python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('16-bit 256:', l['value'], l['ms_per_step'])"
done
