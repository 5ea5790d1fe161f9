#!/bin/bash
for parts in 4 2; do
  echo "== LWKZG_MID_PROOF_PARTS=$parts"
  LWKZG_MID_PROOF_PARTS=$parts python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline --direct-bits default 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('default engine 256:', l['value'], l['ms_per_step'])"
  LWKZG_MID_PROOF_PARTS=$parts python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('16-bit 256:', l['value'], l['ms_per_step'])"
done
LWKZG_MID_PROOF_PARTS=4 LWKZG_MID_PROOF_CHUNKS=8 python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('16-bit 256, 8 chunks / 4 parts:', l['value'], l['ms_per_step'])"
LWKZG_MID_PROOF_PARTS=8 LWKZG_MID_PROOF_CHUNKS=8 python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('16-bit 256, 8 chunks / 8 parts:', l['value'], l['ms_per_step'])"
