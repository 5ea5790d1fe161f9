#!/bin/bash
# the same A/B with more repetitions: even halves against a first and second sub-batch of one chunk each, 256 blobs
bp() { python bench.py --op blob_proof --batch 256 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$1:', l['value'], l['ms_per_step'])"; }
for rep in 1 2 3 4 5 6; do
  LWKZG_MID_PROOF_SPLIT="" bp "even halves"
  LWKZG_MID_PROOF_SPLIT="1,1,2" bp "1,1,2"
done
