#!/bin/bash
# uneven sub-batches of the pipelined 256-blob proof call: a first sub-batch of one chunk starts its MSM a chunk's hashing earlier
bp() { python bench.py --op blob_proof --batch $1 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$2 n=$1:', l['value'], l['ms_per_step'])"; }
for rep in 1 2; do
for split in "" "1,3" "1,1,2" "1,2,1" "2,2" "1,1,1,1"; do
  LWKZG_MID_PROOF_SPLIT=$split bp 256 "split=[$split]"
done
done
for split in "" "1,3" "1,1,2"; do LWKZG_MID_PROOF_SPLIT=$split bp 384 "split=[$split]"; LWKZG_MID_PROOF_SPLIT=$split bp 192 "split=[$split]"; done
LWKZG_MID_PROOF_SPLIT=1,3 timeout 600 python -m pytest tests/test_gpu_proof_parity.py -x -q -m gpu -k "pipelined" 2>&1 | tail -1
