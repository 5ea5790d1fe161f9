#!/bin/bash
# the whole GPU suite on the round's code; evaluation-form quotient with its uniform chain on one lane: A/B again
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
bp() { python bench.py --op blob_proof --batch $1 --steps 30 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_avg_ms']; print('$3 n=$1:', l['value'], l['ms_per_step'], l.get('cold_value'), {a:round(b,3) for a,b in k.items() if 'quot' in a or 'ntt' in a or 'copy_le' in a or 'bitrev' in a or 'parse' in a})"; }
for rep in 1 2; do
  bp 1024 "--mode reference" "reference16"
  bp 1024 "--mode ckzg" "ckzg-evaluation-form16"
done
bp 256 "--mode reference" "reference16"
bp 256 "--mode ckzg" "ckzg-evaluation-form16"
bp 128 "--mode reference" "reference16"
bp 64 "--mode reference" "reference16"
