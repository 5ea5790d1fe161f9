#!/bin/bash
# evaluation-form quotient in the coalesced layout (thread t: elements 256 k + t, domain points in element order): parity, clock; sub-batches on the direct engine's headline (LWKZG_SPLIT)
timeout 1500 python -m pytest tests/test_gpu_lagrange.py -x -q -m gpu 2>&1 | tail -3
bp() { python bench.py --op blob_proof --batch $1 --steps 30 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_avg_ms']; print('$3 n=$1:', l['value'], l['ms_per_step'], {a:round(b,3) for a,b in k.items() if 'quot' in a or 'copy_le' in a or 'parse' in a})"; }
for rep in 1 2; do
  bp 1024 "--mode reference" "reference16"
  bp 1024 "--mode ckzg" "ckzg-evaluation-form16"
done
bp 256 "--mode ckzg" "ckzg-evaluation-form16"
bp 4096 "--mode reference" "reference16"
bp 4096 "--mode ckzg" "ckzg-evaluation-form16"
for split in 1 2 1 2; do
  LWKZG_SPLIT=$split python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('LWKZG_SPLIT=$split headline:', l['value'], l['ms_per_step'], {a:round(b,3) for a,b in l['kernels_avg_ms'].items()})"
done
timeout 600 python tools/leak_check.py 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python tools/leak_check_ckzg.py 2>&1 | grep -v amdgpu.ids | tail -3
