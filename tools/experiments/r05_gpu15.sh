#!/bin/bash
# the evaluation-form quotient (c-kzg proofs on the Lagrange form, no transform): parity, then A/B against the coefficient-form arm; why an 8-blob device-resident proof call takes 4 ms
timeout 1500 python -m pytest tests/test_gpu_lagrange.py -x -q -m gpu 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_proof_parity.py tests/test_gpu_verify_device.py -x -q -m gpu -k "ckzg or mode_c or vectors or lagrange or le" 2>&1 | tail -3
bp() { python bench.py --op blob_proof --batch $1 --steps 30 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_avg_ms']; print('$3 n=$1:', l['value'], l['ms_per_step'], {a:round(b,3) for a,b in k.items() if 'quot' in a or 'ntt' in a or 'copy_le' in a or 'bitrev' in a})"; }
for n in 1024 256; do
  for bits in 16 13 0; do
    bp $n "--mode reference --direct-bits $bits" "reference$bits"
    LWKZG_CKZG_EVAL_PROOFS=0 bp $n "--mode ckzg --direct-bits $bits" "ckzg-coefficient-form$bits"
    bp $n "--mode ckzg --direct-bits $bits" "ckzg-evaluation-form$bits"
  done
done
for n in 1 2 4 8 12 16; do
  python bench.py --op blob_proof --batch $n --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('blob_proof n=$n:', l['value'], l['ms_per_step'], {a:round(b,3) for a,b in l['kernels_avg_ms'].items()})"
done
