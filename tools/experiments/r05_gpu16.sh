#!/bin/bash
# evaluation-form quotient with 5 products per element: parity again, A/B again
timeout 1500 python -m pytest tests/test_gpu_lagrange.py -x -q -m gpu 2>&1 | tail -3
bp() { python bench.py --op blob_proof --batch $1 --steps 30 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_avg_ms']; print('$3 n=$1:', l['value'], l['ms_per_step'], {a:round(b,3) for a,b in k.items() if 'quot' in a or 'ntt' in a or 'copy_le' in a or 'bitrev' in a or 'parse' in a})"; }
for rep in 1 2; do
  bp 1024 "--mode reference" "reference16"
  LWKZG_CKZG_EVAL_PROOFS=0 bp 1024 "--mode ckzg" "ckzg-coefficient-form16"
  bp 1024 "--mode ckzg" "ckzg-evaluation-form16"
done
bp 4096 "--mode reference" "reference16"
bp 4096 "--mode ckzg" "ckzg-evaluation-form16"
python tools/config_sweep.py 2>/dev/null | tail -1 > gpurun_out/r05/gpu16_sweep.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/gpu16_sweep.json').read())
for r in d['blob_proof']: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items()})
PY
