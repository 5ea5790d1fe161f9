#!/bin/bash
# validation on quads of lanes (k_subgroup_coop_asm) + windowed square root: parity, timing A/B; hipHostRegister A/B; small-batch soak
timeout 900 python -m pytest tests/test_gpu_coop.py -x -q -m gpu -k "validation" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_proof_parity.py tests/test_gpu_verify_device.py tests/test_gpu_fuzz_seeds.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "verify or proof or setup or bad" 2>&1 | tail -3
for coop in 1 0; do
  echo "== LWKZG_VALIDATE_COOP=$coop"
  LWKZG_VALIDATE_COOP=$coop python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('blob_proof 256:', l['value'], l['ms_per_step'], {k:v for k,v in l['kernels_avg_ms'].items() if 'valid' in k or 'subgroup' in k or 'decompress' in k})"
  LWKZG_VALIDATE_COOP=$coop LWKZG_MID_PROOF_HOST=0 python bench.py --op blob_proof --batch 256 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('blob_proof 256 (GPU hash):', l['value'], l['ms_per_step'])"
  LWKZG_VALIDATE_COOP=$coop python bench.py --op verify_batch --batch 4096 --steps 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('verify 4096 host:', l['value'], l['ms_per_step'], {k:v for k,v in l['kernels_avg_ms'].items() if 'valid' in k or 'subgroup' in k or 'decompress' in k})"
done
for reg in 0 512; do
  echo "== LWKZG_HOST_REGISTER=$reg"
  LWKZG_HOST_REGISTER=$reg python tools/host_api_timing.py 2>&1 | grep -E "n=512|n=1024|n=4096"
done
timeout 900 python tools/soak_small.py --rounds 300 2>&1 | tail -2
