set -x
mkdir -p gpurun_out/r04d
# parity on the bucket engine (every test that runs on it) first
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_proof_parity.py tests/test_gpu_setups.py tests/test_gpu_setups_unstructured.py -m gpu -x -q -k "bucket or engine or adversarial or digits or unstructured" 2>&1 | tail -8 > gpurun_out/r04d/bucket_tests.txt
tail -4 gpurun_out/r04d/bucket_tests.txt
for arm in 1 0; do
  LWKZG_BUCKET_ASM=$arm python bench.py --direct-bits 0 --no-config-legs --no-cpu-baseline --steps 10 > gpurun_out/r04d/bench_bucket_asm$arm.json 2> gpurun_out/r04d/err$arm.txt
  cp bench_detail.json gpurun_out/r04d/detail_bucket_asm$arm.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r04d/detail_bucket_asm$arm.json"))
print("asm=$arm", round(d["value"]), {k: round(v["avg_ms"],3) for k,v in d["kernels"].items()})
PY
done
