mkdir -p gpurun_out/r04l
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_setups_unstructured.py tests/test_gpu_proof_parity.py tests/test_gpu_setups.py -m gpu -x -q -k "bucket or engine or adversarial or repairs" 2>&1 | tail -5
for i in 1 2; do
python bench.py --direct-bits 0 --no-config-legs --no-cpu-baseline --steps 10 > gpurun_out/r04l/bench_bucket.json 2> gpurun_out/r04l/err.txt
cp bench_detail.json gpurun_out/r04l/detail_bucket.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04l/detail_bucket.json"))
print("bucket", round(d["value"]), {k: round(v["avg_ms"],3) for k,v in d["kernels"].items()})
PY
done
