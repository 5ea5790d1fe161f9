timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_setups_unstructured.py tests/test_gpu_lagrange.py -m gpu -x -q -k "bucket or adversarial or repairs or lagrange_buckets" 2>&1 | tail -3
for rep in 1 2 3; do
python bench.py --direct-bits 0 --no-config-legs --no-cpu-baseline --steps 10 > /tmp/l.json 2>/dev/null
python - <<PY
import json
d=json.load(open("bench_detail.json"))
print("bucket", round(d["value"]), {k: round(v["avg_ms"],3) for k,v in d["kernels"].items()})
PY
done
