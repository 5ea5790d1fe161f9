set -x
mkdir -p gpurun_out/r04e
timeout 900 python -m pytest tests/test_gpu_setups_unstructured.py tests/test_gpu_parity.py -m gpu -x -q -k "bucket or repairs or adversarial or engine" 2>&1 | tail -6 > gpurun_out/r04e/bucket_tests.txt
tail -3 gpurun_out/r04e/bucket_tests.txt
for split in 2 3 4 1; do
  LWKZG_SPLIT=$split python bench.py --direct-bits 0 --no-config-legs --no-cpu-baseline --steps 10 > gpurun_out/r04e/bench_split$split.json 2> gpurun_out/r04e/err$split.txt
  cp bench_detail.json gpurun_out/r04e/detail_split$split.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r04e/detail_split$split.json"))
print("split=$split", round(d["value"]), {k: round(v["avg_ms"],3) for k,v in d["kernels"].items()})
PY
done
