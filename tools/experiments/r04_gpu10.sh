mkdir -p gpurun_out/r04j
python -m pytest tests/test_gpu_setups_unstructured.py tests/test_gpu_setups.py tests/test_gpu_lagrange.py tests/test_gpu_multi.py tests/test_gpu_fuzz_seeds.py tests/test_gpu_zz_env.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04j/tests.txt
tail -4 gpurun_out/r04j/tests.txt
python bench.py > gpurun_out/r04j/bench_line.json 2> gpurun_out/r04j/bench_err.txt
cp bench_detail.json gpurun_out/r04j/bench_detail.json
python - <<'PY'
import json
l=json.load(open("gpurun_out/r04j/bench_line.json"))
print(l["value"], l["roofline"]["avg_launch_ms"], l["default_engine"]["value"], l["bucket_engine"]["value"])
for k,v in l["configs"].items(): print(k, v)
d=json.load(open("gpurun_out/r04j/bench_detail.json"))
print(d["configs"]["ckzg_commit_b1024_lagrange"])
PY
