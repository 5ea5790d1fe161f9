mkdir -p gpurun_out/r04i
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r04i/gpu_tests.txt
tail -5 gpurun_out/r04i/gpu_tests.txt
python bench.py > gpurun_out/r04i/bench_line.json 2> gpurun_out/r04i/bench_err.txt
wc -c gpurun_out/r04i/bench_line.json
cp bench_detail.json gpurun_out/r04i/bench_detail.json
python - <<'PY'
import json
l=json.load(open("gpurun_out/r04i/bench_line.json"))
print(l["value"], l["roofline"]["avg_launch_ms"], l["default_engine"]["value"], l["bucket_engine"]["value"])
for k,v in l["configs"].items(): print(k, v)
PY
