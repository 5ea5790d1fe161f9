#!/bin/bash
# node-level device entry points, reserve semantics, bench with the new legs
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -4
LWKZG_BENCH_DETAIL=gpurun_out/r05/bench_detail_a.json python bench.py > gpurun_out/r05/bench_line_a.json 2> gpurun_out/r05/bench_err_a.txt; echo "bench rc=$?"
tail -3 gpurun_out/r05/bench_err_a.txt
python - <<'PY'
import json
l = json.load(open("gpurun_out/r05/bench_line_a.json"))
print(len(json.dumps(l)), "bytes")
for k in ("value", "ms_per_step", "api_latency_ms", "box", "host_abi"):
    print(k, l.get(k))
print("default", l.get("default_engine", {}).get("value"), "bucket", l.get("bucket_engine", {}).get("value"))
for k, v in l.get("configs", {}).items():
    print(" ", k, v.get("value"), v.get("unit"), v.get("ms_per_step"), v.get("kernel"), v.get("kernel_ms"), v.get("kernel_sum_over_wall"), v.get("error"))
PY
