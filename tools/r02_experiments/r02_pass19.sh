#!/bin/bash
# A/B of a library build: headline bench (16-bit), default engine, and the parity tests of the MSM paths
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/p19
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/p19/bench.json 2> gpurun_out/p19/bench.err
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/p19/bench.json"))
print("headline", round(j["value"]), "ops/s; accumulate avg ms", j["roofline"]["avg_launch_ms"], "; default", j.get("default_engine", {}).get("value"), "; bucket", j.get("bucket_engine", {}).get("value"), "; host_abi", j.get("host_abi", {}).get("value"))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
