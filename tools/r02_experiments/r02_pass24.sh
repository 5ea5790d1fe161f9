#!/bin/bash
# the NTT on fr28 arithmetic: transform tests, the c-kzg-mode suites, and the c-kzg-mode bench line with its kernel clock
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/p24; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ntt or ckzg or mode_c or vectors or modes or fft" 2>&1 | tail -4 > $O/tests.txt
timeout 300 python bench.py --mode ckzg --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench_ckzg.json 2> $O/bench.err
python3 - <<'PY' >> $O/tests.txt
import json
j = json.load(open("gpurun_out/p24/bench_ckzg.json"))
print("ckzg mode", round(j["value"]), "ops/s", j["ms_per_step"], "ms/step; kernels:", {k: round(v["total_ms"] / max(1, v["launches"]), 4) for k, v in j.get("kernel_profile", {}).items()} if "kernel_profile" in j else "")
PY
cat $O/tests.txt; tail -3 $O/bench.err
