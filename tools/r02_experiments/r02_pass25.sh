#!/bin/bash
# full GPU suite + c-kzg-mode bench lines (commit and proofs) on the fr28 transform
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/p25; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
timeout 300 python bench.py --mode ckzg --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench_ckzg.json 2> $O/bench.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench_ref.json 2>> $O/bench.err
timeout 300 python bench.py --mode ckzg --direct-bits default --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench_ckzg_default.json 2>> $O/bench.err
timeout 300 python bench.py --mode ckzg --op blob_proof --batch 256 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ckzg_proof256.json 2>> $O/bench.err
python3 - <<'PY' >> $O/tests.txt
import json
for f in ("bench_ref", "bench_ckzg", "bench_ckzg_default", "bench_ckzg_proof256"):
    j = json.load(open("gpurun_out/p25/%s.json" % f))
    print(f, round(j["value"]), j["unit"], round(j["ms_per_step"], 3), {k: round(v["avg_ms"], 4) for k, v in j.get("kernels", {}).items()})
PY
cat $O/tests.txt
