#!/bin/bash
# r06 call 13: the default bench run with the headline first (leg order, step spread, table build on a clean process)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
( time LWKZG_BENCH_DETAIL=$O/g13_bench_detail.json python bench.py > $O/g13_bench_line.json 2> $O/g13_bench_err.txt ) 2>&1 | tail -3
tail -c 1500 $O/g13_bench_err.txt
python - <<'PY'
import json, os
O = os.path.expandvars("$GRAFT_REPO_ROOT/gpurun_out/r06")
l = json.loads(open(O + "/g13_bench_line.json").read().strip().splitlines()[-1])
print("value", l["value"], "ms_per_step", l["ms_per_step"], "step_ms", l.get("step_ms"), "build_s", l.get("direct_table_build_s"), "load_s", l.get("setup_load_s"))
print("default", l.get("default_engine"), "\nbucket", l.get("bucket_engine"), "\napi", l.get("api_latency_ms"), "\nhost_abi", l.get("host_abi"))
print({k: (v.get("value"), v.get("ms_per_step")) for k, v in (l.get("configs") or {}).items()})
print(l["roofline"]); print(l.get("cpu_baseline")); print(len(open(O + "/g13_bench_line.json").read()))
d = json.load(open(O + "/g13_bench_detail.json"))
print(d.get("leg_order")); print(d["default_engine"].get("table_build_s_after_freeing_the_headline_table"), d.get("direct_table_build_breakdown_ms"))
PY
