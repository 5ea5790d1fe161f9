cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_full
rm -rf $O; mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -m gpu -q -x --durations=8 > $O/gpu_test_log.txt 2>&1
echo "pytest rc=$?" >> $O/gpu_test_log.txt
tail -14 $O/gpu_test_log.txt
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_full/bench_line.json").read().strip().splitlines()[-1])
print(round(d["value"]), d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["int_mad"]["frac_of_theoretical"])
print("init", d["hip_first_use_init_s"], "load", d["setup_load_s"], d["setup_load_breakdown_ms"], d["default_table_build_breakdown_ms"])
print("table", d["direct_table_build_s"], d["direct_table_build_breakdown_ms"])
for k,v in d["configs"].items():
    if isinstance(v,dict): print(k, {kk:vv for kk,vv in v.items() if kk in ("value","ms_per_step","error")})
for k in ("default_engine","bucket_engine","host_abi"): print(k, round(d[k]["value"]))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["gpu_outputs_match_oracle"])
PY
