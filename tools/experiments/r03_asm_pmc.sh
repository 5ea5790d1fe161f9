# issue-side counters of the hand-scheduled kernel and of the compiler-scheduled arm (LWKZG_DIRECT_ASM=0): same passes as tools/collect_profiles.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pmc
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_setups.py -m gpu -q -x -k "direct_commitments or adversarial or tau2_load" > $O/parity.txt 2>&1; tail -2 $O/parity.txt
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs"
for arm in asm cpp; do
  if [ $arm = cpp ]; then export LWKZG_DIRECT_ASM=0; else unset LWKZG_DIRECT_ASM; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/${arm}_sq1 -o sq -- $P > $O/${arm}_sq1.json 2> $O/${arm}_sq1.err
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/${arm}_sq2 -o sq -- $P > $O/${arm}_sq2.json 2> $O/${arm}_sq2.err
  rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/${arm}_grbm -o grbm -- $P > $O/${arm}_grbm.json 2> $O/${arm}_grbm.err
done
unset LWKZG_DIRECT_ASM
python3 tools/pmc_issue_summary.py k_direct_accumulate_asm $O/issue_asm.json $O/asm_sq1/sq_counter_collection.csv $O/asm_sq2/sq_counter_collection.csv $O/asm_grbm/grbm_counter_collection.csv > /dev/null
python3 tools/pmc_issue_summary.py k_direct_accumulate $O/issue_cpp.json $O/cpp_sq1/sq_counter_collection.csv $O/cpp_sq2/sq_counter_collection.csv $O/cpp_grbm/grbm_counter_collection.csv > /dev/null
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config-legs > $O/bench_asm.json 2> $O/bench_asm.err
python - <<'PY'
import json
for a in ("asm","cpp"):
    d=json.load(open("gpurun_out/r03_pmc/issue_%s.json"%a))
    print(a, {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k not in ("counters_per_launch","launches_seen")}, "INSTS_VALU", d["counters_per_launch"].get("SQ_INSTS_VALU"))
d=json.loads(open("gpurun_out/r03_pmc/bench_asm.json").read().strip().splitlines()[-1])
print(round(d["value"]), d["ms_per_step"], {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()}, round(d["default_engine"]["value"]))
PY
