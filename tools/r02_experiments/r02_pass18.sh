#!/bin/bash
# the clock the chip really holds in the micro-benchmark loops: GRBM_GUI_ACTIVE per dispatch of tools/ubench_mad_bin
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/p18; mkdir -p $O
rocprofv3 --pmc GRBM_GUI_ACTIVE -d $O/pmc -o u --output-format csv -- $GRAFT_REPO_ROOT/tools/ubench_mad_bin > $O/run.log 2>&1
f=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" > $O/ubench_clock.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
g = collections.OrderedDict()
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]) if "Grid_Size" in r else 0)
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    g.setdefault(key, []).append((float(r["Counter_Value"]) / 8, dur))
for (k, grid), v in g.items():
    cyc = min(c for c, d in v); dur = min(d for c, d in v)
    waves_per_simd = grid / 256 / 256
    instr_per_simd = waves_per_simd * 16384 * 8
    print("%-28s waves/SIMD %d  %.3f ms  clock %.2f GHz  actual cycles per wave-instruction per SIMD %.2f  ns %.2f" % (k, waves_per_simd, dur * 1e3, cyc / dur / 1e9, cyc / instr_per_simd, dur * 1e9 / instr_per_simd))
PY
cat $O/ubench_clock.txt; head -3 "$f"
rm -rf $O/pmc
