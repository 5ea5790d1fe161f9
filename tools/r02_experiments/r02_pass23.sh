#!/bin/bash
# Why is an MSM launch slower after a quiet phase? Cycles (GRBM_GUI_ACTIVE) and wall time of k_direct_accumulate
# back to back, after 20 ms of idleness, and after the hash phase of a proof call: same cycles at a lower clock, or more cycles?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/p23; mkdir -p $O
run() {  # name, bench args
  name=$1; shift
  rocprofv3 --pmc GRBM_GUI_ACTIVE -d $O/$name -o p --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs "$@" > $O/$name.log 2>&1
  python3 - "$name" $(find $O/$name -name "*counter_collection.csv" | head -1) >> $O/summary.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "k_direct_accumulate" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
rows = rows[2:]  # warm-up launches
cyc = [float(r["Counter_Value"]) / 8 for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 for r in rows]
print("%-22s launches %2d  avg %.3f ms  %.3e cycles  clock %.3f GHz   (min %.3f ms / max %.3f ms)" % (sys.argv[1], len(rows), sum(dur) / len(dur) * 1e3, sum(cyc) / len(cyc), sum(cyc) / sum(dur) / 1e9, min(dur) * 1e3, max(dur) * 1e3))
PY
  rm -rf $O/$name
}
run back_to_back
run idle_1ms --idle-ms 1
run idle_20ms --idle-ms 20
run after_hash --op blob_proof --batch 1024
cat $O/summary.txt
