cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_fold
rm -rf $O; mkdir -p $O
cd $R
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_CYCLES --output-format csv -d $O/sq -o sq -- $P > $O/sq.json 2> $O/sq.err
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- $P > $O/kt.json 2> $O/kt.err
python3 - <<'PY'
import csv,glob,collections,re
for f in glob.glob("gpurun_out/r03_fold/sq/*_counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m=re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
        if m and m.group(0) in ("k_direct_fold_lanes","k_direct_accumulate_asm","k_finalize_compress"):
            agg[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[m.group(0)]["dur_us"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for k,v in agg.items():
        print(k, {c:"%.4g"%(sum(x)/len(x)) for c,x in v.items()})
rows=list(csv.DictReader(open(glob.glob("gpurun_out/r03_fold/kt/*kernel_trace.csv")[0])))
rows=[r for r in rows if "lwk::" in r["Kernel_Name"]][-12:]
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    print(re.search(r"k_[a-z0-9_]+",r["Kernel_Name"]).group(0), "start %.1f us dur %.1f us"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3), r.get("Workgroup_Size_X"), r.get("Grid_Size_X"), r.get("LDS_Block_Size"), r.get("VGPR_Count"))
PY
