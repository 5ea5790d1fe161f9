# scalar-memory counters of the headline kernel, and the wait counters of the confined-gather build (one gpurun call)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mem_pmc2
rm -rf $O; mkdir -p $O
cd $R
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs"
i=0
for set in "SmemLatency" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM" "SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- $P > $O/p$i.json 2> $O/p$i.err
done
for set in "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU" "VmemLatency" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  LWKZG_LIBRARY=$R/lambdaworks_kzg_amd/lib_confine/liblambdaworks_kzg.so rocprofv3 --pmc $set --output-format csv -d $O/c$i -o p -- $P > $O/c$i.json 2> $O/c$i.err
done
python3 - <<'PY'
import csv, glob, collections
for tag in ("p", "c"):
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob("gpurun_out/mem_pmc2/%s*/p_counter_collection.csv" % tag)):
        for r in csv.DictReader(open(f)):
            if "k_direct_accumulate_asm" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("shipped build" if tag == "p" else "confined gathers (1 MB)")
    for k, v in acc.items():
        print("  %-40s %16.4g  (launches %d)" % (k, sum(v) / len(v), len(v)))
PY
