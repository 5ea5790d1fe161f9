mkdir -p gpurun_out/r04n
for rep in 1 2; do for stage in 1 0; do
LWKZG_SORT_STAGE=$stage python bench.py --direct-bits 0 --no-config-legs --no-cpu-baseline --steps 10 > gpurun_out/r04n/bench_s$stage.json 2> gpurun_out/r04n/err.txt
cp bench_detail.json gpurun_out/r04n/detail_s$stage.json
python - <<PY
import json
d=json.load(open("gpurun_out/r04n/detail_s$stage.json"))
print("stage=$stage", round(d["value"]), {k: round(v["avg_ms"],3) for k,v in d["kernels"].items()})
PY
done; done
cd /tmp && export TMPDIR=/tmp
for stage in 1 0; do
LWKZG_SORT_STAGE=$stage rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04n/write_s$stage -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --direct-bits 0 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for stage in (1, 0):
    for f in glob.glob("gpurun_out/r04n/write_s%d/**/*counter_collection.csv" % stage, recursive=True):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_digit_sort" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
        print("stage=%d k_digit_sort WRITE_SIZE per launch: avg %.0f max %.0f (n=%d)" % (stage, sum(v)/len(v), max(v), len(v)))
PY
