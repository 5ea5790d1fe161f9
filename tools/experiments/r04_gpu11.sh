mkdir -p gpurun_out/r04k
./tools/gather_sectors_bin 96 > gpurun_out/r04k/gather_sectors.txt 2>&1
cat gpurun_out/r04k/gather_sectors.txt
cd /tmp && export TMPDIR=/tmp
for ctr in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $ctr | tr ' ' '_')
  rocprofv3 --pmc $ctr --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r04k/pmc_$tag -o run -f csv -- $GRAFT_REPO_ROOT/tools/gather_sectors_bin 96 > $GRAFT_REPO_ROOT/gpurun_out/r04k/pmc_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/r04k/pmc_*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(f)
    for k, d in sorted(agg.items()):
        print("  ", k, {c: sum(v) / len(v) for c, v in d.items()})
PY
