#!/bin/bash
# Copies what tools/collect_profiles.sh left under gpurun_out/r01b into profiles/ (tracked), keeping only our kernels' rows.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r01b; P=profiles; TAG=${1:-r01}
for f in bench_line bench_line_bucket_path bench_line_ckzg_mode bench_line_blob_proof_b256 bench_line_blob_proof_b1024; do tail -1 $O/$f.json > $P/${TAG}_$f.json; done
cp $O/config_sweep_direct16.json $P/${TAG}_config_sweep_direct16.json; cp $O/config_sweep_bucket.json $P/${TAG}_config_sweep_bucket.json
grep -v "amdgpu.ids" $O/verify_timing.txt > $P/${TAG}_verify_timing.txt
python3 - "$TAG" <<'PY'
import csv, sys
tag = sys.argv[1]
rows = list(csv.reader(open('gpurun_out/r01b/kt/kt_kernel_stats.csv')))
out = [rows[0]] + [r for r in rows[1:] if 'lwk::' in r[0] or 'rocclr' in r[0]]
csv.writer(open('profiles/%s_bench_kernel_stats.csv' % tag, 'w'), quoting=csv.QUOTE_ALL).writerows(out)
for t in ('fetch', 'write'):
    rows = list(csv.reader(open('gpurun_out/r01b/%s/%s_counter_collection.csv' % (t, t))))
    keep = [rows[0]] + [r for r in rows[1:] if 'lwk::' in r[8]]
    csv.writer(open('profiles/%s_pmc_%s_size.csv' % (tag, t), 'w'), quoting=csv.QUOTE_ALL).writerows(keep)
PY
python3 tools/pmc_summary.py $O/fetch/fetch_counter_collection.csv $O/write/write_counter_collection.csv $TAG 1024 16 | grep direct_acc
{ echo "# default engine (bucket path)"; grep -v amdgpu $O/host_api_timing.txt; echo; echo "# LWKZG_DIRECT=16 (direct table)"; grep -v amdgpu $O/host_api_timing_direct.txt; } > $P/${TAG}_host_api_timing.txt
