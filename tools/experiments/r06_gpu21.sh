#!/bin/bash
# r06 call 21: where the host-pointer verification of 4096 blobs spends its 16-18 ms: the library's phase clock and a kernel + memcpy timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
LWKZG_TIMING=1 python tools/verify_device_loop.py --n 4096 --calls 6 --host --tag "host" 2> $O/g21_timing.txt | tail -1 | cut -c1-150
grep -v "Miller\|verification:" $O/g21_timing.txt | tail -12
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt_host21 -o kt -- python3 tools/verify_device_loop.py --n 4096 --calls 3 --host --no-profile > $O/g21_kt_out.txt 2> $O/g21_kt_err.txt
ls $O/kt_host21/* | head
python3 - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06/kt_host21"
kt = glob.glob(O + "/**/*kernel_trace.csv", recursive=True)
mc = glob.glob(O + "/**/*memory_copy_trace.csv", recursive=True)
ev = []
for f in kt:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-40:]))
for f in mc:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M %s %s B" % (r.get("Direction", "?"), r.get("Size", "?"))))
ev.sort()
# the last call: events after the last gap > 5 ms
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 50):i]) > 3_000_000: cut = i
last = ev[cut:]
t0 = last[0][0]
out = open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06/g21_host_timeline.txt", "w")
for s, e, n in last:
    out.write("%9.3f %9.3f %8.3f  %s\n" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
out.close()
print("events", len(last), "span ms", (max(e[1] for e in last) - t0) / 1e6)
PY
rm -rf $O/kt_host21
head -60 $O/g21_host_timeline.txt
