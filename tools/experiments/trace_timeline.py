"""kernel + memory-copy timeline of the LAST call in a rocprofv3 --kernel-trace --memory-copy-trace directory (calls are separated by gaps > 3 ms);
a third argument N takes the last N events instead (calls that follow each other without a gap)."""
import csv, glob, sys
O, out_path = sys.argv[1], sys.argv[2]
ev = []
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-44:]))
for f in glob.glob(O + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M " + r.get("Direction", "?")))
ev.sort()
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 60):i]) > 3_000_000:
        cut = i
last = ev[cut:]
if len(sys.argv) > 3:
    last = ev[-int(sys.argv[3]):]
t0 = last[0][0]
with open(out_path, "w") as out:
    for s, e, n in last:
        out.write("%9.3f %9.3f %8.3f  %s\n" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
print("events", len(last), "span ms %.3f" % ((max(e[1] for e in last) - t0) / 1e6))
