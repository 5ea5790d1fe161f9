#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r05/trace_proof256 -o t -- python3 bench.py --op blob_proof --batch 256 --steps 6 --warmup 3 --no-cpu-baseline --no-extra-legs > gpurun_out/r05/trace_proof256_line.json 2> gpurun_out/r05/trace_proof256_err.txt
ls -la gpurun_out/r05/trace_proof256/*/ | head
python - <<'PY'
import csv, glob
kt = glob.glob("gpurun_out/r05/trace_proof256/*/*kernel_trace.csv")[0]
mc = glob.glob("gpurun_out/r05/trace_proof256/*/*memory_copy_trace.csv")[0]
ev = []
for r in csv.DictReader(open(kt)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:], "q%s" % r.get("Queue_Id", "")))
for r in csv.DictReader(open(mc)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", ""), r.get("Bytes", r.get("Size", ""))), ""))
ev.sort()
# the last call: find the last k_finalize_compress and go back to the previous one
fins = [i for i, e in enumerate(ev) if "finalize_compress" in e[2]]
lo, hi = fins[-2] + 1, fins[-1] + 1
t0 = ev[lo][0]
for s, e, name, q in ev[lo:hi]:
    print("%8.3f %8.3f  %7.3f ms  %s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, name, q))
PY
