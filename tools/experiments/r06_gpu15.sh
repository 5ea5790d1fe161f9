#!/bin/bash
# r06 call 15: what are the 0.17 ms of ONE blob_to_kzg_commitment? kernels and copies on one time axis
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
cat > /tmp/one_blob.py <<'PY'
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
blob = B.synthetic_blob(1)
c = K.blob_to_kzg_commitment(blob, ts)
for _ in range(30): K.blob_to_kzg_commitment(blob, ts)
t = time.perf_counter()
for _ in range(200): K.blob_to_kzg_commitment(blob, ts)
print("commit: %.4f ms per call" % ((time.perf_counter() - t) / 200 * 1e3))
PY
python /tmp/one_blob.py
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt_one -o kt -- python3 /tmp/one_blob.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
O = os.path.expandvars("$GRAFT_REPO_ROOT/gpurun_out/r06/kt_one")
ev = []
for r in csv.DictReader(open(glob.glob(O + "/*kernel_trace.csv")[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
mc = glob.glob(O + "/*memory_copy_trace.csv")
if mc:
    for r in csv.DictReader(open(mc[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", "?"), r.get("Bytes", "?"))))
ev.sort()
ev = ev[-40:]
t0 = ev[0][0]
for s, e, n in ev: print("%9.1f %9.1f %7.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
