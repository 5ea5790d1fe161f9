#!/bin/bash
# r06 call 10: per-kernel times of a staged 4096-blob host-pointer commitment call against the device-resident call
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
python - <<'PY' 2>&1 | tail -20
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch
import blobs as B
import lambdaworks_kzg_amd as K
from lambdaworks_kzg_amd import capi
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
n = 4096
data = B.synthetic_batch(0, n)
K.blob_to_kzg_commitment_batch(data, ts); K.blob_to_kzg_commitment_batch(data, ts)
capi.profile_reset(); capi.profile_enable(True)
t = time.perf_counter(); K.blob_to_kzg_commitment_batch(data, ts); el = (time.perf_counter() - t) * 1e3
capi.profile_enable(False)
print("host call %.2f ms" % el, {k: (v["launches"], round(v["total_ms"], 3)) for k, v in capi.profile_report().items()})
d = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda(); o = torch.empty(48 * n, dtype=torch.uint8, device="cuda"); s = torch.zeros(n, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(2): K.blob_to_kzg_commitment_batch_device(o.data_ptr(), d.data_ptr(), n, ts, st, s.data_ptr())
torch.cuda.synchronize()
capi.profile_reset(); capi.profile_enable(True)
t = time.perf_counter(); K.blob_to_kzg_commitment_batch_device(o.data_ptr(), d.data_ptr(), n, ts, st, s.data_ptr()); torch.cuda.synchronize(); el = (time.perf_counter() - t) * 1e3
capi.profile_enable(False)
print("device call %.2f ms" % el, {k: (v["launches"], round(v["total_ms"], 3)) for k, v in capi.profile_report().items()})
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt_host4096 -o kt -- python3 - <<'PY' > /dev/null 2>&1
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
data = B.synthetic_batch(0, 4096)
for _ in range(3): K.blob_to_kzg_commitment_batch(data, ts)
PY
ls $O/kt_host4096/ | head; python tools/timeline.py $(ls $O/kt_host4096/*kernel_trace.csv | head -1) 30 | tail -32
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.expandvars("$GRAFT_REPO_ROOT/gpurun_out/r06/kt_host4096/*memory_copy_trace.csv"))
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows = [r for r in rows if int(r["Bytes"] if "Bytes" in r else 0) > 1000000][-6:]
    for r in rows: print({k: r[k] for k in r if k in ("Direction", "Bytes", "Start_Timestamp", "End_Timestamp")}, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms")
PY
