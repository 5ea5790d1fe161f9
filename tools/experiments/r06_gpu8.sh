#!/bin/bash
# r06 call 8: how long does the upload call of a staged slice take inside the library?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
LWKZG_TIMING=1 python - <<'PY' 2>&1 | tail -40
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
n = 4096
data = B.synthetic_batch(0, n)
for rep in range(3):
    t = time.perf_counter(); K.blob_to_kzg_commitment_batch(data, ts); print("call %d: %.2f ms" % (rep, (time.perf_counter() - t) * 1e3), flush=True)
import numpy as np, torch
arr = np.frombuffer(data, dtype=np.uint8).copy()
buf = bytes(arr)     # another 512 MiB object
for rep in range(2):
    t = time.perf_counter(); K.blob_to_kzg_commitment_batch(buf, ts); print("other buffer, call %d: %.2f ms" % (rep, (time.perf_counter() - t) * 1e3), flush=True)
PY
