#!/bin/bash
# cooperative kernel with four-wave workgroups and row prefetch: single-blob timing over rpq, small-batch sweep
for rpq in 2 4 8; do
  echo "== rpq $rpq 16-bit"; LWKZG_COOP_RPQ=$rpq python tools/single_blob_timing.py 2>&1 | head -2
  echo "== rpq $rpq 13-bit"; LWKZG_DIRECT=13 LWKZG_COOP_RPQ=$rpq python tools/single_blob_timing.py 2>&1 | head -2
done
python - <<'PY'
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
from lambdaworks_kzg_amd import capi
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
ts.reserve(256)
for n in (1, 2, 4, 8, 16, 32):
    data = b"".join(B.synthetic_blob(i) for i in range(n))
    K.blob_to_kzg_commitment_batch(data, ts)
    t = []
    for _ in range(15):
        t0 = time.perf_counter(); K.blob_to_kzg_commitment_batch(data, ts); t.append((time.perf_counter() - t0) * 1e3)
    capi.profile_reset(); capi.profile_enable(True); K.blob_to_kzg_commitment_batch(data, ts); capi.profile_enable(False)
    print("batch %d (default engine): min %.3f ms median %.3f; kernels %s" % (n, min(t), sorted(t)[7], {k: round(v["total_ms"], 3) for k, v in capi.profile_report().items()}))
PY
