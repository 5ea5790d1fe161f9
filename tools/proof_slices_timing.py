"""Host-pointer proof batches: wall clock and per-kernel totals (library profile), LWKZG_TIMING=1 for the slice clock."""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
from lambdaworks_kzg_amd import capi
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
ts.reserve(256); ts.enable_direct_table(16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
data = B.synthetic_batch(0, n)
cm = b"".join(K.blob_to_kzg_commitment_batch(data, ts))
zs = data[:32 * n]
ops = {"commit": lambda: K.blob_to_kzg_commitment_batch(data, ts),
       "blob_proof": lambda: K.compute_blob_kzg_proof_batch(data, cm, ts),
       "point_proof": lambda: K.compute_kzg_proof_batch(data, zs, ts)}
for name, fn in ops.items():
    fn()
for rep in range(3):
    for name, fn in ops.items():
        t = time.perf_counter(); fn(); print("%s n=%d: %.2f ms" % (name, n, (time.perf_counter() - t) * 1e3))
for name, fn in ops.items():
    capi.profile_reset(); capi.profile_enable(True); fn(); capi.profile_enable(False)
    print(name, {k: round(v["total_ms"], 2) for k, v in capi.profile_report().items()})
