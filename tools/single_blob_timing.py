"""One blob through the reference's own symbols: wall clock and the library's per-kernel clock (direct table, 16 bits)."""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
from lambdaworks_kzg_amd import capi
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
bits = int(os.environ.get('LWKZG_DIRECT', '16'))
if bits:
    ts.reserve(256); ts.enable_direct_table(bits)
blob = B.synthetic_blob(1)
c = K.blob_to_kzg_commitment(blob, ts)
p = K.compute_blob_kzg_proof(blob, c, ts)
z = blob[32:64]
ops = {"blob_to_kzg_commitment": lambda: K.blob_to_kzg_commitment(blob, ts),
       "compute_blob_kzg_proof": lambda: K.compute_blob_kzg_proof(blob, c, ts),
       "compute_kzg_proof": lambda: K.compute_kzg_proof(blob, z, ts),
       "verify_blob_kzg_proof": lambda: K.verify_blob_kzg_proof(blob, c, p, ts)}
for name, fn in ops.items():
    fn(); t = []
    for _ in range(20):
        t0 = time.perf_counter(); fn(); t.append((time.perf_counter() - t0) * 1e3)
    capi.profile_reset(); capi.profile_enable(True); fn(); capi.profile_enable(False)
    print("%s: min %.3f ms median %.3f ms; kernels %s" % (name, min(t), sorted(t)[10], {k: round(v["total_ms"], 3) for k, v in capi.profile_report().items()}))
