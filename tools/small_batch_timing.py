"""Host-pointer commitment batches of 1 .. 64 blobs on the engine a plain load selects: wall clock per call and the library's per-kernel clock
(where the cooperative kernel hands over to the throughput kernels). `LWKZG_COOP=0 python tools/small_batch_timing.py` is the A/B arm."""
import sys
import time
sys.path.insert(0, '.')
sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
from lambdaworks_kzg_amd import capi

ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
ts.reserve(256)
print("engine: direct_bits = %d" % ts.direct_table_bits())
for n in (1, 2, 3, 4, 6, 8, 12, 16, 32, 64):
    data = b"".join(B.synthetic_blob(i) for i in range(n))
    K.blob_to_kzg_commitment_batch(data, ts)
    t = []
    for _ in range(21):
        t0 = time.perf_counter()
        K.blob_to_kzg_commitment_batch(data, ts)
        t.append((time.perf_counter() - t0) * 1e3)
    capi.profile_reset()
    capi.profile_enable(True)
    K.blob_to_kzg_commitment_batch(data, ts)
    capi.profile_enable(False)
    t.sort()
    print("batch %2d: min %.3f ms median %.3f ms (%.0f blobs/s); kernels %s" % (n, t[0], t[10], n / t[10] * 1e3, {k: round(v["total_ms"], 3) for k, v in capi.profile_report().items()}))
