"""Host-pointer ABI on batches longer than one workspace chunk (LWKZG_DIRECT=16|15|14 enables the direct table first)."""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
if os.environ.get('LWKZG_DIRECT'):
    ts.reserve(1024); ts.enable_direct_table(int(os.environ['LWKZG_DIRECT']))
base = B.synthetic_batch(0, 1024)
ref = K.blob_to_kzg_commitment_batch(base, ts)
for n in (1024, 2048, 4096):
    data = base * (n // 1024)
    best = 1e9
    for rep in range(3):
        t = time.perf_counter(); out = K.blob_to_kzg_commitment_batch(data, ts); best = min(best, time.perf_counter() - t)
    assert out == ref * (n // 1024)
    print("host commit n=%d: %.2f ms (%.0f/s)" % (n, best * 1e3, n / best), flush=True)
