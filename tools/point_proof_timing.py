import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
ts.reserve(1024); ts.enable_direct_table(16)
for n in (1, 16, 256, 1024):
    data = B.synthetic_batch(0, n)
    zs = b"".join((1000 + i).to_bytes(32, "big") for i in range(n))
    K.compute_kzg_proof_batch(data, zs, ts)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); out = K.compute_kzg_proof_batch(data, zs, ts); best = min(best, time.perf_counter() - t)
    print("compute_kzg_proof_batch n=%d: %.2f ms (%.0f/s)" % (n, best * 1e3, n / best), flush=True)
