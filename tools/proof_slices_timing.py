import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
ts.reserve(256); ts.enable_direct_table(16)
n = int(sys.argv[1])
data = B.synthetic_batch(0, n)
cm = b"".join(K.blob_to_kzg_commitment_batch(data, ts))
for _ in range(3):
    t = time.perf_counter(); K.compute_blob_kzg_proof_batch(data, cm, ts); print("total %.2f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr)
