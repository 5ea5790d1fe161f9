import os, sys, time
os.environ["LWKZG_TIMING"] = "1"
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
n = 1024
data = B.synthetic_batch(0, n)
comms = b"".join(K.blob_to_kzg_commitment_batch(data, ts))
proofs = b"".join(K.compute_blob_kzg_proof_batch(data, comms, ts))
for m in (2048, 4096):
    reps = m // n
    bl, cm, pr = data * reps, comms * reps, proofs * reps
    for rep in range(3):
        t = time.perf_counter(); ok = K.verify_blob_kzg_proof_batch(bl, cm, pr, m, ts)
        print("n=%d ok=%s %.2f ms" % (m, ok, (time.perf_counter() - t) * 1e3), flush=True)
