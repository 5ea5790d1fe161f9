import os, sys, time, statistics
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
n = 1024
data = B.synthetic_batch(0, n)
comms = b"".join(K.blob_to_kzg_commitment_batch(data, ts))
proofs = b"".join(K.compute_blob_kzg_proof_batch(data, comms, ts))
out=[]
for m in (128, 256, 512, 1024):
    bl, cm, pr = data[:m*B.BYTES_PER_BLOB], comms[:48*m], proofs[:48*m]
    K.verify_blob_kzg_proof_batch(bl, cm, pr, m, ts)
    tsx=[]
    for rep in range(9):
        t = time.perf_counter(); ok = K.verify_blob_kzg_proof_batch(bl, cm, pr, m, ts); tsx.append((time.perf_counter()-t)*1e3)
    out.append((m, round(statistics.median(tsx),2), round(min(tsx),2)))
print(os.environ.get("LWKZG_LIBRARY","")[-16:], out)
