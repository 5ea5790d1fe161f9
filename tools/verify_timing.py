"""Phase timing of verify_blob_kzg_proof_batch (LWKZG_TIMING=1 prints the library's own phase clock to stderr)."""
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
for m in (2, 4, 8, 16, 32, 64, 65, 128, 512, 1024):
    bl, cm, pr = data[:m * B.BYTES_PER_BLOB], comms[:48 * m], proofs[:48 * m]   # slicing copies: keep it out of the clock
    for rep in range(2):
        t = time.perf_counter()
        ok = K.verify_blob_kzg_proof_batch(bl, cm, pr, m, ts)
        print("n=%d ok=%s %.2f ms" % (m, ok, (time.perf_counter() - t) * 1e3), flush=True)
for rep in range(3):
    t = time.perf_counter(); ok = K.verify_blob_kzg_proof(data[:B.BYTES_PER_BLOB], comms[:48], proofs[:48], ts)
    print("verify_blob_kzg_proof ok=%s %.2f ms" % (ok, (time.perf_counter() - t) * 1e3), flush=True)
from lambdaworks_kzg_amd import capi
for m in (64,):
    capi.profile_reset(); capi.profile_enable(True)
    K.verify_blob_kzg_proof_batch(data[:m * B.BYTES_PER_BLOB], comms[:48 * m], proofs[:48 * m], m, ts)
    capi.profile_enable(False)
    print("kernels n=%d:" % m, {k: round(v["total_ms"], 3) for k, v in capi.profile_report().items()})
pr1, y1 = K.compute_kzg_proof(data[:B.BYTES_PER_BLOB], data[32:64], ts)
for rep in range(3):
    t = time.perf_counter(); ok = K.verify_kzg_proof(comms[:48], data[32:64], y1, pr1, ts)
    print("verify_kzg_proof ok=%s %.2f ms" % (ok, (time.perf_counter() - t) * 1e3), flush=True)
