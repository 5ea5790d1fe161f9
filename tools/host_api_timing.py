"""Wall-clock of the host-pointer C ABI (PCIe included). LWKZG_DIRECT=16|15|14 enables the direct table first."""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
print('engine after load: direct_bits = %d' % ts.direct_table_bits())
if os.environ.get('LWKZG_DIRECT'):
    ts.reserve(256)
    t = time.perf_counter(); ts.enable_direct_table(int(os.environ['LWKZG_DIRECT']))
    print('direct table (%s bits) built in %.2f s' % (os.environ['LWKZG_DIRECT'], time.perf_counter() - t))
def best(fn, reps=3):
    out, ts_ = None, []
    for _ in range(reps):
        t = time.perf_counter(); out = fn(); ts_.append(time.perf_counter() - t)
    return out, min(ts_)


for n in (1, 16, 256, 512, 1024, 4096):
    data = B.synthetic_batch(0, n)
    K.blob_to_kzg_commitment_batch(data, ts)
    comms, tc = best(lambda: K.blob_to_kzg_commitment_batch(data, ts))
    cm = b"".join(comms)
    K.compute_blob_kzg_proof_batch(data, cm, ts)
    pr, tp = best(lambda: K.compute_blob_kzg_proof_batch(data, cm, ts))
    prj = b"".join(pr)
    ok, tv = best(lambda: K.verify_blob_kzg_proof_batch(data, cm, prj, n, ts))
    zs = data[:32 * n]
    K.compute_kzg_proof_batch(data, zs, ts)
    _, tz = best(lambda: K.compute_kzg_proof_batch(data, zs, ts))
    print("host API n=%d (best of 3): commit %.2f ms (%.0f/s)  blob_proof %.2f ms (%.0f/s)  point_proof %.2f ms (%.0f/s)  verify_batch %.2f ms ok=%s" % (n, tc*1e3, n/tc, tp*1e3, n/tp, tz*1e3, n/tz, tv*1e3, ok))
