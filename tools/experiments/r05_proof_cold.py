"""What do the first calls of a burst of 256-blob device-resident proof calls cost, one by one? Per-call wall clock (synchronised) of the
first eight calls in a fresh process (after commitments only), and again after three seconds of idleness (host threads cold by the
library's rule, LWKZG_HOST_WARM_MS = 2000). 16-bit table."""
import json, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import torch
import blobs as B
import lambdaworks_kzg_amd as K
from lambdaworks_kzg_amd import capi
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
ts.reserve(1024)
ts.enable_direct_table(16)
n = 256
d_b = torch.frombuffer(bytearray(B.synthetic_batch(7000, n)), dtype=torch.uint8).cuda()
d_c = torch.empty(48 * n, dtype=torch.uint8, device='cuda')
d_p = torch.empty(48 * n, dtype=torch.uint8, device='cuda')
d_s = torch.zeros(n, dtype=torch.int32, device='cuda')
stream = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    capi.blob_to_kzg_commitment_batch_device(d_c.data_ptr(), d_b.data_ptr(), n, ts, stream, d_s.data_ptr())
torch.cuda.synchronize()
time.sleep(3.0)
def calls(k):
    out = []
    for _ in range(k):
        t0 = time.perf_counter()
        capi.compute_blob_kzg_proof_batch_device(d_p.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), n, ts, stream, d_s.data_ptr())
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) * 1e3, 3))
    return out
res = {"first_calls_of_the_process_ms": calls(8)}
time.sleep(3.0)
res["after_3s_idle_ms"] = calls(8)
time.sleep(3.0)
t0 = time.perf_counter()
for _ in range(3):
    capi.compute_blob_kzg_proof_batch_device(d_p.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), n, ts, stream, d_s.data_ptr())
torch.cuda.synchronize()
res["three_calls_no_sync_after_3s_idle_ms_per_call"] = round((time.perf_counter() - t0) * 1e3 / 3, 3)
print(json.dumps(res))
