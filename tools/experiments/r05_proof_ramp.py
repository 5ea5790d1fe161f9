"""How long does a burst of 256-blob device-resident proof calls take to reach its steady rate? Calls issued back to back (no
synchronisation between them, as bench.py's legs do), wall clock per group of four calls, from a cold start (the process has done
commitments only) -- and again after one second of idleness. 16-bit table."""
import json, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import torch
import blobs as B
import lambdaworks_kzg_amd as K
from lambdaworks_kzg_amd import capi
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
ts.reserve(1024)
ts.enable_direct_table(int(sys.argv[1]) if len(sys.argv) > 1 else 16)
n = 256
d_b = torch.frombuffer(bytearray(B.synthetic_batch(7000, n)), dtype=torch.uint8).cuda()
d_c = torch.empty(48 * n, dtype=torch.uint8, device='cuda')
d_p = torch.empty(48 * n, dtype=torch.uint8, device='cuda')
d_s = torch.zeros(n, dtype=torch.int32, device='cuda')
stream = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    capi.blob_to_kzg_commitment_batch_device(d_c.data_ptr(), d_b.data_ptr(), n, ts, stream, d_s.data_ptr())
torch.cuda.synchronize()
def burst(groups, per=4):
    out = []
    for _ in range(groups):
        t0 = time.perf_counter()
        for _ in range(per):
            capi.compute_blob_kzg_proof_batch_device(d_p.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), n, ts, stream, d_s.data_ptr())
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) * 1e3 / per, 3))
    return out
res = {"ms_per_call_by_group_of_4_cold_start": burst(16)}
time.sleep(1.0)
res["after_1s_idle"] = burst(8)
t0 = time.perf_counter()
for _ in range(40):
    capi.compute_blob_kzg_proof_batch_device(d_p.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), n, ts, stream, d_s.data_ptr())
torch.cuda.synchronize()
res["forty_calls_no_sync_ms_per_call"] = round((time.perf_counter() - t0) * 1e3 / 40, 3)
capi.profile_reset(); capi.profile_enable(True)
t0 = time.perf_counter()
for _ in range(40):
    capi.compute_blob_kzg_proof_batch_device(d_p.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), n, ts, stream, d_s.data_ptr())
torch.cuda.synchronize()
res["forty_calls_with_kernel_events_ms_per_call"] = round((time.perf_counter() - t0) * 1e3 / 40, 3)
capi.profile_enable(False)
print(json.dumps(res))
