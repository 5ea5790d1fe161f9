import sys, time, json, ctypes as C
sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import torch, blobs as B
import lambdaworks_kzg_amd.capi as K
ts = K.TrustedSetup.from_file("tests/golden/trusted_setup.txt")
ts.reserve(1024)
n = 1024
data = B.synthetic_batch(1000, n)
d = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
out = torch.empty(48*n, dtype=torch.uint8, device="cuda")
def run(reps=5):
    K.blob_to_kzg_commitment_batch_device(out.data_ptr(), d.data_ptr(), n, ts, None, None); torch.cuda.synchronize()
    t=time.time()
    for _ in range(reps): K.blob_to_kzg_commitment_batch_device(out.data_ptr(), d.data_ptr(), n, ts, None, None)
    torch.cuda.synchronize(); return (time.time()-t)/reps*1e3
base = run(); ref = bytes(out.cpu().numpy().tobytes())
print("bucket path ms/1024:", base, flush=True)
for bits in (14, 15, 16):
    t=time.time()
    try: ts.enable_direct_table(bits)
    except K.KzgError as e: print(bits, "failed", e, K.lib().lwkzg_last_error()); continue
    tb=time.time()-t
    K.lib().lwkzg_profile_enable(1); K.lib().lwkzg_profile_reset()
    ms = run()
    buf=C.create_string_buffer(1<<16); K.lib().lwkzg_profile_report(buf, len(buf)); K.lib().lwkzg_profile_enable(0)
    ok = bytes(out.cpu().numpy().tobytes()) == ref
    print(bits, "build s %.2f" % tb, "ms/1024 %.2f" % ms, "parity", ok, flush=True)
    print(buf.value.decode()[:600], flush=True)
    for nn in (1, 16, 256):
        K.blob_to_kzg_commitment_batch_device(out.data_ptr(), d.data_ptr(), nn, ts, None, None); torch.cuda.synchronize()
        t=time.time()
        for _ in range(10): K.blob_to_kzg_commitment_batch_device(out.data_ptr(), d.data_ptr(), nn, ts, None, None)
        torch.cuda.synchronize(); print("   n", nn, "ms %.3f" % ((time.time()-t)/10*1e3), bytes(out[:48*nn].cpu().numpy().tobytes())==ref[:48*nn], flush=True)
ts.enable_direct_table(0)
