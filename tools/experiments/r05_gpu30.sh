#!/bin/bash
# c-kzg device-resident verification evaluates in one launch from the blobs (range check inside): parity, then the leg in c-kzg mode
timeout 1500 python -m pytest tests/test_gpu_verify_device.py tests/test_gpu_lagrange.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q -m gpu -k "verify or ckzg" 2>&1 | tail -2
python - <<'PY'
import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests/golden')
import torch, blobs as B, lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt'); ts.reserve(1024); ts.set_mode(K.MODE_CKZG); ts.enable_direct_table(16)
n = 4096
data = B.synthetic_batch(9000, n, big_endian=False)
cj = b"".join(K.blob_to_kzg_commitment_batch(data, ts)); pj = b"".join(K.compute_blob_kzg_proof_batch(data, cj, ts))
dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
d_b, d_c, d_p = dev(data), dev(cj), dev(pj)
for _ in range(2): assert K.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, ts, None)
t0 = time.perf_counter()
for _ in range(5): assert K.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, ts, None)
el = (time.perf_counter() - t0) / 5
print("c-kzg device verify 4096: %.2f ms, %.0f blobs/s" % (el * 1e3, n / el))
PY
