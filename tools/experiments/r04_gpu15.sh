mkdir -p gpurun_out/r04o
timeout 1200 python -m pytest tests/test_gpu_proof_parity.py tests/test_gpu_lagrange.py -m gpu -x -q 2>&1 | tail -4
for mid in 384 0 512 1024; do
LWKZG_MID_PROOF_HOST=$mid python - <<'PY'
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import torch, blobs as B
import lambdaworks_kzg_amd as K
from lambdaworks_kzg_amd import capi
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
ts.enable_direct_table(16)
out = []
for n in (128, 192, 256, 384, 512, 768):
    data = B.synthetic_batch(0, n)
    d_b = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    d_c = torch.empty(48 * n, dtype=torch.uint8, device='cuda'); d_o = torch.empty(48 * n, dtype=torch.uint8, device='cuda')
    K.blob_to_kzg_commitment_batch_device(d_c.data_ptr(), d_b.data_ptr(), n, ts)
    torch.cuda.synchronize()
    for _ in range(3): K.compute_blob_kzg_proof_batch_device(d_o.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), n, ts)
    torch.cuda.synchronize()
    t = []
    for _ in range(10):
        t0 = time.perf_counter(); K.compute_blob_kzg_proof_batch_device(d_o.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), n, ts); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    t.sort(); out.append((n, round(t[5] * 1e3, 3)))
print("mid limit", os.environ["LWKZG_MID_PROOF_HOST"], out)
PY
done
python bench.py --no-cpu-baseline > gpurun_out/r04o/line.json 2> gpurun_out/r04o/err.txt
python - <<'PY'
import json
l=json.load(open("gpurun_out/r04o/line.json"))
print(l["value"]); 
for k in ("blob_proof_b256","blob_proof_b256_two_streams","commit_prove_b256"): print(k, l["configs"][k])
PY
