"""load / commit (reference AND c-kzg mode: the Lagrange form with its second table) / small and mid-size proof calls (pinned staging) /
free, twelve times in one process: device memory must come back every time (round 4's new allocations: lag.points, lag.table, the second
direct table, the pinned buffers of the host-assisted challenge)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import torch, blobs as B, lambdaworks_kzg_amd as K
torch.cuda.init()
f0 = torch.cuda.mem_get_info()[0]
n = 200
be, le = B.synthetic_batch(1, n), B.synthetic_batch(1, n, big_endian=False)
d_be = torch.frombuffer(bytearray(be), dtype=torch.uint8).cuda()
d_le = torch.frombuffer(bytearray(le), dtype=torch.uint8).cuda()
d_c = torch.empty(48 * n, dtype=torch.uint8, device='cuda'); d_o = torch.empty(48 * n, dtype=torch.uint8, device='cuda')
f1 = torch.cuda.mem_get_info()[0]
for i in range(12):
    ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
    K.blob_to_kzg_commitment(be[:131072], ts)
    ts.set_mode(K.MODE_CKZG)
    forms = ts.direct_table_forms()
    K.blob_to_kzg_commitment_batch_device(d_c.data_ptr(), d_le.data_ptr(), n, ts)
    K.compute_blob_kzg_proof_batch_device(d_o.data_ptr(), d_le.data_ptr(), d_c.data_ptr(), n, ts)       # mid-size path
    K.compute_blob_kzg_proof_batch_device(d_o.data_ptr(), d_le.data_ptr(), d_c.data_ptr(), 7, ts)       # small path
    K.commit_and_prove_batch_device(d_c.data_ptr(), d_o.data_ptr(), d_le.data_ptr(), n, ts)
    torch.cuda.synchronize()
    ts.free()
    torch.cuda.synchronize()
    print(i, "forms", forms, (f1 - torch.cuda.mem_get_info()[0]) >> 20, "MiB below the level before the first load", flush=True)
