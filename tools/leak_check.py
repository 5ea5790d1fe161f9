"""Load / use / free a settings object twelve times in one process and print how far the device's free memory sits below where it started:
a plateau (the runtime's own pools), not a slope. r06: every second cycle also runs a long host-pointer verification (1100 blobs: the
device buffer of verify_prepare_staged, the verify scratch, the long-call workspace), so that what those allocate is inside the cycle."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import torch, blobs as B, lambdaworks_kzg_amd as K
torch.cuda.init()
f0 = torch.cuda.mem_get_info()[0]
data = B.synthetic_batch(4242, 1100)
comms = proofs = None
for i in range(12):
    ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
    K.blob_to_kzg_commitment(B.synthetic_blob(1), ts)
    if i % 2 == 1:
        if comms is None:
            comms = b"".join(K.blob_to_kzg_commitment_batch(data, ts))
            proofs = b"".join(K.compute_blob_kzg_proof_batch(data, comms, ts))
        assert K.verify_blob_kzg_proof_batch(data, comms, proofs, 1100, ts) is True
    ts.free()
    torch.cuda.synchronize()
    print(i, (f0 - torch.cuda.mem_get_info()[0]) >> 20, "MiB below start", flush=True)
