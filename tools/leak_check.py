import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests/golden')
import torch, blobs as B, lambdaworks_kzg_amd as K
torch.cuda.init()
f0 = torch.cuda.mem_get_info()[0]
for i in range(12):
    ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
    K.blob_to_kzg_commitment(B.synthetic_blob(1), ts)
    ts.free()
    torch.cuda.synchronize()
    print(i, (f0 - torch.cuda.mem_get_info()[0]) >> 20, "MiB below start", flush=True)
