"""How cold is the host-assisted challenge path after a quiet spell, and how long does it take to warm? 256 device-resident blobs per
compute_blob_kzg_proof call (the mid-size path: hashing on the host threads, pipelined with the copy out), calls issued back to back
after an idle gap; per-call wall clock of the first twelve calls after gaps of 0 .. 1000 ms, against LWKZG_MID_PROOF_HOST=0 (the GPU
hash kernel) run as a child with the same schedule."""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, '.')
sys.path.insert(0, 'tests/golden')


def run():
    import torch
    import blobs as B
    import lambdaworks_kzg_amd as K
    from lambdaworks_kzg_amd import capi
    ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
    n = 256
    ts.reserve(n)
    data = B.synthetic_batch(7000, n)
    d_b = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    d_c = torch.empty(48 * n, dtype=torch.uint8, device='cuda')
    d_p = torch.empty(48 * n, dtype=torch.uint8, device='cuda')
    d_s = torch.zeros(n, dtype=torch.int32, device='cuda')
    capi.blob_to_kzg_commitment_batch_device(d_c.data_ptr(), d_b.data_ptr(), n, ts)
    torch.cuda.synchronize()

    def call():
        t0 = time.perf_counter()
        capi.compute_blob_kzg_proof_batch_device(d_p.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), n, ts, None, d_s.data_ptr())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    for _ in range(30):
        call()
    out = {}
    for gap in (0, 2, 5, 10, 20, 50, 100, 300, 1000):
        rows = []
        for rep in range(3):
            for _ in range(30):
                call()
            time.sleep(gap * 1e-3)
            rows.append([round(call(), 2) for _ in range(12)])
        out[gap] = rows
    print(json.dumps({"mid_proof_host": os.environ.get("LWKZG_MID_PROOF_HOST", "default"), "per_call_ms_after_gap_ms": out}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        run()
    else:
        for env in ({}, {"LWKZG_MID_PROOF_HOST": "0"}):
            e = dict(os.environ)
            e.update(env)
            print(subprocess.check_output([sys.executable, __file__, "child"], env=e, text=True).strip().split("\n")[-1])
