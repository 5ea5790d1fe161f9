#!/usr/bin/env python3
"""BASELINE configs[3] in its device-resident form, in a loop of its own (what profiles/r06_verify_b4096_device_timeline.txt is a
rocprofv3 --kernel-trace of): lwkzg_verify_blob_kzg_proof_batch_device on N blobs that are already in HBM.

    python tools/verify_device_loop.py [--n 4096] [--calls 6] [--host]   (--host: the host-pointer ABI on the same batch, where the
                                                                           hash runs on host threads and the GPU kernels run solo)
Prints one JSON line: median / min ms per call and the library's per-kernel averages (lwkzg_profile_*).
"""
import argparse, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
import blobs as B
import lambdaworks_kzg_amd as K
from lambdaworks_kzg_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--calls", type=int, default=6)
ap.add_argument("--host", action="store_true")
ap.add_argument("--tag", default="")
ap.add_argument("--no-profile", action="store_true", help="no hipEvent pairs around the launches (the timeline run: rocprofv3 times them)")
a = ap.parse_args()
dev = torch.device("cuda:0")
ts = K.TrustedSetup.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"))
n = a.n
h_blobs = B.synthetic_batch(9000, n)
h_comms = b"".join(K.blob_to_kzg_commitment_batch(h_blobs, ts))
h_proofs = b"".join(K.compute_blob_kzg_proof_batch(h_blobs, h_comms, ts))
to_dev = lambda b: torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).to(dev)
d_b, d_c, d_p = to_dev(h_blobs), to_dev(h_comms), to_dev(h_proofs)
stream = torch.cuda.current_stream(dev).cuda_stream


def call():
    if a.host:
        return K.verify_blob_kzg_proof_batch(h_blobs, h_comms, h_proofs, n, ts)
    return K.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, ts, stream)


assert call() and call()
torch.cuda.synchronize(dev)
if not a.no_profile:
    capi.profile_reset(); capi.profile_enable(True)
t = []
for _ in range(a.calls):
    t0 = time.perf_counter()
    assert call()
    t.append((time.perf_counter() - t0) * 1e3)
kern = {}
if not a.no_profile:
    capi.profile_enable(False)
    kern = {k: round(v["total_ms"] / max(1, v["launches"]), 4) for k, v in capi.profile_report().items()}
print(json.dumps({"tag": a.tag, "form": "host" if a.host else "device", "n": n, "calls": a.calls, "median_ms": round(statistics.median(t), 3),
                  "min_ms": round(min(t), 3), "blobs_per_s": round(n / statistics.median(t) * 1e3), "kernels_avg_ms": kern}))
