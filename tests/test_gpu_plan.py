"""GPU: every schedule of csrc/plan.h's ProofSchedule forced once on the same 256 blobs (fresh processes: LWKZG_EXPERIMENTAL=1
LWKZG_PROOF_SCHEDULE=k), the schedule read back through lwkzg_last_proof_schedule, the proofs byte-equal to the default call's and to the
oracle's for three blobs -- among them a commitment at infinity in an odd encoding, which every schedule has to send through its
re-hash over the canonical bytes (/root/reference/src/lib.rs:361-404)."""
import json
import os
import subprocess
import sys

import pytest

import blobs as B
from conftest import ROOT, SETUP_PATH

pytestmark = pytest.mark.gpu

_WORKER = r"""
import sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, blobs as B, lambdaworks_kzg_amd as K
from lambdaworks_kzg_amd import capi
ts = K.TrustedSetup.from_file(%r)
n = 256
blobs = [B.synthetic_blob(91000 + i) for i in range(n)]
blobs[7] = bytes(B.BYTES_PER_BLOB)
data = b"".join(blobs)
comms = K.blob_to_kzg_commitment_batch(data, ts)
comms[7] = bytes([0xc0, 0x01]) + bytes(46)          # infinity with a stray bit: valid, not canonical
d_b = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
d_c = torch.frombuffer(bytearray(b"".join(comms)), dtype=torch.uint8).cuda()
d_o = torch.empty(48 * n, dtype=torch.uint8, device="cuda"); d_s = torch.zeros(n, dtype=torch.int32, device="cuda")
ts.reserve(n)
took = []
for rep in range(2):
    K.compute_blob_kzg_proof_batch_device(d_o.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), n, ts, None, d_s.data_ptr())
    torch.cuda.synchronize(); took.append(capi.last_proof_schedule())
print(json.dumps({"took": took, "status": int(d_s.abs().sum().item()), "proofs": bytes(d_o.cpu().numpy().tobytes()).hex()}))
"""


def _run(**env):
    code = _WORKER % (ROOT, os.path.join(ROOT, "tests", "golden"), SETUP_PATH)
    out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, **env), timeout=900).decode()
    return json.loads(out.strip().splitlines()[-1])


def test_every_schedule_once(K, gpu_setup, oracle, oracle_setup):
    default = _run()
    assert default["status"] == 0 and default["took"][1] in (2, 3)        # reserved + warm: a host-assisted mid-size schedule
    for k in range(5):
        r = _run(LWKZG_EXPERIMENTAL="1", LWKZG_PROOF_SCHEDULE=str(k))
        assert r["took"] == [k, k], (k, r["took"])
        assert r["status"] == 0 and r["proofs"] == default["proofs"], k
    proofs = bytes.fromhex(default["proofs"])
    for i in (0, 7, 255):
        blob = bytes(B.BYTES_PER_BLOB) if i == 7 else B.synthetic_blob(91000 + i)
        rc, c = oracle.blob_to_kzg_commitment(blob, oracle_setup, oracle.MODE_R)
        rc2, want = oracle.compute_blob_kzg_proof(blob, c, oracle_setup, oracle.MODE_R)
        assert rc == 0 and rc2 == 0 and proofs[48 * i:48 * i + 48] == want, i
