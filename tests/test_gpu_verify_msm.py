"""GPU: the verification's three linear combinations as the latency-shaped bucket MSM of csrc/vmsm.hip (r06; the reference's
g1_lincomb x 3, /root/reference/src/lib.rs:679-685) against its two other arms, in fresh processes: the scan fallback of a bucket whose
list overflows (LWKZG_VMSM_LIST_CAP=1: every lane takes it) and r05's per-point multiples + Straus pieces (LWKZG_VERIFY_MSM=0). The
partial sums are affine points: all arms must agree BYTE FOR BYTE, per shard, in both semantics, host-pointer and device-resident. The
2600-blob case also holds the two forms of a host-pointer shard longer than one chunk against each other (LWKZG_HOST_STAGE)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(mode, **env):
    e = dict(os.environ, LWKZG_EXPERIMENTAL="1", **env)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tests", "verify_arm_worker.py"), mode], env=e).decode()
    return json.loads(out.strip().splitlines()[-1])


@pytest.mark.parametrize("mode", ["reference", "ckzg"])
def test_all_arms_of_the_linear_combinations_agree(mode):
    shipped = _run(mode)
    for n, r in shipped.items():
        assert r["ok"] is True and r["ok_sharded"] is True, n
        assert r["ok_swapped"] in (None, False), n
        assert r["partial_device_form"] == r["partials"][0], n      # same shard, device-resident inputs
    scan = _run(mode, LWKZG_VMSM_LIST_CAP="1")
    r05 = _run(mode, LWKZG_VERIFY_MSM="0")
    assert scan == shipped
    assert r05 == shipped
    sliced = _run(mode, LWKZG_HOST_STAGE="0")    # long host-pointer shards: r05's slices (every blob hashed on the host threads) instead of the staged form
    assert sliced == shipped


def test_experiment_knobs_need_the_switch():
    """without LWKZG_EXPERIMENTAL=1 an experiment knob is ignored (knobs.h): the arm variable alone changes nothing, and the library says so"""
    e = dict(os.environ, LWKZG_VERIFY_MSM="0", LWKZG_VERBOSE="1")
    e.pop("LWKZG_EXPERIMENTAL", None)
    p = subprocess.run([sys.executable, "-c",
                        "import sys, json; sys.path.insert(0, %r); import lambdaworks_kzg_amd as K; print(json.dumps(K.knob_report()))" % ROOT],
                       env=e, capture_output=True, text=True, check=True)
    rep = json.loads(p.stdout.strip().splitlines()[-1])
    assert rep["experimental"] is False and rep["verify_msm"] == 1
    assert "LWKZG_VERIFY_MSM ignored" in p.stderr


@pytest.mark.parametrize("n", [2, 64, 70, 512])
def test_identical_and_opposite_points_meet_in_one_bucket(K, gpu_setup, oracle, n):
    """the bucket sums take the complete branches: a batch of IDENTICAL blobs puts the same row of the same point into a bucket again and
    again (P + P: the doubling branch of the mixed addition, in the tree and in the scan too), and proofs / commitments at infinity drop
    out. Honest batches verify, one altered proof does not -- device-resident and host-pointer (the small ones on the host threads)."""
    import torch
    import blobs as B
    from conftest import R
    blob = B.synthetic_blob(77001)
    zero = bytes(B.BYTES_PER_BLOB)
    for blobs in ([blob] * n, [blob, zero] * (n // 2), [zero] * n):
        data = b"".join(blobs)
        cj = b"".join(K.blob_to_kzg_commitment_batch(data, gpu_setup))
        pj = b"".join(K.compute_blob_kzg_proof_batch(data, cj, gpu_setup))
        db = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        dc = torch.frombuffer(bytearray(cj), dtype=torch.uint8).cuda()
        dp = torch.frombuffer(bytearray(pj), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        assert K.verify_blob_kzg_proof_batch_device(db.data_ptr(), dc.data_ptr(), dp.data_ptr(), n, gpu_setup) is True
        assert K.verify_blob_kzg_proof_batch(data, cj, pj, n, gpu_setup) is True
        other = oracle.g1_generator_mul(12345 % R)
        bad = pj[:48 * (n - 1)] + other
        dpb = torch.frombuffer(bytearray(bad), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        assert K.verify_blob_kzg_proof_batch_device(db.data_ptr(), dc.data_ptr(), dpb.data_ptr(), n, gpu_setup) is False
        assert K.verify_blob_kzg_proof_batch(data, cj, bad, n, gpu_setup) is False
