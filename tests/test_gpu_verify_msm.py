"""GPU: the verification's three linear combinations as the latency-shaped bucket MSM of csrc/vmsm.hip (r06; the reference's
g1_lincomb x 3, /root/reference/src/lib.rs:679-685) against its two other arms, in fresh processes: the scan fallback of a bucket whose
list overflows (LWKZG_VMSM_LIST_CAP=1: every lane takes it) and r05's per-point multiples + Straus pieces (LWKZG_VERIFY_MSM=0). The
partial sums are affine points: all arms must agree BYTE FOR BYTE, per shard, in both semantics, host-pointer and device-resident."""
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
