"""CPU: the schedule of a device-resident compute_blob_kzg_proof call as a PURE function (csrc/plan.h: plan_proof_call).
The C++ function is compiled as plain C++ and enumerated over a grid (tests/plan_table.cpp); every row must equal a second statement of
the rules written here from INTEGRATION.md section 5 / DESIGN.md section 8, and a handful of rows are pinned literally -- the ones the
usage rules in INTEGRATION.md quote. What the schedules are: /root/reference/src/lib.rs:361-404 per blob, five ways of putting the
Fiat-Shamir hash and the commitment validation in front of the quotient MSM."""
import os
import subprocess

from conftest import ROOT

SMALL, COLD, MID, PIPED, GPU = range(5)


def rules(n, warm, busy, direct, staging, ks):
    small, mid, chunks_knob, pipe, pipe_min = 64, 384, 4, True, 192
    if ks == 1:
        pipe = False
    if ks == 2:
        small = mid = 0
    heavy = n <= 512
    if staging and n <= small:
        return (SMALL, 0, 0, 0, heavy)
    if n <= mid and not busy and not warm:
        return (COLD, 0, 0, 0, heavy)
    if staging and n <= mid and not busy:
        per = -(-n // chunks_knob)
        chunks = -(-n // per)
        if pipe and direct and n >= pipe_min and chunks >= 2 and chunks % 2 == 0 and n <= 1024:
            parts = 4 if n >= 320 else 2
            while parts > 1 and chunks % parts:
                parts -= 1
            return (PIPED, per, chunks, parts, heavy)
        return (MID, per, chunks, 0, heavy)
    return (GPU, 0, 0, 0, heavy)


def test_plan_table(tmp_path):
    exe = str(tmp_path / "plan_table")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "plan_table.cpp"), "-o", exe])
    rows = {}
    for line in subprocess.check_output([exe], text=True).splitlines():
        lhs, rhs = line.split(" -> ")
        key = tuple(int(x) for x in lhs.split())
        val = tuple(int(x) for x in rhs.split())
        rows[key] = (val[0], val[1], val[2], val[3], bool(val[4]))
    assert len(rows) == 27 * 16 * 3
    for key, val in rows.items():
        assert val == rules(*key), (key, val)
    # the rows the documentation quotes (default knobs; n, warm, busy, direct, staging, knob set)
    assert rows[(1, 1, 0, 1, 1, 0)][0] == SMALL and rows[(64, 0, 1, 0, 1, 0)][0] == SMALL            # small calls: always the host, warm or not
    assert rows[(256, 0, 0, 1, 1, 0)][0] == COLD                                                      # cold host threads: the GPU hash once
    assert rows[(256, 1, 0, 1, 1, 0)] == (PIPED, 64, 4, 2, True)                                      # the bench's 256-blob call: two sub-batches of two chunks
    assert rows[(384, 1, 0, 1, 1, 0)] == (PIPED, 96, 4, 4, True)
    assert rows[(128, 1, 0, 1, 1, 0)][0] == MID and rows[(256, 1, 0, 0, 1, 0)][0] == MID             # below the pipelining minimum; no direct table under the MSM
    assert rows[(256, 1, 1, 1, 1, 0)][0] == GPU                                                       # the other context busy: two caller streams keep the GPU hash
    assert rows[(1024, 1, 0, 1, 1, 0)] == (GPU, 0, 0, 0, False) and rows[(4096, 1, 0, 1, 1, 0)][0] == GPU
    for key, val in rows.items():
        if not key[4]:
            assert val[0] in (COLD, GPU), key                                                         # without staging no host-assisted schedule


def test_staged_verification_split(tmp_path):
    """plan.h: plan_staged_verification -- who hashes which blobs of a long host-pointer verification
    (/root/reference/src/lib.rs:525-614 per blob: compute_challenge), as a pure function of the batch and the host threads' measured hashing
    rate. Invariants on every row, and the rows the documentation quotes."""
    exe = str(tmp_path / "plan_table")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "plan_table.cpp"), "-o", exe])
    rows = {}
    for line in subprocess.check_output([exe, "staged"], text=True).splitlines():
        lhs, rhs = line.split(" -> ")
        n, rate = (int(x) for x in lhs.split())
        v = [int(x) for x in rhs.split()]
        rows[(n, rate)] = (v[0], v[1], v[2], v[3], v[4:])
    assert len(rows) == 70
    for (n, rate), (n_gpu, n_host, slice_, every, launches) in rows.items():
        assert n_gpu + n_host == n and n_gpu % slice_ == 0 and slice_ == 512 and every == 3       # whole slices for the GPU; three slices upload in 3.6 ms, a launch runs 3.1
        head = n_gpu // slice_
        assert (launches[-1] == head if head else launches == [])                                 # the last launch goes out when the head's last slice has landed
        assert all(b - a == every for a, b in zip(launches, launches[1:])) and (not launches or launches[0] <= every)
        assert n_host <= max(0.8 * rate / 56.0 * n, 0) + slice_                                   # never more than the host threads hash beside the upload (+ the ragged end)
        if rate >= 36 and n >= 4096:
            assert n_host == 1536 + (n % slice_)                                                  # a fast enough host takes what is uploaded in the last 3.4 ms, no more
    assert rows[(4096, 36)] == (2560, 1536, 512, 3, [2, 5])     # DESIGN.md section 8 / profiles/r06_experiments.md section 9: two launches, after the 2nd and the 5th slice
    assert rows[(4096, 30)] == (2560, 1536, 512, 3, [2, 5])
    assert rows[(4096, 18)] == (3072, 1024, 512, 3, [3, 6])     # a slower host takes less
    assert rows[(4096, 0)] == (4096, 0, 512, 3, [2, 5, 8])      # no host threads worth the name: everything on the GPU, a 3.2 ms tail
    assert rows[(1100, 36)] == (512, 588, 512, 3, [1])          # tests/test_gpu_parity.py::test_verify_long_batch_pipelined_path
    assert rows[(2600, 36)] == (1536, 1064, 512, 3, [3])        # tests/verify_arm_worker.py's long shard is 2100 blobs of such a batch; the whole batch: this row
    assert rows[(16384, 36)][:2] == (14848, 1536)
