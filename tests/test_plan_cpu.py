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
