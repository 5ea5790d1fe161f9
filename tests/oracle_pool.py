"""Runs the CPU oracle over many blobs on the host cores (spawned worker processes: the parent has the GPU open, and a
forked child of such a process must not touch it; the workers import the oracle only, never torch).

Test infrastructure: the oracle is the checker here, nothing in the product path imports this."""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup.txt")

_state = {}


def _init(setup_path=SETUP):
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    _state["O"] = O
    _state["s"] = O.Settings.from_file(setup_path, check_subgroup=False)


def _proof(args):
    blob, comm, mode = args
    O = _state["O"]
    return O.compute_blob_kzg_proof(blob, comm, _state["s"], mode)


def _commit(args):
    blob, mode = args
    O = _state["O"]
    return O.blob_to_kzg_commitment(blob, _state["s"], mode)


def usable_cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        pass
    return n


class OraclePool:
    """with OraclePool() as p: p.blob_proofs(blobs, comms, mode) -> [(rc, proof48), ...] in input order."""

    def __init__(self, procs=None, setup_path=SETUP):
        self.procs = procs or usable_cores()
        self.setup_path = setup_path

    def __enter__(self):
        self.pool = mp.get_context("spawn").Pool(self.procs, initializer=_init, initargs=(self.setup_path,))
        return self

    def __exit__(self, *a):
        self.pool.close()
        self.pool.join()

    def blob_proofs(self, blobs, comms, mode):
        return self.pool.map(_proof, [(b, c, mode) for b, c in zip(blobs, comms)], chunksize=4)

    def commitments(self, blobs, mode):
        return self.pool.map(_commit, [(b, mode) for b in blobs], chunksize=4)
