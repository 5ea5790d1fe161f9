"""Shared helpers of the GPU proof-parity tests: oracle results for whole batches, computed once per (mode, range) on
the host cores and reused by every engine under test; hashlib's view of the Fiat-Shamir challenge; the tau closed
forms. Test infrastructure only."""
import hashlib

import blobs as B
from conftest import R, TAU

_cache = {}


def oracle_batch(oracle, first, n, mode_c, setup_path=None):
    """(blobs, commitments, proofs) of synthetic blobs first .. first + n - 1 in the given mode, from the CPU oracle
    (on the tau = 1337 setup unless another setup file is named)."""
    key = (first, n, bool(mode_c), setup_path)
    if key not in _cache:
        from oracle_pool import SETUP, OraclePool
        omode = oracle.MODE_C if mode_c else oracle.MODE_R
        blobs = [B.synthetic_blob(first + i, big_endian=not mode_c) for i in range(n)]
        with OraclePool(setup_path=setup_path or SETUP) as p:
            comms = p.commitments(blobs, omode)
            assert all(rc == 0 for rc, _ in comms)
            comms = [c for _, c in comms]
            proofs = p.blob_proofs(blobs, comms, omode)
            assert all(rc == 0 for rc, _ in proofs)
        _cache[key] = (blobs, comms, [p for _, p in proofs])
    return _cache[key]


def challenge_int(blob, commitment, mode_c):
    """compute_challenge (src/utils.rs:120-154) with hashlib: the digest read big-endian (reference) or little-endian
    (c-kzg), reduced mod r."""
    msg = b"FSBLOBVERIFY_V1_" + (4096).to_bytes(8, "little") + (0).to_bytes(8, "little") + blob + commitment
    return int.from_bytes(hashlib.sha256(msg).digest(), "little" if mode_c else "big") % R


def reference_mode_proof_closed_form(oracle, blob, commitment, tau=TAU):
    """Reference mode, powers-of-tau setup (default tau = 1337): proof = [(p(tau) - p(z)) / (tau - z)] G with z from
    hashlib -- no MSM, no oracle Pippenger: an independent O(n) check of compute_blob_kzg_proof."""
    coeffs = B.blob_scalars(blob)
    z = challenge_int(blob, commitment, False)
    pt = pz = 0
    for c in reversed(coeffs):
        pt = (pt * tau + c) % R
        pz = (pz * z + c) % R
    q = (pt - pz) * pow((tau - z) % R, R - 2, R) % R
    return oracle.g1_generator_mul(q)
