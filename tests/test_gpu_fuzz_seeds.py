"""GPU: replay of the reference's fuzz seeds (/root/reference/fuzz/*/corpus/, 115 inputs) through the nine symbols the way
the reference's harnesses call them (fuzz/base_fuzz.h:17-34, fuzz/*/fuzz.c), in both modes: return code and output bytes
against the CPU oracle's answers stored in tests/golden/fuzz_seeds.json (tests/test_oracle_golden.py re-derives them)."""
import ctypes as C

import pytest

import fuzz_cases as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seeds():
    return F.load_seeds()[1]


def _call(K, ts, target, data):
    """what the harness of `target` does with `data`: (rc, output) with output = hex, bool or None"""
    l = K.lib()
    if target == "blob_to_kzg_commitment":
        out = C.create_string_buffer(48)
        rc = l.blob_to_kzg_commitment(out, data, ts.ref())
        return rc, out.raw.hex() if rc == 0 else None
    if target == "compute_kzg_proof":
        pr, y = C.create_string_buffer(48), C.create_string_buffer(32)
        rc = l.compute_kzg_proof(pr, y, data[:F.BLOB], data[F.BLOB:], ts.ref())
        return rc, (pr.raw + y.raw).hex() if rc == 0 else None
    if target == "compute_blob_kzg_proof":
        pr = C.create_string_buffer(48)
        rc = l.compute_blob_kzg_proof(pr, data[:F.BLOB], data[F.BLOB:], ts.ref())
        return rc, pr.raw.hex() if rc == 0 else None
    ok = C.c_bool(True)
    if target == "verify_kzg_proof":
        rc = l.verify_kzg_proof(C.byref(ok), data[:48], data[48:80], data[80:112], data[112:160], ts.ref())
    elif target == "verify_blob_kzg_proof":
        rc = l.verify_blob_kzg_proof(C.byref(ok), data[:F.BLOB], data[F.BLOB:F.BLOB + 48], data[F.BLOB + 48:], ts.ref())
    else:
        count, blobs, comms, proofs = F.batch_split(data or b"")
        rc = l.verify_blob_kzg_proof_batch(C.byref(ok), blobs, comms, proofs, count, ts.ref())
    return rc, bool(ok.value) if rc == 0 else None


@pytest.mark.parametrize("mode", ["reference", "ckzg"])
def test_fuzz_seed_replay(K, engine_setup, seeds, mode):
    ts = engine_setup
    ts.set_mode(K.MODE_CKZG if mode == "ckzg" else K.MODE_REFERENCE)
    try:
        called = 0
        for e, data in seeds:
            if not e["harness_calls"]:
                continue
            rc, out = _call(K, ts, e["target"], data)
            assert {"rc": rc, "out": out} == e["expect"][mode], (e["target"], e["name"])
            called += 1
        assert called == 65
    finally:
        ts.set_mode(-1)
