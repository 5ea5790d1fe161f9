"""GPU: the whole path on trusted setups OTHER than tau = 1337 (VERDICT r02 "missing 1"), on every engine.

The reference's own integration test commits against a random SRS (/root/reference/tests/lib_test.rs:68 ->
src/utils.rs:84-107 create_srs); SURVEY 0.4 asks for "non-power-of-tau random points against a CPU MSM so the kernel
isn't only right for this SRS". Two fixtures (tests/golden/make_setups.py):

* trusted_setup_tau2.txt: powers of a 255-bit tau' -- commitments, both kinds of proof, verification and batch
  verification against the CPU oracle (pinned on this file by tests/test_oracle_golden.py) and the tau' closed forms;
* trusted_setup_unstructured.txt: P_i = [k_i]G, no power structure -- loaded through load_trusted_setup bytes AND as a
  hand-built KZGSettings; commitments against oracle.msm_affine and the closed form sum s_i k_i.

Each on the engine a plain load selects, the bucket engine and the direct tables of 10 / 14 / 16 bits (the seven table
builders of direct.hip split into the generic-plan kernel, <14> and <16>), and two settings objects with different
semantics side by side (lwkzg_settings_set_mode)."""
import ctypes as C
import random

import pytest

import blobs as B
import make_setups as M
from conftest import R, SETUP_PATH, SETUP_TAU2_PATH, SETUP_UNSTRUCTURED_PATH, tau_closed_form, unstructured_closed_form
from proof_cases import oracle_batch, reference_mode_proof_closed_form

pytestmark = pytest.mark.gpu

ENGINES = ["default", 0, 10, 14, 16]


@pytest.fixture(autouse=True)
def _reference_mode(K):
    K.set_mode(K.MODE_REFERENCE)
    yield
    K.set_mode(K.MODE_REFERENCE)


def _select_engine(K, ts_ref, engine, gpu_setup):
    """Put the settings behind `ts_ref` on `engine`; returns an undo callable. The 16-bit table (240-275 GB) needs the
    session's default table out of the way."""
    import torch
    from lambdaworks_kzg_amd import capi
    if engine == "default":
        return lambda: None
    stepped_aside = engine >= 15
    if stepped_aside:
        gpu_setup.enable_direct_table(0)
    rc = K.lib().lwkzg_enable_direct_table(ts_ref, engine)
    if rc != K.C_KZG_OK:
        if stepped_aside:
            gpu_setup.enable_direct_table(gpu_setup.default_bits)
        assert rc == K.C_KZG_MALLOC
        free_b = torch.cuda.mem_get_info()[0]
        assert free_b < capi.direct_table_bytes(engine) + (12 << 30), (engine, free_b)
        pytest.skip("direct table of width %d does not fit on this device (%d GB free)" % (engine, free_b >> 30))
    assert K.lib().lwkzg_direct_table_bits(ts_ref) == engine

    def undo():
        K.lib().lwkzg_enable_direct_table(ts_ref, 0)
        torch.cuda.empty_cache()
        if stepped_aside:
            gpu_setup.enable_direct_table(gpu_setup.default_bits)
    return undo


@pytest.fixture(scope="module", params=ENGINES, ids=lambda e: "engine_%s" % e)
def tau2_setup(request, K, gpu_setup):
    ts = K.TrustedSetup.from_file(SETUP_TAU2_PATH)
    undo = _select_engine(K, ts.ref(), request.param, gpu_setup)
    yield ts
    undo()
    ts.free()


@pytest.fixture(scope="module")
def oracle_tau2(oracle):
    return oracle.Settings.from_file(SETUP_TAU2_PATH, check_subgroup=False)


@pytest.fixture(scope="module")
def oracle_unstructured(oracle):
    return oracle.Settings.from_file(SETUP_UNSTRUCTURED_PATH, check_subgroup=False)


def _adversarial_sets():
    return [[R - 1] * 4096,
            [5, R - 5] * 2048,
            [0] * 1234 + [R - 2] + [0] * 2861,
            [0] * 4096,
            [(1 << 254) | 1] * 4096,
            [sum(1 << (13 * j + 12) for j in range(19)) % R] * 4096,      # every 13-bit window at its sign boundary
            [sum(1 << (16 * j + 15) for j in range(15)) % R] * 4096,      # every 16-bit window at its sign boundary
            [sum(1 << (10 * j + 9) for j in range(25)) % R] * 4096,       # every 10-bit window at its sign boundary
            list(range(1, 4097))]


# ---- powers of tau' ----------------------------------------------------------------------------------------------

def test_tau2_load_and_commitments(K, tau2_setup, oracle, oracle_tau2):
    ts = tau2_setup
    assert ts.g1_values_bytes() == oracle_tau2.g1_blst()                    # srs.rs:131-153 bytes of all 4096 points
    n = 70
    data = B.synthetic_batch(31000, n)
    got = K.blob_to_kzg_commitment_batch(data, ts)
    for i in range(n):
        blob = data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB]
        assert got[i] == tau_closed_form(oracle, B.blob_scalars(blob), tau=M.TAU2), i
        if i < 3:
            assert (0, got[i]) == oracle.blob_to_kzg_commitment(blob, oracle_tau2, oracle.MODE_R)
    sets = _adversarial_sets()
    blobs = [b"".join(s.to_bytes(32, "big") for s in ss) for ss in sets]
    for ss, g in zip(sets, K.blob_to_kzg_commitment_batch(b"".join(blobs), ts)):
        assert g == tau_closed_form(oracle, ss, tau=M.TAU2)
    # full-range 32-byte scalars (reduced mod r by the parse kernel, utils.rs:27-41), one launch geometry with 1024 blobs
    full = B.synthetic_batch(32000, 1024, full_range=True)
    got = K.blob_to_kzg_commitment_batch(full, ts)
    for i in list(range(0, 1024, 97)) + [1023]:
        blob = full[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB]
        assert got[i] == tau_closed_form(oracle, [v % R for v in B.blob_scalars(blob)], tau=M.TAU2), i
    assert K.blob_to_kzg_commitment(blobs[0], ts) == tau_closed_form(oracle, sets[0], tau=M.TAU2)   # the single-blob symbol


@pytest.mark.parametrize("mode_c", [False, True], ids=["reference", "ckzg"])
def test_tau2_device_resident_proofs_every_blob_vs_oracle(K, tau2_setup, oracle, mode_c):
    from test_gpu_proof_parity import device_commit_and_prove
    K.set_mode(K.MODE_CKZG if mode_c else K.MODE_REFERENCE)
    n = 70
    blobs, want_c, want_p = oracle_batch(oracle, 33000, n, mode_c, setup_path=SETUP_TAU2_PATH)
    comms, proofs = device_commit_and_prove(K, tau2_setup, b"".join(blobs), n)
    for i in range(n):
        assert comms[48 * i:48 * i + 48] == want_c[i], ("commitment", i)
        assert proofs[48 * i:48 * i + 48] == want_p[i], ("proof", i)
        if not mode_c and i < 8:
            assert want_p[i] == reference_mode_proof_closed_form(oracle, blobs[i], want_c[i], tau=M.TAU2)
    # the host-pointer batch entry points and the one-pass commit-and-prove give the same bytes
    joined = b"".join(blobs)
    assert K.compute_blob_kzg_proof_batch(joined, b"".join(want_c), tau2_setup) == want_p
    import torch
    d_blobs = torch.frombuffer(bytearray(joined), dtype=torch.uint8).cuda()
    d_c = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    K.commit_and_prove_batch_device(d_c.data_ptr(), d_p.data_ptr(), d_blobs.data_ptr(), n, tau2_setup)
    torch.cuda.synchronize()
    assert bytes(d_c.cpu().numpy().tobytes()) == b"".join(want_c) and bytes(d_p.cpu().numpy().tobytes()) == b"".join(want_p)


@pytest.mark.parametrize("mode_c", [False, True], ids=["reference", "ckzg"])
def test_tau2_point_proofs_and_verification(K, tau2_setup, oracle, oracle_tau2, mode_c):
    """compute_kzg_proof against the oracle; the verify side (pairing against THIS setup's [tau']G2) accepts what the
    oracle's known-tau' verifier accepts and rejects what it rejects; batch verification of 5 and 70 blobs"""
    ts = tau2_setup
    K.set_mode(K.MODE_CKZG if mode_c else K.MODE_REFERENCE)
    omode = oracle.MODE_C if mode_c else oracle.MODE_R
    order = "little" if mode_c else "big"
    rnd = random.Random(77)
    n = 70
    blobs, comms, proofs = oracle_batch(oracle, 33000, n, mode_c, setup_path=SETUP_TAU2_PATH)
    for i in range(3):
        z = rnd.randrange(R).to_bytes(32, order)
        rc, want_pr, want_y = oracle.compute_kzg_proof(blobs[i], z, oracle_tau2, omode)
        assert rc == 0
        pr, y = K.compute_kzg_proof(blobs[i], z, ts)
        assert (pr, y) == (want_pr, want_y)
        assert K.verify_kzg_proof(comms[i], z, y, pr, ts) is True
        assert oracle.verify_kzg_proof_known_tau(comms[i], z, y, pr, M.TAU2, omode) == (0, True)
        bad_y = ((int.from_bytes(y, order) + 1) % R).to_bytes(32, order)
        assert K.verify_kzg_proof(comms[i], z, bad_y, pr, ts) is False
        assert oracle.verify_kzg_proof_known_tau(comms[i], z, bad_y, pr, M.TAU2, omode) == (0, False)
        assert K.verify_blob_kzg_proof(blobs[i], comms[i], proofs[i], ts) is True
        assert K.verify_blob_kzg_proof(blobs[i], comms[i], proofs[(i + 1) % n], ts) is False
    for k in (5, n):
        data, cm, pr = b"".join(blobs[:k]), b"".join(comms[:k]), b"".join(proofs[:k])
        assert K.verify_blob_kzg_proof_batch(data, cm, pr, k, ts) is True
        swapped = pr[48:96] + pr[:48] + pr[96:]
        assert K.verify_blob_kzg_proof_batch(data, cm, swapped, k, ts) is False
    # a proof that is right for the tau = 1337 setup is wrong here: the verdict depends on this setup's G2 point
    ts1337 = K.TrustedSetup.from_file(SETUP_PATH)
    try:
        ts1337.enable_direct_table(0)
        c1 = K.blob_to_kzg_commitment(blobs[0], ts1337)
        p1 = K.compute_blob_kzg_proof(blobs[0], c1, ts1337)
        assert K.verify_blob_kzg_proof(blobs[0], c1, p1, ts1337) is True
        assert K.verify_blob_kzg_proof(blobs[0], c1, p1, ts) is False
    finally:
        ts1337.free()


# ---- two consumers, two semantics, one process -------------------------------------------------------------------

def test_two_settings_objects_with_different_modes(K, gpu_setup, tau2_setup, oracle, oracle_setup, oracle_tau2):
    """lwkzg_settings_set_mode: the mode belongs to the settings object; the process-wide default only applies to
    objects that were not given one. Same bytes as the oracle in the respective mode, interleaved calls."""
    a, b = gpu_setup, tau2_setup
    be, le = B.synthetic_blob(35000), B.synthetic_blob(35001, big_endian=False)
    try:
        assert a.set_mode(K.MODE_REFERENCE) == K.MODE_REFERENCE and b.set_mode(K.MODE_CKZG) == K.MODE_REFERENCE
        K.set_mode(K.MODE_CKZG)                                        # the default no longer matters for either
        assert (a.get_mode(), b.get_mode()) == (K.MODE_REFERENCE, K.MODE_CKZG)
        for _ in range(2):
            ca = K.blob_to_kzg_commitment(be, a)
            cb = K.blob_to_kzg_commitment(le, b)
            assert (0, ca) == oracle.blob_to_kzg_commitment(be, oracle_setup, oracle.MODE_R)
            assert (0, cb) == oracle.blob_to_kzg_commitment(le, oracle_tau2, oracle.MODE_C)
            assert (0, K.compute_blob_kzg_proof(be, ca, a)) == oracle.compute_blob_kzg_proof(be, ca, oracle_setup, oracle.MODE_R)
            assert (0, K.compute_blob_kzg_proof(le, cb, b)) == oracle.compute_blob_kzg_proof(le, cb, oracle_tau2, oracle.MODE_C)
        # error codes follow the object's mode too: a non-canonical element is reduced in reference mode, BADARGS in c-kzg mode
        bad = b"\xff" * 32 + le[32:]
        with pytest.raises(K.KzgError) as e:
            K.blob_to_kzg_commitment(bad, b)
        assert e.value.rc == K.C_KZG_BADARGS
        K.blob_to_kzg_commitment(bad, a)
        # the empty batch: False for the reference-mode object, True for the c-kzg one
        assert K.verify_blob_kzg_proof_batch(b"", b"", b"", 0, a) is False
        assert K.verify_blob_kzg_proof_batch(b"", b"", b"", 0, b) is True
        assert b.set_mode(-1) == K.MODE_CKZG and b.get_mode() == K.MODE_CKZG       # back on the (c-kzg) default
        K.set_mode(K.MODE_REFERENCE)
        assert b.get_mode() == K.MODE_REFERENCE and a.get_mode() == K.MODE_REFERENCE
        with pytest.raises(K.KzgError):
            a.set_mode(7)
    finally:
        a.set_mode(-1)
        b.set_mode(-1)
        K.set_mode(K.MODE_REFERENCE)
