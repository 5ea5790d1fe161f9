"""GPU: c-kzg mode on the LAGRANGE form of the setup (the conversion the reference left commented out,
/root/reference/src/lib.rs:760-770, src/srs.rs:117-124): L_i = [l_i(tau)]G derived on the device, commitments as an MSM over the
blob's evaluations as they stand (no transform), the quotient of a proof moved to that form by one forward transform when the
Lagrange table is the only direct table. Every answer must be the one the transform path (inverse NTT + monomial tables) gives,
the c-kzg vectors' and the oracle's -- on a Lagrange-only table, on both forms side by side, on the Lagrange buckets, and across
the moves lwkzg_settings_set_mode makes."""
import ctypes as C

import pytest

import blobs as B
from conftest import R, SETUP_PATH, TAU, hx

pytestmark = pytest.mark.gpu

MONO, LAG, BOTH = 1, 2, 3
W4096 = pow(7, (R - 1) // 4096, R)


def _brp(i):
    return int(format(i, "012b")[::-1], 2)


@pytest.fixture(scope="module")
def mono(K):
    """the transform path: a 10-bit monomial table only, answering in c-kzg mode"""
    ts = K.TrustedSetup.from_file(SETUP_PATH)
    ts.enable_direct_table_forms(10, MONO)
    assert ts.direct_table_forms() == MONO
    ts.mode_override = True
    yield ts
    ts.free()


@pytest.fixture(scope="module", params=["lagrange_only_10", "both_11", "lagrange_buckets"])
def lag(request, K):
    ts = K.TrustedSetup.from_file(SETUP_PATH)
    if request.param == "lagrange_only_10":
        ts.enable_direct_table_forms(10, LAG)
        assert ts.direct_table_forms() == LAG and ts.direct_table_bits() == 10
    elif request.param == "both_11":
        ts.enable_direct_table_forms(11, BOTH)
        assert ts.direct_table_forms() == BOTH and ts.direct_table_bits() == 11
    else:
        ts.enable_direct_table(0)
        assert ts.direct_table_forms() == 0
    yield ts
    ts.free()


@pytest.fixture(autouse=True)
def _ckzg_mode(K):
    K.set_mode(K.MODE_CKZG)          # the process-wide default: never moves a table
    yield
    K.set_mode(K.MODE_REFERENCE)


def test_unit_blobs_commit_to_the_lagrange_points(K, lag, oracle):
    """blob = e_i (a single evaluation equal to one) commits to L_i itself: [l_i(tau)]G with l_i(tau) = (tau^4096 - 1) / 4096 * w_i / (tau - w_i)"""
    zt = (pow(TAU, 4096, R) - 1) * pow(4096, -1, R) % R
    idx = [0, 1, 2, 5, 2047, 2048, 4094, 4095]
    blobs = b""
    for i in idx:
        b = bytearray(B.BYTES_PER_BLOB)
        b[32 * i] = 1                                  # little-endian 1
        blobs += bytes(b)
    got = K.blob_to_kzg_commitment_batch(blobs, lag)
    for k, i in enumerate(idx):
        wi = pow(W4096, _brp(i), R)
        assert got[k] == oracle.g1_generator_mul(zt * wi % R * pow(TAU - wi, -1, R) % R), i


@pytest.mark.parametrize("n", [1, 3, 64, 200, 700, 1024])
def test_commitments_equal_the_transform_path(K, lag, mono, oracle, oracle_setup, n):
    import torch
    data = B.synthetic_batch(91000 + n, n, big_endian=False)
    want = K.blob_to_kzg_commitment_batch(data, mono)
    assert K.blob_to_kzg_commitment_batch(data, lag) == want
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    d_out = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    d_st = torch.full((n,), 9, dtype=torch.int32, device="cuda")
    K.blob_to_kzg_commitment_batch_device(d_out.data_ptr(), d_in.data_ptr(), n, lag, None, d_st.data_ptr())
    torch.cuda.synchronize()
    assert bytes(d_out.cpu().numpy().tobytes()) == b"".join(want) and int(d_st.abs().sum()) == 0
    for i in {0, n - 1}:
        assert oracle.blob_to_kzg_commitment(data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB], oracle_setup, oracle.MODE_C) == (0, want[i])
    if n == 1:
        assert K.blob_to_kzg_commitment(data, lag) == want[0]            # the reference's own symbol (coalescing front)


def test_full_range_and_edge_elements(K, lag, mono):
    """elements up to r - 1, the zero blob (infinity), sparse blobs; one element >= r: BADARGS for that blob only"""
    import random
    rnd = random.Random(5)
    blobs = [b"".join(rnd.randrange(R).to_bytes(32, "little") for _ in range(4096)),
             b"".join((R - 1).to_bytes(32, "little") for _ in range(4096)),
             bytes(B.BYTES_PER_BLOB),
             b"".join((rnd.randrange(R) if k % 97 == 0 else 0).to_bytes(32, "little") for k in range(4096))]
    data = b"".join(blobs)
    assert K.blob_to_kzg_commitment_batch(data, lag) == K.blob_to_kzg_commitment_batch(data, mono)
    # r05: and their proofs, whose quotient is taken in evaluation form on the Lagrange form (full-range p_i, y and q_i; z = -1 and z = 1 are domain points)
    comms = b"".join(K.blob_to_kzg_commitment_batch(data, mono))
    assert K.compute_blob_kzg_proof_batch(data, comms, lag) == K.compute_blob_kzg_proof_batch(data, comms, mono)
    zb = b"".join(z.to_bytes(32, "little") for z in (R - 1, 1, 0, rnd.randrange(R)))
    assert K.compute_kzg_proof_batch(data, zb, lag) == K.compute_kzg_proof_batch(data, zb, mono)
    bad = bytearray(data)
    bad[2 * B.BYTES_PER_BLOB + 32 * 77:2 * B.BYTES_PER_BLOB + 32 * 78] = R.to_bytes(32, "little")
    out = C.create_string_buffer(48 * 4)
    first = C.c_size_t(99)
    assert K.lib().lwkzg_blob_to_kzg_commitment_batch(out, bytes(bad), 4, lag.ref(), C.byref(first)) == K.C_KZG_BADARGS and first.value == 2


def test_ckzg_vectors_on_the_lagrange_form(K, lag, vectors):
    n = 0
    for c in vectors["suites"]["blob_to_kzg_commitment"]:
        blob = B.make_blob(c["input"]["blob"])
        if len(blob) != B.BYTES_PER_BLOB:
            continue
        if c["output"] is None:
            with pytest.raises(K.KzgError) as e:
                K.blob_to_kzg_commitment(blob, lag)
            assert e.value.rc == K.C_KZG_BADARGS
        else:
            assert K.blob_to_kzg_commitment(blob, lag) == hx(c["output"])
        n += 1
    assert n == 8
    n = 0
    for c in vectors["suites"]["compute_kzg_proof"]:                      # (z on the domain among them)
        blob, z = B.make_blob(c["input"]["blob"]), hx(c["input"]["z"])
        if len(blob) != B.BYTES_PER_BLOB or len(z) != 32:
            continue
        if c["output"] is None:
            with pytest.raises(K.KzgError) as e:
                K.compute_kzg_proof(blob, z, lag)
            assert e.value.rc == K.C_KZG_BADARGS
        else:
            pr, y = K.compute_kzg_proof(blob, z, lag)
            assert pr == hx(c["output"][0]) and y == hx(c["output"][1])
        n += 1
    assert n == 42
    n = 0
    for c in vectors["suites"]["compute_blob_kzg_proof"]:
        blob, cm = B.make_blob(c["input"]["blob"]), hx(c["input"]["commitment"])
        if len(blob) != B.BYTES_PER_BLOB or len(cm) != 48:
            continue
        if c["output"] is None:
            with pytest.raises(K.KzgError) as e:
                K.compute_blob_kzg_proof(blob, cm, lag)
            assert e.value.rc == K.C_KZG_BADARGS
        else:
            assert K.compute_blob_kzg_proof(blob, cm, lag) == hx(c["output"])
        n += 1
    assert n == 10


@pytest.mark.parametrize("n", [1, 70, 600])
def test_proofs_equal_the_transform_path(K, lag, mono, oracle, oracle_setup, n):
    """blob proofs (host, device, one-pass commit-and-prove) and point proofs; on a Lagrange-only table the quotient takes one forward
    transform to the Lagrange form, everywhere else it stays on the monomial tables"""
    import torch
    from lambdaworks_kzg_amd import capi
    data = B.synthetic_batch(92000 + n, n, big_endian=False)
    comms = b"".join(K.blob_to_kzg_commitment_batch(data, mono))
    want = K.compute_blob_kzg_proof_batch(data, comms, mono)
    assert K.compute_blob_kzg_proof_batch(data, comms, lag) == want
    assert oracle.compute_blob_kzg_proof(data[:B.BYTES_PER_BLOB], comms[:48], oracle_setup, oracle.MODE_C) == (0, want[0])
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    d_cm = torch.frombuffer(bytearray(comms), dtype=torch.uint8).cuda()
    d_out = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    d_c2 = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    d_st = torch.full((n,), 9, dtype=torch.int32, device="cuda")
    K.compute_blob_kzg_proof_batch_device(d_out.data_ptr(), d_in.data_ptr(), d_cm.data_ptr(), n, lag, None, d_st.data_ptr())
    torch.cuda.synchronize()
    assert bytes(d_out.cpu().numpy().tobytes()) == b"".join(want) and int(d_st.abs().sum()) == 0
    d_out.zero_()
    K.commit_and_prove_batch_device(d_c2.data_ptr(), d_out.data_ptr(), d_in.data_ptr(), n, lag, None, d_st.data_ptr())
    torch.cuda.synchronize()
    assert bytes(d_c2.cpu().numpy().tobytes()) == comms and bytes(d_out.cpu().numpy().tobytes()) == b"".join(want)
    # point proofs: arbitrary z, z on the domain (w_5 and w_0 = 1), z = 0
    zs = [(12345 + 7 * i) % R for i in range(n)]
    zs[0] = pow(W4096, _brp(5), R)
    if n > 2:
        zs[1], zs[2] = 1, 0
    zb = b"".join(z.to_bytes(32, "little") for z in zs)
    want_pz = K.compute_kzg_proof_batch(data, zb, mono)
    assert K.compute_kzg_proof_batch(data, zb, lag) == want_pz
    rc, pr, y = oracle.compute_kzg_proof(data[:B.BYTES_PER_BLOB], zb[:32], oracle_setup, oracle.MODE_C)
    assert rc == 0 and (pr, y) == want_pz[0]
    if n == 1:
        assert K.compute_blob_kzg_proof(data, comms, lag) == want[0] and K.compute_kzg_proof(data, zb, lag) == want_pz[0]
    assert K.verify_blob_kzg_proof_batch(data, comms, b"".join(want), n, lag) is True


def test_evaluation_form_quotient_with_z_on_the_domain(K, lag, mono, oracle, oracle_setup):
    """r05: on a usable Lagrange form a c-kzg-mode proof takes its quotient in EVALUATION form (fr_ops.hip: k_eval_quotient_evalform; one batch
    inversion per blob, y by the barycentric formula). z = w_m makes the batch inversion's product zero: the kernel finds m, sets y = p_m and
    takes q_m from the other quotients. m at every position that matters to a workgroup of 256 threads x 16 elements (first and last of
    a thread, of a wave, of the blob; w = 1 and w = -1), beside ordinary z in the same launch, against the coefficient-form path
    (a monomial-only table) and the oracle"""
    ms = [0, 1, 2, 15, 16, 17, 1023, 1024, 2047, 2048, 4094, 4095, 777, 3001]
    n = 2 * len(ms)
    data = B.synthetic_batch(93100, n, big_endian=False)
    zs = []
    for j, m in enumerate(ms):
        zs += [pow(W4096, _brp(m), R), (R - 1 - 1000 * j) % R]
    zb = b"".join(z.to_bytes(32, "little") for z in zs)
    want = K.compute_kzg_proof_batch(data, zb, mono)
    got = K.compute_kzg_proof_batch(data, zb, lag)
    assert got == want
    for j in (0, 1, 6, 22):
        rc, pr, y = oracle.compute_kzg_proof(data[j * B.BYTES_PER_BLOB:(j + 1) * B.BYTES_PER_BLOB], zb[32 * j:32 * j + 32], oracle_setup, oracle.MODE_C)
        assert rc == 0 and (pr, y) == got[j]
    for j, m in enumerate(ms):   # y = p(w_m) is the blob's own element m
        assert got[2 * j][1] == data[2 * j * B.BYTES_PER_BLOB + 32 * m:2 * j * B.BYTES_PER_BLOB + 32 * m + 32]
    # one blob per call (the cooperative kernel's path) with z on the domain
    assert K.compute_kzg_proof(data[:B.BYTES_PER_BLOB], zb[:32], lag) == want[0]


def test_reference_mode_on_a_lagrange_form_table_still_answers(K, lag, oracle):
    """monomial semantics on a settings object whose direct table may be in the Lagrange form only: the monomial buckets answer"""
    from conftest import tau_closed_form
    K.set_mode(K.MODE_REFERENCE)
    data = B.synthetic_batch(93000, 5)
    got = K.blob_to_kzg_commitment_batch(data, lag)
    for i in range(5):
        assert got[i] == tau_closed_form(oracle, B.blob_scalars(data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB]))
    assert K.compute_blob_kzg_proof_batch(data, b"".join(got), lag) == K.compute_blob_kzg_proof_batch(data, b"".join(got), lag)


def test_settings_mode_moves_the_tables_and_the_default_mode_does_not(K, oracle, oracle_setup):
    """lwkzg_settings_set_mode brings the tables to the mode's form: a second table beside the first when it fits (12 bits: it does),
    nothing when the form is there already; lwkzg_enable_direct_table builds for the mode in force; the process-wide default alone
    builds the Lagrange form lazily, at the first c-kzg call"""
    K.set_mode(K.MODE_REFERENCE)
    ts = K.TrustedSetup.from_file(SETUP_PATH)
    try:
        assert ts.direct_table_forms() == MONO                              # loaded in reference mode: monomial
        data = B.synthetic_batch(94000, 9, big_endian=False)
        K.set_mode(K.MODE_CKZG)                                             # the default: no table moves yet
        assert ts.direct_table_forms() == MONO
        import torch
        from lambdaworks_kzg_amd import capi
        roomy = torch.cuda.mem_get_info()[0] // 4 >= capi.direct_table_bytes(ts.direct_table_bits(), 112)
        first = K.blob_to_kzg_commitment_batch(data, ts)                    # first c-kzg call: Lagrange form derived; a second table beside the
        assert ts.direct_table_forms() == (BOTH if roomy else MONO)        # first only within a quarter of the free memory (the load's own rule)
        assert K.blob_to_kzg_commitment_batch(data, ts) == first
        assert oracle.blob_to_kzg_commitment(data[:B.BYTES_PER_BLOB], oracle_setup, oracle.MODE_C) == (0, first[0])
        ts.enable_direct_table_forms(12, MONO)
        assert ts.direct_table_forms() == MONO and ts.direct_table_bits() == 12
        assert K.blob_to_kzg_commitment_batch(data, ts) == first            # transform path
        prev = ts.set_mode(K.MODE_CKZG)                                     # explicit: the Lagrange table appears beside
        assert prev == K.MODE_CKZG or prev == K.MODE_REFERENCE
        K.set_mode(K.MODE_REFERENCE)
        ts.set_mode(-1)
        ts.set_mode(K.MODE_CKZG)
        assert ts.direct_table_forms() == BOTH and ts.direct_table_bits() == 12
        assert K.blob_to_kzg_commitment_batch(data, ts) == first
        ts.enable_direct_table(0)
        assert ts.direct_table_forms() == 0
        assert K.blob_to_kzg_commitment_batch(data, ts) == first            # Lagrange buckets
        ts.enable_direct_table(10)                                          # built for the mode in force (c-kzg): Lagrange first, monomial beside
        assert ts.direct_table_forms() == BOTH
        assert K.blob_to_kzg_commitment_batch(data, ts) == first
        with pytest.raises(K.KzgError) as e:
            ts.enable_direct_table_forms(10, 4)
        assert e.value.rc == K.C_KZG_BADARGS
    finally:
        ts.free()
        K.set_mode(K.MODE_REFERENCE)


def test_hand_built_settings_and_two_caller_streams_on_the_lagrange_form(K, mono, oracle, oracle_setup):
    """the Lagrange form behind a KZGSettings filled in by hand (fs == NULL: the context lives in the library's registry) and behind the
    settings' SECOND context (two caller streams: the twin holds copies of every table address, refreshed when the tables move)"""
    import torch
    ts0 = K.TrustedSetup.from_file(SETUP_PATH)
    g1 = C.create_string_buffer(ts0.g1_values_bytes())
    hand = K.KZGSettings()
    hand.fs, hand.g1_values, hand.g2_values = None, C.cast(g1, C.c_void_p), ts0.s.g2_values
    n = 40
    data = B.synthetic_batch(95000, n, big_endian=False)
    want = K.blob_to_kzg_commitment_batch(data, mono)
    try:
        out = C.create_string_buffer(48 * n)
        bad = C.c_size_t(0)
        assert K.lib().lwkzg_blob_to_kzg_commitment_batch(out, data, n, C.byref(hand), C.byref(bad)) == K.C_KZG_OK
        assert [out.raw[48 * i:48 * i + 48] for i in range(n)] == want
        assert K.lib().lwkzg_direct_table_forms(C.byref(hand)) & LAG or torch.cuda.mem_get_info()[0] // 4 < 36 << 30
        assert K.lib().lwkzg_enable_direct_table_forms(C.byref(hand), 10, LAG) == K.C_KZG_OK
        assert K.lib().lwkzg_blob_to_kzg_commitment_batch(out, data, n, C.byref(hand), C.byref(bad)) == K.C_KZG_OK
        assert [out.raw[48 * i:48 * i + 48] for i in range(n)] == want
    finally:
        K.lib().lwkzg_release_context(C.byref(hand))
    # two caller streams on a Lagrange-only table: the second stream's calls run on the twin context
    ts0.enable_direct_table_forms(10, LAG)
    ts0.reserve(n, caller_streams=2)
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    d_cm = torch.frombuffer(bytearray(b"".join(want)), dtype=torch.uint8).cuda()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = [torch.empty(48 * n, dtype=torch.uint8, device="cuda") for _ in range(6)]
    prf = [torch.empty(48 * n, dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for k in range(6):
        K.blob_to_kzg_commitment_batch_device(outs[k].data_ptr(), d_in.data_ptr(), n, ts0, (s1 if k % 2 == 0 else s2).cuda_stream, None)
    K.compute_blob_kzg_proof_batch_device(prf[0].data_ptr(), d_in.data_ptr(), d_cm.data_ptr(), n, ts0, s1.cuda_stream, None)
    K.compute_blob_kzg_proof_batch_device(prf[1].data_ptr(), d_in.data_ptr(), d_cm.data_ptr(), n, ts0, s2.cuda_stream, None)
    torch.cuda.synchronize()
    for o in outs:
        assert bytes(o.cpu().numpy().tobytes()) == b"".join(want)
    want_p = b"".join(K.compute_blob_kzg_proof_batch(data, b"".join(want), mono))
    assert bytes(prf[0].cpu().numpy().tobytes()) == want_p and bytes(prf[1].cpu().numpy().tobytes()) == want_p
    ts0.enable_direct_table_forms(11, BOTH)                               # the tables move under an existing twin
    K.blob_to_kzg_commitment_batch_device(outs[0].data_ptr(), d_in.data_ptr(), n, ts0, s1.cuda_stream, None)
    K.blob_to_kzg_commitment_batch_device(outs[1].data_ptr(), d_in.data_ptr(), n, ts0, s2.cuda_stream, None)
    torch.cuda.synchronize()
    assert bytes(outs[0].cpu().numpy().tobytes()) == b"".join(want) == bytes(outs[1].cpu().numpy().tobytes())
    ts0.free()
