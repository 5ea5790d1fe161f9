"""GPU: parity of the HIP path (through the C ABI) with the CPU oracle, the committed golden vectors
and the tau = 1337 closed form. Bit-exact: every output is a canonical byte encoding."""
import ctypes as C
import random

import pytest

import blobs as B
from conftest import R, SETUP_PATH, TAU, hx, tau_closed_form

pytestmark = pytest.mark.gpu


def _dev(data):
    import torch
    return torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()


def _host(t):
    return bytes(t.cpu().numpy().tobytes())


@pytest.fixture(autouse=True)
def _reference_mode(K):
    K.set_mode(K.MODE_REFERENCE)
    yield
    K.set_mode(K.MODE_REFERENCE)


# ---- setup load (a12, a13, a14) --------------------------------------------------------------------

def test_setup_load_matches_reference_layout(K, gpu_setup, oracle, oracle_setup):
    # g1_point_to_blst_p1 (srs.rs:131-153) bytes, produced by GPU decompression of all 4096 points
    assert gpu_setup.g1_values_bytes() == oracle_setup.g1_blst()
    fs = gpu_setup.fft_settings()
    assert fs.max_width == 4096
    w = pow(7, (R - 1) // 4096, R)
    import struct
    def fr_at(ptr, i):
        l = struct.unpack("<4Q", C.string_at(ptr + 32 * i, 32))
        return (l[0] << 192) | (l[1] << 128) | (l[2] << 64) | l[3]
    for i in (0, 1, 2, 2047, 2048, 4095, 4096):
        assert fr_at(fs.expanded_roots_of_unity, i) == pow(w, i, R)
        assert fr_at(fs.reverse_roots_of_unity, i) == pow(w, -i, R)
    assert fr_at(fs.roots_of_unity, 1) == pow(w, 2048, R)   # bit-reversed order


def test_setup_from_bytes_and_bad_inputs(K, oracle_setup):
    g1, g2 = oracle_setup.g1_compressed(), oracle_setup.g2_compressed()
    ts = K.TrustedSetup.from_bytes(g1, g2)
    assert ts.g1_values_bytes() == oracle_setup.g1_blst()
    ts.free()
    bad = bytearray(g1)
    bad[48 * 7 + 20] ^= 1                                   # x no longer on the curve / wrong subgroup
    with pytest.raises(K.KzgError) as e:
        K.TrustedSetup.from_bytes(bytes(bad), g2)
    assert e.value.rc == K.C_KZG_ERROR
    notsub = bytearray(g1)
    notsub[48:96] = bytes([0x80]) + bytes(47)               # (0, 2): on the curve, not in G1 (compression.rs:155-165)
    with pytest.raises(K.KzgError):
        K.TrustedSetup.from_bytes(bytes(notsub), g2)


def test_hand_built_settings_without_context(K, gpu_setup, oracle, oracle_setup):
    # a caller may fill KZGSettings itself (fs = NULL), exactly the reference's own layout (lib.rs:754-758)
    s = K.KZGSettings()
    g1 = C.create_string_buffer(oracle_setup.g1_blst())
    s.fs, s.g1_values, s.g2_values = None, C.cast(g1, C.c_void_p), gpu_setup.s.g2_values
    blob = B.synthetic_blob(3)
    out = C.create_string_buffer(48)
    assert K.lib().blob_to_kzg_commitment(out, blob, C.byref(s)) == K.C_KZG_OK
    assert out.raw == oracle.blob_to_kzg_commitment(blob, oracle_setup, oracle.MODE_R)[1]
    assert K.lib().lwkzg_direct_table_bits(C.byref(s)) == gpu_setup.default_bits     # the engine a loaded setup would get
    assert K.lib().lwkzg_release_context(C.byref(s)) == K.C_KZG_OK                     # the arrays stay the caller's
    assert K.lib().lwkzg_release_context(gpu_setup.ref()) == K.C_KZG_BADARGS           # a loaded setup is freed, not released
    assert K.lib().blob_to_kzg_commitment(out, blob, C.byref(s)) == K.C_KZG_OK         # still usable: a new context is built
    assert out.raw == oracle.blob_to_kzg_commitment(blob, oracle_setup, oracle.MODE_R)[1]
    assert K.lib().lwkzg_release_context(C.byref(s)) == K.C_KZG_OK


def test_hand_built_settings_rebuilt_in_place_get_a_fresh_context(K, gpu_setup, oracle_setup):
    """the context cached for a hand-built KZGSettings is keyed by the g1_values pointer AND a digest of its first and
    last point: a caller that rewrites the array in place (here: first and last setup point exchanged) must not get the
    stale table. A genuine 32-byte FFTSettings behind fs (another producer's) is never dereferenced past its end."""
    g1 = bytearray(oracle_setup.g1_blst())
    buf = (C.c_uint8 * len(g1)).from_buffer(g1)
    fs_foreign = (C.c_uint64 * 4)(4096, 0, 0, 0)            # a c-kzg style FFTSettings that is not ours: exactly 32 bytes
    s = K.KZGSettings()
    s.fs, s.g1_values, s.g2_values = C.cast(fs_foreign, C.c_void_p), C.cast(buf, C.c_void_p), gpu_setup.s.g2_values
    one = (1).to_bytes(32, "big")
    blob_first = one + bytes(B.BYTES_PER_BLOB - 32)          # p(x) = 1        -> commitment = g1[0]
    blob_last = bytes(B.BYTES_PER_BLOB - 32) + one           # p(x) = x^4095   -> commitment = g1[4095]
    comp = oracle_setup.g1_compressed()
    out = C.create_string_buffer(48)
    assert K.lib().blob_to_kzg_commitment(out, blob_first, C.byref(s)) == K.C_KZG_OK and out.raw == comp[:48]
    assert K.lib().blob_to_kzg_commitment(out, blob_last, C.byref(s)) == K.C_KZG_OK and out.raw == comp[48 * 4095:]
    g1[:144], g1[144 * 4095:] = bytes(g1[144 * 4095:]), bytes(g1[:144])      # same address, other contents
    assert K.lib().blob_to_kzg_commitment(out, blob_first, C.byref(s)) == K.C_KZG_OK and out.raw == comp[48 * 4095:]
    assert K.lib().blob_to_kzg_commitment(out, blob_last, C.byref(s)) == K.C_KZG_OK and out.raw == comp[:48]
    assert K.lib().lwkzg_release_context(C.byref(s)) == K.C_KZG_OK      # (free_trusted_setup would free() the caller's arrays)


# ---- stage level: NTT, MSM -------------------------------------------------------------------------

def test_ntt_kernel_vs_oracle(K, gpu_setup, oracle):
    import torch
    from lambdaworks_kzg_amd import capi
    rnd = random.Random(11)
    vecs = [b"".join(rnd.randrange(R).to_bytes(32, "big") for _ in range(4096)) for _ in range(3)]
    vecs.append(b"".join((R - 1).to_bytes(32, "big") for _ in range(4096)))
    d_in = _dev(b"".join(vecs))
    d_out = torch.empty_like(d_in)
    for inverse in (False, True):
        capi.fr_ntt4096_device(d_out.data_ptr(), d_in.data_ptr(), len(vecs), inverse, gpu_setup)
        torch.cuda.synchronize()
        got = _host(d_out)
        for i, v in enumerate(vecs):
            assert got[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB] == oracle.fr_ntt4096(v, inverse=inverse)


def test_ntt_roundtrip_and_linearity_batch_256(K, gpu_setup):
    """size-independent properties at batch size: INTT(NTT(x)) == x, NTT(x + y) == NTT(x) + NTT(y) (mod r)"""
    import numpy as np
    import torch
    from lambdaworks_kzg_amd import capi
    n = 256
    x = np.frombuffer(B.synthetic_batch(8000, n), dtype=np.uint8)            # canonical (< 2^248) big-endian elements
    y = np.frombuffer(B.synthetic_batch(9000, n), dtype=np.uint8)
    d_x, d_y = torch.from_numpy(x.copy()).cuda(), torch.from_numpy(y.copy()).cuda()
    d_fx, d_fy, d_back = torch.empty_like(d_x), torch.empty_like(d_x), torch.empty_like(d_x)
    capi.fr_ntt4096_device(d_fx.data_ptr(), d_x.data_ptr(), n, False, gpu_setup)
    capi.fr_ntt4096_device(d_back.data_ptr(), d_fx.data_ptr(), n, True, gpu_setup)
    capi.fr_ntt4096_device(d_fy.data_ptr(), d_y.data_ptr(), n, False, gpu_setup)
    torch.cuda.synchronize()
    assert torch.equal(d_back, d_x)
    # linearity on a few vectors, big integers on the host
    fx, fy = _host(d_fx), _host(d_fy)
    for v in (0, 100, 255):
        sx, sy = B.blob_scalars(x[v * B.BYTES_PER_BLOB:(v + 1) * B.BYTES_PER_BLOB].tobytes()), B.blob_scalars(y[v * B.BYTES_PER_BLOB:(v + 1) * B.BYTES_PER_BLOB].tobytes())
        ssum = b"".join(((a + b) % R).to_bytes(32, "big") for a, b in zip(sx, sy))
        d_s = _dev(ssum)
        d_fs = torch.empty_like(d_s)
        capi.fr_ntt4096_device(d_fs.data_ptr(), d_s.data_ptr(), 1, False, gpu_setup)
        torch.cuda.synchronize()
        want = b"".join(((a + b) % R).to_bytes(32, "big") for a, b in
                        zip(B.blob_scalars(fx[v * B.BYTES_PER_BLOB:(v + 1) * B.BYTES_PER_BLOB]), B.blob_scalars(fy[v * B.BYTES_PER_BLOB:(v + 1) * B.BYTES_PER_BLOB])))
        assert _host(d_fs) == want


def test_msm_kernel_vs_oracle_and_closed_form(K, engine_setup, oracle, oracle_setup):
    import torch
    gpu_setup = engine_setup
    from lambdaworks_kzg_amd import capi
    rnd = random.Random(12)
    sets = [[rnd.randrange(R) for _ in range(4096)],
            [rnd.randrange(1 << 256) for _ in range(4096)],          # non-canonical: reduced mod r
            [0] * 4096,
            [1] * 4096,
            [R - 1] * 4096,                                          # every digit pattern identical
            [0] * 4095 + [5],
            [(1 << 12)] * 4096, [(1 << 12) + 1] * 4096,              # signed-digit boundary 2^(c-1) and +1
            [(1 << 13) - 1] * 4096, [(1 << 255) - 1] * 4096]
    data = b"".join(b"".join(s.to_bytes(32, "big") for s in ss) for ss in sets)
    d_in = _dev(data)
    d_out = torch.empty(48 * len(sets), dtype=torch.uint8, device="cuda")
    capi.g1_lincomb_setup_device(d_out.data_ptr(), d_in.data_ptr(), len(sets), gpu_setup)
    torch.cuda.synchronize()
    got = _host(d_out)
    for i, ss in enumerate(sets):
        assert got[48 * i:48 * i + 48] == tau_closed_form(oracle, ss), i
    # first set also against the oracle's own Pippenger (reference algorithm shape)
    rc, cm = oracle.blob_to_kzg_commitment(data[:B.BYTES_PER_BLOB], oracle_setup, oracle.MODE_R)
    assert got[:48] == cm


# ---- blob_to_kzg_commitment (a1-a7) ------------------------------------------------------------------

def test_lib_test_rs_behaviours(K, gpu_setup, oracle_setup):
    g1 = oracle_setup.g1_compressed()
    one, two = (1).to_bytes(32, "big"), (2).to_bytes(32, "big")
    blob1 = one + bytes(B.BYTES_PER_BLOB - 32)
    blobx = bytes(32) + one + bytes(B.BYTES_PER_BLOB - 64)
    # tests/lib_test.rs:19-87
    pr, y = K.compute_kzg_proof(blob1, one, gpu_setup)
    assert pr == bytes([0xc0]) + bytes(47) and y == one
    # tests/lib_test.rs:89-167
    pr, y = K.compute_kzg_proof(blobx, two, gpu_setup)
    assert y == two and pr == g1[:48]
    assert K.blob_to_kzg_commitment(blobx, gpu_setup) == g1[48:96]
    assert K.blob_to_kzg_commitment(bytes(B.BYTES_PER_BLOB), gpu_setup) == bytes([0xc0]) + bytes(47)


def test_commitment_reference_mode_vs_oracle(K, engine_setup, oracle, oracle_setup):
    gpu_setup = engine_setup
    blobs = [B.synthetic_blob(i) for i in range(5)]
    blobs += [B.make_blob(n) for n in ("pow2", "pow3", "pow5", "r_minus_1", "delta_3211", "zero", "all_ff", "r_at_2111")]
    got = K.blob_to_kzg_commitment_batch(b"".join(blobs), gpu_setup)
    for b, g in zip(blobs, got):
        rc, want = oracle.blob_to_kzg_commitment(b, oracle_setup, oracle.MODE_R)
        assert rc == 0 and g == want
    # single-blob symbol == batch symbol
    assert K.blob_to_kzg_commitment(blobs[0], gpu_setup) == got[0]


@pytest.mark.parametrize("n", [1, 2, 7, 64])
def test_commitment_batch_sizes_closed_form(K, gpu_setup, oracle, n):
    blobs = [B.synthetic_blob(100 + i) for i in range(n)]
    got = K.blob_to_kzg_commitment_batch(b"".join(blobs), gpu_setup)
    for b, g in zip(blobs, got):
        assert g == tau_closed_form(oracle, B.blob_scalars(b))


def test_commitment_ckzg_vectors(K, gpu_setup, vectors):
    K.set_mode(K.MODE_CKZG)
    n = 0
    for c in vectors["suites"]["blob_to_kzg_commitment"]:
        blob = B.make_blob(c["input"]["blob"])
        if len(blob) != B.BYTES_PER_BLOB:
            continue
        if c["output"] is None:
            with pytest.raises(K.KzgError) as e:
                K.blob_to_kzg_commitment(blob, gpu_setup)
            assert e.value.rc == K.C_KZG_BADARGS
        else:
            assert K.blob_to_kzg_commitment(blob, gpu_setup) == hx(c["output"])
        n += 1
    assert n == 8


def test_ckzg_batch_reports_first_bad(K, gpu_setup):
    K.set_mode(K.MODE_CKZG)
    blobs = B.make_blob("pow2") + B.make_blob("pow3") + B.make_blob("r_at_2111") + B.make_blob("all_ff")
    out = C.create_string_buffer(48 * 4)
    bad = C.c_size_t(99)
    rc = K.lib().lwkzg_blob_to_kzg_commitment_batch(out, blobs, 4, gpu_setup.ref(), C.byref(bad))
    assert rc == K.C_KZG_BADARGS and bad.value == 2


def test_large_host_batch_is_sliced_and_reports_first_bad(K, gpu_setup):
    """long host-pointer batches (>= 512 blobs with the direct table, more than one 1024-blob chunk on the default engine)
    go up in 512-blob slices on two streams: same bytes as the device path, and the first rejected blob is still
    reported by its index in the whole batch"""
    import torch
    K.set_mode(K.MODE_CKZG)
    n = 1100
    data = bytearray(B.synthetic_batch(3000, n, big_endian=False))
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    d_out = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    K.blob_to_kzg_commitment_batch_device(d_out.data_ptr(), d_in.data_ptr(), n, gpu_setup, None, None)
    torch.cuda.synchronize()
    assert b"".join(K.blob_to_kzg_commitment_batch(bytes(data), gpu_setup)) == _host(d_out)
    bad_blob = B.make_blob("all_ff")                        # every element >= r: not canonical in c-kzg mode
    data[317 * B.BYTES_PER_BLOB:318 * B.BYTES_PER_BLOB] = bad_blob
    data[1090 * B.BYTES_PER_BLOB:1091 * B.BYTES_PER_BLOB] = bad_blob
    out = C.create_string_buffer(48 * n)
    bad = C.c_size_t(9999)
    rc = K.lib().lwkzg_blob_to_kzg_commitment_batch(out, bytes(data), n, gpu_setup.ref(), C.byref(bad))
    assert rc == K.C_KZG_BADARGS and bad.value == 317


# ---- proofs (a8-a11) --------------------------------------------------------------------------------

def test_compute_kzg_proof_ckzg_vectors(K, gpu_setup, vectors):
    K.set_mode(K.MODE_CKZG)
    n = 0
    for c in vectors["suites"]["compute_kzg_proof"]:
        blob, z = B.make_blob(c["input"]["blob"]), hx(c["input"]["z"])
        if len(blob) != B.BYTES_PER_BLOB or len(z) != 32:
            continue
        if c["output"] is None:
            with pytest.raises(K.KzgError) as e:
                K.compute_kzg_proof(blob, z, gpu_setup)
            assert e.value.rc == K.C_KZG_BADARGS
        else:
            pr, y = K.compute_kzg_proof(blob, z, gpu_setup)
            assert pr == hx(c["output"][0]) and y == hx(c["output"][1])
        n += 1
    assert n == 42


def test_compute_blob_kzg_proof_ckzg_vectors(K, gpu_setup, vectors):
    K.set_mode(K.MODE_CKZG)
    n = 0
    for c in vectors["suites"]["compute_blob_kzg_proof"]:
        blob, cm = B.make_blob(c["input"]["blob"]), hx(c["input"]["commitment"])
        if len(blob) != B.BYTES_PER_BLOB or len(cm) != 48:
            continue
        if c["output"] is None:
            with pytest.raises(K.KzgError) as e:
                K.compute_blob_kzg_proof(blob, cm, gpu_setup)
            assert e.value.rc == K.C_KZG_BADARGS
        else:
            assert K.compute_blob_kzg_proof(blob, cm, gpu_setup) == hx(c["output"])
        n += 1
    assert n == 10


def test_proofs_reference_mode_vs_oracle(K, engine_setup, oracle, oracle_setup):
    gpu_setup = engine_setup
    blobs = [B.synthetic_blob(40 + i) for i in range(3)] + [B.make_blob("pow2"), B.make_blob("r_minus_1"), B.make_blob("zero")]
    comms = K.blob_to_kzg_commitment_batch(b"".join(blobs), gpu_setup)
    proofs = K.compute_blob_kzg_proof_batch(b"".join(blobs), b"".join(comms), gpu_setup)
    for b, cm, pr in zip(blobs, comms, proofs):
        rc, want = oracle.compute_blob_kzg_proof(b, cm, oracle_setup, oracle.MODE_R)
        assert rc == 0 and pr == want
    # SURVEY 4.3 smoke value
    assert proofs[3].hex() == ("b352a02445cc2f74ecf7bdb12380fd2debce3f352407514b8645d940a26079d9"
                               "b7167c68ac5957dc22c23d2aabbe471b")
    rnd = random.Random(3)
    zs = [rnd.randrange(R).to_bytes(32, "big") for _ in blobs]
    zs[1] = (R + 7).to_bytes(32, "big")            # reference mode reduces z
    res = K.compute_kzg_proof_batch(b"".join(blobs), b"".join(zs), gpu_setup)
    for b, z, (pr, y) in zip(blobs, zs, res):
        rc, wpr, wy = oracle.compute_kzg_proof(b, z, oracle_setup, oracle.MODE_R)
        assert rc == 0 and pr == wpr and y == wy
        # and the proof verifies (closed form for the known tau)
        rc, cm = oracle.blob_to_kzg_commitment(b, oracle_setup, oracle.MODE_R)
        zz = (int.from_bytes(z, "big") % R).to_bytes(32, "big")
        assert oracle.verify_kzg_proof_known_tau(cm, zz, y, pr, TAU, oracle.MODE_R) == (0, True)


def test_blob_proof_rejects_bad_commitment(K, gpu_setup):
    blob = B.synthetic_blob(1)
    for bad in (bytes(48), bytes([0x80]) + bytes(47), bytes([0x9f]) + b"\xff" * 47):
        with pytest.raises(K.KzgError) as e:
            K.compute_blob_kzg_proof(blob, bad, gpu_setup)
        assert e.value.rc == K.C_KZG_ERROR          # reference mode: every failure is C_KZG_ERROR
    K.set_mode(K.MODE_CKZG)
    with pytest.raises(K.KzgError) as e:
        K.compute_blob_kzg_proof(B.make_blob("pow2"), bytes(48), gpu_setup)
    assert e.value.rc == K.C_KZG_BADARGS


# ---- full-size properties (BASELINE config 2: batch up to 1024) ------------------------------------------

def test_batch_1024_device_resident_closed_form_and_determinism(K, engine_setup, oracle):
    import numpy as np
    import torch
    gpu_setup = engine_setup
    n = 1024
    data = B.synthetic_batch(1000, n)
    d_blobs = _dev(data)
    d_out = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    d_status = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    K.blob_to_kzg_commitment_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), n, gpu_setup, None, d_status.data_ptr())
    torch.cuda.synchronize()
    first = _host(d_out)
    assert int(d_status.abs().sum()) == 0
    # determinism: atomics reorder bucket contents, the canonical output must not change
    K.blob_to_kzg_commitment_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), n, gpu_setup, None, d_status.data_ptr())
    torch.cuda.synchronize()
    assert _host(d_out) == first
    # closed form for EVERY blob of the batch
    for i in range(n):
        blob = data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB]
        assert first[48 * i:48 * i + 48] == tau_closed_form(oracle, B.blob_scalars(blob)), i
    # linearity: commit(a) + commit(b) == commit(a + b) -- checked through scalars: blob c = a + b mod r
    a = B.blob_scalars(data[:B.BYTES_PER_BLOB])
    b = B.blob_scalars(data[B.BYTES_PER_BLOB:2 * B.BYTES_PER_BLOB])
    csum = b"".join(((x + y) % R).to_bytes(32, "big") for x, y in zip(a, b))
    got = K.blob_to_kzg_commitment(csum, gpu_setup)
    xa, _ = oracle.g1_decompress(first[:48])
    xb, _ = oracle.g1_decompress(first[48:96])
    s, inf = oracle.g1_add_affine(xa, False, xb, False)
    assert not inf and oracle.g1_compress(s) == got
    # partition invariance (what 1/2/4/8-GPU sharding relies on): any sub-range gives the same bytes
    from lambdaworks_kzg_amd.dist import shard_range
    for world in (2, 8):
        for r in (0, world - 1):
            st, cnt = shard_range(n, world, r)
            sub = K.blob_to_kzg_commitment_batch(data[st * B.BYTES_PER_BLOB:(st + min(cnt, 3)) * B.BYTES_PER_BLOB], gpu_setup)
            assert b"".join(sub) == first[48 * st:48 * (st + min(cnt, 3))]


def test_setup_image_export_import_roundtrip(K, gpu_setup, oracle):
    import torch
    from lambdaworks_kzg_amd import capi
    img = torch.empty(capi.setup_image_bytes(), dtype=torch.uint8, device="cuda")
    gpu_setup.export_device_image(img.data_ptr())
    ts2 = K.TrustedSetup.from_device_image(img.data_ptr())
    assert ts2.g1_values_bytes() == gpu_setup.g1_values_bytes()
    assert ts2.g2_values_bytes() == gpu_setup.g2_values_bytes()
    blob = B.synthetic_blob(9)
    assert K.blob_to_kzg_commitment(blob, ts2) == tau_closed_form(oracle, B.blob_scalars(blob))
    # what every rank > 0 of the multi-GPU bench does: build its own direct table from the imported points
    assert ts2.direct_table_bits() == gpu_setup.default_bits     # an imported setup gets the default engine like a loaded one
    ts2.enable_direct_table(14)
    data = B.synthetic_batch(40, 5)
    assert K.blob_to_kzg_commitment_batch(data, ts2) == K.blob_to_kzg_commitment_batch(data, gpu_setup)
    ts2.free()


# ---- verify side: GPU per-blob work + host pairing ----------------------------------

def test_verify_lib_test_rs_behaviours(K, gpu_setup, oracle_setup):
    # tests/lib_test.rs:19-87, 89-167 (single proofs) and :169-260 (2-blob batch), reference mode
    g1 = oracle_setup.g1_compressed()
    one, two = (1).to_bytes(32, "big"), (2).to_bytes(32, "big")
    blob1 = one + bytes(B.BYTES_PER_BLOB - 32)
    blobx = bytes(32) + one + bytes(B.BYTES_PER_BLOB - 64)
    pr1, y1 = K.compute_kzg_proof(blob1, one, gpu_setup)
    pr2, y2 = K.compute_kzg_proof(blobx, two, gpu_setup)
    c1, c2 = K.blob_to_kzg_commitment(blob1, gpu_setup), K.blob_to_kzg_commitment(blobx, gpu_setup)
    assert K.verify_kzg_proof(c1, one, y1, pr1, gpu_setup) is True
    assert K.verify_kzg_proof(c2, two, y2, pr2, gpu_setup) is True
    assert K.verify_kzg_proof(c2, two, one, pr2, gpu_setup) is False
    # the toy proofs are z-independent (infinity and G), so the Fiat-Shamir batch accepts them
    assert K.verify_blob_kzg_proof_batch(blob1 + blobx, c1 + c2, pr1 + pr2, 2, gpu_setup) is True
    assert K.verify_blob_kzg_proof_batch(blob1 + blobx, c1 + c2, pr2 + pr1, 2, gpu_setup) is False
    assert K.verify_blob_kzg_proof_batch(b"", b"", b"", 0, gpu_setup) is False       # lib.rs:538-543


def test_verify_roundtrip_reference_mode(K, gpu_setup):
    blobs = [B.synthetic_blob(70 + i) for i in range(5)]
    comms = K.blob_to_kzg_commitment_batch(b"".join(blobs), gpu_setup)
    proofs = K.compute_blob_kzg_proof_batch(b"".join(blobs), b"".join(comms), gpu_setup)
    for b, c, p in zip(blobs, comms, proofs):
        assert K.verify_blob_kzg_proof(b, c, p, gpu_setup) is True
    assert K.verify_blob_kzg_proof(blobs[0], comms[0], proofs[1], gpu_setup) is False
    assert K.verify_blob_kzg_proof_batch(b"".join(blobs), b"".join(comms), b"".join(proofs), 5, gpu_setup) is True
    bad = list(proofs)
    bad[3] = proofs[2]
    assert K.verify_blob_kzg_proof_batch(b"".join(blobs), b"".join(comms), b"".join(bad), 5, gpu_setup) is False
    with pytest.raises(K.KzgError) as e:
        K.verify_blob_kzg_proof(blobs[0], bytes(48), proofs[0], gpu_setup)
    assert e.value.rc == K.C_KZG_ERROR


def test_verify_kzg_proof_ckzg_vectors(K, gpu_setup, vectors):
    K.set_mode(K.MODE_CKZG)
    n = 0
    for c in vectors["suites"]["verify_kzg_proof"]:
        i = c["input"]
        cm, z, y, pr = hx(i["commitment"]), hx(i["z"]), hx(i["y"]), hx(i["proof"])
        if (len(cm), len(z), len(y), len(pr)) != (48, 32, 32, 48):
            continue
        if c["output"] is None:
            with pytest.raises(K.KzgError) as e:
                K.verify_kzg_proof(cm, z, y, pr, gpu_setup)
            assert e.value.rc == K.C_KZG_BADARGS
        else:
            assert K.verify_kzg_proof(cm, z, y, pr, gpu_setup) is c["output"], c["case"]
        n += 1
    assert n == 85


def test_verify_blob_kzg_proof_ckzg_vectors(K, gpu_setup, vectors):
    K.set_mode(K.MODE_CKZG)
    n = 0
    for c in vectors["suites"]["verify_blob_kzg_proof"]:
        i = c["input"]
        blob, cm, pr = B.make_blob(i["blob"]), hx(i["commitment"]), hx(i["proof"])
        if (len(blob), len(cm), len(pr)) != (B.BYTES_PER_BLOB, 48, 48):
            continue
        if c["output"] is None:
            with pytest.raises(K.KzgError) as e:
                K.verify_blob_kzg_proof(blob, cm, pr, gpu_setup)
            assert e.value.rc == K.C_KZG_BADARGS
        else:
            assert K.verify_blob_kzg_proof(blob, cm, pr, gpu_setup) is c["output"], c["case"]
        n += 1
    assert n >= 18


def test_verify_blob_kzg_proof_batch_ckzg_vectors(K, gpu_setup, vectors):
    K.set_mode(K.MODE_CKZG)
    n = 0
    for c in vectors["suites"]["verify_blob_kzg_proof_batch"]:
        i = c["input"]
        blobs = [B.make_blob(b) for b in i["blobs"]]
        cms, prs = [hx(x) for x in i["commitments"]], [hx(x) for x in i["proofs"]]
        if any(len(b) != B.BYTES_PER_BLOB for b in blobs) or any(len(x) != 48 for x in cms + prs) or \
                not (len(blobs) == len(cms) == len(prs)):
            assert c["output"] is None      # wrong lengths / counts: not expressible through the C ABI
            continue
        k = len(blobs)
        if c["output"] is None:
            with pytest.raises(K.KzgError) as e:
                K.verify_blob_kzg_proof_batch(b"".join(blobs), b"".join(cms), b"".join(prs), k, gpu_setup)
            assert e.value.rc == K.C_KZG_BADARGS
        else:  # incl. the empty batch (case a271b78b8e869d69): c-kzg accepts it, and so does mode C here
            assert K.verify_blob_kzg_proof_batch(b"".join(blobs), b"".join(cms), b"".join(prs), k, gpu_setup) is c["output"], c["case"]
        n += 1
    assert n >= 10


def test_tiled_long_msm_2_pow_20(K, gpu_setup, oracle):
    """BASELINE configs[4]: 2^20-term MSM over the setup tiled 256x; closed form [sum_k s_k tau^(k mod 4096)] G.
    Also the sharded form: per-shard partial sums added on the host give the same bytes."""
    import numpy as np
    import torch
    from lambdaworks_kzg_amd import capi
    from lambdaworks_kzg_amd.dist import shard_range
    tiles = 256
    data = B.synthetic_batch(5000, tiles)                      # 2^20 canonical 248-bit scalars, big-endian
    d_sc = _dev(data)
    d_out = torch.empty(48, dtype=torch.uint8, device="cuda")
    capi.g1_msm_tiled_device(d_out.data_ptr(), d_sc.data_ptr(), tiles * 4096, gpu_setup)
    torch.cuda.synchronize()
    got = _host(d_out)
    acc = 0
    pw = [pow(TAU, i, R) for i in range(4096)]
    arr = np.frombuffer(data, dtype=np.uint8).reshape(tiles, 4096, 32)
    for t in range(tiles):
        sc = B.blob_scalars(arr[t].tobytes())
        acc = (acc + sum(s * p for s, p in zip(sc, pw))) % R
    assert got == oracle.g1_generator_mul(acc)
    # 8-way sharding of the tiles
    parts = []
    for r in range(8):
        st, cnt = shard_range(tiles, 8, r)
        sub = _dev(data[st * B.BYTES_PER_BLOB:(st + cnt) * B.BYTES_PER_BLOB])
        capi.g1_msm_tiled_device(d_out.data_ptr(), sub.data_ptr(), cnt * 4096, gpu_setup)
        torch.cuda.synchronize()
        parts.append(_host(d_out))
    assert capi.g1_sum_compressed(b"".join(parts)) == got
    assert capi.g1_sum_compressed(b"") == bytes([0xc0]) + bytes(47)


def test_load_free_cycles_and_two_settings(K, oracle, oracle_setup):
    """load/free repeatedly (no leaked device state), and two live settings objects used alternately"""
    import torch
    from conftest import SETUP_PATH
    def cycle():
        ts = K.TrustedSetup.from_file(SETUP_PATH)
        assert K.blob_to_kzg_commitment(B.synthetic_blob(1), ts) == tau_closed_form(oracle, B.blob_scalars(B.synthetic_blob(1)))
        ts.free()
        torch.cuda.synchronize()
    for _ in range(2):       # the runtime's own stream / signal pools fill during the first two cycles and stay (tools/leak_check.py)
        cycle()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(4):
        cycle()
    assert torch.cuda.mem_get_info()[0] >= free0 - (32 << 20)      # nothing left behind per cycle
    a, b = K.TrustedSetup.from_file(SETUP_PATH), K.TrustedSetup.from_bytes(oracle_setup.g1_compressed(), oracle_setup.g2_compressed())
    blob = B.synthetic_blob(2)
    assert K.blob_to_kzg_commitment(blob, a) == K.blob_to_kzg_commitment(blob, b)
    a.free()
    assert K.blob_to_kzg_commitment(blob, b) == tau_closed_form(oracle, B.blob_scalars(blob))
    b.free()


def test_adversarial_blobs_closed_form(K, engine_setup, oracle):
    """digit patterns that stress the bucket machinery: all scalars equal (20 buckets of 4096 entries), two values
    alternating, a single non-zero scalar, scalars whose every window is the signed-digit boundary"""
    gpu_setup = engine_setup
    c = 13
    boundary = sum((1 << (c - 1)) << (c * j) for j in range(19))        # every window = 2^(c-1)
    boundary1 = sum(((1 << (c - 1)) + 1) << (c * j) for j in range(19))  # every window = 2^(c-1)+1 -> negative digits + carries
    sets = [[R - 1] * 4096,
            [5, R - 5] * 2048,
            [0] * 1234 + [R - 2] + [0] * 2861,
            [boundary % R] * 4096,
            [boundary1 % R] * 4096,
            [(1 << 247) | 1] * 4096,
            list(range(1, 4097))]
    # heavy buckets (more than 64 entries) BETWEEN light ones, at every position of the entry array: the digit sort's staged copy-out must
    # step over the slots heavy buckets wrote directly (round 4: it once overwrote them -- the rows of the inverse DFT that
    # lagrange_prepare commits look like this: powers of a root of small order are a few distinct values repeated)
    rnd = random.Random(77)
    few = [rnd.randrange(R) for _ in range(70)]
    sets += [[few[k % 4] for k in range(4096)],                              # 4 distinct values: 80 buckets of 1024 entries
             [few[k % 63] for k in range(4096)],                             # 63 values: buckets of 65-66 entries (just heavy)
             [few[k % 64] for k in range(4096)],                             # 64 values: buckets of exactly 64 (the last light size)
             [few[k % 70] if k % 3 else rnd.randrange(R) for k in range(4096)],   # heavy and light side by side
             [pow(few[0], k, R) if k % 2 else few[1] for k in range(4096)]]  # one value 2048 times + 2048 different ones
    blobs = [b"".join(s.to_bytes(32, "big") for s in ss) for ss in sets]
    got = K.blob_to_kzg_commitment_batch(b"".join(blobs), gpu_setup)
    for ss, g in zip(sets, got):
        assert g == tau_closed_form(oracle, ss)


def test_ckzg_mode_random_blobs_vs_oracle(K, engine_setup, oracle, oracle_setup):
    """c-kzg semantics on random canonical little-endian blobs (not only the formula blobs of the vectors):
    commitment, blob proof and point proof against the oracle, then verification."""
    K.set_mode(K.MODE_CKZG)
    gpu_setup = engine_setup
    blobs = [B.synthetic_blob(300 + i, big_endian=False) for i in range(3)]
    comms = K.blob_to_kzg_commitment_batch(b"".join(blobs), gpu_setup)
    proofs = K.compute_blob_kzg_proof_batch(b"".join(blobs), b"".join(comms), gpu_setup)
    rnd = random.Random(17)
    for b, c, p in zip(blobs, comms, proofs):
        assert oracle.blob_to_kzg_commitment(b, oracle_setup, oracle.MODE_C) == (0, c)
        assert oracle.compute_blob_kzg_proof(b, c, oracle_setup, oracle.MODE_C) == (0, p)
        z = rnd.randrange(R).to_bytes(32, "little")
        pr, y = K.compute_kzg_proof(b, z, gpu_setup)
        assert oracle.compute_kzg_proof(b, z, oracle_setup, oracle.MODE_C) == (0, pr, y)
        assert K.verify_kzg_proof(c, z, y, pr, gpu_setup) is True
        assert K.verify_blob_kzg_proof(b, c, p, gpu_setup) is True
    assert K.verify_blob_kzg_proof_batch(b"".join(blobs), b"".join(comms), b"".join(proofs), 3, gpu_setup) is True


def test_c_consumer_links_and_matches_oracle(K, oracle, oracle_setup, tmp_path):
    """A plain C program (gcc, the public header, -llambdaworks_kzg) drives the nine symbols like the reference's
    fuzz harnesses do; its output is compared with the oracle."""
    import os
    import subprocess
    from conftest import ROOT, SETUP_PATH
    lib_dir = os.path.join(ROOT, "lambdaworks_kzg_amd", "lib")
    exe = str(tmp_path / "harness")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_harness.c"), "-o", exe, "-L", lib_dir, "-llambdaworks_kzg",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    blob = B.synthetic_blob(77)
    bpath = tmp_path / "blob.bin"
    bpath.write_bytes(blob)
    env = dict(os.environ)
    env.pop("LWKZG_MODE", None)
    out = subprocess.check_output([exe, SETUP_PATH, str(bpath)], env=env, timeout=1200).decode().split("\n")
    kv = dict(l.split(" ", 1) for l in out if " " in l)
    rc, cm = oracle.blob_to_kzg_commitment(blob, oracle_setup, oracle.MODE_R)
    assert kv["commitment"] == cm.hex()
    assert kv["blob_proof"] == oracle.compute_blob_kzg_proof(blob, cm, oracle_setup, oracle.MODE_R)[1].hex()
    z = (2).to_bytes(32, "big")
    rc, pr, y = oracle.compute_kzg_proof(blob, z, oracle_setup, oracle.MODE_R)
    assert kv["proof"] == pr.hex() and kv["y"] == y.hex()
    assert kv["verify_blob"] == "1" and kv["verify"] == "1" and kv["verify_wrong_y"] == "0"
    assert kv["g1_0_x_limb0"] == "17f1d3a73197d794"


def test_reference_lib_test_rs_mirror_in_c(K, tmp_path):
    """tests/lib_test_mirror.c: the reference's own integration tests (tests/lib_test.rs) as a C program against the
    drop-in library -- all nine symbols, incl. load_trusted_setup from bytes and verify_blob_kzg_proof_batch"""
    import os
    import subprocess
    from conftest import ROOT
    lib_dir = os.path.join(ROOT, "lambdaworks_kzg_amd", "lib")
    exe = str(tmp_path / "lib_test_mirror")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "lib_test_mirror.c"), "-o", exe, "-L", lib_dir, "-llambdaworks_kzg",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ)
    env.pop("LWKZG_MODE", None)
    out = subprocess.run([exe, SETUP_PATH], env=env, timeout=1200, capture_output=True)
    assert out.returncode == 0, out.stderr.decode()
    assert b"all assertions held" in out.stdout


def test_noncanonical_infinity_commitment_takes_gpu_hash_fallback(K, gpu_setup, oracle, oracle_setup):
    """decompress_g1_point does not inspect the remaining bits of an infinity encoding (compression.rs:73-75) and
    compute_challenge hashes the RE-compressed point (utils.rs:138): a commitment 0xc0 | junk must hash as c0 00..00.
    On the host-pointer path this is the case where the host digest (made from the caller's bytes) is discarded and
    the GPU hash over the canonical bytes is used."""
    blob = B.synthetic_blob(55)
    junk = bytes([0xc0]) + bytes(range(1, 48))
    got = K.compute_blob_kzg_proof(blob, junk, gpu_setup)
    assert oracle.compute_blob_kzg_proof(blob, junk, oracle_setup, oracle.MODE_R) == (0, got)
    assert got == K.compute_blob_kzg_proof(blob, bytes([0xc0]) + bytes(47), gpu_setup)
    # mixed batch: one canonical, one not
    c0 = K.blob_to_kzg_commitment(blob, gpu_setup)
    both = K.compute_blob_kzg_proof_batch(blob + blob, c0 + junk, gpu_setup)
    assert both[1] == got and oracle.compute_blob_kzg_proof(blob, c0, oracle_setup, oracle.MODE_R) == (0, both[0])


# ---- opt-in direct (giant table) MSM: bit-identical to the default path --------------------------------------

# 13 bits is what a plain load selects on an empty device (gpu_setup); the other widths are opted into here. The two
# widest tables (135 / 240 GB) do not fit beside the session's default table, which steps aside while they are tested.
@pytest.fixture(scope="module", params=[10, 12, 14, 15, 16])
def direct_setup(request, K, gpu_setup, bucket_setup):
    import torch
    wide = request.param >= 15
    if wide:
        gpu_setup.enable_direct_table(0)
    ts = K.TrustedSetup.from_file(SETUP_PATH)
    assert ts.direct_table_bits() in (10, 11, 12, 13)           # the library's own choice (engine.hip: direct_from_env)
    try:
        ts.enable_direct_table(request.param)
    except K.KzgError as e:
        ts.free()
        if wide:
            gpu_setup.enable_direct_table(gpu_setup.default_bits)
        assert e.rc == K.C_KZG_MALLOC
        from lambdaworks_kzg_amd import capi
        free_b = torch.cuda.mem_get_info()[0]
        # on a device with room for it, "does not fit" is a leak somewhere in this suite, not a reason to skip
        assert free_b < capi.direct_table_bytes(request.param) + (12 << 30), (request.param, free_b)
        pytest.skip("direct table of width %d does not fit on this device (%d GB free)" % (request.param, free_b >> 30))
    assert ts.direct_table_bits() == request.param
    assert ts.direct_row_bytes() in (112, 128)                   # aligned rows when they leave headroom, packed otherwise
    yield ts, request.param
    ts.enable_direct_table(0)
    assert ts.direct_table_bits() == 0
    # on the bucket engine the same object still answers
    blob = B.synthetic_blob(77)
    assert K.blob_to_kzg_commitment(blob, ts) == K.blob_to_kzg_commitment(blob, bucket_setup)
    ts.free()
    torch.cuda.empty_cache()
    if wide:
        gpu_setup.enable_direct_table(gpu_setup.default_bits)


def test_direct_table_bad_width(K, gpu_setup):
    for bad in (9, 17, -1, 64):
        with pytest.raises(K.KzgError) as e:
            gpu_setup.enable_direct_table(bad)
        assert e.value.rc == K.C_KZG_BADARGS
        assert gpu_setup.direct_table_bits() == gpu_setup.default_bits      # the engine it had stays in place
    nw = K.lib().lwkzg_direct_num_windows
    assert [nw(b) for b in (9, 10, 11, 12, 13, 14, 15, 16, 17)] == [0, 26, 24, 22, 20, 19, 17, 16, 0]


def test_direct_adversarial_digits_closed_form(K, direct_setup, oracle):
    ts, c = direct_setup
    nw = (255 + c - 1) // c
    half = 1 << (c - 1)
    boundary = sum(half << (c * j) for j in range(nw - 1))            # every signed window = 2^(c-1): last positive row
    boundary1 = sum((half + 1) << (c * j) for j in range(nw - 1))     # every signed window negative, carries ripple up
    ones = (1 << (c * (nw - 1))) - 1                                  # all-ones: digits -1 then 0,0,... with a final carry
    sets = [[R - 1] * 4096,
            [5, R - 5] * 2048,
            [0] * 1234 + [R - 2] + [0] * 2861,
            [0] * 4096,
            [boundary % R] * 4096,
            [boundary1 % R] * 4096,
            [ones % R] * 4096,
            [(1 << 254) + ones] * 4096 if (1 << 254) + ones < R else [(1 << 254)] * 4096,
            [(1 << 247) | 1] * 4096,
            list(range(1, 4097))]
    blobs = [b"".join(s.to_bytes(32, "big") for s in ss) for ss in sets]
    got = K.blob_to_kzg_commitment_batch(b"".join(blobs), ts)
    for ss, g in zip(sets, got):
        assert g == tau_closed_form(oracle, ss)


def _one_colliding_lane_blob(rnd, c, negate):
    """Random full-range scalars, except that ONE lane of the direct kernel meets a table row equal to (negate: opposite to) its
    accumulator, once, late: with P_k = [tau^k]G, s_i = +-e tau^2048 (i < 256: the lane's first scalar in every launch geometry),
    s_(i + 256 m) = 0 for m = 1 .. 7 (skipped), s_(i + 2048) = e -- after the 255 bits of s_i the lane holds +-[e]P_(i + 2048) and the
    next row it gathers is [e]P_(i + 2048)."""
    ss = [rnd.randrange(R) for _ in range(4096)]
    i = rnd.randrange(256)
    e = rnd.randrange(1, 1 << (c - 1))
    for m in range(1, 8):
        ss[i + 256 * m] = 0
    ss[i + 2048] = e
    ss[i] = e * pow(TAU, 2048, R) % R
    if negate:
        ss[i] = (R - ss[i]) % R
    return ss


@pytest.mark.parametrize("n", [8, 40, 200, 1024])
def test_direct_exactly_one_colliding_lane_per_blob(K, direct_setup, oracle, n):
    """the hand-scheduled accumulation has no P = +-Q branches; a lane that meets one must raise its blob's redo flag so
    that the complete-branches kernel recomputes the blob. One such lane per blob, with the accumulator at its loop bounds (the
    17th+ addition of the lane). Every commitment against the closed form. (The round-3 stream, whose flag test dropped the carry
    of k MOD0 for k >= 17, passes this test too: with the row coordinates as the table build leaves them -- Montgomery outputs,
    below p but for one in about a thousand -- P = U2 - X1 stays under 17 p; the missed case needs a row coordinate >= p, which
    tests/test_direct_asm_cpu.py produces on the simulated lane. What this test pins is the whole redo path, one lane at a time.)"""
    ts, c = direct_setup
    rnd = random.Random(4100 + n + c)
    sets = [_one_colliding_lane_blob(rnd, c, negate=(b % 2 == 1)) for b in range(n)]
    data = b"".join(b"".join(x.to_bytes(32, "big") for x in ss) for ss in sets)
    got = K.blob_to_kzg_commitment_batch(data, ts)
    bad = [b for b in range(n) if got[b] != tau_closed_form(oracle, sets[b])]
    assert not bad, (len(bad), bad[:8])


@pytest.mark.parametrize("n", [1, 3, 64, 200, 700, 1024])
def test_direct_commitments_match_default_path(K, direct_setup, bucket_setup, oracle, n):
    """every launch geometry of the direct kernel (16 .. 1 workgroups per blob): every commitment against the tau closed
    form, and against the bucket engine (an independent algorithm)"""
    ts, _ = direct_setup
    gpu_setup = bucket_setup
    data = B.synthetic_batch(5000 + n, n)
    want = K.blob_to_kzg_commitment_batch(data, gpu_setup)
    got = K.blob_to_kzg_commitment_batch(data, ts)
    assert got == want
    for i in range(n):
        assert got[i] == tau_closed_form(oracle, B.blob_scalars(data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB])), i
    if n >= 700:    # the sliced proof paths on the direct engine (700: the 128 + 384 + rest schedule)
        cm = b"".join(got)
        assert K.compute_blob_kzg_proof_batch(data, cm, ts) == K.compute_blob_kzg_proof_batch(data, cm, gpu_setup)
        zs = data[:32 * n]
        assert K.compute_kzg_proof_batch(data, zs, ts) == K.compute_kzg_proof_batch(data, zs, gpu_setup)


@pytest.mark.parametrize("n,first,mode_c", [(70, 21200, False), (256, 22000, False), (256, 22000, True), (1024, 23000, False)])
def test_direct_device_resident_blob_proofs_every_blob_vs_oracle(K, direct_setup, oracle, n, first, mode_c):
    """the path `bench.py --op blob_proof` times, on every direct-table width: all proofs of the batch against the CPU
    oracle (the oracle's results are shared with tests/test_gpu_proof_parity.py, which runs the default and bucket engines)"""
    from proof_cases import oracle_batch
    from test_gpu_proof_parity import device_commit_and_prove
    ts, _ = direct_setup
    K.set_mode(K.MODE_CKZG if mode_c else K.MODE_REFERENCE)
    blobs, want_c, want_p = oracle_batch(oracle, first, n, mode_c)
    comms, proofs = device_commit_and_prove(K, ts, b"".join(blobs), n)
    for i in range(n):
        assert comms[48 * i:48 * i + 48] == want_c[i], ("commitment", i)
        assert proofs[48 * i:48 * i + 48] == want_p[i], ("proof", i)


def test_direct_proofs_both_modes_match_default_path(K, direct_setup, bucket_setup, oracle, oracle_setup):
    ts, _ = direct_setup
    gpu_setup = bucket_setup
    rnd = random.Random(23)
    for mode, be, omode in ((K.MODE_REFERENCE, True, oracle.MODE_R), (K.MODE_CKZG, False, oracle.MODE_C)):
        K.set_mode(mode)
        blobs = [B.synthetic_blob(900 + i, big_endian=be) for i in range(5)]
        joined = b"".join(blobs)
        comms = K.blob_to_kzg_commitment_batch(joined, ts)
        assert comms == K.blob_to_kzg_commitment_batch(joined, gpu_setup)
        proofs = K.compute_blob_kzg_proof_batch(joined, b"".join(comms), ts)
        assert proofs == K.compute_blob_kzg_proof_batch(joined, b"".join(comms), gpu_setup)
        assert oracle.compute_blob_kzg_proof(blobs[0], comms[0], oracle_setup, omode) == (0, proofs[0])
        z = rnd.randrange(R).to_bytes(32, "big" if be else "little")
        pr, y = K.compute_kzg_proof(blobs[1], z, ts)
        assert oracle.compute_kzg_proof(blobs[1], z, oracle_setup, omode) == (0, pr, y)
        assert K.verify_blob_kzg_proof_batch(joined, b"".join(comms), b"".join(proofs), 5, ts) is True


def test_direct_tiled_long_msm(K, direct_setup, bucket_setup, oracle):
    """2^18-term tiled MSM through the direct table == the bucket path == the closed form"""
    import numpy as np
    import torch
    from lambdaworks_kzg_amd import capi
    ts, _ = direct_setup
    gpu_setup = bucket_setup
    tiles = 64
    data = B.synthetic_batch(7000, tiles)
    d_sc = _dev(data)
    d_out = torch.empty(48, dtype=torch.uint8, device="cuda")
    capi.g1_msm_tiled_device(d_out.data_ptr(), d_sc.data_ptr(), tiles * 4096, ts)
    torch.cuda.synchronize()
    got = _host(d_out)
    capi.g1_msm_tiled_device(d_out.data_ptr(), d_sc.data_ptr(), tiles * 4096, gpu_setup)
    torch.cuda.synchronize()
    assert _host(d_out) == got
    pw = [pow(TAU, i, R) for i in range(4096)]
    arr = np.frombuffer(data, dtype=np.uint8).reshape(tiles, 4096, 32)
    acc = 0
    for t in range(tiles):
        acc = (acc + sum(s * p for s, p in zip(B.blob_scalars(arr[t].tobytes()), pw))) % R
    assert got == oracle.g1_generator_mul(acc)



def test_verify_long_batch_pipelined_path(K, gpu_setup):
    """more than one chunk (1024 blobs): up-front validation of all points + sliced, pipelined per-blob pass. Accepts a
    valid batch, rejects one with a single wrong proof far into it, errors on an invalid commitment encoding."""
    n = 1100
    data = B.synthetic_batch(4000, n)
    comms = b"".join(K.blob_to_kzg_commitment_batch(data, gpu_setup))
    proofs = b"".join(K.compute_blob_kzg_proof_batch(data, comms, gpu_setup))
    assert K.verify_blob_kzg_proof_batch(data, comms, proofs, n, gpu_setup) is True
    wrong = bytearray(proofs)
    wrong[48 * 1077:48 * 1078] = proofs[48 * 3:48 * 4]          # a valid G1 point, but not this blob's proof
    assert K.verify_blob_kzg_proof_batch(data, comms, bytes(wrong), n, gpu_setup) is False
    badc = bytearray(comms)
    badc[48 * 1050] &= 0x7f                                      # compression flag cleared: not a valid encoding
    with pytest.raises(K.KzgError) as e:
        K.verify_blob_kzg_proof_batch(data, bytes(badc), proofs, n, gpu_setup)
    assert e.value.rc == K.C_KZG_ERROR
    # r06: such a batch lands in one device buffer, the head hashed by the GPU and the tail by the host threads (engine.hip:
    # verify_prepare_staged) -- the same three outcomes with the defect in the HEAD of the batch, and with an invalid proof encoding
    wrong = bytearray(proofs)
    wrong[48 * 7:48 * 8] = proofs[48 * 300:48 * 301]
    assert K.verify_blob_kzg_proof_batch(data, comms, bytes(wrong), n, gpu_setup) is False
    badc = bytearray(comms)
    badc[48 * 33] &= 0x7f
    with pytest.raises(K.KzgError) as e:
        K.verify_blob_kzg_proof_batch(data, bytes(badc), proofs, n, gpu_setup)
    assert e.value.rc == K.C_KZG_ERROR
    badp = bytearray(proofs)
    badp[48 * 600 + 47] ^= 1                                     # x moves off the curve or out of the subgroup
    with pytest.raises(K.KzgError) as e:
        K.verify_blob_kzg_proof_batch(data, comms, bytes(badp), n, gpu_setup)
    assert e.value.rc == K.C_KZG_ERROR
    swapped = data[:B.BYTES_PER_BLOB * 5] + data[B.BYTES_PER_BLOB * 6:B.BYTES_PER_BLOB * 7] + data[B.BYTES_PER_BLOB * 5:B.BYTES_PER_BLOB * 6] + data[B.BYTES_PER_BLOB * 7:]
    assert K.verify_blob_kzg_proof_batch(swapped, comms, proofs, n, gpu_setup) is False
    assert K.verify_blob_kzg_proof_batch(data, comms, proofs, n, gpu_setup) is True   # and the settings object is none the worse for it


def test_direct_table_that_does_not_fit_leaves_the_engine_in_place(K, direct_setup, gpu_setup):
    """a second settings object loaded beside the 240 GB table gets a narrower default engine (a quarter of what is
    still free) or the bucket engine; asking it for a table the device can no longer hold returns C_KZG_MALLOC and
    leaves the engine it had (only meaningful while the 240 GB table of the fixture is resident)"""
    ts, bits = direct_setup
    if bits != 16:
        pytest.skip("needs the 240 GB table resident")
    other = K.TrustedSetup.from_file(SETUP_PATH)
    had = other.direct_table_bits()
    assert had in (0, 10, 11, 12) and had < gpu_setup.default_bits
    with pytest.raises(K.KzgError) as e:
        other.enable_direct_table(16)
    assert e.value.rc == K.C_KZG_MALLOC
    assert other.direct_table_bits() == had
    blob = B.synthetic_blob(31337)
    assert K.blob_to_kzg_commitment(blob, other) == K.blob_to_kzg_commitment(blob, ts) == K.blob_to_kzg_commitment(blob, gpu_setup)
    other.free()
    other.free()


def test_concurrent_callers_on_one_settings_object(K, gpu_setup):
    """the reference's KZGSettings is read-only after load, so callers may share it across threads;
    here calls serialise on the context's mutexes: four threads mixing commitments, proofs and verifications must get
    exactly the single-threaded answers, with no deadlock between the verify-side and the engine locks"""
    import threading
    blobs = [B.synthetic_blob(600 + i) for i in range(6)]
    comms = [K.blob_to_kzg_commitment(b, gpu_setup) for b in blobs]
    proofs = [K.compute_blob_kzg_proof(b, c, gpu_setup) for b, c in zip(blobs, comms)]
    joined = (b"".join(blobs), b"".join(comms), b"".join(proofs))
    errors = []

    def worker(seed):
        try:
            for it in range(6):
                i = (seed + it) % 6
                assert K.blob_to_kzg_commitment(blobs[i], gpu_setup) == comms[i]
                assert K.compute_blob_kzg_proof(blobs[i], comms[i], gpu_setup) == proofs[i]
                assert K.verify_blob_kzg_proof(blobs[i], comms[i], proofs[i], gpu_setup) is True
                assert K.verify_blob_kzg_proof(blobs[i], comms[i], proofs[(i + 1) % 6], gpu_setup) is False
                assert K.verify_blob_kzg_proof_batch(joined[0], joined[1], joined[2], 6, gpu_setup) is True
                assert K.blob_to_kzg_commitment_batch(joined[0], gpu_setup) == comms
        except Exception as e:      # pragma: no cover - reported below
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "deadlock"
    assert errors == []


def test_long_host_proof_batches_are_sliced(K, gpu_setup):
    """host-pointer proof batches of >= 512 blobs stream through in 512-blob slices on two streams with the commitments
    validated once up front: same bytes as short calls over the same inputs; a non-canonical infinity encoding deep in
    the batch still takes the hash over the canonical bytes; the first rejected input is reported by its index in the
    whole batch"""
    n, piece = 1100, 275
    K.set_mode(K.MODE_REFERENCE)
    data = B.synthetic_batch(5000, n)
    comms = bytearray(b"".join(K.blob_to_kzg_commitment_batch(data, gpu_setup)))
    junk = bytes([0xc0]) + bytes(range(1, 48))
    comms[48 * 700:48 * 701] = junk                      # valid (infinity), not canonical, second slice
    comms = bytes(comms)
    blob = lambda i: data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB]
    want = []
    for lo in range(0, n, piece):                        # short calls: the single-pass path
        want += K.compute_blob_kzg_proof_batch(data[lo * B.BYTES_PER_BLOB:(lo + piece) * B.BYTES_PER_BLOB],
                                               comms[48 * lo:48 * (lo + piece)], gpu_setup)
    got = K.compute_blob_kzg_proof_batch(data, comms, gpu_setup)
    assert got == want
    assert got[700] == K.compute_blob_kzg_proof(blob(700), bytes([0xc0]) + bytes(47), gpu_setup)
    for i in (0, 511, 512, 1099):
        assert K.verify_blob_kzg_proof(blob(i), comms[48 * i:48 * i + 48], got[i], gpu_setup) is True
    # invalid commitment far into the batch, in front of another one
    badc = bytearray(comms)
    badc[48 * 900] &= 0x7f
    badc[48 * 1000] &= 0x7f
    out, bad = C.create_string_buffer(48 * n), C.c_size_t(9999)
    rc = K.lib().lwkzg_compute_blob_kzg_proof_batch(out, data, bytes(badc), n, gpu_setup.ref(), C.byref(bad))
    assert rc == K.C_KZG_ERROR and bad.value == 900
    assert out.raw == bytes(48 * n)                      # nothing written on failure

    # evaluation proofs at caller-chosen points
    zs = b"".join(B.synthetic_blob(7000 + i)[:32] for i in range(n))
    wantp = []
    for lo in range(0, n, piece):
        wantp += K.compute_kzg_proof_batch(data[lo * B.BYTES_PER_BLOB:(lo + piece) * B.BYTES_PER_BLOB],
                                           zs[32 * lo:32 * (lo + piece)], gpu_setup)
    gotp = K.compute_kzg_proof_batch(data, zs, gpu_setup)
    assert gotp == wantp
    for i in (3, 600, 1099):
        assert K.verify_kzg_proof(comms[48 * i:48 * i + 48], zs[32 * i:32 * i + 32], gotp[i][1], gotp[i][0], gpu_setup) is True

    # c-kzg mode: non-canonical blob element / evaluation point -> BADARGS with the first index
    K.set_mode(K.MODE_CKZG)
    try:
        data_le = bytearray(B.synthetic_batch(5100, n, big_endian=False))
        zs_le = bytearray(b"".join(B.synthetic_blob(7100 + i, big_endian=False)[:32] for i in range(n)))
        good = K.compute_kzg_proof_batch(bytes(data_le), bytes(zs_le), gpu_setup)
        short = K.compute_kzg_proof_batch(bytes(data_le[:300 * B.BYTES_PER_BLOB]), bytes(zs_le[:32 * 300]), gpu_setup)
        assert good[:300] == short
        zs_bad = bytearray(zs_le)
        zs_bad[32 * 800:32 * 801] = b"\xff" * 32
        data_bad = bytearray(data_le)
        data_bad[1050 * B.BYTES_PER_BLOB:1051 * B.BYTES_PER_BLOB] = B.make_blob("all_ff")
        outp, outy = C.create_string_buffer(48 * n), C.create_string_buffer(32 * n)
        rc = K.lib().lwkzg_compute_kzg_proof_batch(outp, outy, bytes(data_bad), bytes(zs_bad), n, gpu_setup.ref(), C.byref(bad))
        assert rc == K.C_KZG_BADARGS and bad.value == 800
        rc = K.lib().lwkzg_compute_kzg_proof_batch(outp, outy, bytes(data_bad), bytes(zs_le), n, gpu_setup.ref(), C.byref(bad))
        assert rc == K.C_KZG_BADARGS and bad.value == 1050
    finally:
        K.set_mode(K.MODE_REFERENCE)


def test_host_proof_batch_below_one_chunk_short_first_slices(K, gpu_setup):
    """512 .. 1023 blobs: slices of 128, 384 and the rest; same bytes as short single-pass calls"""
    n, piece = 700, 350
    data = B.synthetic_batch(8000, n)
    comms = b"".join(K.blob_to_kzg_commitment_batch(data, gpu_setup))
    zs = b"".join(B.synthetic_blob(8800 + i)[:32] for i in range(n))
    want_b, want_p = [], []
    for lo in range(0, n, piece):
        sl = data[lo * B.BYTES_PER_BLOB:(lo + piece) * B.BYTES_PER_BLOB]
        want_b += K.compute_blob_kzg_proof_batch(sl, comms[48 * lo:48 * (lo + piece)], gpu_setup)
        want_p += K.compute_kzg_proof_batch(sl, zs[32 * lo:32 * (lo + piece)], gpu_setup)
    assert K.compute_blob_kzg_proof_batch(data, comms, gpu_setup) == want_b
    assert K.compute_kzg_proof_batch(data, zs, gpu_setup) == want_p


@pytest.mark.parametrize("n", [2, 7, 64, 65, 200, 700])
def test_verify_batch_every_path_random_tamper(K, gpu_setup, oracle, n):
    """batch verification through each of its internal routes (host-thread validation + linear combinations up to 64
    blobs, validation kernels + piece-split GPU linear combinations above): accepts the honest batch; rejects it with
    one proof, one commitment or one blob swapped for another valid one at a random place; points at infinity
    (zero blob, zero quotient) take part"""
    rnd = random.Random(1000 + n)
    blobs = [B.synthetic_blob(12000 + 31 * n + i) for i in range(n)]
    blobs[rnd.randrange(n)] = bytes(B.BYTES_PER_BLOB)                     # commitment and proof at infinity
    const = bytearray(B.BYTES_PER_BLOB)
    const[31] = 5
    blobs[rnd.randrange(1, n) if blobs[0] == bytes(B.BYTES_PER_BLOB) else 0] = bytes(const)   # constant polynomial: proof at infinity
    data = b"".join(blobs)
    comms = K.blob_to_kzg_commitment_batch(data, gpu_setup)
    proofs = K.compute_blob_kzg_proof_batch(data, b"".join(comms), gpu_setup)
    assert bytes([0xc0]) + bytes(47) in proofs
    cj, pj = b"".join(comms), b"".join(proofs)
    assert K.verify_blob_kzg_proof_batch(data, cj, pj, n, gpu_setup) is True
    other = oracle.g1_generator_mul(rnd.randrange(2, R))                  # a valid G1 point that proves nothing here
    i = rnd.randrange(n)
    bad_p = pj[:48 * i] + other + pj[48 * i + 48:]
    assert K.verify_blob_kzg_proof_batch(data, cj, bad_p, n, gpu_setup) is False
    j = rnd.randrange(n)
    bad_c = cj[:48 * j] + other + cj[48 * j + 48:]
    assert K.verify_blob_kzg_proof_batch(data, bad_c, pj, n, gpu_setup) is False
    k = rnd.randrange(n)
    bad_b = data[:k * B.BYTES_PER_BLOB] + B.synthetic_blob(99000 + n) + data[(k + 1) * B.BYTES_PER_BLOB:]
    assert K.verify_blob_kzg_proof_batch(bad_b, cj, pj, n, gpu_setup) is False
    # single-blob entry point agrees on a sample
    for t in rnd.sample(range(n), min(n, 3)):
        assert K.verify_blob_kzg_proof(blobs[t], comms[t], proofs[t], gpu_setup) is True
        assert K.verify_blob_kzg_proof(blobs[t], comms[t], other, gpu_setup) is False


def test_verify_single_blob_noncanonical_infinity_commitment(K, gpu_setup):
    """verify_blob_kzg_proof starts its GPU pass on the caller's commitment bytes and repeats it when the decompression
    (on a thread beside it) finds that they were not the canonical encoding: an infinity encoding with stray bits, for
    the zero blob, must verify exactly like c0 00..00; an invalid commitment or proof is still an error"""
    zero = bytes(B.BYTES_PER_BLOB)
    inf = bytes([0xc0]) + bytes(47)
    junk = bytes([0xc0]) + bytes(range(1, 48))
    assert K.blob_to_kzg_commitment(zero, gpu_setup) == inf
    assert K.compute_blob_kzg_proof(zero, inf, gpu_setup) == inf
    for _ in range(3):
        assert K.verify_blob_kzg_proof(zero, inf, inf, gpu_setup) is True
        assert K.verify_blob_kzg_proof(zero, junk, inf, gpu_setup) is True
        assert K.verify_blob_kzg_proof(zero, junk, junk, gpu_setup) is True
    blob = B.synthetic_blob(4242)
    c = K.blob_to_kzg_commitment(blob, gpu_setup)
    p = K.compute_blob_kzg_proof(blob, c, gpu_setup)
    assert K.verify_blob_kzg_proof(blob, c, p, gpu_setup) is True
    assert K.verify_blob_kzg_proof(blob, junk, p, gpu_setup) is False
    for badc, badp in ((bytes(48), p), (c, bytes(48)), (bytes([c[0] & 0x7f]) + c[1:], p)):
        with pytest.raises(K.KzgError) as e:
            K.verify_blob_kzg_proof(blob, badc, badp, gpu_setup)
        assert e.value.rc == K.C_KZG_ERROR


def test_verify_kzg_proof_random_differential_vs_oracle(K, gpu_setup, oracle, oracle_setup):
    """verify_kzg_proof on random inputs, the way the reference's fuzz harnesses drive it (fuzz/*/fuzz.c): honest proofs at
    random points, then one field perturbed -- y off by one, z replaced, commitment / proof replaced by another subgroup point,
    infinity in either place, a point outside the subgroup, non-canonical scalars. Verdict AND return code against the
    oracle's closed-form verifier (the setup's tau is known); both modes."""
    rnd = random.Random(4844)
    inf = bytes([0xc0]) + bytes(47)
    not_in_g1 = bytes([0x80]) + bytes(47)                     # (0, 2): on the curve, not in the subgroup (compression.rs:155-165)
    cases = agree = 0
    for mode, be, omode in ((K.MODE_REFERENCE, True, oracle.MODE_R), (K.MODE_CKZG, False, oracle.MODE_C)):
        K.set_mode(mode)
        order = "big" if be else "little"
        blobs = [B.synthetic_blob(97000 + i, big_endian=be) for i in range(3)] + [bytes(B.BYTES_PER_BLOB)]
        comms = K.blob_to_kzg_commitment_batch(b"".join(blobs), gpu_setup)
        for b, c in zip(blobs, comms):
            for _ in range(6):
                z = rnd.randrange(R).to_bytes(32, order)
                pr, y = K.compute_kzg_proof(b, z, gpu_setup)
                other = oracle.g1_generator_mul(rnd.randrange(2, R))
                yy = ((int.from_bytes(y, order) + 1) % R).to_bytes(32, order)
                variants = [(c, z, y, pr), (c, z, yy, pr), (c, rnd.randrange(R).to_bytes(32, order), y, pr), (other, z, y, pr),
                            (c, z, y, other), (inf, z, y, pr), (c, z, y, inf), (not_in_g1, z, y, pr), (c, z, y, not_in_g1),
                            (c, (R + 5).to_bytes(32, order), y, pr), (c, z, b"\xff" * 32, pr)]
                for cm, zz, yv, pf in variants:
                    want_rc, want_ok = oracle.verify_kzg_proof_known_tau(cm, zz, yv, pf, TAU, omode)
                    try:
                        got = (0, K.verify_kzg_proof(cm, zz, yv, pf, gpu_setup))
                    except K.KzgError as e:
                        got = (e.rc, False)
                    assert got == (want_rc, want_ok), (mode, cm.hex()[:8], zz.hex()[:8], got, (want_rc, want_ok))
                    cases += 1
                    agree += int(got[1])
    assert cases == 2 * 4 * 6 * 11 and agree >= 2 * 4 * 6       # every honest proof accepted, at least
