"""GPU: the node-level entry points (lwkzg_multi_*, csrc/multi.hip) -- one process, several devices, contiguous blob shards, the
setup image delivered device to device, no reduction. On a one-GPU box "several devices" are several contexts on device 0
(an ordinal may repeat), which exercises everything but the xGMI hop of hipMemcpyPeer: sharding, per-device threads, the image
hand-off, first_bad arithmetic, the one-r batch verification over shards. Bytes must equal the single-device calls
(reference: the plain C callers of /root/reference/fuzz/base_fuzz.h:17-34 and src/lib.rs:253-283, which this serves)."""
import os
import subprocess

import pytest

import blobs as B
from conftest import ROOT, SETUP_PATH, tau_closed_form

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def single(K):
    ts = K.TrustedSetup.from_file(SETUP_PATH)
    yield ts
    ts.free()


@pytest.fixture(scope="module", params=[[0], [0, 0], [0, 0, 0]], ids=lambda d: "x%d" % len(d))
def multi(request, K):
    from lambdaworks_kzg_amd import capi
    m = capi.MultiSetup.from_file(SETUP_PATH, request.param)
    assert m.devices() == request.param
    yield m
    m.free()


@pytest.mark.parametrize("mode_c", [False, True])
def test_multi_batches_equal_the_single_device_calls(K, single, multi, oracle, mode_c):
    n = 7                                                     # uneven shards over 2 and 3 devices (3 + 4; 2 + 2 + 3)
    mode = K.MODE_CKZG if mode_c else K.MODE_REFERENCE
    data = B.synthetic_batch(81000, n, big_endian=not mode_c)
    single.set_mode(mode)
    multi.set_mode(mode)
    try:
        want_c = K.blob_to_kzg_commitment_batch(data, single)
        got_c = multi.blob_to_kzg_commitment_batch(data)
        assert got_c == want_c
        if not mode_c:
            assert got_c[3] == tau_closed_form(oracle, B.blob_scalars(data[3 * B.BYTES_PER_BLOB:4 * B.BYTES_PER_BLOB]))
        cm = b"".join(got_c)
        want_p = K.compute_blob_kzg_proof_batch(data, cm, single)
        assert multi.compute_blob_kzg_proof_batch(data, cm) == want_p
        zs = b"".join((i + 5).to_bytes(32, "little" if mode_c else "big") for i in range(n))
        assert multi.compute_kzg_proof_batch(data, zs) == K.compute_kzg_proof_batch(data, zs, single)
        pr = b"".join(want_p)
        assert multi.verify_blob_kzg_proof_batch(data, cm, pr, n) is True
        swapped = pr[48 * (n - 1):] + pr[48:48 * (n - 1)] + pr[:48]
        assert multi.verify_blob_kzg_proof_batch(data, cm, swapped, n) is False
        # the empty batch: the mode's own answer (reference: false, src/lib.rs:538-543; c-kzg: true), no device touched
        assert multi.verify_blob_kzg_proof_batch(b"", b"", b"", 0) is mode_c
        assert multi.blob_to_kzg_commitment_batch(b"") == []
    finally:
        single.set_mode(-1)
        multi.set_mode(-1)


def test_multi_first_bad_is_the_lowest_index_of_the_whole_batch(K, multi):
    n = 9
    data = bytearray(B.synthetic_batch(82000, n, big_endian=False))
    for bad in (7, 4):                                        # two non-canonical blobs, on different shards when there are several
        data[bad * B.BYTES_PER_BLOB:bad * B.BYTES_PER_BLOB + 32] = b"\xff" * 32
    multi.set_mode(K.MODE_CKZG)
    try:
        with pytest.raises(K.KzgError) as e:
            multi.blob_to_kzg_commitment_batch(bytes(data))
        assert e.value.rc == K.C_KZG_BADARGS and multi.first_bad.value == 4
    finally:
        multi.set_mode(-1)


def test_multi_bad_arguments(K):
    from lambdaworks_kzg_amd import capi
    for devices in ([], [99], [-1], [0] * 65):
        with pytest.raises((K.KzgError, ValueError)) as e:
            capi.MultiSetup.from_file(SETUP_PATH, devices)
        if isinstance(e.value, K.KzgError):
            assert e.value.rc == K.C_KZG_BADARGS


def test_multi_engines_per_device_and_tiled_msm(K, single, multi, oracle):
    """every device picks / is given its own MSM engine; the long MSM (BASELINE configs[4]) is whole tiles per device and one
    host addition of 48-byte partial sums"""
    import torch
    from lambdaworks_kzg_amd import capi
    tiles = 5
    sc = B.synthetic_batch(83000, tiles)
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    d_out = torch.empty(48, dtype=torch.uint8, device="cuda")
    capi.g1_msm_tiled_device(d_out.data_ptr(), d_sc.data_ptr(), tiles * 4096, single)
    torch.cuda.synchronize()
    want = bytes(d_out.cpu().numpy().tobytes())
    assert multi.g1_msm_tiled(sc) == want
    multi.enable_direct_table(10)
    try:
        for k in range(multi.device_count()):
            assert K.lib().lwkzg_direct_table_bits(multi.settings(k).ref()) == 10
        assert multi.g1_msm_tiled(sc) == want
        data = B.synthetic_batch(83100, 4)
        assert multi.blob_to_kzg_commitment_batch(data) == K.blob_to_kzg_commitment_batch(data, single)
        with pytest.raises(K.KzgError) as e:
            multi.enable_direct_table(9)
        assert e.value.rc == K.C_KZG_BADARGS
    finally:
        multi.enable_direct_table(0)
    assert multi.blob_to_kzg_commitment_batch(B.synthetic_batch(83100, 4)) == K.blob_to_kzg_commitment_batch(B.synthetic_batch(83100, 4), single)


@pytest.mark.parametrize("devices,mode", [("0,0", 0), ("0,0,0", 1)])
def test_multi_from_a_c_program(K, single, tmp_path, devices, mode):
    """tests/multi_harness.c: no Python between the caller and the devices"""
    lib_dir = os.path.join(ROOT, "lambdaworks_kzg_amd", "lib")
    exe = str(tmp_path / "multi_harness")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "multi_harness.c"),
                           "-o", exe, "-L", lib_dir, "-llambdaworks_kzg", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    n = 10
    data = B.synthetic_batch(84000, n, big_endian=(mode == 0))
    (tmp_path / "blobs.bin").write_bytes(data)
    env = dict(os.environ)
    env.pop("LWKZG_MODE", None)
    out = subprocess.check_output([exe, SETUP_PATH, str(tmp_path / "blobs.bin"), devices, str(mode)], env=env, timeout=1200).decode().split("\n")
    kv = dict(l.split(" ", 1) for l in out if " " in l)
    single.set_mode(mode)
    try:
        cm = K.blob_to_kzg_commitment_batch(data, single)
        assert kv["devices"] == str(len(devices.split(",")))
        assert kv["commitments"] == b"".join(cm).hex()
        assert kv["blob_proofs"] == b"".join(K.compute_blob_kzg_proof_batch(data, b"".join(cm), single)).hex()
        zs = b"".join((i + 2).to_bytes(32, "little" if mode else "big") for i in range(n))
        pz = K.compute_kzg_proof_batch(data, zs, single)
        assert kv["point_proofs"] == b"".join(p for p, _ in pz).hex() and kv["ys"] == b"".join(y for _, y in pz).hex()
        assert kv["verify_batch"] == "1" and kv["verify_batch_swapped"] == "0"
        assert kv["commitment0_on_last_device"] == cm[0].hex()
        if mode == 0:
            import torch
            from lambdaworks_kzg_amd import capi
            d_sc = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
            d_out = torch.empty(48, dtype=torch.uint8, device="cuda")
            capi.g1_msm_tiled_device(d_out.data_ptr(), d_sc.data_ptr(), n * 4096, single)
            torch.cuda.synchronize()
            assert kv["tiled_msm"] == bytes(d_out.cpu().numpy().tobytes()).hex()
        else:
            assert kv["bad_blob_rc"] == "%d first_bad %d" % (K.C_KZG_BADARGS, n - 2)
    finally:
        single.set_mode(-1)


@pytest.mark.parametrize("counts_of", [lambda g: [5] * g, lambda g: [0] + [3] * (g - 1) if g > 1 else [4], lambda g: [70] + [1] * (g - 1)])
def test_multi_device_resident_shards(K, single, multi, counts_of):
    """lwkzg_multi_*_device: every device's shard already in HBM (on a one-GPU box: several contexts on device 0, each with its own
    buffers), results in place; commitments, proofs and the one-r batch verification equal the single-device calls on the concatenated
    batch; an empty shard is allowed; a rejected blob is reported by its index through the shards"""
    import torch
    g = multi.device_count()
    counts = counts_of(g)
    n = sum(counts)
    data = B.synthetic_batch(84000 + n, n)
    want_c = K.blob_to_kzg_commitment_batch(data, single)
    want_p = K.compute_blob_kzg_proof_batch(data, b"".join(want_c), single)
    d_blobs, d_comm, d_proof, off = [], [], [], 0
    for c in counts:
        chunk = data[off * B.BYTES_PER_BLOB:(off + c) * B.BYTES_PER_BLOB]
        d_blobs.append(torch.frombuffer(bytearray(chunk) if c else bytearray(1), dtype=torch.uint8).cuda())
        d_comm.append(torch.zeros(max(48 * c, 1), dtype=torch.uint8, device="cuda"))
        d_proof.append(torch.zeros(max(48 * c, 1), dtype=torch.uint8, device="cuda"))
        off += c
    torch.cuda.synchronize()
    ptr = lambda ts: [t.data_ptr() for t in ts]
    multi.blob_to_kzg_commitment_batch_device(ptr(d_comm), ptr(d_blobs), counts)
    got_c = b"".join(bytes(t.cpu().numpy())[:48 * c] for t, c in zip(d_comm, counts))
    assert got_c == b"".join(want_c)
    multi.compute_blob_kzg_proof_batch_device(ptr(d_proof), ptr(d_blobs), ptr(d_comm), counts)
    got_p = b"".join(bytes(t.cpu().numpy())[:48 * c] for t, c in zip(d_proof, counts))
    assert got_p == b"".join(want_p)
    assert multi.verify_blob_kzg_proof_batch_device(ptr(d_blobs), ptr(d_comm), ptr(d_proof), counts) is True
    # one proof replaced by another blob's (a valid point): rejected, whichever shard it sits in
    k = max(range(g), key=lambda j: counts[j])
    keep = d_proof[k][:48].clone()
    d_proof[k][:48] = torch.frombuffer(bytearray(want_p[(sum(counts[:k]) + 1) % n]), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    assert multi.verify_blob_kzg_proof_batch_device(ptr(d_blobs), ptr(d_comm), ptr(d_proof), counts) is (n == 1)
    d_proof[k][:48] = keep
    # a commitment that is not a point: an error, reported with its index through the shards
    bad_at = sum(counts[:k]) + counts[k] - 1
    d_comm[k][48 * (counts[k] - 1):48 * counts[k]] = 0
    torch.cuda.synchronize()
    with pytest.raises(K.KzgError) as e:
        multi.compute_blob_kzg_proof_batch_device(ptr(d_proof), ptr(d_blobs), ptr(d_comm), counts)
    assert e.value.rc == K.C_KZG_ERROR and multi.first_bad.value == bad_at
    with pytest.raises(K.KzgError):
        multi.verify_blob_kzg_proof_batch_device(ptr(d_blobs), ptr(d_comm), ptr(d_proof), counts)


def test_eight_contexts_on_one_device(K, single):
    """lwkzg_multi_load with EIGHT entries -- the node's shape -- here all on device 0 (fresh process, 10-bit tables:
    eight default tables do not fit one device). Uneven shards (70 blobs over eight contexts), the one-r batch verification over eight
    shards, eight partial sums of the tiled MSM: bytes and verdicts of the single-device calls."""
    import sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import blobs as B, lambdaworks_kzg_amd as K\n"
        "from lambdaworks_kzg_amd import capi\n"
        "m = capi.MultiSetup.from_file(%r, [0] * 8)\n"
        "assert m.device_count() == 8 and m.devices() == [0] * 8\n"
        "data = B.synthetic_batch(84000, 70)\n"
        "c = m.blob_to_kzg_commitment_batch(data); p = m.compute_blob_kzg_proof_batch(data, b''.join(c))\n"
        "ok = m.verify_blob_kzg_proof_batch(data, b''.join(c), b''.join(p), 70)\n"
        "bad = m.verify_blob_kzg_proof_batch(data, b''.join(c), b''.join(p[1:] + p[:1]), 70)\n"
        "t = m.g1_msm_tiled(B.synthetic_batch(84100, 9))\n"
        "print(b''.join(c).hex(), b''.join(p).hex(), ok, bad, t.hex()); m.free()\n"
    ) % (ROOT, os.path.join(ROOT, "tests", "golden"), SETUP_PATH)
    env = dict(os.environ, LWKZG_DIRECT_BITS="10")
    out = subprocess.check_output([sys.executable, "-c", code], env=env, timeout=900).decode().split()
    c_hex, p_hex, ok, bad, t_hex = out[-5:]
    data = B.synthetic_batch(84000, 70)
    want_c = K.blob_to_kzg_commitment_batch(data, single)
    want_p = K.compute_blob_kzg_proof_batch(data, b"".join(want_c), single)
    assert c_hex == b"".join(want_c).hex() and p_hex == b"".join(want_p).hex()
    assert ok == "True" and bad == "False"
    import torch
    from lambdaworks_kzg_amd import capi
    sc = B.synthetic_batch(84100, 9)
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    d_out = torch.empty(48, dtype=torch.uint8, device="cuda")
    capi.g1_msm_tiled_device(d_out.data_ptr(), d_sc.data_ptr(), 9 * 4096, single)
    torch.cuda.synchronize()
    assert t_hex == bytes(d_out.cpu().numpy().tobytes()).hex()
