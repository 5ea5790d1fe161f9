"""GPU: the cooperative kernel (k_coop_msm_asm: four lanes per group addition, direct.hip) is what a commitment or a quotient MSM of up to
eight blobs runs on. Every test goes through the C ABI and compares with the tau closed form / the CPU oracle: the digit sets that
stress the signed recoding and the infinity masks, one blob at a time and in batches of 2 / 4 / 8 (the window groups change with the
batch), on the narrow, the default and the widest table and on the bucket-free widths between; partial sums that are EQUAL (the redo
flag and the complete-branches second pass); the device-resident entry point (inversion on the GPU) against the host-pointer one
(inversion on the host) on the same blobs; proofs of one blob."""
import random

import pytest

import blobs as B
from conftest import R, SETUP_PATH, TAU, tau_closed_form

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[10, 11, 13, 14, 16])
def table(request, K, gpu_setup):
    """a settings object of its own on a table of the given width (13: the library's own choice on an empty device)"""
    import torch
    wide = request.param >= 15
    if wide:
        gpu_setup.enable_direct_table(0)       # the session's default table steps aside for the 240 GB one
    ts = K.TrustedSetup.from_file(SETUP_PATH)
    try:
        ts.enable_direct_table(request.param)
    except K.KzgError as e:
        ts.free()
        if wide:
            gpu_setup.enable_direct_table(gpu_setup.default_bits)
        assert e.rc == K.C_KZG_MALLOC
        pytest.skip("direct table of width %d does not fit on this device" % request.param)
    assert ts.direct_table_bits() == request.param
    yield ts, request.param
    ts.free()
    torch.cuda.empty_cache()
    if wide:
        gpu_setup.enable_direct_table(gpu_setup.default_bits)


def _digit_sets(c):
    nw = (255 + c - 1) // c
    half = 1 << (c - 1)
    boundary = sum(half << (c * j) for j in range(nw - 1))            # every signed window = 2^(c-1): the last positive row
    boundary1 = sum((half + 1) << (c * j) for j in range(nw - 1))     # every signed window negative, carries ripple up
    ones = (1 << (c * (nw - 1))) - 1                                  # digits -1, 0, 0, ... and a carry into the top window
    rnd = random.Random(500 + c)
    return [[R - 1] * 4096,
            [5, R - 5] * 2048,
            [0] * 1234 + [R - 2] + [0] * 2861,                        # one row group of one quad; every other quad at infinity
            [0] * 4096,                                               # the empty sum
            [boundary % R] * 4096,
            [boundary1 % R] * 4096,
            [ones % R] * 4096,
            [(1 << 254) + ones] * 4096 if (1 << 254) + ones < R else [1 << 254] * 4096,   # the top window at work
            [(1 << 247) | 1] * 4096,
            [rnd.randrange(R) if k % 7 else 0 for k in range(4096)],
            [rnd.randrange(1 << c) for _ in range(4096)],             # one window each: whole window groups with nothing to add
            list(range(1, 4097))]


def _pack(ss):
    return b"".join(s.to_bytes(32, "big") for s in ss)


@pytest.mark.parametrize("batch", [1, 2, 4, 8])
def test_digit_sets_closed_form(K, table, oracle, batch):
    ts, c = table
    sets = _digit_sets(c)
    want = [tau_closed_form(oracle, ss) for ss in sets]
    for lo in range(0, len(sets), batch):
        grp = sets[lo:lo + batch]
        if len(grp) < batch:
            grp = grp + sets[:batch - len(grp)]
        got = K.blob_to_kzg_commitment_batch(b"".join(_pack(ss) for ss in grp), ts)
        for k, g in enumerate(got):
            assert g == want[(lo + k) % len(sets)], (c, batch, lo + k)


def _equal_partial_sums_blob(rnd, c, pairs):
    """P_(i+1) = [tau]P_i, so s_i = tau e and s_(i+1) = e (both inside the lowest windows) give the quads of points i and i + 1 the
    same sum [e]P_(i+1): the first level of the wave's tree meets P = Q. The rest of the blob is random."""
    ss = [rnd.randrange(R) for _ in range(4096)]
    for _ in range(pairs):
        i = 2 * rnd.randrange(2048)
        e = rnd.randrange(1, 1 << (2 * c - 13))
        ss[i], ss[i + 1] = e * TAU, e
    return ss


@pytest.mark.parametrize("batch", [1, 3, 8])
def test_equal_partial_sums_take_the_second_pass(K, table, oracle, batch):
    """the quad addition has no P = Q branch: the wave raises its blob's redo flag and the complete-branches kernel recomputes the blob
    (only that blob: its neighbours in the batch are honest and keep the cooperative kernel's sums)"""
    ts, c = table
    rnd = random.Random(900 + c + batch)
    sets = [_equal_partial_sums_blob(rnd, c, 1 + b % 3) if b % 2 == 0 else [rnd.randrange(R) for _ in range(4096)] for b in range(batch)]
    got = K.blob_to_kzg_commitment_batch(b"".join(_pack(ss) for ss in sets), ts)
    for b in range(batch):
        assert got[b] == tau_closed_form(oracle, sets[b]), (c, batch, b)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8])
def test_device_entry_point_equals_host_entry_point(K, gpu_setup, oracle, n):
    """the same blobs through lwkzg_blob_to_kzg_commitment_batch_device (k_finalize_compress on the GPU) and through the host-pointer
    call (XYZZ sums back, inversion and compression on the host): the same 48 bytes, and the closed form's"""
    import torch
    from lambdaworks_kzg_amd import capi
    data = b"".join(B.synthetic_blob(3000 + n * 10 + i, full_range=(i % 2 == 1)) for i in range(n))
    host = K.blob_to_kzg_commitment_batch(data, gpu_setup)
    d_blobs = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    d_out = torch.zeros(48 * n, dtype=torch.uint8, device="cuda")
    capi.blob_to_kzg_commitment_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), n, gpu_setup)
    torch.cuda.synchronize()
    dev = bytes(d_out.cpu().numpy())
    for i in range(n):
        assert dev[48 * i:48 * i + 48] == host[i]
        assert host[i] == tau_closed_form(oracle, [s % R for s in B.blob_scalars(data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB])])
    if n == 1:
        assert K.blob_to_kzg_commitment(data, gpu_setup) == host[0]


def test_single_blob_proofs_vs_oracle(K, table, oracle, oracle_setup):
    """compute_blob_kzg_proof / compute_kzg_proof of ONE blob (the reference's call shape): the quotient's MSM is the cooperative kernel,
    the inversion the host's; reference mode, against the oracle"""
    ts, c = table
    O = oracle
    for seed in (41, 42):
        blob = B.synthetic_blob(seed + c, full_range=(seed == 42))
        rc, cm = O.blob_to_kzg_commitment(blob, oracle_setup, O.MODE_R)
        assert rc == 0 and K.blob_to_kzg_commitment(blob, ts) == cm
        rc, want = O.compute_blob_kzg_proof(blob, cm, oracle_setup, O.MODE_R)
        assert rc == 0 and K.compute_blob_kzg_proof(blob, cm, ts) == want
        z = blob[64:96]
        rc, pw, yw = O.compute_kzg_proof(blob, z, oracle_setup, O.MODE_R)
        assert rc == 0 and K.compute_kzg_proof(blob, z, ts) == (pw, yw)


def _compress(pt):
    from conftest import P
    x, y = pt
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if y > P - y else 0)
    return bytes(b)


@pytest.mark.parametrize("n", [40, 300, 1500])
def test_validation_on_quads_of_lanes_judges_every_point_like_the_oracle(K, gpu_setup, oracle, n):
    """r05: a proof call's commitments are validated by square root + the quad-of-lanes subgroup test (k_subgroup_coop_asm) + canonical bytes.
    A batch of commitments of every kind -- points of G1, points of E(Fp) OUTSIDE G1 (random x with a square x^3 + 4: the cofactor is
    ~2^125, so almost all of them), points of small order (a cofactor-cleared... r-cleared point), infinity, x not on the curve, bytes
    without the compression flag -- goes through the device-resident proof call: the per-blob status is non-zero exactly where the CPU
    oracle's decompression (which includes the reference's [r]P == O test, compression.rs:22-27) rejects; n = 40: the host-assisted small
    path validates on the host instead (same verdicts); 1500: a call longer than one chunk"""
    import sys, os
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_subgroup_asm as S
    import gen_direct_asm as G
    rnd = random.Random(8800 + n)
    comms = []
    for i in range(n):
        kind = i % 6
        if kind in (0, 1):
            comms.append(oracle.g1_generator_mul(rnd.randrange(1, R)))
        elif kind == 2:
            comms.append(_compress(S.curve_point_outside_g1(rnd)))
        elif kind == 3 and i % 30 == 3:
            comms.append(_compress(G.ec_mul(R, S.curve_point_outside_g1(rnd))))          # order divides the cofactor
        elif kind == 4 and i % 12 == 4:
            comms.append(bytes([0xc0]) + bytes(47))                                       # infinity
        elif kind == 5 and i % 12 == 5:
            comms.append(bytes([0x80]) + rnd.randbytes(47))                               # a random x: on the curve or not, in G1 almost never
        else:
            comms.append(oracle.g1_generator_mul(rnd.randrange(1, R)))
    want_ok = [oracle.g1_decompress(c) is not None for c in comms]
    assert 0.4 < sum(want_ok) / n < 0.95
    data = B.synthetic_batch(9900, min(n, 8)) * ((n + 7) // 8)
    data = data[:n * B.BYTES_PER_BLOB]
    d_blobs = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    d_comm = torch.frombuffer(bytearray(b"".join(comms)), dtype=torch.uint8).cuda()
    d_out = torch.zeros(48 * n, dtype=torch.uint8, device="cuda")
    d_st = torch.full((n,), 9, dtype=torch.int32, device="cuda")
    K.compute_blob_kzg_proof_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), d_comm.data_ptr(), n, gpu_setup, None, d_st.data_ptr())
    torch.cuda.synchronize()
    st = d_st.cpu().tolist()
    wrong = [i for i in range(n) if (st[i] == 0) != want_ok[i]]
    assert not wrong, (wrong[:10], [st[i] for i in wrong[:10]])
    assert all(s in (0, K.C_KZG_ERROR) for s in st)
