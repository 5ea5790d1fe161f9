"""GPU: the device-resident verification entry points (lwkzg_verify_blob_kzg_proof_batch_device, lwkzg_verify_shard_begin_device):
blobs, commitments and proofs as device pointers, the verdict on the host. Same verdicts and return codes as the host-pointer forms
(/root/reference/src/lib.rs:525-614, 639-692): the three c-kzg verify suites' vectors that the fixed-size ABI can express, honest
batches and batches with one swapped proof / commitment / blob at 1 .. 2300 blobs (one chunk, several chunks), non-canonical
commitment encodings (the re-hash over the canonical bytes), invalid points, the empty batch in both modes, the sharded form with
device-resident shards against the single batch, and inputs produced on the caller's own stream."""
import random

import pytest

import blobs as B
from conftest import R, hx

pytestmark = pytest.mark.gpu


def _dev(torch, data):
    return torch.frombuffer(bytearray(data) if data else bytearray(1), dtype=torch.uint8).cuda()


def _verify_dev(K, torch, data, comms, proofs, n, ts, stream=None):
    db, dc, dp = _dev(torch, data), _dev(torch, comms), _dev(torch, proofs)
    torch.cuda.synchronize()
    return K.verify_blob_kzg_proof_batch_device(db.data_ptr(), dc.data_ptr(), dp.data_ptr(), n, ts, stream)


def test_ckzg_verify_vectors_through_the_device_forms(K, gpu_setup, vectors):
    """verify_blob_kzg_proof and verify_blob_kzg_proof_batch vectors (c-kzg mode), every case the C ABI can express: verdict or BADARGS"""
    import torch
    K.set_mode(K.MODE_CKZG)
    try:
        n_cases = 0
        for c in vectors["suites"]["verify_blob_kzg_proof_batch"]:
            i = c["input"]
            blobs = [B.make_blob(b) for b in i["blobs"]]
            cms, prs = [hx(x) for x in i["commitments"]], [hx(x) for x in i["proofs"]]
            if any(len(b) != B.BYTES_PER_BLOB for b in blobs) or any(len(x) != 48 for x in cms + prs) or not (len(blobs) == len(cms) == len(prs)):
                continue
            k = len(blobs)
            if c["output"] is None:
                with pytest.raises(K.KzgError) as e:
                    _verify_dev(K, torch, b"".join(blobs), b"".join(cms), b"".join(prs), k, gpu_setup)
                assert e.value.rc == K.C_KZG_BADARGS, c["case"]
            else:
                assert _verify_dev(K, torch, b"".join(blobs), b"".join(cms), b"".join(prs), k, gpu_setup) is c["output"], c["case"]
            n_cases += 1
        for c in vectors["suites"]["verify_blob_kzg_proof"]:
            i = c["input"]
            blob, cm, pr = B.make_blob(i["blob"]), hx(i["commitment"]), hx(i["proof"])
            if len(blob) != B.BYTES_PER_BLOB or len(cm) != 48 or len(pr) != 48:
                continue
            if c["output"] is None:
                with pytest.raises(K.KzgError) as e:
                    _verify_dev(K, torch, blob, cm, pr, 1, gpu_setup)
                assert e.value.rc == K.C_KZG_BADARGS, c["case"]
            else:
                assert _verify_dev(K, torch, blob, cm, pr, 1, gpu_setup) is c["output"], c["case"]
            n_cases += 1
        assert n_cases >= 30
    finally:
        K.set_mode(K.MODE_REFERENCE)


@pytest.mark.parametrize("n", [1, 2, 7, 65, 700, 1024, 2300])
def test_device_batches_honest_and_tampered(K, gpu_setup, oracle, n):
    """the device form accepts what the host form accepts and rejects what it rejects: one proof, one commitment or one blob swapped
    for another valid one at a random place; points at infinity (zero blob, constant blob) take part"""
    import torch
    rnd = random.Random(7000 + n)
    blobs = [B.synthetic_blob(52000 + 31 * n + i) for i in range(n)]
    if n > 2:
        blobs[rnd.randrange(n)] = bytes(B.BYTES_PER_BLOB)
        const = bytearray(B.BYTES_PER_BLOB)
        const[31] = 5
        blobs[(blobs.index(bytes(B.BYTES_PER_BLOB)) + 1) % n] = bytes(const)
    data = b"".join(blobs)
    cj = b"".join(K.blob_to_kzg_commitment_batch(data, gpu_setup))
    pj = b"".join(K.compute_blob_kzg_proof_batch(data, cj, gpu_setup))
    assert _verify_dev(K, torch, data, cj, pj, n, gpu_setup) is True
    assert K.verify_blob_kzg_proof_batch(data, cj, pj, n, gpu_setup) is True
    other = oracle.g1_generator_mul(rnd.randrange(2, R))
    i = rnd.randrange(n)
    assert _verify_dev(K, torch, data, cj, pj[:48 * i] + other + pj[48 * i + 48:], n, gpu_setup) is False
    j = rnd.randrange(n)
    assert _verify_dev(K, torch, data, cj[:48 * j] + other + cj[48 * j + 48:], pj, n, gpu_setup) is False
    k = rnd.randrange(n)
    bad_b = data[:k * B.BYTES_PER_BLOB] + B.synthetic_blob(99100 + n) + data[(k + 1) * B.BYTES_PER_BLOB:]
    assert _verify_dev(K, torch, bad_b, cj, pj, n, gpu_setup) is False
    # an invalid point is an error, not a verdict
    for badc, badp in ((bytes(48) + cj[48:], pj), (cj, pj[:-48] + bytes(48)), (bytes([cj[0] & 0x7f]) + cj[1:], pj)):
        with pytest.raises(K.KzgError) as e:
            _verify_dev(K, torch, data, badc, badp, n, gpu_setup)
        assert e.value.rc == K.C_KZG_ERROR


@pytest.mark.parametrize("n", [3, 300, 1500])
def test_ckzg_device_batches_evaluate_straight_from_the_blobs(K, gpu_setup, oracle, n):
    """r05, c-kzg mode on the Lagrange form: the device form evaluates y_i = p_i(z_i) of ALL blobs in one launch that reads the blobs'
    little-endian evaluations as they stand (k_eval_quotient_evalform without a quotient, the front end's range check inside it). Honest
    batches verify in both forms, a swapped blob does not, and ONE element >= r anywhere in the batch is BADARGS, not a verdict"""
    import torch
    K.set_mode(K.MODE_CKZG)
    try:
        rnd = random.Random(7100 + n)
        data = B.synthetic_batch(53000 + n, n, big_endian=False)
        cj = b"".join(K.blob_to_kzg_commitment_batch(data, gpu_setup))
        pj = b"".join(K.compute_blob_kzg_proof_batch(data, cj, gpu_setup))
        assert _verify_dev(K, torch, data, cj, pj, n, gpu_setup) is True
        assert K.verify_blob_kzg_proof_batch(data, cj, pj, n, gpu_setup) is True
        k = rnd.randrange(n)
        swapped = data[:k * B.BYTES_PER_BLOB] + B.synthetic_blob(99200 + n, big_endian=False) + data[(k + 1) * B.BYTES_PER_BLOB:]
        assert _verify_dev(K, torch, swapped, cj, pj, n, gpu_setup) is False
        for where in (0, n // 2, n - 1):
            e_at = rnd.randrange(4096)
            off = where * B.BYTES_PER_BLOB + 32 * e_at
            bad = data[:off] + R.to_bytes(32, "little") + data[off + 32:]
            with pytest.raises(K.KzgError) as e:
                _verify_dev(K, torch, bad, cj, pj, n, gpu_setup)
            assert e.value.rc == K.C_KZG_BADARGS, (where, e_at)
            if n > 1024:   # r06: the host-pointer form of a long batch evaluates from its uploaded blobs the same way (verify_prepare_staged)
                with pytest.raises(K.KzgError) as e:
                    K.verify_blob_kzg_proof_batch(bad, cj, pj, n, gpu_setup)
                assert e.value.rc == K.C_KZG_BADARGS, (where, e_at)
        if n > 1024:
            assert K.verify_blob_kzg_proof_batch(swapped, cj, pj, n, gpu_setup) is False
    finally:
        K.set_mode(K.MODE_REFERENCE)


def test_noncanonical_infinity_encoding_is_rehashed_on_the_device(K, gpu_setup):
    """a valid commitment in a non-canonical encoding (infinity with stray bits, for the zero blob) among honest ones: the challenge of
    that blob is taken again over the canonical bytes, as the host form does"""
    import torch
    zero = bytes(B.BYTES_PER_BLOB)
    inf = bytes([0xc0]) + bytes(47)
    junk = bytes([0xc0]) + bytes(range(1, 48))
    blobs = [B.synthetic_blob(61000 + i) for i in range(5)]
    blobs[2] = zero
    data = b"".join(blobs)
    comms = K.blob_to_kzg_commitment_batch(data, gpu_setup)
    proofs = K.compute_blob_kzg_proof_batch(data, b"".join(comms), gpu_setup)
    assert comms[2] == inf and proofs[2] == inf
    for c2, p2 in ((inf, inf), (junk, inf), (junk, junk)):
        cs, ps = list(comms), list(proofs)
        cs[2], ps[2] = c2, p2
        assert _verify_dev(K, torch, data, b"".join(cs), b"".join(ps), 5, gpu_setup) is True
        assert K.verify_blob_kzg_proof_batch(data, b"".join(cs), b"".join(ps), 5, gpu_setup) is True


def test_empty_batch_both_modes(K, gpu_setup):
    import torch
    assert _verify_dev(K, torch, b"", b"", b"", 0, gpu_setup) is False       # lib.rs:538-543
    K.set_mode(K.MODE_CKZG)
    try:
        assert _verify_dev(K, torch, b"", b"", b"", 0, gpu_setup) is True    # c-kzg vector a271b78b8e869d69
    finally:
        K.set_mode(K.MODE_REFERENCE)


@pytest.mark.parametrize("n,world", [(5, 8), (200, 3), (2300, 2)])
def test_device_resident_shards_equal_the_single_batch(K, gpu_setup, n, world):
    """the sharded form with every shard's inputs on the device: the records are byte-equal to the host-pointer shards', the verdicts
    those of the single batch (honest / one wrong proof / proofs exchanged across shards)"""
    import torch
    from lambdaworks_kzg_amd import capi
    from lambdaworks_kzg_amd import dist as D
    data = B.synthetic_batch(83000 + n, n)
    comms = b"".join(K.blob_to_kzg_commitment_batch(data, gpu_setup))
    proofs = b"".join(K.compute_blob_kzg_proof_batch(data, comms, gpu_setup))

    def verdict(data, comms, proofs, compare_records):
        shards, counts, keep = [], [], []
        for r in range(world):
            st, cnt = D.shard_range(n, world, r)
            db, dc, dp = (_dev(torch, x) for x in (data[st * B.BYTES_PER_BLOB:(st + cnt) * B.BYTES_PER_BLOB], comms[48 * st:48 * (st + cnt)],
                                                    proofs[48 * st:48 * (st + cnt)]))
            keep.append((db, dc, dp))
            torch.cuda.synchronize()
            shards.append(capi.VerifyShard.from_device(db.data_ptr(), dc.data_ptr(), dp.data_ptr(), cnt, gpu_setup))
            counts.append(cnt)
            if compare_records:
                h = capi.VerifyShard(data[st * B.BYTES_PER_BLOB:(st + cnt) * B.BYTES_PER_BLOB], comms[48 * st:48 * (st + cnt)],
                                     proofs[48 * st:48 * (st + cnt)], cnt, gpu_setup)
                assert h.records == shards[-1].records, r
                h.free()
        records = b"".join(s.records for s in shards)
        partials = [s.partial(records, n, sum(counts[:r])) for r, s in enumerate(shards)]
        ok = capi.verify_shards_finish(b"".join(partials), world, n, gpu_setup)
        for s in shards:
            s.free()
        return ok

    assert verdict(data, comms, proofs, True) is True
    k = n - 1
    assert verdict(data, comms, proofs[:48 * k] + proofs[:48], False) is False
    if n >= 2 * world:
        i, j = 1, n - 2
        sw = lambda buf, w: buf[:w * i] + buf[w * j:w * (j + 1)] + buf[w * (i + 1):w * j] + buf[w * i:w * (i + 1)] + buf[w * (j + 1):]
        assert verdict(sw(data, B.BYTES_PER_BLOB), sw(comms, 48), sw(proofs, 48), False) is True
        assert verdict(data, comms, sw(proofs, 48), False) is False


def test_inputs_produced_on_the_callers_stream(K, gpu_setup):
    """commitments and proofs computed on a caller stream and verified at once, no synchronisation in between: the verification orders
    itself behind the stream it is given"""
    import torch
    from lambdaworks_kzg_amd import capi
    n = 300
    data = B.synthetic_batch(91000, n)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        db = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda(non_blocking=False)
        dc = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
        dp = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
        capi.commit_and_prove_batch_device(dc.data_ptr(), dp.data_ptr(), db.data_ptr(), n, gpu_setup, st.cuda_stream)
        assert K.verify_blob_kzg_proof_batch_device(db.data_ptr(), dc.data_ptr(), dp.data_ptr(), n, gpu_setup, st.cuda_stream) is True
        # one proof overwritten ON the stream, verified again without waiting
        dp[48 * 17:48 * 18] = dp[48 * 3:48 * 4].clone()
        assert K.verify_blob_kzg_proof_batch_device(db.data_ptr(), dc.data_ptr(), dp.data_ptr(), n, gpu_setup, st.cuda_stream) is False
    torch.cuda.synchronize()
