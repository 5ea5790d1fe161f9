"""GPU: the device-resident proof path (what `bench.py --op blob_proof` times, BASELINE configs[2]) against the CPU
oracle on EVERY blob of the batch, at the benchmark's sizes and at partial-workgroup sizes, on both engines a plain load
can select and in both modes; the Fiat-Shamir kernel against hashlib at scale.
Reference: /root/reference/src/lib.rs:361-404, /root/reference/src/utils.rs:120-154."""
import pytest

import blobs as B
from conftest import R, tau_closed_form
from proof_cases import challenge_int, oracle_batch, reference_mode_proof_closed_form

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


def device_commit_and_prove(K, ts, data, n):
    import torch
    d_blobs = _dev(data)
    d_comm = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    d_out = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    d_st = torch.full((n,), 9, dtype=torch.int32, device="cuda")
    K.blob_to_kzg_commitment_batch_device(d_comm.data_ptr(), d_blobs.data_ptr(), n, ts, None, d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    d_st.fill_(9)
    K.compute_blob_kzg_proof_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), d_comm.data_ptr(), n, ts, None, d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    return _host(d_comm), _host(d_out)


# 64 = one full hash workgroup; 65, 70 = a second, mostly empty one; 256, 1024 = the sizes bench.py quotes
@pytest.mark.parametrize("n,first", [(64, 21000), (65, 21100), (70, 21200), (256, 22000), (1024, 23000)])
@pytest.mark.parametrize("mode_c", [False, True], ids=["reference", "ckzg"])
def test_device_resident_blob_proofs_every_blob_vs_oracle(K, engine_setup, oracle, n, first, mode_c):
    K.set_mode(K.MODE_CKZG if mode_c else K.MODE_REFERENCE)
    blobs, want_c, want_p = oracle_batch(oracle, first, n, mode_c)
    comms, proofs = device_commit_and_prove(K, engine_setup, b"".join(blobs), n)
    for i in range(n):
        assert comms[48 * i:48 * i + 48] == want_c[i], ("commitment", i)
        assert proofs[48 * i:48 * i + 48] == want_p[i], ("proof", i)


@pytest.mark.parametrize("n", [70, 1500])
@pytest.mark.parametrize("mode_c", [False, True], ids=["reference", "ckzg"])
def test_fiat_shamir_kernel_vs_hashlib(K, gpu_setup, n, mode_c):
    """k_challenge_pairs at a partial-workgroup size and beyond one chunk, every digest against hashlib."""
    import torch
    from lambdaworks_kzg_amd import capi
    K.set_mode(K.MODE_CKZG if mode_c else K.MODE_REFERENCE)
    data = B.synthetic_batch(31000, n, big_endian=not mode_c)
    comms = b"".join(bytes([0x80 | (i % 32)]) + (i * 0x9E3779B97F4A7C15 % (1 << 376)).to_bytes(47, "big") for i in range(n))  # hashed as given
    d_blobs, d_comm = _dev(data), _dev(comms)
    d_z = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    capi.compute_challenges_device(d_z.data_ptr(), d_blobs.data_ptr(), d_comm.data_ptr(), n, gpu_setup)
    torch.cuda.synchronize()
    got = _host(d_z)
    for i in range(n):
        want = challenge_int(data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB], comms[48 * i:48 * i + 48], mode_c)
        assert int.from_bytes(got[32 * i:32 * i + 32], "little" if mode_c else "big") == want, i


def test_device_resident_proof_call_longer_than_one_chunk(K, engine_setup, oracle, oracle_setup):
    """n > 1024 in ONE device-resident call: all blobs are hashed and validated up front, then the MSMs run chunk by chunk.
    Every proof against the hashlib + tau closed form; chunk-boundary blobs against the oracle's own pipeline."""
    n = 2100
    data = B.synthetic_batch(40000, n)
    comms, proofs = device_commit_and_prove(K, engine_setup, data, n)
    for i in range(n):
        blob = data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB]
        if i % 16 == 0 or i in (1023, 1024, 1025, 2047, 2048, 2099):
            assert comms[48 * i:48 * i + 48] == tau_closed_form(oracle, B.blob_scalars(blob)), i
            assert proofs[48 * i:48 * i + 48] == reference_mode_proof_closed_form(oracle, blob, comms[48 * i:48 * i + 48]), i
    for i in (0, 1023, 1024, 2048, 2099):
        blob = data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB]
        assert oracle.compute_blob_kzg_proof(blob, comms[48 * i:48 * i + 48], oracle_setup, oracle.MODE_R) == (0, proofs[48 * i:48 * i + 48])
    # no status array: the library keeps its own (the call must still work)
    import torch
    d_blobs, d_comm = _dev(data), _dev(comms)
    d_out = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    K.compute_blob_kzg_proof_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), d_comm.data_ptr(), n, engine_setup, None, None)
    torch.cuda.synchronize()
    assert _host(d_out) == proofs


def test_device_calls_on_two_caller_streams_overlap_safely(K, oracle):
    """Two device-resident calls in flight on two different caller streams: the first runs on the settings' context, the
    second -- arriving while that workspace is busy -- on its twin (own streams and workspace, same tables), so that the
    hash of one overlaps the MSM of the other; a third call on the first stream chains behind the first by events. All
    results must be the single-stream ones, repeatedly, and across an engine switch and the final free."""
    import torch
    from conftest import SETUP_PATH
    ts = K.TrustedSetup.from_file(SETUP_PATH)
    n = 96
    a, b = B.synthetic_batch(50000, n), B.synthetic_batch(51000, n)
    d_a, d_b = _dev(a), _dev(b)
    want_a = b"".join(K.blob_to_kzg_commitment_batch(a, ts))
    want_b = b"".join(K.blob_to_kzg_commitment_batch(b, ts))
    for i in (0, n - 1):
        assert want_a[48 * i:48 * i + 48] == tau_closed_form(oracle, B.blob_scalars(a[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB]))
    want_pa = [reference_mode_proof_closed_form(oracle, a[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB], want_a[48 * i:48 * i + 48]) for i in range(n)]
    want_pb = [reference_mode_proof_closed_form(oracle, b[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB], want_b[48 * i:48 * i + 48]) for i in range(n)]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for engine in (None, 12, 0):
        if engine is not None:
            ts.enable_direct_table(engine)      # with the twin alive: both contexts must see the new table
        for _ in range(4):
            o_a = torch.zeros(48 * n, dtype=torch.uint8, device="cuda")
            o_b = torch.zeros(48 * n, dtype=torch.uint8, device="cuda")
            p_a = torch.zeros(48 * n, dtype=torch.uint8, device="cuda")
            p_b = torch.zeros(48 * n, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            K.blob_to_kzg_commitment_batch_device(o_a.data_ptr(), d_a.data_ptr(), n, ts, s1.cuda_stream, None)
            K.blob_to_kzg_commitment_batch_device(o_b.data_ptr(), d_b.data_ptr(), n, ts, s2.cuda_stream, None)
            K.compute_blob_kzg_proof_batch_device(p_a.data_ptr(), d_a.data_ptr(), o_a.data_ptr(), n, ts, s1.cuda_stream, None)
            K.compute_blob_kzg_proof_batch_device(p_b.data_ptr(), d_b.data_ptr(), o_b.data_ptr(), n, ts, s2.cuda_stream, None)
            s2.synchronize()
            K.compute_blob_kzg_proof_batch_device(p_b.data_ptr(), d_b.data_ptr(), o_b.data_ptr(), n, ts, s1.cuda_stream, None)   # crosses streams
            torch.cuda.synchronize()
            assert _host(o_a) == want_a and _host(o_b) == want_b
            assert _host(p_a) == b"".join(want_pa) and _host(p_b) == b"".join(want_pb)
    ts.free()


@pytest.mark.parametrize("which", ["default", "bucket"])
def test_sixteen_threads_on_the_reference_symbol_are_coalesced(K, gpu_setup, bucket_setup, oracle, oracle_setup, which):
    """The reference's contract is lock-free concurrent calls on one KZGSettings (src/lib.rs:253-283, SURVEY 8b
    "Threading"). Sixteen threads hammering blob_to_kzg_commitment -- one blob per call -- are merged into shared launch
    sets by the library: every answer is the closed form's / the oracle's, and on the default engine the aggregate rate is
    above the north star's 10k ops/s although no caller ever passes more than one blob."""
    import threading
    import time
    ts = gpu_setup if which == "default" else bucket_setup
    n_threads, per_thread = 16, 150
    blobs = [B.synthetic_blob(90000 + i) for i in range(64)]
    want = [tau_closed_form(oracle, B.blob_scalars(b)) for b in blobs]
    for i in (0, 63):
        assert oracle.blob_to_kzg_commitment(blobs[i], oracle_setup, oracle.MODE_R) == (0, want[i])
    assert K.blob_to_kzg_commitment(blobs[0], ts) == want[0]          # warm: workspace, pinned staging
    errors, done = [], []

    def worker(t):
        try:
            for it in range(per_thread):
                i = (t * 7 + it * 13) % 64
                assert K.blob_to_kzg_commitment(blobs[i], ts) == want[i], (t, it, i)
            done.append(t)
        except Exception as e:      # pragma: no cover - reported below
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    el = time.perf_counter() - t0
    assert not any(t.is_alive() for t in threads), "deadlock"
    assert errors == [] and len(done) == n_threads
    rate = n_threads * per_thread / el
    print("coalesced single-blob calls, %s engine: %.0f ops/s from %d threads" % (which, rate, n_threads))
    if which == "default":
        assert rate >= 10000, rate
    # an invalid blob among valid ones (c-kzg mode: non-canonical element) is rejected for its caller only
    K.set_mode(K.MODE_CKZG)
    good = B.synthetic_blob(91000, big_endian=False)
    want_good = K.blob_to_kzg_commitment(good, ts)
    res = {}

    def mixed(t):
        try:
            if t % 4 == 0:
                K.blob_to_kzg_commitment(B.make_blob("all_ff"), ts)
                res[t] = "accepted"
            else:
                res[t] = K.blob_to_kzg_commitment(good, ts)
        except K.KzgError as e:
            res[t] = e.rc

    threads = [threading.Thread(target=mixed, args=(t,)) for t in range(12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    for t in range(12):
        assert res[t] == (K.C_KZG_BADARGS if t % 4 == 0 else want_good), (t, res[t])


def test_everything_at_once_on_one_settings_object(K, gpu_setup, oracle):
    """Threads of every kind on ONE KZGSettings for a few seconds: coalesced single-blob commitments, host-pointer batches,
    device-resident commitments + proofs on two caller streams (main context and twin), batch verification. Every answer
    must be the single-threaded one; nothing may deadlock between the context, verify and coalescing locks."""
    import threading
    import time
    import torch
    ts = gpu_setup
    blobs = [B.synthetic_blob(95000 + i) for i in range(48)]
    joined = b"".join(blobs)
    want_c = K.blob_to_kzg_commitment_batch(joined, ts)
    for i in (0, 47):
        assert want_c[i] == tau_closed_form(oracle, B.blob_scalars(blobs[i]))
    cj = b"".join(want_c)
    want_p = K.compute_blob_kzg_proof_batch(joined, cj, ts)
    for i in (0, 47):
        assert want_p[i] == reference_mode_proof_closed_form(oracle, blobs[i], want_c[i])
    pj = b"".join(want_p)
    d_blobs = _dev(joined)
    d_comm = _dev(cj)
    import os
    stop = time.time() + float(os.environ.get("LWKZG_TEST_STRESS_SECONDS", "4"))     # a longer run: tools/r02_experiments/r02_longsoak.sh
    errors, counts = [], {}

    def run(name, fn):
        k = 0
        try:
            while time.time() < stop:
                fn(k)
                k += 1
        except Exception as e:      # pragma: no cover - reported below
            errors.append("%s: %r" % (name, e))
        counts[name] = counts.get(name, 0) + k

    def single(k):
        i = k % 48
        assert K.blob_to_kzg_commitment(blobs[i], ts) == want_c[i]
        if k % 3 == 0:
            assert K.compute_blob_kzg_proof(blobs[i], want_c[i], ts) == want_p[i]

    def host_batch(k):
        lo = (k * 5) % 32
        assert K.blob_to_kzg_commitment_batch(joined[lo * B.BYTES_PER_BLOB:(lo + 16) * B.BYTES_PER_BLOB], ts) == want_c[lo:lo + 16]
        assert K.compute_blob_kzg_proof_batch(joined[lo * B.BYTES_PER_BLOB:(lo + 4) * B.BYTES_PER_BLOB], cj[48 * lo:48 * (lo + 4)], ts) == want_p[lo:lo + 4]

    def device(stream):
        o = torch.zeros(48 * 48, dtype=torch.uint8, device="cuda")
        p = torch.zeros(48 * 48, dtype=torch.uint8, device="cuda")

        def fn(k):
            if k % 3 == 2:      # every third round: both in one pass (the hash's first 2048 blocks beside the commitment MSM)
                K.commit_and_prove_batch_device(o.data_ptr(), p.data_ptr(), d_blobs.data_ptr(), 48, ts, stream.cuda_stream, None)
            else:
                K.blob_to_kzg_commitment_batch_device(o.data_ptr(), d_blobs.data_ptr(), 48, ts, stream.cuda_stream, None)
                K.compute_blob_kzg_proof_batch_device(p.data_ptr(), d_blobs.data_ptr(), d_comm.data_ptr(), 48, ts, stream.cuda_stream, None)
            stream.synchronize()
            assert _host(o) == cj and _host(p) == pj
        return fn

    def verify(k):
        assert K.verify_blob_kzg_proof_batch(joined[:8 * B.BYTES_PER_BLOB], cj[:48 * 8], pj[:48 * 8], 8, ts) is True
        assert K.verify_blob_kzg_proof_batch(joined[:8 * B.BYTES_PER_BLOB], cj[:48 * 8], pj[48:48 * 9], 8, ts) is False

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    jobs = [("single", single)] * 4 + [("host_batch", host_batch)] * 2 + [("device_s1", device(s1)), ("device_s2", device(s2)), ("verify", verify)]
    threads = [threading.Thread(target=run, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "deadlock"
    assert errors == [], errors
    assert all(counts.get(n, 0) > 0 for n, _ in jobs), counts
    print("mixed concurrency: " + ", ".join("%s x%d" % kv for kv in sorted(counts.items())))


def test_concurrent_single_blob_proofs_are_coalesced(K, gpu_setup, oracle, oracle_setup):
    """compute_blob_kzg_proof, one blob per call, from twelve threads (a block builder proving its blobs in parallel):
    merged into shared host-pointer batches by the library. Every proof equals the oracle's; a thread that passes an
    invalid commitment gets its error while the others -- members of the same merged batch -- get their proofs."""
    import threading
    import time
    ts = gpu_setup
    blobs = [B.synthetic_blob(98000 + i) for i in range(24)]
    comms = K.blob_to_kzg_commitment_batch(b"".join(blobs), ts)
    want = [reference_mode_proof_closed_form(oracle, b, c) for b, c in zip(blobs, comms)]
    for i in (0, 23):
        assert oracle.compute_blob_kzg_proof(blobs[i], comms[i], oracle_setup, oracle.MODE_R) == (0, want[i])
    assert K.compute_blob_kzg_proof(blobs[0], comms[0], ts) == want[0]
    errors, bad_seen = [], []
    n_threads, per_thread = 12, 40

    def worker(t):
        try:
            for it in range(per_thread):
                i = (t * 5 + it * 7) % 24
                if t == 3 and it % 4 == 0:           # an invalid commitment in the middle of everybody else's calls
                    try:
                        K.compute_blob_kzg_proof(blobs[i], bytes(48), ts)
                        errors.append("invalid commitment accepted")
                    except K.KzgError as e:
                        bad_seen.append(e.rc)
                else:
                    assert K.compute_blob_kzg_proof(blobs[i], comms[i], ts) == want[i], (t, it, i)
        except Exception as e:      # pragma: no cover - reported below
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    el = time.perf_counter() - t0
    assert not any(t.is_alive() for t in threads), "deadlock"
    assert errors == []
    assert bad_seen == [K.C_KZG_ERROR] * (per_thread // 4)
    print("coalesced single-blob proofs: %.0f proofs/s from %d threads" % (n_threads * per_thread / el, n_threads))


@pytest.mark.parametrize("mode_name", ["reference", "ckzg"])
def test_concurrent_single_point_proofs_are_coalesced(K, gpu_setup, oracle, oracle_setup, mode_name):
    """compute_kzg_proof, one blob per call, from twelve threads: merged the same way (blobs and evaluation points side by
    side). Every (proof, y) equals the oracle's; in c-kzg mode a thread that passes a non-canonical z gets C_KZG_BADARGS
    while the members of the same merged batch get their answers."""
    import threading
    import time
    ts = gpu_setup
    ckzg = mode_name == "ckzg"
    K.set_mode(K.MODE_CKZG if ckzg else K.MODE_REFERENCE)
    try:
        omode = oracle.MODE_C if ckzg else oracle.MODE_R
        blobs = [B.synthetic_blob(99000 + i, big_endian=not ckzg) for i in range(16)]
        zs = [B.synthetic_blob(99500 + i, big_endian=not ckzg)[:32] for i in range(16)]
        want = []
        for b, z in zip(blobs, zs):
            rc, pr, y = oracle.compute_kzg_proof(b, z, oracle_setup, omode)
            assert rc == 0
            want.append((pr, y))
        assert K.compute_kzg_proof(blobs[0], zs[0], ts) == want[0]
        errors, bad_seen = [], []
        n_threads, per_thread = 12, 40

        def worker(t):
            try:
                for it in range(per_thread):
                    i = (t * 5 + it * 7) % 16
                    if ckzg and t == 3 and it % 4 == 0:
                        try:
                            K.compute_kzg_proof(blobs[i], b"\xff" * 32, ts)
                            errors.append("non-canonical z accepted")
                        except K.KzgError as e:
                            bad_seen.append(e.rc)
                    else:
                        assert K.compute_kzg_proof(blobs[i], zs[i], ts) == want[i], (t, it, i)
            except Exception as e:      # pragma: no cover - reported below
                errors.append(repr(e))

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
        el = time.perf_counter() - t0
        assert not any(t.is_alive() for t in threads), "deadlock"
        assert errors == []
        assert bad_seen == ([K.C_KZG_BADARGS] * (per_thread // 4) if ckzg else [])
        print("coalesced single point proofs (%s): %.0f proofs/s from %d threads" % (mode_name, n_threads * per_thread / el, n_threads))
    finally:
        K.set_mode(K.MODE_REFERENCE)


@pytest.mark.parametrize("mode_name", ["reference", "ckzg"])
@pytest.mark.parametrize("n", [1, 65, 256, 1024, 2100])
def test_commit_and_prove_in_one_pass_equals_the_two_calls(K, gpu_setup, bucket_setup, oracle, oracle_setup, n, mode_name):
    """lwkzg_commit_and_prove_batch_device -- the challenge hash's commitment-independent 2048 blocks beside the
    commitment MSM, the last two blocks (k_challenge_finish) once the commitments exist -- returns the bytes of
    blob_to_kzg_commitment followed by compute_blob_kzg_proof: against the two device calls on every blob (default
    engine, and the bucket engine at 65), against the CPU oracle on a spread, and the challenges against hashlib
    through the separate hash entry point (n = 2100: a call of three chunks)."""
    import torch
    ckzg = mode_name == "ckzg"
    K.set_mode(K.MODE_CKZG if ckzg else K.MODE_REFERENCE)
    try:
        data = B.synthetic_batch(120000 + n, n, big_endian=not ckzg)
        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        for ts in ([gpu_setup, bucket_setup] if n == 65 else [gpu_setup]):
            c1 = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
            p1 = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
            c2 = torch.zeros(48 * n, dtype=torch.uint8, device="cuda")
            p2 = torch.zeros(48 * n, dtype=torch.uint8, device="cuda")
            st = torch.full((n,), 7, dtype=torch.int32, device="cuda")
            K.blob_to_kzg_commitment_batch_device(c1.data_ptr(), d_in.data_ptr(), n, ts, None, None)
            K.compute_blob_kzg_proof_batch_device(p1.data_ptr(), d_in.data_ptr(), c1.data_ptr(), n, ts, None, None)
            K.commit_and_prove_batch_device(c2.data_ptr(), p2.data_ptr(), d_in.data_ptr(), n, ts, None, st.data_ptr())
            torch.cuda.synchronize()
            assert int(st.abs().sum().item()) == 0
            assert torch.equal(c1, c2), "commitments differ"
            assert torch.equal(p1, p2), "proofs differ"
        comm = bytes(c2.cpu().numpy().tobytes())
        proof = bytes(p2.cpu().numpy().tobytes())
        omode = oracle.MODE_C if ckzg else oracle.MODE_R
        for i in sorted({0, n // 2, n - 1}):
            blob = data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB]
            assert oracle.blob_to_kzg_commitment(blob, oracle_setup, omode) == (0, comm[48 * i:48 * i + 48])
            assert oracle.compute_blob_kzg_proof(blob, comm[48 * i:48 * i + 48], oracle_setup, omode) == (0, proof[48 * i:48 * i + 48])
    finally:
        K.set_mode(K.MODE_REFERENCE)


def test_commit_and_prove_flags_a_non_canonical_blob(K, gpu_setup):
    """c-kzg mode: a blob with an element >= r is marked in the status array by the fused entry point as well; its
    neighbours get their results."""
    import torch
    K.set_mode(K.MODE_CKZG)
    try:
        n = 5
        blobs = [bytearray(B.synthetic_blob(130000 + i, big_endian=False)) for i in range(n)]
        blobs[3][32 * 100:32 * 101] = b"\xff" * 32
        data = b"".join(bytes(b) for b in blobs)
        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        c = torch.zeros(48 * n, dtype=torch.uint8, device="cuda")
        p = torch.zeros(48 * n, dtype=torch.uint8, device="cuda")
        st = torch.zeros(n, dtype=torch.int32, device="cuda")
        K.commit_and_prove_batch_device(c.data_ptr(), p.data_ptr(), d_in.data_ptr(), n, gpu_setup, None, st.data_ptr())
        torch.cuda.synchronize()
        assert [int(x != 0) for x in st.cpu().tolist()] == [0, 0, 0, 1, 0]
        for i in (0, 4):
            want_c = K.blob_to_kzg_commitment(bytes(blobs[i]), gpu_setup)
            assert bytes(c[48 * i:48 * i + 48].cpu().numpy().tobytes()) == want_c
            assert bytes(p[48 * i:48 * i + 48].cpu().numpy().tobytes()) == K.compute_blob_kzg_proof(bytes(blobs[i]), want_c, gpu_setup)
    finally:
        K.set_mode(K.MODE_REFERENCE)


# ---- small device-resident calls: the host-assisted challenge (engine.hip: small_proof_host_fn), VERDICT r03 item 6b -----------------

@pytest.mark.parametrize("n,first", [(1, 21000), (16, 21000), (64, 21000)])   # (65, 70 above: also the host-assisted path; 256, 1024: the GPU chains)
@pytest.mark.parametrize("mode_c", [False, True], ids=["reference", "ckzg"])
def test_small_device_resident_blob_proofs_vs_oracle(K, engine_setup, oracle, n, first, mode_c):
    """up to 128 blobs the Fiat-Shamir challenges and the commitment validation of a device-resident call come from the host threads
    (copy out, hipLaunchHostFunc, digests back); every proof against the oracle (the blobs of the 64-blob case above, shared)"""
    K.set_mode(K.MODE_CKZG if mode_c else K.MODE_REFERENCE)
    blobs, want_c, want_p = oracle_batch(oracle, first, 64, mode_c)
    comms, proofs = device_commit_and_prove(K, engine_setup, b"".join(blobs[:n]), n)
    for i in range(n):
        assert comms[48 * i:48 * i + 48] == want_c[i], ("commitment", i)
        assert proofs[48 * i:48 * i + 48] == want_p[i], ("proof", i)


@pytest.mark.parametrize("n", [12, 200], ids=["small_12", "mid_200"])
@pytest.mark.parametrize("mode_c", [False, True], ids=["reference", "ckzg"])
def test_small_device_resident_call_rejects_only_the_bad_commitment_and_rehashes_odd_encodings(K, gpu_setup, oracle, oracle_setup, mode_c, n):
    """one commitment that is not on the curve, one in the wrong subgroup, one infinity with stray flag bits (valid, non-canonical: the
    challenge must be taken over the canonical c0 00.. bytes): per-blob status, the other lanes unharmed, and the same answers as the
    host-pointer ABI gives blob by blob; back-to-back small calls on one stream keep their own arguments. n = 200 takes the mid-size path
    (host hashing in chunks beside the copy out, validation on the GPU)."""
    import torch
    mode = K.MODE_CKZG if mode_c else K.MODE_REFERENCE
    K.set_mode(mode)
    data = bytearray(B.synthetic_batch(33000, n, big_endian=not mode_c))
    data[5 * B.BYTES_PER_BLOB:6 * B.BYTES_PER_BLOB] = bytes(B.BYTES_PER_BLOB)           # the zero polynomial: commitment = infinity
    data = bytes(data)
    comms = bytearray(b"".join(K.blob_to_kzg_commitment_batch(data, gpu_setup)))
    assert comms[48 * 5] == 0xc0
    comms[48 * 2 + 20] ^= 1                                                              # off the curve (or out of the subgroup)
    comms[48 * 9:48 * 10] = bytes([0x80]) + bytes(47)                                    # (0, 2): on the curve, not in G1
    comms = bytes(comms)
    want, want_rc = [], []
    for i in range(n):
        try:
            want.append(K.compute_blob_kzg_proof(data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB], comms[48 * i:48 * i + 48], gpu_setup))
            want_rc.append(0)
        except K.KzgError as e:
            want.append(None)
            want_rc.append(e.rc)
    assert [i for i in range(n) if want_rc[i]] == [2, 9] and want_rc[2] == (K.C_KZG_BADARGS if mode_c else K.C_KZG_ERROR)
    omode = oracle.MODE_C if mode_c else oracle.MODE_R
    assert oracle.compute_blob_kzg_proof(data[:B.BYTES_PER_BLOB], comms[:48], oracle_setup, omode) == (0, want[0])
    d_blobs, d_comm = _dev(data), _dev(comms)
    outs = [torch.empty(48 * n, dtype=torch.uint8, device="cuda") for _ in range(3)]
    stats = [torch.full((n,), 9, dtype=torch.int32, device="cuda") for _ in range(3)]
    for k in range(3):                                                                   # three calls enqueued before anything is awaited
        m = n - k
        K.compute_blob_kzg_proof_batch_device(outs[k].data_ptr(), d_blobs.data_ptr(), d_comm.data_ptr(), m, gpu_setup, None, stats[k].data_ptr())
    torch.cuda.synchronize()
    for k in range(3):
        m = n - k
        st = stats[k].cpu().tolist()
        got = _host(outs[k])
        assert st[:m] == want_rc[:m], (k, st)
        for i in range(m):
            if want[i] is not None:
                assert got[48 * i:48 * i + 48] == want[i], (k, i)


@pytest.mark.parametrize("n", [128, 256, 300, 384])
@pytest.mark.parametrize("mode_c", [False, True], ids=["reference", "ckzg"])
def test_pipelined_mid_size_call_takes_odd_encodings_through_its_second_pass(K, gpu_setup, oracle, oracle_setup, mode_c, n):
    """r05: a mid-size device-resident call starts each sub-batch's quotient and MSM as soon as ITS chunks are hashed -- over the caller's
    commitment bytes, without waiting for the validation; a commitment that is valid but NOT canonically encoded (infinity with stray
    bits, for the zero blob) must get its challenge, quotient and proof again once the canonical bytes exist. One such blob in every
    sub-batch (and none in a control call): every proof against the host-pointer call blob by blob, and the oracle on a sample."""
    import torch
    mode = K.MODE_CKZG if mode_c else K.MODE_REFERENCE
    K.set_mode(mode)
    data = bytearray(B.synthetic_batch(36000 + n, n, big_endian=not mode_c))
    zeros = sorted(set([3, n // 4 + 1, n // 2 + 2, 3 * n // 4 + 3, n - 1]))
    for i in zeros:
        data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB] = bytes(B.BYTES_PER_BLOB)
    data = bytes(data)
    comms = bytearray(b"".join(K.blob_to_kzg_commitment_batch(data, gpu_setup)))
    canonical = bytes(comms)
    for k, i in enumerate(zeros):
        assert comms[48 * i] == 0xc0
        comms[48 * i + 1:48 * i + 48] = bytes((7 * k + j) % 251 + 1 for j in range(47))      # stray bits behind the infinity flag
    comms = bytes(comms)
    omode = oracle.MODE_C if mode_c else oracle.MODE_R
    want = K.compute_blob_kzg_proof_batch(data, canonical, gpu_setup)
    for i in (0, zeros[1], n - 2):
        assert oracle.compute_blob_kzg_proof(data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB], canonical[48 * i:48 * i + 48], oracle_setup, omode) == (0, want[i])
    d_blobs = _dev(data)
    for cm in (comms, canonical, comms):
        d_comm = _dev(cm)
        d_out = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
        d_st = torch.full((n,), 9, dtype=torch.int32, device="cuda")
        K.compute_blob_kzg_proof_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), d_comm.data_ptr(), n, gpu_setup, None, d_st.data_ptr())
        torch.cuda.synchronize()
        assert int(d_st.abs().sum()) == 0
        got = _host(d_out)
        bad = [i for i in range(n) if got[48 * i:48 * i + 48] != want[i]]
        assert not bad, bad[:8]
