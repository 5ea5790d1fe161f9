"""GPU: multi-GPU readiness on the one device this box has.
 * the 8-way sharded form of verify_blob_kzg_proof_batch at BASELINE configs[3]'s size (4096 blobs): the shards are run
   one after the other on this GPU through the same C entry points the ranks of an 8-GPU job call, and must give the
   single-batch verdicts; the common Fiat-Shamir scalar and its powers are checked against hashlib through the partial
   sums' scalar part;
 * a rehearsal with TWO fresh processes (gloo; RCCL needs one GPU per rank) of everything a rank does: setup export ->
   broadcast -> import, sharded commitments, the sharded long MSM, the sharded verification.
No scaling number comes out of this: 8-GPU throughput is unmeasured on hardware.
Reference: /root/reference/src/lib.rs:525-692, /root/reference/src/utils.rs:166-206."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import pytest

import blobs as B
from conftest import R, ROOT, TAU, tau_closed_form

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reference_mode(K):
    K.set_mode(K.MODE_REFERENCE)
    yield
    K.set_mode(K.MODE_REFERENCE)


def _sharded_verdict(capi, D, data, comms, proofs, n, world, ts, check_scalars=False):
    """the ranks of a `world`-GPU job, one after the other on this GPU"""
    shards, counts = [], []
    for r in range(world):
        st, cnt = D.shard_range(n, world, r)
        shards.append(capi.VerifyShard(data[st * B.BYTES_PER_BLOB:(st + cnt) * B.BYTES_PER_BLOB], comms[48 * st:48 * (st + cnt)],
                                       proofs[48 * st:48 * (st + cnt)], cnt, ts))
        counts.append(cnt)
    records = b"".join(s.records for s in shards)                  # what the all_gather delivers
    assert len(records) == 160 * n
    partials = []
    for r, s in enumerate(shards):
        partials.append(s.partial(records, n, sum(counts[:r])))
    if check_scalars:
        # r = SHA-256("RCKZGBATCH___V1_" | le64(4096) | le64(n) | records) read big-endian, reduced (utils.rs:166-206, 148-154);
        # bytes 291..323 of a partial sum are sum_{i in shard} r^i y_i
        rr = int.from_bytes(hashlib.sha256(b"RCKZGBATCH___V1_" + (4096).to_bytes(8, "little") + n.to_bytes(8, "little") + records).digest(), "big") % R
        for k in range(world):
            st, cnt = D.shard_range(n, world, k)
            want = sum(pow(rr, i, R) * int.from_bytes(records[160 * i + 80:160 * i + 112], "big") for i in range(st, st + cnt)) % R
            assert int.from_bytes(partials[k][291:323], "big") == want, k
        for i in (0, n // 2, n - 1):                               # the records themselves: C | z | y | pi with z from hashlib
            blob = data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB]
            z = int.from_bytes(hashlib.sha256(b"FSBLOBVERIFY_V1_" + (4096).to_bytes(8, "little") + bytes(8) + blob + comms[48 * i:48 * i + 48]).digest(), "big") % R
            y = 0
            for c in reversed(B.blob_scalars(blob)):
                y = (y * z + c) % R
            assert records[160 * i:160 * (i + 1)] == comms[48 * i:48 * i + 48] + z.to_bytes(32, "big") + y.to_bytes(32, "big") + proofs[48 * i:48 * i + 48]
    ok = capi.verify_shards_finish(b"".join(partials), world, n, ts)
    for s in shards:
        s.free()
    return ok


def test_verify_batch_4096_blobs_single_and_8_shards(K, gpu_setup):
    """BASELINE configs[3] at its stated size on one GPU: one 4096-blob verify_blob_kzg_proof_batch (honest, and with one
    wrong proof deep in the batch), and the 8 x 512 sharded form giving the same verdicts"""
    from lambdaworks_kzg_amd import capi
    from lambdaworks_kzg_amd import dist as D
    n = 4096
    data = B.synthetic_batch(70000, n)
    comms = b"".join(K.blob_to_kzg_commitment_batch(data, gpu_setup))
    proofs = b"".join(K.compute_blob_kzg_proof_batch(data, comms, gpu_setup))
    assert K.verify_blob_kzg_proof_batch(data, comms, proofs, n, gpu_setup) is True
    assert _sharded_verdict(capi, D, data, comms, proofs, n, 8, gpu_setup, check_scalars=True) is True
    wrong = bytearray(proofs)
    wrong[48 * 3011:48 * 3012] = proofs[48 * 7:48 * 8]              # a valid G1 point, not blob 3011's proof (shard 5)
    wrong = bytes(wrong)
    assert K.verify_blob_kzg_proof_batch(data, comms, wrong, n, gpu_setup) is False
    assert _sharded_verdict(capi, D, data, comms, wrong, n, 8, gpu_setup) is False
    # two blobs (with their commitments and proofs) exchanged ACROSS shards: still one honest batch, another transcript
    i, j = 100, 4000
    sw = lambda buf, w: buf[:w * i] + buf[w * j:w * (j + 1)] + buf[w * (i + 1):w * j] + buf[w * i:w * (i + 1)] + buf[w * (j + 1):]
    assert _sharded_verdict(capi, D, sw(data, B.BYTES_PER_BLOB), sw(comms, 48), sw(proofs, 48), n, 8, gpu_setup) is True
    # only the proofs exchanged: rejected
    assert _sharded_verdict(capi, D, data, comms, sw(proofs, 48), n, 8, gpu_setup) is False


@pytest.mark.parametrize("n,world", [(2, 2), (5, 8), (64, 3), (200, 8), (2300, 2)])
def test_sharded_verification_equals_single_batch(K, gpu_setup, n, world):
    """shard counts that leave some ranks empty (5 blobs on 8 ranks), every validation route (host threads up to 64 blobs
    per shard, GPU kernels above, the pipelined path above 1024), in both modes; invalid input on one shard is an error"""
    from lambdaworks_kzg_amd import capi
    from lambdaworks_kzg_amd import dist as D
    for mode, be in ((K.MODE_REFERENCE, True), (K.MODE_CKZG, False)):
        K.set_mode(mode)
        data = B.synthetic_batch(80000 + n, n, big_endian=be)
        comms = b"".join(K.blob_to_kzg_commitment_batch(data, gpu_setup))
        proofs = b"".join(K.compute_blob_kzg_proof_batch(data, comms, gpu_setup))
        assert K.verify_blob_kzg_proof_batch(data, comms, proofs, n, gpu_setup) is True
        assert _sharded_verdict(capi, D, data, comms, proofs, n, world, gpu_setup, check_scalars=be) is True
        k = n - 1
        wrong = proofs[:48 * k] + proofs[:48]
        assert K.verify_blob_kzg_proof_batch(data, comms, wrong, n, gpu_setup) is False
        assert _sharded_verdict(capi, D, data, comms, wrong, n, world, gpu_setup) is False
        with pytest.raises(capi.KzgError) as e:
            _sharded_verdict(capi, D, data, bytes([comms[0] & 0x7f]) + comms[1:], proofs, n, world, gpu_setup)
        assert e.value.rc == (K.C_KZG_ERROR if be else K.C_KZG_BADARGS)
    # the empty batch: OK with ok = false in reference mode (lib.rs:538-543), ok = true in c-kzg mode (vector a271b78b8e869d69)
    K.set_mode(K.MODE_REFERENCE)
    assert capi.verify_shards_finish(b"", 0, 0, gpu_setup) is False
    assert D.verify_blob_kzg_proof_batch_sharded(b"", b"", b"", 0, gpu_setup) is False
    K.set_mode(K.MODE_CKZG)
    assert capi.verify_shards_finish(b"", 0, 0, gpu_setup) is True
    assert D.verify_blob_kzg_proof_batch_sharded(b"", b"", b"", 0, gpu_setup) is True
    K.set_mode(K.MODE_REFERENCE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_process_rehearsal_of_a_multi_gpu_job(K, gpu_setup, oracle, tmp_path):
    """two FRESH processes on this one device, gloo backend: load -> lwkzg_setup_export_device -> broadcast ->
    lwkzg_setup_import_device -> sharded commitments / tiled MSM / sharded verification; bytes equal to the single-process
    results of this process"""
    import numpy as np
    import torch
    from lambdaworks_kzg_amd import capi
    worker = os.path.join(ROOT, "tests", "dist_gpu_worker.py")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_gpu_worker as W
    port = _free_port()
    env = dict(os.environ)
    env.pop("LWKZG_MODE", None)
    outs = [str(tmp_path / ("rank%d.json" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), outs[r]], env=env, stderr=subprocess.PIPE) for r in range(2)]
    errs = [p.communicate(timeout=900)[1].decode(errors="replace") for p in procs]
    assert [p.returncode for p in procs] == [0, 0], "\n".join(errs)
    res = [json.load(open(o)) for o in outs]
    # the imported setup is the loaded one
    want_sha = hashlib.sha256(gpu_setup.g1_values_bytes()).hexdigest()
    assert res[0]["g1_values_sha"] == res[1]["g1_values_sha"] == want_sha
    assert res[1]["direct_bits"] == res[0]["direct_bits"]           # the importing rank chose its engine like a loading one
    # commitments: both ranks hold the whole gathered batch, equal to the single-process bytes and the closed form
    data = B.synthetic_batch(60000, W.N_COMMIT)
    single = b"".join(K.blob_to_kzg_commitment_batch(data, gpu_setup))
    assert res[0]["commitments"] == res[1]["commitments"] == single.hex()
    for i in (0, W.N_COMMIT // 2, W.N_COMMIT - 1):
        assert single[48 * i:48 * i + 48] == tau_closed_form(oracle, B.blob_scalars(data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB]))
    # the tiled MSM
    tiles = B.synthetic_batch(61000, W.N_TILES)
    d_sc = torch.frombuffer(bytearray(tiles), dtype=torch.uint8).cuda()
    d_out = torch.empty(48, dtype=torch.uint8, device="cuda")
    capi.g1_msm_tiled_device(d_out.data_ptr(), d_sc.data_ptr(), W.N_TILES * 4096, gpu_setup)
    torch.cuda.synchronize()
    want_msm = bytes(d_out.cpu().numpy().tobytes())
    assert res[0]["tiled_msm"] == res[1]["tiled_msm"] == want_msm.hex()
    pw = [pow(TAU, i, R) for i in range(4096)]
    acc = 0
    for t in range(W.N_TILES):
        acc = (acc + sum(s * p for s, p in zip(B.blob_scalars(tiles[t * B.BYTES_PER_BLOB:(t + 1) * B.BYTES_PER_BLOB]), pw))) % R
    assert want_msm == oracle.g1_generator_mul(acc)
    # the sharded verification: same verdicts on both ranks, equal to the single batch; the error reaches both ranks
    for r in res:
        assert r["verify_honest"] is True and r["verify_tampered"] is False and r["verify_invalid"] == K.C_KZG_ERROR
    vdata = B.synthetic_batch(62000, W.N_VERIFY)
    comms = bytes.fromhex(res[0]["comms"]) + bytes.fromhex(res[1]["comms"])
    proofs = bytes.fromhex(res[0]["proofs"]) + bytes.fromhex(res[1]["proofs"])
    assert comms == b"".join(K.blob_to_kzg_commitment_batch(vdata, gpu_setup))
    assert proofs == b"".join(K.compute_blob_kzg_proof_batch(vdata, comms, gpu_setup))
    assert K.verify_blob_kzg_proof_batch(vdata, comms, proofs, W.N_VERIFY, gpu_setup) is True


def test_rccl_path_at_world_size_1(K, gpu_setup, tmp_path):
    """The RCCL transport of the multi-GPU layer, executed on the one GPU there is: a FRESH process initialises the "nccl"
    backend (= RCCL) at world size 1 and runs broadcast_trusted_setup (+ an explicit export -> broadcast -> import),
    gather_shards, msm_tiled_sharded and verify_blob_kzg_proof_batch_sharded through it -- device tensors in
    dist.broadcast / dist.all_gather, which the gloo rehearsal above never touches. Bytes equal to the plain calls.
    (N > 1 over xGMI is unmeasured on hardware: one-GPU boxes.)"""
    import torch
    from lambdaworks_kzg_amd import capi
    worker = os.path.join(ROOT, "tests", "dist_gpu_worker.py")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_gpu_worker as W
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("LWKZG_MODE", None)
    out = str(tmp_path / "rank0.json")
    p = subprocess.run([sys.executable, worker, "0", "1", str(_free_port()), out, "nccl"], env=env, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-4000:]
    res = json.load(open(out))
    assert res["backend"] == "nccl" and res["nccl_version"]
    want_sha = hashlib.sha256(gpu_setup.g1_values_bytes()).hexdigest()
    assert res["g1_values_sha"] == res["imported_g1_values_sha"] == want_sha
    assert res["imported_commitment"] == K.blob_to_kzg_commitment(B.synthetic_blob(60000), gpu_setup).hex()
    data = B.synthetic_batch(60000, W.N_COMMIT)
    assert res["commitments"] == b"".join(K.blob_to_kzg_commitment_batch(data, gpu_setup)).hex()
    tiles = B.synthetic_batch(61000, W.N_TILES)
    d_sc = torch.frombuffer(bytearray(tiles), dtype=torch.uint8).cuda()
    d_out = torch.empty(48, dtype=torch.uint8, device="cuda")
    capi.g1_msm_tiled_device(d_out.data_ptr(), d_sc.data_ptr(), W.N_TILES * 4096, gpu_setup)
    torch.cuda.synchronize()
    assert res["tiled_msm"] == bytes(d_out.cpu().numpy().tobytes()).hex()
    assert res["verify_honest"] is True and res["verify_tampered"] is False and res["verify_invalid"] == K.C_KZG_ERROR
    vdata = B.synthetic_batch(62000, W.N_VERIFY)
    assert bytes.fromhex(res["comms"]) == b"".join(K.blob_to_kzg_commitment_batch(vdata, gpu_setup))


def test_bench_gpus_2_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2` with NO torch.distributed.run in front of it must start its ranks itself (fresh child
    processes, created before the parent touches the GPU) and print the one parseable line: two ranks on this one device over gloo
    (RCCL refuses two ranks on one device), small batch, the bucket-free 10-bit table. The launcher form is what the rehearsal above
    and the driver use."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LWKZG_BENCH_DETAIL=str(tmp_path / "detail.json"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LWKZG_MODE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--batch", "64", "--direct-bits", "10", "--no-extra-legs", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-4000:]
    lines = [l for l in p.stdout.decode().split("\n") if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 8000, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["dist"]["initialised"] is True and line["dist"]["ranks"] == 2 and line["dist"]["backend"] == "gloo"
    assert line["config"]["direct_bits"] == 10 and line["config"]["direct_bits_min_over_ranks"] == 10
    assert line["roofline"]["kernel"].startswith("k_direct_accumulate") and line["roofline"]["avg_launch_ms"] > 0
    detail = json.load(open(env["LWKZG_BENCH_DETAIL"]))
    assert detail["value"] == pytest.approx(line["value"], rel=1e-5) and "scaling_note" in detail


def test_bench_gpus_8_rehearsal_on_one_device(tmp_path):
    """VERDICT r05 item 2, "rehearse first contact": the command the driver runs on an 8-GPU node, with its EIGHT ranks on this one
    device over gloo (RCCL refuses several ranks per device), the bucket engine (eight 41 GB tables do not fit one device) and a small
    batch. What it pins: eight ranks rendezvous, the setup image is broadcast once and imported seven times, every rank times its own
    shard, rank 0's line aggregates eight per-rank values and says dist.ranks == 8. NOT a scaling number: eight ranks share one GPU."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LWKZG_BENCH_DETAIL=str(tmp_path / "detail8.json"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LWKZG_MODE", "LWKZG_DIRECT_BITS"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--batch", "64", "--direct-bits", "0", "--no-extra-legs", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-4000:]
    lines = [l for l in p.stdout.decode().split("\n") if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 8000, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0
    d = line["dist"]
    assert d["initialised"] is True and d["ranks"] == 8 and d["backend"] == "gloo"
    assert len(d["per_rank_value"]) == 8 and all(v > 0 for v in d["per_rank_value"]) and len(d["devices"]) == 8
    assert line["config"]["direct_bits"] == 0
