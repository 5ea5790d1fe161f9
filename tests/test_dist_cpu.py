"""CPU: the multi-GPU layer's host logic -- contiguous sharding and the setup-image broadcast /
result gather -- with world_size 2 over gloo (the GPU path uses the same code over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lambdaworks_kzg_amd import dist as D


def test_shard_range_partitions_exactly():
    for n in (0, 1, 2, 7, 8, 9, 64, 1000, 1024, 4096):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                start, count = D.shard_range(n, world, r)
                assert count >= 0
                seen.extend(range(start, start + count))
                for k in range(start, start + count):
                    assert D.owner_of(k, n, world) == r      # blob k -> GPU floor(k * G / B), SURVEY 8e
            assert seen == list(range(n))
            counts = [D.shard_range(n, world, r)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1


def test_shard_range_rejects_bad_rank():
    with pytest.raises(ValueError):
        D.shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the "setup image": rank 0 owns the bytes, everyone ends up with them (one broadcast)
        img = torch.arange(4096, dtype=torch.int64).to(torch.uint8) if rank == 0 else torch.zeros(4096, dtype=torch.uint8)
        D.broadcast_bytes(img, src=0)
        ok_img = bool((img == torch.arange(4096, dtype=torch.int64).to(torch.uint8)).all())
        # every rank "commits" its own shard: result of item k is 48 bytes of value k % 251
        start, count = D.shard_range(n_items, world, rank)
        local = torch.stack([torch.full((48,), k % 251, dtype=torch.uint8) for k in range(start, start + count)]) \
            if count else torch.zeros((0, 48), dtype=torch.uint8)
        full = D.gather_shards(local, n_items, 48)
        want = torch.stack([torch.full((48,), k % 251, dtype=torch.uint8) for k in range(n_items)]).reshape(-1)
        q.put((rank, ok_img, bool((full == want).all())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [5, 8])
def test_world_size_2_broadcast_and_gather(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True)]


class _FakeShard:
    """Stands in for capi.VerifyShard on a machine without a GPU: record i of the batch is 160 bytes derived from the
    blob's first byte; the partial sum is a digest of (whole transcript, first index, shard length), so the finish step
    can check that every rank saw the same complete, ordered transcript and its own position in it."""

    def __init__(self, blobs, commitments, proofs, n_local, ts):
        import hashlib
        from lambdaworks_kzg_amd import capi
        if n_local and blobs[:1] == b"\xff":
            raise capi.KzgError("lwkzg_verify_shard_begin", capi.C_KZG_ERROR)
        if n_local and blobs[:1] == b"\xfe":
            raise TypeError("not a KzgError: a ctypes argument of the wrong type, a failed assertion ...")
        self.fail_partial = n_local and blobs[:1] == b"\xfd"
        self.n = n_local
        self.records = b"".join(hashlib.sha256(blobs[i:i + 1] + commitments[i:i + 1]).digest() * 5 for i in range(n_local))
        self.freed = False

    def partial(self, records_all, n_total, first):
        import hashlib
        from lambdaworks_kzg_amd import capi
        if self.fail_partial:
            raise capi.KzgError("lwkzg_verify_shard_partial", capi.C_KZG_MALLOC)
        assert records_all[160 * first:160 * (first + self.n)] == self.records
        return (hashlib.sha256(records_all + bytes([first, self.n, n_total])).digest() * 11)[:328]

    def free(self):
        self.freed = True


def _verify_worker(rank, world, port, case, q):
    import hashlib
    from lambdaworks_kzg_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the batch: 5 one-byte "blobs", rank 0 holds 3 of them (shard_range), rank 1 two -- or nothing at all
        n_total = 0 if case.startswith("empty") else 5
        if case == "empty_ckzg":
            capi.set_mode(capi.MODE_CKZG)
        start, count = D.shard_range(n_total, world, rank)
        blobs = bytes(range(10 + start, 10 + start + count))
        if case == "bad_rank_1" and rank == 1:
            blobs = b"\xff" + blobs[1:]
        if case == "type_error_rank_0" and rank == 0:
            blobs = b"\xfe" + blobs[1:]
        if case == "partial_fails_rank_1" and rank == 1:
            blobs = b"\xfd" + blobs[1:]
        comms = bytes(range(50 + start, 50 + start + count))

        def finish(partials, n_shards, n_tot, ts):
            all_rec = b"".join(hashlib.sha256(bytes([10 + i]) + bytes([50 + i])).digest() * 5 for i in range(n_tot))
            want = b"".join((hashlib.sha256(all_rec + bytes([D.shard_range(n_tot, n_shards, r)[0], D.shard_range(n_tot, n_shards, r)[1], n_tot])).digest() * 11)[:328]
                            for r in range(n_shards))
            return partials == want and n_shards == world and n_tot == n_total

        try:
            verdict = D.verify_blob_kzg_proof_batch_sharded(blobs, comms, comms, count, None, _shard=_FakeShard, _finish=finish)
        except capi.KzgError as e:
            verdict = ("error", e.rc)
        except TypeError:
            verdict = ("error", "TypeError")
        q.put((rank, verdict))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["honest", "bad_rank_1", "type_error_rank_0", "partial_fails_rank_1", "empty", "empty_ckzg"])
def test_sharded_batch_verification_gathers_one_transcript(case):
    """world size 2 over gloo: both ranks see the whole transcript in order, the partial sums arrive in rank order, and
    both return the same verdict; a shard rejected on one rank -- by the library, by a Python-side exception, or only in
    its partial sums after the first data collective -- makes BOTH ranks raise (nobody hangs in a collective); the empty
    batch is False as in the reference (lib.rs:538-543) and True in c-kzg mode (the reference's own vector a271b78b8e869d69)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verify_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = {"honest": (True, True), "bad_rank_1": (("error", 2), ("error", 2)),
            "type_error_rank_0": (("error", "TypeError"), ("error", 2)),     # the failing rank raises its own exception
            "partial_fails_rank_1": (("error", 3), ("error", 3)),
            "empty": (False, False), "empty_ckzg": (True, True)}[case]
    assert res == [(0, want[0]), (1, want[1])]
