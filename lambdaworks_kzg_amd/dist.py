"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

SURVEY section 8e: batches of blobs are independent units -> contiguous shards, every GPU holds the full
setup, NO data-path collective for commitments and proofs. The only collective there is ONE broadcast of the
prepared setup image (g1_values | g2_values | fixed-base table | twiddles, about 10 MB) from the rank that loaded
and validated the trusted setup, over xGMI. Batch verification has the one real exchange of the path: the
Fiat-Shamir scalar of the random linear combination hashes every blob's (C, z, y, pi), so the ranks all-gather
160 bytes per blob and 328 bytes of partial sums per rank (verify_blob_kzg_proof_batch_sharded). The 2^20-term MSM
gathers one 48-byte partial sum per rank. torch is plumbing here (device memory + the process group).
"""
import torch
import torch.distributed as dist

from . import capi


def shard_range(n_items, world_size, rank):
    """Contiguous partition: item k belongs to rank floor(k * world_size / n_items).
    Returns (start, count) for `rank`; the shards cover [0, n_items) exactly once."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    # ONE rule for the C library's node-level entry points (csrc/multi.hip: lwkzg_shard_range) and for this module: part k owns
    # [ceil(k n / G), ceil((k + 1) n / G)). Closed form here -- a launcher or a CPU-only rank that only wants its slice must not need the
    # native library; tests/test_capi_cpu.py::test_one_shard_rule_from_c_and_python holds the two against each other.
    lo = -(-rank * n_items // world_size)
    hi = -(-(rank + 1) * n_items // world_size)
    return lo, hi - lo


def owner_of(item, n_items, world_size):
    return item * world_size // n_items


def broadcast_bytes(buf, src=0, group=None):
    """Broadcast a uint8 tensor in place (CPU tensor under gloo, device tensor under RCCL)."""
    if buf.dtype != torch.uint8:
        raise TypeError("broadcast_bytes expects a uint8 tensor")
    dist.broadcast(buf, src=src, group=group)
    return buf


def broadcast_trusted_setup(ts, device, src=0, group=None):
    """Rank `src` passes its loaded TrustedSetup, every other rank passes None.
    Returns a TrustedSetup usable on this rank's GPU. One RCCL broadcast, no other traffic."""
    rank = dist.get_rank(group)
    nbytes = capi.setup_image_bytes()
    image = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if rank == src:
        if ts is None:
            raise ValueError("source rank must hold a loaded trusted setup")
        ts.export_device_image(image.data_ptr())
    broadcast_bytes(image, src=src, group=group)
    if rank == src:
        return ts
    torch.cuda.synchronize(device)
    out = capi.TrustedSetup.from_device_image(image.data_ptr())
    return out


def gather_shards(local, n_items, item_bytes, group=None):
    """All ranks contribute their shard's result bytes (uint8 tensor, count x item_bytes); every rank
    gets the full n_items x item_bytes tensor in item order. Used by tests and by callers that want
    results in one place; the bench does not need it."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [shard_range(n_items, world, r)[1] for r in range(world)]
    mx = max(counts) if counts else 0
    dev = local.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
    pad = torch.zeros(mx * item_bytes, dtype=torch.uint8, device=dev)
    pad[: counts[rank] * item_bytes] = local.reshape(-1)[: counts[rank] * item_bytes].to(dev)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][: counts[r] * item_bytes] for r in range(world)])


def msm_tiled_sharded(scalars_be, n_terms, ts, device, group=None):
    """BASELINE config "2^20-point G1 MSM (tiled trusted setup), 1->8 GPUs": sum_k s_k * g1[k mod 4096].
    `scalars_be` is THIS rank's shard: a uint8 device tensor holding a whole number of 4096-scalar tiles
    (shard_range over the tiles). Each GPU produces one partial sum; the only exchange is an all_gather of
    the 48-byte compressed partial sums, added on the host. Returns the 48-byte result."""
    tiles_here = scalars_be.numel() // (32 * 4096)
    out = torch.empty(48, dtype=torch.uint8, device=device)
    if tiles_here:
        capi.g1_msm_tiled_device(out.data_ptr(), scalars_be.data_ptr(), tiles_here * 4096, ts)
        torch.cuda.synchronize(device)
    else:
        out.zero_()
        out[0] = 0xC0     # the empty sum: point at infinity
    if group is None and not dist.is_initialized():
        return bytes(out.cpu().numpy().tobytes())
    world = dist.get_world_size(group)
    if dist.get_backend(group) != "nccl":     # gloo (single-GPU plumbing tests) gathers host tensors
        out = out.cpu()
    parts = [torch.empty_like(out) for _ in range(world)]
    dist.all_gather(parts, out, group=group)
    return capi.g1_sum_compressed(b"".join(bytes(p.cpu().numpy().tobytes()) for p in parts))


def _all_gather_bytes(local, counts, item_bytes, device, group=None):
    """Every rank contributes counts[rank] items of item_bytes bytes (a bytes object); returns the concatenation in
    rank order on every rank. One all_gather of equal-sized (padded) tensors."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mx = max(max(counts), 1) * item_bytes
    pad = torch.zeros(mx, dtype=torch.uint8)
    if counts[rank]:
        pad[: counts[rank] * item_bytes] = torch.frombuffer(bytearray(local), dtype=torch.uint8)
    pad = pad.to(device)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return b"".join(bytes(parts[r][: counts[r] * item_bytes].cpu().numpy().tobytes()) for r in range(world))


def _empty_batch_verdict(ts):
    """n == 0: False in reference mode (/root/reference/src/lib.rs:538-543), True in c-kzg mode (the reference's own
    vector verify_blob_kzg_proof_batch_case_a271b78b8e869d69)."""
    mode = ts.get_mode() if ts is not None else capi.get_mode()
    return mode == capi.MODE_CKZG


def verify_blob_kzg_proof_batch_sharded(blobs, commitments, proofs, n_local, ts, group=None, _shard=None, _finish=None):
    """BASELINE config "verify_blob_kzg_proof_batch, 4096 blobs sharded across 8 GPUs" as the reference computes it
    (/root/reference/src/lib.rs:525-692, src/utils.rs:166-206): ONE batch, ONE Fiat-Shamir scalar r over the transcript
    of all blobs, ONE random linear combination, ONE pairing check -- SURVEY 8e's gather form.

    `blobs` / `commitments` / `proofs` are THIS rank's contiguous shard (bytes), `n_local` its length; rank k's shard
    follows rank k - 1's in the batch. Steps: (1) per blob on this rank's GPU: validation of C_i and pi_i, z_i, y_i;
    (2) all_gather of the 160-byte transcript records; (3) this rank's terms of the three linear combinations with the
    common r; (4) all_gather of the 328-byte partial sums (+ one status byte); every rank adds them and does the pairing
    check, so every rank returns the same verdict. Two data collectives, 160 bytes per blob + 329 bytes per rank.

    Errors: ANY failure of step (1) or (3) on ANY rank -- an invalid point or blob, an allocation or device error, a
    Python-side exception -- is carried through the next collective as a return code, so every rank reaches every
    collective and EVERY rank raises afterwards (the failing rank its own exception, the others a KzgError naming the
    lowest failing rank, as the reference returns at the first offending blob). An empty global batch returns False in
    reference mode, as the reference does for n == 0 (lib.rs:538-543), and True in c-kzg mode."""
    make_shard = _shard or capi.VerifyShard
    finish = _finish or capi.verify_shards_finish
    distributed = not (group is None and not dist.is_initialized())
    err_rc = 0
    err_first = None
    shard = None
    if not distributed and _shard is None and _finish is None:   # one process, one shard: the reference's own symbol
        return bool(capi.verify_blob_kzg_proof_batch(blobs, commitments, proofs, n_local, ts))
    try:
        shard = make_shard(blobs, commitments, proofs, n_local, ts)
    except Exception as e:                                       # noqa: BLE001 -- whatever it is, the other ranks must hear of it
        err_rc = getattr(e, "rc", 0) or capi.C_KZG_ERROR
        err_first = e
    try:
        if not distributed:
            if err_rc:
                raise err_first
            if n_local == 0:
                return _empty_batch_verdict(ts)
            return bool(finish(shard.partial(shard.records, n_local, 0), 1, n_local, ts))
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")

        def raise_for(rcs, what):
            bad = next(r for r in range(world) if rcs[r])
            if err_first is not None and bad == rank:
                raise err_first
            raise capi.KzgError("verify_blob_kzg_proof_batch_sharded (rank %d %s)" % (bad, what), rcs[bad])

        # shard lengths and error codes of every rank: the one small collective every rank always reaches
        meta = torch.tensor([n_local, err_rc], dtype=torch.int64, device=device)
        metas = [torch.empty_like(meta) for _ in range(world)]
        dist.all_gather(metas, meta, group=group)
        counts = [int(m[0]) for m in metas]
        rcs = [int(m[1]) for m in metas]
        if any(rcs):
            raise_for(rcs, "rejected its shard")
        n_total = sum(counts)
        if n_total == 0:
            return _empty_batch_verdict(ts)
        first = sum(counts[:rank])
        records_all = _all_gather_bytes(shard.records, counts, capi.VERIFY_RECORD_BYTES, device, group)
        # step (3) can fail on one rank only (allocation, device error): its code travels in the payload's last byte
        try:
            partial = shard.partial(records_all, n_total, first)
            if len(partial) != capi.VERIFY_PARTIAL_BYTES:
                raise ValueError("partial sums of %d bytes" % len(partial))
            partial += b"\0"
        except Exception as e:                                   # noqa: BLE001
            err_first = e
            partial = bytes(capi.VERIFY_PARTIAL_BYTES) + bytes([getattr(e, "rc", 0) or capi.C_KZG_ERROR])
        padded = _all_gather_bytes(partial, [1] * world, capi.VERIFY_PARTIAL_BYTES + 1, device, group)
        step = capi.VERIFY_PARTIAL_BYTES + 1
        rcs = [padded[step * r + step - 1] for r in range(world)]
        if any(rcs):
            raise_for(rcs, "failed in its partial sums")
        partials = b"".join(padded[step * r: step * r + step - 1] for r in range(world))
        return bool(finish(partials, world, n_total, ts))
    finally:
        if shard is not None:
            shard.free()
