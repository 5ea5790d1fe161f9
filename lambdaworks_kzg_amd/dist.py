"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

SURVEY section 8e: batches of blobs are independent units -> contiguous shards, every GPU holds the full
setup, NO data-path collective. The only collective is ONE broadcast of the prepared setup image
(g1_values | g2_values | fixed-base table | twiddles, about 8.6 MB) from the rank that loaded and
validated the trusted setup, over xGMI. torch is plumbing here (device memory + the process group).
"""
import torch
import torch.distributed as dist

from . import capi


def shard_range(n_items, world_size, rank):
    """Contiguous partition: item k belongs to rank floor(k * world_size / n_items).
    Returns (start, count) for `rank`; the shards cover [0, n_items) exactly once."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    start = -(-rank * n_items // world_size)          # ceil(rank * n / G)
    end = -(-(rank + 1) * n_items // world_size)
    return start, end - start


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
    pad = torch.zeros(mx * item_bytes, dtype=torch.uint8, device=local.device)
    pad[: counts[rank] * item_bytes] = local.reshape(-1)[: counts[rank] * item_bytes]
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
    parts = [torch.empty_like(out) for _ in range(world)]
    dist.all_gather(parts, out, group=group)
    return capi.g1_sum_compressed(b"".join(bytes(p.cpu().numpy().tobytes()) for p in parts))


def verify_blob_kzg_proof_batch_sharded(blobs, commitments, proofs, n_local, ts, group=None, _verify=None):
    """BASELINE config "verify_blob_kzg_proof_batch, 4096 blobs sharded across 8 GPUs": every rank verifies ITS
    contiguous shard as an independent batch (own Fiat-Shamir challenges, own random linear combination, own pairing
    check on its host), then the verdicts are AND-ed with one all_reduce of a single byte. Sound because each
    sub-batch check is sound on its own; no point or scalar crosses GPUs (SURVEY 8e allows the gather variant, this
    one needs less traffic). `blobs`/`commitments`/`proofs` are this rank's shard as bytes, `n_local` its length."""
    verify = _verify or capi.verify_blob_kzg_proof_batch
    ok = True if n_local == 0 else bool(verify(blobs, commitments, proofs, n_local, ts))
    if group is None and not dist.is_initialized():
        return ok
    flag = torch.tensor([1 if ok else 0], dtype=torch.uint8)
    backend = dist.get_backend(group)
    if backend == "nccl":
        flag = flag.cuda()
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item())
