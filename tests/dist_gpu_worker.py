"""One rank of the multi-GPU rehearsals that tests/test_gpu_dist.py runs on ONE device: two processes over gloo (RCCL
needs a GPU per rank), and ONE process over "nccl" (= RCCL on ROCm) at world size 1, which is how the RCCL code path
-- device tensors through dist.broadcast / dist.all_gather -- executes on a one-GPU box. Fresh process; everything a rank of an 8-GPU job does, in order: rendezvous, setup image
export -> broadcast -> import, its shard of a commitment batch, its tiles of a long MSM, its shard of a sharded batch
verification (honest and tampered). Results go to a JSON file the test compares with the single-process answers.

    python tests/dist_gpu_worker.py RANK WORLD PORT OUT.json [BACKEND=gloo]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

N_COMMIT, N_TILES, N_VERIFY = 96, 16, 160


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    import blobs as B
    import lambdaworks_kzg_amd as K
    from lambdaworks_kzg_amd import capi
    from lambdaworks_kzg_amd import dist as D
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    res = {"rank": rank, "backend": dist.get_backend()}
    try:
        if backend == "nccl":
            res["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        K.set_device(0)
        K.set_mode(K.MODE_REFERENCE)
        # rank 0 loads and validates; the others import what the broadcast delivered
        ts = K.TrustedSetup.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt")) if rank == 0 else None
        ts = D.broadcast_trusted_setup(ts, dev, src=0)
        res["direct_bits"] = ts.direct_table_bits()
        res["g1_values_sha"] = __import__("hashlib").sha256(ts.g1_values_bytes()).hexdigest()
        if world == 1:
            # at world size 1 the source rank keeps its own setup: import what went through the broadcast explicitly, and
            # commit with it, so that the receiving side of broadcast_trusted_setup has run on this backend too
            image = torch.empty(capi.setup_image_bytes(), dtype=torch.uint8, device=dev)
            ts.export_device_image(image.data_ptr())
            D.broadcast_bytes(image, src=0)
            torch.cuda.synchronize(dev)
            ts_imp = capi.TrustedSetup.from_device_image(image.data_ptr())
            res["imported_g1_values_sha"] = __import__("hashlib").sha256(ts_imp.g1_values_bytes()).hexdigest()
            res["imported_commitment"] = K.blob_to_kzg_commitment(B.synthetic_blob(60000), ts_imp).hex()
            ts_imp.free()

        # commitments: contiguous shard, no data-path collective; gathered here only so that the test can compare
        st, cnt = D.shard_range(N_COMMIT, world, rank)
        data = B.synthetic_batch(60000 + st, cnt)
        d_blobs = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        d_out = torch.empty(48 * cnt, dtype=torch.uint8, device=dev)
        K.blob_to_kzg_commitment_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), cnt, ts, None, None)
        torch.cuda.synchronize()
        full = D.gather_shards(d_out, N_COMMIT, 48)
        res["commitments"] = bytes(full.cpu().numpy().tobytes()).hex()

        # the long MSM over the tiled setup: this rank's tiles, one 48-byte partial sum gathered
        t_st, t_cnt = D.shard_range(N_TILES, world, rank)
        tiles = np.frombuffer(B.synthetic_batch(61000 + t_st, t_cnt), dtype=np.uint8)
        d_tiles = torch.from_numpy(tiles.copy()).to(dev)
        res["tiled_msm"] = D.msm_tiled_sharded(d_tiles, t_cnt * 4096, ts, dev).hex()

        # sharded batch verification: one transcript, one r, one pairing check
        v_st, v_cnt = D.shard_range(N_VERIFY, world, rank)
        vdata = B.synthetic_batch(62000 + v_st, v_cnt)
        comms = b"".join(K.blob_to_kzg_commitment_batch(vdata, ts))
        proofs = b"".join(K.compute_blob_kzg_proof_batch(vdata, comms, ts))
        res["verify_honest"] = D.verify_blob_kzg_proof_batch_sharded(vdata, comms, proofs, v_cnt, ts)
        bad = bytearray(proofs)
        if rank == world - 1:                       # one wrong (but valid) proof on the last rank only
            bad[48 * 5:48 * 6] = proofs[48 * 6:48 * 7]
        res["verify_tampered"] = D.verify_blob_kzg_proof_batch_sharded(vdata, comms, bytes(bad), v_cnt, ts)
        badc = bytearray(comms)
        if rank == 0:                               # an invalid encoding on rank 0 only: EVERY rank must raise
            badc[0] &= 0x7f
        try:
            D.verify_blob_kzg_proof_batch_sharded(vdata, bytes(badc), proofs, v_cnt, ts)
            res["verify_invalid"] = "no error"
        except capi.KzgError as e:
            res["verify_invalid"] = e.rc
        res["proofs"] = proofs.hex()
        res["comms"] = comms.hex()
        ts.free()
    finally:
        dist.destroy_process_group()
    with open(out_path, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
