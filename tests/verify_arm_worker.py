"""Run by tests/test_gpu_verify_msm.py in fresh processes (the library reads its environment once): the partial sums of a sharded batch
verification -- sum r^i pi_i, sum r^i z_i pi_i, sum r^i C_i as affine points, 328 bytes per shard -- and the verdicts, for a fixed set of
batches, printed as JSON. Every arm of the linear combinations (vmsm.hip as shipped; its overflow scan forced with LWKZG_VMSM_LIST_CAP=1;
r05's k_point_multiples + k_lincomb3 with LWKZG_VERIFY_MSM=0) must print the same bytes: they compute the same group elements."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402
import blobs as B  # noqa: E402
import lambdaworks_kzg_amd as K  # noqa: E402


def main():
    mode = K.MODE_CKZG if (len(sys.argv) > 1 and sys.argv[1] == "ckzg") else K.MODE_REFERENCE
    K.set_mode(mode)
    be = mode == K.MODE_REFERENCE
    ts = K.TrustedSetup.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"))
    out = {}
    for n in (1, 9, 70, 300, 1500, 2600):
        blobs = [B.synthetic_blob(61000 + 17 * n + i, big_endian=be) for i in range(n)]
        if n > 2:
            blobs[n // 3] = bytes(B.BYTES_PER_BLOB)            # commitment and proof at infinity
            const = bytearray(B.BYTES_PER_BLOB)
            const[31 if be else 0] = 5
            blobs[n // 2] = bytes(const)                        # proof at infinity
        data = b"".join(blobs)
        cj = b"".join(K.blob_to_kzg_commitment_batch(data, ts))
        pj = b"".join(K.compute_blob_kzg_proof_batch(data, cj, ts))
        cut = max(1, (2 * n) // 3)
        if n == 2600:
            # a first shard of 2100 blobs through host pointers is LONGER THAN ONE CHUNK: the staged form (engine.hip: verify_prepare_staged --
            # the head of the batch hashed by the GPU as it lands, the tail by the host threads) against the sliced one (LWKZG_HOST_STAGE=0).
            # Two all-zero blobs, one in each part, come with their commitment at infinity in a NON-canonical encoding (stray bits behind the
            # flags, which the reference does not inspect): their challenges must be taken over the canonical bytes on either side
            cut = 2100
            stray = bytes([0xc5]) + bytes([7] * 47)
            cjb = bytearray(cj)
            for i in (100, 2000):
                assert data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB] != bytes(B.BYTES_PER_BLOB)
            blobs[100] = blobs[2000] = bytes(B.BYTES_PER_BLOB)
            data = b"".join(blobs)
            cj = b"".join(K.blob_to_kzg_commitment_batch(data, ts))
            pj = b"".join(K.compute_blob_kzg_proof_batch(data, cj, ts))
            cjb = bytearray(cj)
            for i in (100, 2000):
                assert cj[48 * i:48 * i + 48] == bytes([0xc0]) + bytes(47)
                cjb[48 * i:48 * i + 48] = stray
            cj = bytes(cjb)
        # host-pointer shards: the whole batch as one, and cut in two uneven shards (the second one's powers start at r^first)
        recs = []
        shards = []
        for lo, hi in ((0, cut), (cut, n)):
            sh = K.VerifyShard(data[lo * B.BYTES_PER_BLOB:hi * B.BYTES_PER_BLOB], cj[48 * lo:48 * hi], pj[48 * lo:48 * hi], hi - lo, ts)
            shards.append((sh, lo))
            recs.append(sh.records)
        rec_all = b"".join(recs)
        parts = [sh.partial(rec_all, n, lo) for sh, lo in shards]
        ok_sharded = K.verify_shards_finish(b"".join(parts), 2, n, ts)
        # the device-resident form of the same first shard
        to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
        db, dc, dp = to_dev(data[:cut * B.BYTES_PER_BLOB]), to_dev(cj[:48 * cut]), to_dev(pj[:48 * cut])
        torch.cuda.synchronize()
        shd = K.VerifyShard.from_device(db.data_ptr(), dc.data_ptr(), dp.data_ptr(), cut, ts)
        assert shd.records == recs[0]
        part_dev = shd.partial(rec_all, n, 0)
        # verdicts: honest, and with two proofs swapped
        ok = K.verify_blob_kzg_proof_batch(data, cj, pj, n, ts)
        swapped = pj
        if n >= 9:
            swapped = pj[48:96] + pj[:48] + pj[96:]
        ok_sw = K.verify_blob_kzg_proof_batch(data, cj, swapped, n, ts) if n >= 9 else None
        out[str(n)] = {"partials": [p.hex() for p in parts], "partial_device_form": part_dev.hex(), "ok_sharded": ok_sharded, "ok": ok,
                       "ok_swapped": ok_sw}
        for sh, _ in shards:
            sh.free()
        shd.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
