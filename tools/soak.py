#!/usr/bin/env python3
"""Differential soak on the GPU: wide direct table vs a narrow one (the generic-plan kernel) vs the bucket engine vs the tau = 1337 closed form, on inputs the unit tests
do not sweep: full-range random 32-byte elements (values >= r included: reference mode reduces them), sparse blobs,
blobs with few distinct scalars, and proofs. Prints one JSON summary; exit code 1 on any mismatch.

    python tools/soak.py [--batches 24] [--direct-bits 16]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--direct-bits", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1024)
    args = ap.parse_args()
    import numpy as np
    import torch
    import blobs as B
    import lambdaworks_kzg_amd as K
    from lambdaworks_kzg_amd import capi
    from oracle import oracle as O
    R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    TAU = 1337
    setup = os.path.join(ROOT, "tests", "golden", "trusted_setup.txt")
    ts_b = K.TrustedSetup.from_file(setup)
    ts_b.enable_direct_table(0)              # the bucket engine, forced (a plain load selects a direct table by itself)
    ts_m = K.TrustedSetup.from_file(setup)
    ts_m.enable_direct_table(12)             # a narrow table: the kernel that takes its window plan as an argument
    ts_d = K.TrustedSetup.from_file(setup)
    ts_b.reserve(args.batch)
    ts_d.reserve(args.batch)
    ts_d.enable_direct_table(args.direct_bits)
    pw = [pow(TAU, i, R) for i in range(4096)]

    def closed(blob):
        acc = 0
        for i in range(4096):
            acc = (acc + int.from_bytes(blob[32 * i:32 * i + 32], "big") % R * pw[i]) % R
        return O.g1_generator_mul(acc)

    n = args.batch
    rng = np.random.default_rng(20240)
    d_out_b = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    d_out_d = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    d_out_m = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(n, dtype=torch.int32, device="cuda")
    blobs_done = mismatches = closed_checked = proofs_checked = fused_checked = 0
    t0 = time.time()
    for it in range(args.batches):
        kind = it % 4
        arr = rng.integers(0, 256, size=(n, 4096, 32), dtype=np.uint8)            # full range, many >= r
        if kind == 1:                                                             # sparse: ~1 % non-zero elements
            arr[rng.random((n, 4096)) > 0.01] = 0
        elif kind == 2:                                                           # 3 distinct scalars per blob
            pick = rng.integers(0, 3, size=(n, 4096))
            vals = rng.integers(0, 256, size=(n, 3, 32), dtype=np.uint8)
            arr = np.take_along_axis(vals, pick[:, :, None].repeat(32, axis=2), axis=1)
        elif kind == 3:                                                           # small scalars: only the low windows
            arr[:, :, :28] = 0
        data = arr.tobytes()
        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        capi.blob_to_kzg_commitment_batch_device(d_out_b.data_ptr(), d_in.data_ptr(), n, ts_b, None, d_st.data_ptr())
        capi.blob_to_kzg_commitment_batch_device(d_out_d.data_ptr(), d_in.data_ptr(), n, ts_d, None, d_st.data_ptr())
        capi.blob_to_kzg_commitment_batch_device(d_out_m.data_ptr(), d_in.data_ptr(), n, ts_m, None, d_st.data_ptr())
        torch.cuda.synchronize()
        a, b = bytes(d_out_b.cpu().numpy().tobytes()), bytes(d_out_d.cpu().numpy().tobytes())
        mm = bytes(d_out_m.cpu().numpy().tobytes())
        if a != b or mm != b:
            mismatches += sum(a[48 * i:48 * i + 48] != b[48 * i:48 * i + 48] or mm[48 * i:48 * i + 48] != b[48 * i:48 * i + 48] for i in range(n))
        for i in (0, n - 1):
            if b[48 * i:48 * i + 48] != closed(data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB]):
                mismatches += 1
            closed_checked += 1
        # proofs on a slice (both paths; the quotient's scalars are full range by construction)
        m = 64
        d_pb = torch.empty(48 * m, dtype=torch.uint8, device="cuda")
        d_pd = torch.empty(48 * m, dtype=torch.uint8, device="cuda")
        capi.compute_blob_kzg_proof_batch_device(d_pb.data_ptr(), d_in.data_ptr(), d_out_b.data_ptr(), m, ts_b, None, d_st.data_ptr())
        capi.compute_blob_kzg_proof_batch_device(d_pd.data_ptr(), d_in.data_ptr(), d_out_d.data_ptr(), m, ts_d, None, d_st.data_ptr())
        torch.cuda.synchronize()
        pa, pb = bytes(d_pb.cpu().numpy().tobytes()), bytes(d_pd.cpu().numpy().tobytes())
        mismatches += sum(pa[48 * i:48 * i + 48] != pb[48 * i:48 * i + 48] for i in range(m))
        # commitment and proof in one pass (the hash's first 2048 blocks beside the commitment MSM) against the two calls
        d_fc = torch.empty(48 * m, dtype=torch.uint8, device="cuda")
        d_fp = torch.empty(48 * m, dtype=torch.uint8, device="cuda")
        capi.commit_and_prove_batch_device(d_fc.data_ptr(), d_fp.data_ptr(), d_in.data_ptr(), m, ts_d, None, d_st.data_ptr())
        torch.cuda.synchronize()
        fused_checked += m
        if bytes(d_fc.cpu().numpy().tobytes()) != b[:48 * m] or bytes(d_fp.cpu().numpy().tobytes()) != pb:
            mismatches += 1
        ok = K.verify_blob_kzg_proof_batch(data[:8 * B.BYTES_PER_BLOB], b[:48 * 8], pb[:48 * 8], 8, ts_d)
        mismatches += 0 if ok else 1
        proofs_checked += m
        blobs_done += n
    print(json.dumps({"blobs": blobs_done, "batches": args.batches, "direct_bits": args.direct_bits,
                      "closed_form_checked": closed_checked, "proofs_compared": proofs_checked, "one_pass_pairs_compared": fused_checked,
                      "mismatches": mismatches,
                      "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if mismatches else 0)


if __name__ == "__main__":
    main()
