#!/usr/bin/env python3
"""Differential soak of the paths a handful of blobs take (round 5): the cooperative kernel (k_coop_msm_asm) on the default table and on a second
width against the large-batch kernels, the bucket engine and the tau = 1337 closed form; inversion + compression on the host against the device
kernel; one-blob proofs against the batch path. Inputs the unit tests do not sweep: full-range random elements (values >= r included), sparse
blobs, blobs with few distinct scalars, small scalars (whole window groups with nothing to add), batches of 1 .. 8. Prints one JSON summary; exit
code 1 on any mismatch.

    python tools/soak_small.py [--rounds 400] [--direct-bits 16]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=400)
    ap.add_argument("--direct-bits", type=int, default=16)
    args = ap.parse_args()
    import numpy as np
    import torch
    import lambdaworks_kzg_amd as K
    from lambdaworks_kzg_amd import capi
    from oracle import oracle as O
    R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    TAU = 1337
    setup = os.path.join(ROOT, "tests", "golden", "trusted_setup.txt")
    ts_default = K.TrustedSetup.from_file(setup)            # what a plain load selects (13 bits on an empty device)
    ts_wide = K.TrustedSetup.from_file(setup)
    ts_wide.enable_direct_table(args.direct_bits)
    ts_bucket = K.TrustedSetup.from_file(setup)
    ts_bucket.enable_direct_table(0)
    pw = [pow(TAU, i, R) for i in range(4096)]

    def closed(blob):
        acc = 0
        for i in range(4096):
            acc = (acc + int.from_bytes(blob[32 * i:32 * i + 32], "big") % R * pw[i]) % R
        return O.g1_generator_mul(acc)

    rng = np.random.default_rng(50505)
    stats = {"calls": 0, "blobs": 0, "mismatches": 0, "closed_form_checked": 0, "proofs_checked": 0, "device_vs_host_checked": 0}
    t0 = time.time()
    for it in range(args.rounds):
        n = int(rng.integers(1, 9))
        kind = it % 5
        if kind == 0:       # full range
            data = rng.integers(0, 256, size=(n, 4096, 32), dtype=np.uint8)
        elif kind == 1:     # sparse: most elements zero
            data = rng.integers(0, 256, size=(n, 4096, 32), dtype=np.uint8)
            data[rng.random((n, 4096)) < 0.97] = 0
        elif kind == 2:     # few distinct values
            vals = rng.integers(0, 256, size=(n, 5, 32), dtype=np.uint8)
            idx = rng.integers(0, 5, size=(n, 4096))
            data = np.stack([vals[b][idx[b]] for b in range(n)])
        elif kind == 3:     # small scalars: only the lowest windows carry digits
            data = np.zeros((n, 4096, 32), dtype=np.uint8)
            data[:, :, 29:] = rng.integers(0, 256, size=(n, 4096, 3), dtype=np.uint8)
        else:               # canonical 248-bit scalars (the bench's construction)
            data = rng.integers(0, 256, size=(n, 4096, 32), dtype=np.uint8)
            data[:, :, 0] = 0
        blob_bytes = data.tobytes()
        got = K.blob_to_kzg_commitment_batch(blob_bytes, ts_default)
        wide = K.blob_to_kzg_commitment_batch(blob_bytes, ts_wide)
        buck = K.blob_to_kzg_commitment_batch(blob_bytes, ts_bucket)
        stats["calls"] += 3
        stats["blobs"] += n
        if got != wide or got != buck:
            stats["mismatches"] += 1
            print("MISMATCH between engines at round %d (n=%d kind=%d)" % (it, n, kind), file=sys.stderr)
        # device entry point (k_finalize_compress on the GPU) against the host-pointer call (inversion on the host)
        d_b = torch.frombuffer(bytearray(blob_bytes), dtype=torch.uint8).cuda()
        d_o = torch.zeros(48 * n, dtype=torch.uint8, device="cuda")
        capi.blob_to_kzg_commitment_batch_device(d_o.data_ptr(), d_b.data_ptr(), n, ts_default)
        torch.cuda.synchronize()
        if bytes(d_o.cpu().numpy()) != b"".join(got):
            stats["mismatches"] += 1
            print("MISMATCH device vs host at round %d" % it, file=sys.stderr)
        stats["device_vs_host_checked"] += n
        if it % 8 == 0:     # the closed form on the first blob
            if got[0] != closed(blob_bytes[:131072]):
                stats["mismatches"] += 1
                print("MISMATCH vs closed form at round %d" % it, file=sys.stderr)
            stats["closed_form_checked"] += 1
        if it % 4 == 1:     # one-blob proofs (cooperative kernel + host finishing) against the batch path on the bucket engine
            cj = b"".join(got)
            want = K.compute_blob_kzg_proof_batch(blob_bytes, cj, ts_bucket)
            for b in range(min(n, 2)):
                one = K.compute_blob_kzg_proof(blob_bytes[b * 131072:(b + 1) * 131072], got[b], ts_default)
                if one != want[b]:
                    stats["mismatches"] += 1
                    print("MISMATCH proof at round %d blob %d" % (it, b), file=sys.stderr)
                stats["proofs_checked"] += 1
    stats["seconds"] = round(time.time() - t0, 1)
    stats["direct_bits"] = [ts_default.direct_table_bits(), ts_wide.direct_table_bits(), 0]
    print(json.dumps(stats))
    return 1 if stats["mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
