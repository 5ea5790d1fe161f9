#!/usr/bin/env python3
"""Differential soak of the c-kzg-mode front end (blob bytes -> inverse transform -> MSM) on the GPU: batches of random
canonical little-endian evaluation-form blobs -- elements up to r - 1, sparse, few distinct values -- committed on the
default engine and on the bucket engine, a sample of every batch checked against the tau = 1337 closed form evaluated
straight from the EVALUATIONS by the barycentric formula (no transform on the checking side):

    p(tau) = (tau^n - 1) / n * sum_i f_i w_i / (tau - w_i),   w_i = the i-th root of unity in bit-reversed order.

Prints one JSON summary; exit code 1 on any mismatch.   python tools/soak_ckzg.py [--batches 20] [--sample 16]
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

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
TAU, N = 1337, 4096


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--sample", type=int, default=16)
    args = ap.parse_args()
    import numpy as np
    import torch
    import lambdaworks_kzg_amd as K
    from lambdaworks_kzg_amd import capi
    from oracle import oracle as O
    setup = os.path.join(ROOT, "tests", "golden", "trusted_setup.txt")
    ts_d = K.TrustedSetup.from_file(setup)            # the default engine
    ts_b = K.TrustedSetup.from_file(setup)
    ts_b.enable_direct_table(0)                       # buckets
    K.set_mode(K.MODE_CKZG)
    w = pow(7, (R - 1) // N, R)
    brp = [int(format(i, "012b")[::-1], 2) for i in range(N)]
    roots = [pow(w, brp[i], R) for i in range(N)]
    # weights w_i / (tau - w_i) * (tau^n - 1) / n, once
    pref = (pow(TAU, N, R) - 1) * pow(N, -1, R) % R
    weight = [roots[i] * pow(TAU - roots[i], -1, R) % R * pref % R for i in range(N)]

    def closed(blob):
        acc = 0
        for i in range(N):
            acc += int.from_bytes(blob[32 * i:32 * i + 32], "little") * weight[i]
        return O.g1_generator_mul(acc % R)

    n = args.batch
    rng = np.random.default_rng(424242)
    d_out_d = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    d_out_b = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(n, dtype=torch.int32, device="cuda")
    near_r = np.stack([np.frombuffer((R - k).to_bytes(32, "little"), dtype=np.uint8) for k in (1, 2, 3, 4)])
    blobs_done = mismatches = closed_checked = 0
    t0 = time.time()
    for it in range(args.batches):
        kind = it % 4
        arr = rng.integers(0, 256, size=(n, N, 32), dtype=np.uint8)
        arr[:, :, 31] &= 0x3F                                                  # < 2^254 < r: canonical
        if kind == 1:                                                          # every tenth element r - 1 .. r - 4
            sel = rng.random((n, N)) < 0.1
            arr[sel] = near_r[rng.integers(0, 4, size=int(sel.sum()))]
        elif kind == 2:                                                        # sparse
            arr[rng.random((n, N)) > 0.01] = 0
        elif kind == 3:                                                        # 3 distinct values per blob
            pick = rng.integers(0, 3, size=(n, N))
            vals = rng.integers(0, 256, size=(n, 3, 32), dtype=np.uint8)
            vals[:, :, 31] &= 0x3F
            arr = np.take_along_axis(vals, pick[:, :, None].repeat(32, axis=2), axis=1)
        data = arr.tobytes()
        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        capi.blob_to_kzg_commitment_batch_device(d_out_d.data_ptr(), d_in.data_ptr(), n, ts_d, None, d_st.data_ptr())
        capi.blob_to_kzg_commitment_batch_device(d_out_b.data_ptr(), d_in.data_ptr(), n, ts_b, None, d_st.data_ptr())
        torch.cuda.synchronize()
        if int(d_st.abs().sum().item()) != 0:
            mismatches += 1                                                    # a canonical blob was rejected
        a, b = bytes(d_out_d.cpu().numpy().tobytes()), bytes(d_out_b.cpu().numpy().tobytes())
        mismatches += sum(a[48 * i:48 * i + 48] != b[48 * i:48 * i + 48] for i in range(n))
        for i in [int(x) for x in rng.choice(n, size=args.sample, replace=False)]:
            if a[48 * i:48 * i + 48] != closed(data[i * 32 * N:(i + 1) * 32 * N]):
                mismatches += 1
            closed_checked += 1
        blobs_done += n
    K.set_mode(K.MODE_REFERENCE)
    print(json.dumps({"soak": "ckzg front end", "blobs": blobs_done, "batches": args.batches, "closed_form_checked": closed_checked,
                      "mismatches": mismatches, "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if mismatches else 0)


if __name__ == "__main__":
    main()
