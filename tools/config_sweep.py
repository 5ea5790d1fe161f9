#!/usr/bin/env python3
"""BASELINE.json configs[1..4] on one MI355X, as SURVEY section 8d asks: >= 3 warm-ups, median of >= 10 runs.

  config 2  commitments, batch B = 1, 2, 4, ..., 1024: device-resident (HIP-side wall clock around the C ABI call +
            synchronize) AND the host-pointer ABI (H2D of blobs / D2H of results included)
  config 3  compute_blob_kzg_proof, batch 256 (+ 1024)
  config 4  verify_blob_kzg_proof_batch (what ONE GPU's shard of the 4096-blob job costs: n = 512, and n = 4096 alone)
  config 5  2^20-term tiled MSM

    python tools/config_sweep.py [--direct-bits 16|15|14|0] > profiles/rNN_config_sweep.json
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def median_ms(fn, sync, warm=3, runs=11):
    for _ in range(warm):
        fn()
    sync()
    ts = []
    for _ in range(runs):
        t = time.perf_counter()
        fn()
        sync()
        ts.append((time.perf_counter() - t) * 1e3)
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--direct-bits", default="16", help="10..16 = that direct table, 0 = bucket engine, default = what the load selected")
    ap.add_argument("--max-verify", type=int, default=4096)
    args = ap.parse_args()
    import torch
    import blobs as B
    import lambdaworks_kzg_amd as K
    from lambdaworks_kzg_amd import capi

    ts = K.TrustedSetup.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"))
    ts.reserve(1024)
    if args.direct_bits != "default":
        ts.enable_direct_table(int(args.direct_bits))
    args.direct_bits = ts.direct_table_bits()
    sync = torch.cuda.synchronize
    out = {"direct_bits": args.direct_bits, "device": torch.cuda.get_device_name(0), "commit": [], "blob_proof": [],
           "verify_batch": [], "tiled_msm": None}

    nmax = 1024
    data = B.synthetic_batch(0, nmax)
    d_blobs = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    d_out = torch.empty(48 * nmax, dtype=torch.uint8, device="cuda")
    d_comm = torch.empty(48 * nmax, dtype=torch.uint8, device="cuda")
    n = 1
    while n <= nmax:
        dev = median_ms(lambda: capi.blob_to_kzg_commitment_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), n, ts), sync)
        sub = data[:n * B.BYTES_PER_BLOB]
        host = median_ms(lambda: K.blob_to_kzg_commitment_batch(sub, ts), lambda: None, warm=2, runs=7)
        out["commit"].append({"batch": n, "device_resident_ms": dev, "device_resident_ops_per_s": n / dev * 1e3,
                              "host_abi_ms": host, "host_abi_ops_per_s": n / host * 1e3})
        n *= 2
    capi.blob_to_kzg_commitment_batch_device(d_comm.data_ptr(), d_blobs.data_ptr(), nmax, ts)
    sync()
    comms = bytes(d_comm.cpu().numpy().tobytes())
    for n in (1, 16, 64, 128, 256, 1024):
        dev = median_ms(lambda: capi.compute_blob_kzg_proof_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), d_comm.data_ptr(), n, ts), sync)
        sub = data[:n * B.BYTES_PER_BLOB]
        host = median_ms(lambda: K.compute_blob_kzg_proof_batch(sub, comms[:48 * n], ts), lambda: None, warm=2, runs=7)
        out["blob_proof"].append({"batch": n, "device_resident_ms": dev, "device_resident_ops_per_s": n / dev * 1e3,
                                  "host_abi_ms": host, "host_abi_ops_per_s": n / host * 1e3})
    proofs = b"".join(K.compute_blob_kzg_proof_batch(data, comms, ts))
    for n in (1, 64, 512, 1024, 4096):
        if n > args.max_verify:
            continue
        reps = (n + nmax - 1) // nmax
        bl = (data * reps)[:n * B.BYTES_PER_BLOB]
        cm = (comms * reps)[:48 * n]
        pr = (proofs * reps)[:48 * n]
        assert K.verify_blob_kzg_proof_batch(bl, cm, pr, n, ts) is True
        host = median_ms(lambda: K.verify_blob_kzg_proof_batch(bl, cm, pr, n, ts), lambda: None, warm=1, runs=5)
        out["verify_batch"].append({"batch": n, "host_abi_ms": host, "host_abi_blobs_per_s": n / host * 1e3})
    tiles = 256
    d_sc = torch.frombuffer(bytearray(B.synthetic_batch(5000, tiles)), dtype=torch.uint8).cuda()
    d_one = torch.empty(48, dtype=torch.uint8, device="cuda")
    ms = median_ms(lambda: capi.g1_msm_tiled_device(d_one.data_ptr(), d_sc.data_ptr(), tiles * 4096, ts), sync)
    out["tiled_msm"] = {"terms": tiles * 4096, "ms": ms, "terms_per_s": tiles * 4096 / ms * 1e3,
                        "algorithmic_GBps": (tiles * 4096 * 128 + 48) / ms / 1e6,
                        "hbm_frac": (tiles * 4096 * 128 + 48) / ms / 1e6 / 8000.0}
    print(json.dumps(out, indent=1))
    ts.free()


if __name__ == "__main__":
    main()
