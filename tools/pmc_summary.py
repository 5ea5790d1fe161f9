#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs as
MI355X_MICROARCH.md prescribes) into profiles/pmc_traffic.json, which bench.py reads to fill
roofline.traffic.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -o write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_summary.py out/fetch_counter_collection.csv out/write_counter_collection.csv rNN [blobs_per_launch [direct_bits]]

Units: the counters are in KiB. gfx950 correction (guide, section HBM): FETCH_SIZE is the L2's memory-side read REQUEST
count times 64 bytes, so 128-byte requests read exactly half: double it. Calibrated on this kernel's own pattern in round
2: with every table row in a 128-byte line of its own a launch shows 1.013 requests per gathered row (the 0.013 are the
scalars), i.e. doubled = one line per row; with packed 112-byte rows the same counter showed 1.75 requests per row (a row
straddles a line boundary seven times out of eight) -- whose raw figure happens to equal the payload, 1.75 x 64 = 112,
which round 1 mistook for a calibration. `traffic_bytes` = 2 x FETCH_SIZE + WRITE_SIZE. Infinity-Cache hits appear to be
counted: this is traffic at the L2's memory side, not necessarily HBM.
"""
import collections
import csv
import json
import os
import re
import sys


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])   # "void lwk::k_direct_accumulate<16>(...)" -> k_direct_accumulate
        name = m.group(0) if m else r["Kernel_Name"]
        agg[name].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch, write, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    blobs_per_launch = int(sys.argv[4]) if len(sys.argv) > 4 else 1024  # bench default: 1024 blobs per step, one launch (direct path)
    direct_bits = int(sys.argv[5]) if len(sys.argv) > 5 else 16         # 0 = the passes ran with --direct-bits 0 (bucket path)
    f, w = per_kernel(fetch), per_kernel(write)
    out = {"round": tag, "batch_blobs_per_launch": blobs_per_launch, "direct_bits": direct_bits, "unit": "bytes per launch (average)", "kernels": {}}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        fr, wr = f.get(k, 0.0), w.get(k, 0.0)
        out["kernels"][k] = {"fetch_raw": fr, "fetch_requests": fr / 64.0, "fetch_doubled": 2 * fr, "write": wr, "traffic_bytes": 2 * fr + wr}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
    for k, v in out["kernels"].items():
        print("  %-26s read requests %.3e (x 128 B = %.3e)  write %.3e" % (k, v["fetch_requests"], v["fetch_doubled"], v["write"]))


if __name__ == "__main__":
    main()
