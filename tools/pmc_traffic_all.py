#!/usr/bin/env python3
"""Per-kernel HBM-side traffic of the all-legs bench run from the two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE):
profiles/<tag>_pmc_traffic_all_legs.json. Same units and gfx950 correction as tools/pmc_summary.py (FETCH_SIZE counts 64-byte
requests in KiB; a 128-byte request reads half: doubled). bench.py reads k_ntt4096's figure for its ntt_roofline.traffic.

    python tools/pmc_traffic_all.py fetch_counter_collection.csv write_counter_collection.csv rNN
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
        m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
        if m:
            agg[m.group(0)].append(float(r["Counter_Value"]) * 1024.0)
    return {k: (sum(v) / len(v), len(v), max(v)) for k, v in agg.items()}


def main():
    f, w, tag = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), sys.argv[3]
    out = {"round": tag,
           "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (every leg)",
           "unit": "bytes per launch: average over every launch of the kernel in the run, and the largest launch (launch sizes differ between legs)",
           "kernels": {}}
    for k in sorted(set(f) | set(w)):
        fr, n, fmax = f.get(k, (0.0, 0, 0.0))
        wr, _, wmax = w.get(k, (0.0, 0, 0.0))
        out["kernels"][k] = {"launches": n, "fetch_raw": fr, "fetch_doubled": 2 * fr, "write": wr, "traffic_bytes": 2 * fr + wr,
                             "largest_launch_traffic_bytes": 2 * fmax + wmax}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "%s_pmc_traffic_all_legs.json" % tag)
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
