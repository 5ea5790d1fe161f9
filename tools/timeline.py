#!/usr/bin/env python3
"""Text timeline of our kernels from a rocprofv3 --kernel-trace CSV: start, end, duration (ms) and queue of the last N
dispatches, to show which kernels ran beside which.

    python tools/timeline.py kt_kernel_trace.csv [N] > profiles/rNN_..._timeline.txt
"""
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    ev = []
    for r in rows:
        m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
        if m:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0), r["Queue_Id"]))
    ev.sort()
    ev = ev[-n:]
    t0 = ev[0][0]
    print("# start_ms   end_ms   dur_ms  queue  kernel   (last %d dispatches of %s)" % (len(ev), sys.argv[1]))
    for s, e, name, q in ev:
        print("%9.3f %9.3f %8.3f  q%-4s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, name))


if __name__ == "__main__":
    main()
