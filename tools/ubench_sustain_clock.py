#!/usr/bin/env python3
"""Effective clock and cycles per instruction of tools/ubench_sustain.hip --quick from a `rocprofv3 --pmc GRBM_GUI_ACTIVE` pass: every dispatch
of that mode issues 1024 x 4096 x 2 = 8,388,608 wave-instructions per SIMD, so cycles per instruction = GRBM_GUI_ACTIVE / 8 XCDs / 8,388,608 and the
clock = the same cycles / the dispatch's duration. Usage: ubench_sustain_clock.py <counter_collection.csv>"""
import csv
import sys
from collections import OrderedDict

INSTR = 1024 * 4096 * 2
groups = OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "k_touch" in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].split("(")[0]
    key = (name, int(r["Grid_Size"]))
    # consecutive dispatches of one kernel and grid form a stream; a kernel that comes back later (k_varied: 28-bit, 32-bit, 8-bit) starts a new one
    if not groups or list(groups.keys())[-1][:2] != key:
        groups[key + (len(groups),)] = []
    groups[list(groups.keys())[-1]].append((float(r["Counter_Value"]) / 8, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9))
print("%-22s %9s %9s %12s %10s %10s" % ("kernel", "waves/SIMD", "launches", "ms per launch", "clock GHz", "cycles/instr"))
for (name, grid, _), v in groups.items():
    v = v[1:] if len(v) > 2 else v          # the first launch of a stream follows another stream's clock
    cyc = sum(c for c, _ in v) / len(v)
    dur = sum(d for _, d in v) / len(v)
    print("%-22s %9.0f %9d %12.3f %10.3f %10.3f" % (name, grid / 256 / 256, len(v), dur * 1e3, cyc / dur / 1e9, cyc / INSTR))
