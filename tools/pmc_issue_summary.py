#!/usr/bin/env python3
"""Issue-side picture of one kernel from separate rocprofv3 --pmc passes (SQ_* and GRBM_* counters, collected by
tools/collect_profiles.sh): instructions per launch, the clock the chip held, VALUBusy, cycles per VALU instruction per
SIMD, and how a wave's cycles split between issuing, waiting for an issue slot and waiting for memory.

    python tools/pmc_issue_summary.py KERNEL OUT.json pass1_counter_collection.csv [pass2.csv ...]

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
GRBM_GUI_ACTIVE is summed over the 8 XCDs; VALUBusy is the gfx9 derived metric 4 * SQ_ACTIVE_INST_VALU / SIMDs / cycles.
"""
import collections
import csv
import json
import re
import sys

SIMDS, XCDS = 1024, 8


def main():
    kernel, out = sys.argv[1], sys.argv[2]
    vals = collections.defaultdict(list)
    durs = []
    for path in sys.argv[3:]:
        for r in csv.DictReader(open(path)):
            m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
            if not m or m.group(0) != kernel:
                continue
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    c = {k: sum(v) / len(v) for k, v in vals.items()}
    res = {"kernel": kernel, "launches_seen": {k: len(v) for k, v in vals.items()}, "counters_per_launch": c,
           "avg_launch_ms_under_the_profiler": sum(durs) / max(1, len(durs))}
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / XCDS
        res["cycles_per_launch"] = cyc
        res["clock_GHz"] = cyc / (res["avg_launch_ms_under_the_profiler"] * 1e-3) / 1e9
        if "SQ_ACTIVE_INST_VALU" in c:
            res["VALUBusy"] = 4 * c["SQ_ACTIVE_INST_VALU"] / SIMDS / cyc
        if "SQ_INSTS_VALU" in c:
            res["cycles_per_VALU_instruction_per_SIMD"] = cyc * SIMDS / c["SQ_INSTS_VALU"]
        if "SQ_BUSY_CYCLES" in c:
            res["SQ_busy_fraction_32_shader_engines"] = c["SQ_BUSY_CYCLES"] / 32 / cyc
    if "SQ_WAVE_CYCLES" in c:
        w = c["SQ_WAVE_CYCLES"]
        res["wave_cycles_split"] = {k: c[n] / w for k, n in (("issuing", "SQ_ACTIVE_INST_ANY"), ("waiting_for_an_issue_slot", "SQ_WAIT_INST_ANY"),
                                                               ("waiting_at_s_waitcnt_or_barrier", "SQ_WAIT_ANY")) if n in c}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
