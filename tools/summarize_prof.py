#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by tools/profile.sh into a short text
summary (per-kernel time statistics + per-kernel mean counter values)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def short(name):
    name = name.split("(")[0]
    return name[:70]


print("== kernel-trace stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("kt/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("%-72s calls=%-5s avg_ns=%-12s min_ns=%-10s max_ns=%-10s pct=%s" % (
                short(row.get("Name", "")), row.get("Calls"), row.get("AverageNs"), row.get("MinNs"),
                row.get("MaxNs"), row.get("Percentage")))

# Steady state: the clocks need tens of milliseconds of load to settle and bench.py keeps warming up (untimed) until
# they have, so the AVERAGE over all launches of a run still contains the ramp and exceeds the driver's ms_per_step.
# The last 20 launches of each kernel are the timed steps of the profiled command: report their mean beside it.
print("\n== kernel-trace, steady state: the LAST 20 launches of each kernel (= the timed steps) ==")
for f in find("kt/**/*kernel_trace.csv"):
    durs = defaultdict(list)
    with open(f) as fh:
        rows = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
    for row in rows:
        durs[short(row.get("Kernel_Name", ""))].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, d in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        if "at::" in k or "elementwise" in k or "Cijk" in k or len(d) < 4:
            continue
        last = d[-20:]
        print("%-72s launches=%-5d avg_all_ns=%-10.0f steady_avg_ns=%-10.0f steady_min_ns=%-10d steady_max_ns=%d" % (
            k, len(d), sum(d) / len(d), sum(last) / len(last), min(last), max(last)))

for sub in ("pmc_sq", "pmc_sq2", "pmc_rd", "pmc_wr"):
    files = find(sub + "/**/*counter_collection.csv")
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Kernel_Name", ""))
                acc[k][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
    print("\n== %s: mean counter value per dispatch ==" % sub)
    for k, d in acc.items():
        if "at::" in k or "elementwise" in k or "Cijk" in k:
            continue
        print(k)
        for c, v in sorted(d.items()):
            print("    %-28s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
