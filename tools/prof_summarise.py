#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel-trace stats + PMC counter_collection CSVs)
into a small per-kernel text summary that is committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 90 else name[:87] + "..."


for f in sorted(glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)):
    print(f"== kernel stats: {os.path.relpath(f, out)}")
    with open(f) as fh:
        for i, row in enumerate(csv.DictReader(fh)):
            if i >= 12:
                break
            print("  %-90s calls=%s total_ns=%s avg_ns=%s pct=%s min_ns=%s max_ns=%s" % (
                short(row.get("Name", "?")), row.get("Calls"), row.get("TotalDurationNs"),
                row.get("AverageNs"), row.get("Percentage"), row.get("MinNs"), row.get("MaxNs")))

for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    print(f"== PMC: {os.path.relpath(f, out)}")
    agg = defaultdict(lambda: defaultdict(list))
    meta = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row.get("Kernel_Name", "?"))
            agg[k][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
            meta[k] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"),
                       row.get("LDS_Block_Size"), row.get("Grid_Size"), row.get("Workgroup_Size"))
    for k, ctrs in agg.items():
        print("  %s  vgpr=%s agpr=%s sgpr=%s lds=%s grid=%s wg=%s" % ((k,) + meta[k]))
        for c, vals in sorted(ctrs.items()):
            print("      %-24s n=%d mean=%.6g min=%.6g max=%.6g" % (c, len(vals), sum(vals) / len(vals),
                                                                  min(vals), max(vals)))
