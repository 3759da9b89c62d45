"""Condense rocprofv3 CSV output (kernel stats + counter passes) into a short per-kernel table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    for key in ("knn_select", "linear_kernel", "edge_kernel", "node_prep", "adj_max"):
        if key in name:
            if key == "linear_kernel":
                return "linear_kernel<" + name.split("linear_kernelILi")[1][:12] + ">" if "ILi" in name else key
            return key
    return name[:60]


print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print(f"{short(row['Name']):44s} calls={row['Calls']:>5s} avg_ns={float(row['AverageNs']):12.0f} "
                  f"total_ns={float(row['TotalDurationNs']):14.0f} pct={row['Percentage']}")

print("\n== per-dispatch durations from the kernel trace (ns): avg over dispatches")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    agg = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            agg[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:44s} n={len(v):4d} avg={sum(v)/len(v):12.0f} min={min(v):10d} max={max(v):10d}")

print("\n== counters (avg per dispatch)")
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in agg.items():
            for cname, vals in cs.items():
                print(f"{os.path.basename(d):6s} {k:44s} {cname:28s} n={len(vals):4d} avg={sum(vals)/len(vals):16.1f}")
