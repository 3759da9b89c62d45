"""Condense rocprofv3 CSV output (kernel stats + counter passes) into a short per-kernel table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


import json
import re

ROLE = {r"linear_hl_kernel<\d+, 0, false": "node_proj", r"linear_hl_kernel<\d+, 1, false": "node_mlp0",
        r"linear_hl_kernel<\d+, 0, true": "node_mlp1", r"edge_kernel": "edge_fused", r"edge_pw_kernel": "edge_fused", r"knn_select_kernel": "knn_select",
        r"node_prep_hl_kernel": "node_prep", r"split_f16_kernel": "split_f16", r"spatial_order_kernel": "spatial_order",
        r"slot_prep_kernel": "slot_prep", r"node_mlp_fused_kernel": "node_mlp"}


def short(name):
    # rocprofv3 leaves kernels with _Float16 parameters mangled: rebuild "kernel<args>" from the Itanium name
    mm = re.search(r"_GLOBAL__N_1\d+([a-z][a-z_0-9]*?_kernel)I((?:L[ib]\d+E)+)E", name)
    if mm:
        args = re.findall(r"L([ib])(\d+)E", mm.group(2))
        name = mm.group(1) + "<" + ", ".join(("true" if v == "1" else "false") if t == "b" else v for t, v in args) + ">"
    else:
        mm = re.search(r"_GLOBAL__N_1\d+([a-z][a-z_0-9]*?_kernel)E", name)
        if mm:
            name = mm.group(1)
    m = re.search(r"(\w+_kernel)<([^>]*)>", name) or re.search(r"(\w+_kernel)", name)
    if not m:
        return name[:60]
    full = m.group(0)
    for key, role in ROLE.items():
        if re.match(key, full):
            return f"{role} [{full}]"
    return full


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

# ---- effective shader clock per kernel (MI355X_MICROARCH.md, "DVFS give-back": GRBM_GUI_ACTIVE / kernel wall time).  rocprofv3 sums the
# counter over the eight XCDs, and its sampling window is wider than the dispatch: the smallest value any dispatch of the pass reports
# (a few-microsecond copy kernel) is taken as that fixed part.  Wall time = the dispatch's own timestamps in the same pass.
XCDS = 8
clocks = {}
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = []
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    rows.append((short(row["Kernel_Name"]), float(row["Counter_Value"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
        if not rows:
            continue
        floor = min(v for _, v, _ in rows)
        print(f"\n== effective clock (GRBM_GUI_ACTIVE pass; counter summed over {XCDS} XCDs, fixed part {floor:.0f} subtracted)")
        agg = defaultdict(list)
        for k, v, ns in rows:
            if ns > 20000:
                agg[k].append(((v - floor) / XCDS / ns, ns))
        for k, v in sorted(agg.items(), key=lambda kv: -sum(n for _, n in kv[1])):
            ghz = sum(g for g, _ in v) / len(v)
            clocks[k.split(" [")[0]] = round(ghz, 3)
            print(f"clock  {k:44s} n={len(v):4d} wall_ns={sum(n for _, n in v)/len(v):10.0f} effective_GHz={ghz:.3f}  (peak figures assume 2.400: x{ghz/2.4:.3f})")

# ---- HBM traffic per launch for bench.py's roofline.traffic (MI355X_MICROARCH.md, HBM section):
# FETCH_SIZE / WRITE_SIZE are in KiB and come from separate passes; on gfx950 FETCH_SIZE reports one half of the
# bytes of a wide (16 B/lane) coalesced read, so it is doubled; WRITE_SIZE is taken as is (uncalibrated).
traffic = {}
vals = defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    agg[short(row["Kernel_Name"]).split(" [")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                vals[k][c] = sum(v) / len(v)
for k, cs in vals.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs and k in ROLE.values():
        traffic[k] = int((2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024)
        print(f"traffic {k:12s} fetch_KiB={cs['FETCH_SIZE']:.0f} (x2 gfx950 correction) write_KiB={cs['WRITE_SIZE']:.0f} "
              f"-> {traffic[k]/1e9:.3f} GB per launch")
# one file for all workloads: {workload: {kernel: bytes per launch, "_head": commit the run was taken at}} (bench.py reads it)
workload = sys.argv[2] if len(sys.argv) > 2 else "north_star"
try:
    import subprocess
    traffic["_head"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip() or None
except Exception:
    traffic["_head"] = None
if len(sys.argv) > 3 and sys.argv[3]:
    traffic["_head"] = sys.argv[3]
if clocks:
    traffic["_clock_ghz"] = {k: v for k, v in clocks.items() if k in ROLE.values()}
with open(os.path.join(out, "pmc_traffic.json"), "w") as fh:
    json.dump({workload: traffic}, fh, indent=1)
