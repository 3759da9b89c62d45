#!/bin/bash
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS of the edge kernels on the north-star shape, per dispatch, for one or more
# libraries:   bash tools/r5_lds_conflicts.sh [path/to/libegnn_hip.so ...]      (default: the shipped library)
# How the 12 % bank-conflict cycles of the edge pass were located in round 5 (a build_variants/ library whose A-fragment read used the
# other bank pair: 16.9 M -> 0) and how the Wst pair swap was confirmed afterwards (profiles/r05_experiments/).
export TMPDIR=/tmp
REPO="$(pwd)"
LIBS=("$@"); [ ${#LIBS[@]} -eq 0 ] && LIBS=("$REPO/egnn_pytorch_amd/libegnn_hip.so")
cat > /tmp/pmc_run.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
torch.set_grad_enabled(False)
sys.argv = ["pw_probe", "shapes=ns", "reps=2"]
exec(open("$REPO/tools/pw_probe.py").read())
PY
cd /tmp
n=0
for lib in "${LIBS[@]}"; do
  n=$((n+1)); OUT="$REPO/gpurun_out/r05_lds/lib$n"; mkdir -p "$OUT"
  EGNN_HIP_LIB="$lib" EGNN_RANGE_CHECK=off rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d "$OUT/pmc1" -o pmc --output-format csv -- python /tmp/pmc_run.py > "$OUT/pmc1.log" 2>&1
  echo "$lib rc=$?"
  python - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "edge_pw" if "edge_pw_kernel" in r["Kernel_Name"] else ("edge_general" if "edge_kernel" in r["Kernel_Name"] else None)
        if k:
            rows[(k, r["Counter_Name"])][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for c, d in sorted(rows.items()):
    vals = [sum(v) for v in d.values()]
    print("   ", c, "dispatches", len(vals), "avg per dispatch", sum(vals) / max(1, len(vals)))
PY
done
