#!/bin/bash
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the north-star edge pass: the shipped kernel against a build whose first-layer A-fragment
# reads avoid the 2-way conflict (build_variants/edge_pw_PW_WSTSWZ1: reads the wrong words -- counters only)
export TMPDIR=/tmp
REPO="$(pwd)"
cat > /tmp/pmc_run.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
torch.set_grad_enabled(False)
sys.argv = ["pw_probe", "shapes=ns", "reps=2"]
exec(open("$REPO/tools/pw_probe.py").read())
PY
cd /tmp
for tag in edge_pw_default edge_pw_PW_WSTSWZ1; do
  OUT="$REPO/gpurun_out/r05_lds/$tag"; mkdir -p "$OUT"
  EGNN_HIP_LIB="$REPO/build_variants/$tag/libegnn_hip.so" EGNN_RANGE_CHECK=off rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d "$OUT/pmc1" -o pmc --output-format csv -- python /tmp/pmc_run.py > "$OUT/pmc1.log" 2>&1
  echo "$tag rc=$?"
  python - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "edge_pw_kernel" in r["Kernel_Name"]:
            rows[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for c, d in rows.items():
    vals = [sum(v) for v in d.values()]
    print("$tag", c, "dispatches", len(vals), "avg per dispatch", sum(vals) / max(1, len(vals)))
PY
done
