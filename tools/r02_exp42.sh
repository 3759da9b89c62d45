#!/bin/bash
# PMC counters of the backward kernels (final versions) on one training step at the north-star shape
mkdir -p gpurun_out/r02_exp42
OUT=$(pwd)/gpurun_out/r02_exp42
REPO=$(pwd)
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "VALUBusy" "MfmaUtil" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" -o pmc --output-format csv -- python $REPO/tools/train_step_probe.py 0 > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i [$grp] rc=$?"
done
cd $REPO
python - <<'PY' | tee gpurun_out/r02_exp42/bwd_pmc.txt
import csv, glob, collections, os
out = "gpurun_out/r02_exp42"
for d in sorted(glob.glob(out + "/pmc*/")):
    for f in glob.glob(d + "*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "edge_bwd_kernel" in k:
                short = "by_dest+W2" if "true, false" in k else ("by_src+S" if "false, true" in k else "other")
            elif "edge_tail" in k:
                short = "tail"
            elif "edge_kernel" in k:
                short = "forward+u"
            else:
                continue
            agg[short][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(short, r["Counter_Name"])] += 1
        for s in sorted(agg):
            print(os.path.basename(d.rstrip("/")), s, {c: round(v / cnt[(s, c)], 1) for c, v in agg[s].items()}, "launches", max(cnt[(s, c)] for c in agg[s]))
PY
rm -rf $OUT/pmc*/
