#!/bin/bash
mkdir -p gpurun_out/r02_exp18
OUT=$(pwd)/gpurun_out/r02_exp18
REPO=$(pwd)
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
EGNN_TEST_VERBOSE=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line -p no:cacheprovider -W ignore::UserWarning -k "edge_bwd_pass" -s 2>&1 | grep -v amdgpu.ids | tee $OUT/pytest_kernel.log | tail -50
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "VALUBusy" "MfmaUtil" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" -o pmc --output-format csv -- python $REPO/tools/train_step_probe.py 0 > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i [$grp] rc=$?"
done
cd $REPO
python - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/r02_exp18"
for d in sorted(glob.glob(out + "/pmc*/")):
    for f in glob.glob(d + "*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "edge_bwd_kernel" not in k: continue
            short = "by_src" if "true" in k or "Lb1" in k else "by_dest"
            agg[short][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(short, r["Counter_Name"])] += 1
        for s in agg:
            print(os.path.basename(d.rstrip("/")), s, {c: (round(v / cnt[(s, c)], 1)) for c, v in agg[s].items()}, "launches", max(cnt[(s, c)] for c in agg[s]))
PY
rm -rf $OUT/pmc*/
