#!/bin/bash
mkdir -p gpurun_out/r02_exp16
OUT=$(pwd)/gpurun_out/r02_exp16
REPO=$(pwd)
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python tools/bwd_tune.py "EDGE_BWD_NT=0" "EDGE_BWD_NT=1" "EDGE_BWD_NT=0" 2>&1 | tee $OUT/bwd_nt.txt
cd /tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "VALUBusy" "TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_64B_sum TCC_EA_WR_UNCACHED_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" -o pmc --output-format csv -- python $REPO/tools/train_step_probe.py 0 > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i [$grp] rc=$?"
done
cd $REPO
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "gpurun_out/r02_exp16")
for d in sorted(glob.glob(out + "/pmc*/")):
    for f in glob.glob(d + "*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "edge_kernel" not in k: continue
            short = "bwd" if "2>(" in k or ", 2>" in k else ("fwd_u" if ", 1>" in k else "fwd")
            agg[short][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(short, r["Counter_Name"])] += 1
        for s in agg:
            print(os.path.basename(d.rstrip("/")), s, {c: (v, cnt[(s, c)]) for c, v in agg[s].items()})
PY
rm -rf $OUT/pmc*/
