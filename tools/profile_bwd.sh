#!/bin/bash
# PMC counters of the backward kernels over one training step at the north-star shape (GPU box, repo root):  bash tools/profile_bwd.sh <tag>
# One rocprofv3 pass per counter group (kernel trace + counters only), summary of the package's backward kernels printed and
# written to gpurun_out/bwd_<tag>/summary.txt.
TAG="${1:-bwd}"
REPO="$(pwd)"
OUT="$REPO/gpurun_out/bwd_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "VALUBusy" "MfmaUtil" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" -o pmc --output-format csv -- python $REPO/tools/train_step_probe.py 1 > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i [$grp] rc=$?"
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
keep = ("edge_bwd_kernel", "edge_bwd_prep", "edge_tail_bwd", "edge_pool", "rows_gather_sum", "dest_lists", "dest_totals", "split_scaled", "absmax",
        "linear_hl_splitk", "silu_bwd", "unsplit_words", "sum_parts")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        k = next((x for x in keep if x in name), None)
        if k is None:
            continue
        if k == "edge_bwd_kernel":
            # template arguments <NM, ST, WANT_W2, WANT_S, ...>: the pass that carries d/d W_2 is the by-destination one
            m = re.search(r"edge_bwd_kernel<\s*\d+,\s*\d+,\s*(true|false),\s*(true|false)", name) or re.search(r"edge_bwd_kernelILi\d+ELi\d+ELb([01])ELb([01])", name)
            w2 = m is not None and m.group(1) in ("true", "1")
            k = "edge_bwd by_dest (+dW2)" if w2 else "edge_bwd by_src (+dWs, ds)"
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as o:
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]
            line = f"{k:30s} {c:18s} launches={len(v):3d} avg={sum(v)/len(v):16.1f}"
            print(line); o.write(line + "\n")
PY
rm -f $OUT/pmc*/*/*kernel_trace.csv $OUT/pmc*/*kernel_trace.csv 2>/dev/null
