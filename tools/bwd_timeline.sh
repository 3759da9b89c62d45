#!/bin/bash
# Ordered kernel timeline of ONE training step (forward + backward) at the north-star shape, with the idle gap before every kernel
# (GPU box, from the repo root):   bash tools/bwd_timeline.sh <tag>
TAG="${1:-bwd}"
REPO="$(pwd)"
OUT="$REPO/gpurun_out/timeline_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d "$OUT/trace" -o trace --output-format csv -- python $REPO/tools/train_step_probe.py 3 > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
starts = [i for i, e in enumerate(ev) if "node_prep_hl_kernel" in e[2]]
# the last training step: from the last forward's node_prep (the backward calls node_prep too: take the last one followed by knn_select)
cand = [i for i in starts if any("knn_select" in e[2] for e in ev[i:i + 6])]
s0 = cand[-1]
t0 = ev[s0][0]
busy = t0
idle = 0.0
with open(out + "/timeline.txt", "w") as o:
    for a, b, name, q in ev[s0:]:
        gap = max(0, a - busy)
        idle += gap
        short = name.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:70]
        o.write(f"+{(a - t0) / 1e3:9.1f} us  dur {(b - a) / 1e3:8.1f}  idle-before {gap / 1e3:6.1f}  q={q}  {short}\n")
        busy = max(busy, b)
    o.write(f"total {(busy - t0) / 1e3:.1f} us, idle {idle / 1e3:.1f} us, {len(ev) - s0} kernels\n")
print(open(out + "/timeline.txt").read())
PY
rm -rf $OUT/trace
