#!/bin/bash
# Kernel trace of the training step at the north-star shape (GPU box, from the repo root):   bash tools/train_trace.sh <tag>
# rocprofv3 --kernel-trace --stats over `bench.py --train-step` (4 forward + backward pairs after the timed inference region);
# prints every kernel's calls / total / average so that the backward's ATen kernels and gaps can be read off next to ours.
TAG="${1:-train}"
REPO="$(pwd)"
OUT="$REPO/gpurun_out/train_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic --train-step > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
grep "^{\"metric\"" "$OUT/trace.log" | tail -1 > "$OUT/bench_line.json"
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out + "/kernels.txt", "w") as o:
    for r in rows:
        line = f'{float(r["TotalDurationNs"])/1e6:9.3f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:10.1f} us  {r["Name"][:150]}'
        o.write(line + "\n")
    o.write(f"total {tot/1e6:.3f} ms\n")
print(open(out + "/kernels.txt").read()[:9000])
print(json.load(open(out + "/bench_line.json")).get("train_step"))
PY
rm -f $OUT/trace/*kernel_trace.csv $OUT/trace/*/*kernel_trace.csv 2>/dev/null
