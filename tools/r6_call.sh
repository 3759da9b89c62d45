#!/bin/bash
# One gpurun call of round 6:   bash tools/r6_call.sh <tag> [suite=1] [trace=1] [extra bench flags ...]
#   * the whole -m gpu suite (suite=1)
#   * the default bench line without the CPU baseline / training step / live traffic passes (what `value` is)
#   * a rocprofv3 kernel trace of a short bench run and tools/timeline.py over its last steps: every launch of a step with the idle gap
#     in front of it (trace=1)
TAG="${1:-r6}"; SUITE="${2:-1}"; TRACE="${3:-1}"; shift 3
REPO="$(pwd)"; OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp
if [ "$SUITE" = 1 ]; then bash tools/gpu_suite.sh "$TAG"; fi
python bench.py --no-cpu-baseline --no-train-step --no-live-traffic "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "deferred", d.get("value_range_check_deferred"))
print({k["kernel"]: k["avg_ms"] for k in d.get("kernels", [])})
PY
if [ "$TRACE" = 1 ]; then
  cd /tmp
  rocprofv3 --kernel-trace -d "$OUT/trace" -o trace --output-format csv -- python "$REPO/bench.py" --steps 8 --warmup 3 --no-cpu-baseline --no-train-step --no-live-traffic "$@" > "$OUT/trace.log" 2>&1
  cd "$REPO"
  CSV="$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)"
  python tools/timeline.py "$CSV" 3 > "$OUT/timeline.txt" 2>&1; tail -45 "$OUT/timeline.txt"
  rm -rf "$OUT/trace"
fi
