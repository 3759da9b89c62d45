#!/bin/bash
# One-call A/B of an environment switch on the default bench line:   bash tools/ab_env.sh <tag> <VAR> <a> <b> [reps] [bench flags ...]
# Alternates the two settings `reps` times (same box, same process image) and prints value / ms_per_step / the deferred-mode value.
TAG="$1"; VAR="$2"; A="$3"; B="$4"; REPS="${5:-3}"; shift 5
OUT="gpurun_out/$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp
for r in $(seq 1 "$REPS"); do
  for v in "$A" "$B"; do
    env "$VAR=$v" python bench.py --no-cpu-baseline --no-train-step --no-live-traffic "$@" 2>> "$OUT/err.log" | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('$VAR=$v', 'value', d['value'], 'ms', d['ms_per_step'], 'deferred', d.get('value_range_check_deferred'), 'sum_kernel_ms', d.get('sum_kernel_ms'))" | tee -a "$OUT/ab.txt"
  done
done
