#!/bin/bash
# per-kernel times of the default bench for several libraries: bash ab_kernel.sh <kernel> <reps> <lib>... [-- bench flags]
KER="$1"; REPS="$2"; shift 2; LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for r in $(seq 1 "$REPS"); do for v in "${LIBS[@]}"; do
  EGNN_HIP_LIB=$v python bench.py --no-cpu-baseline --no-train-step --no-live-traffic "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline()); e = [k['avg_ms'] for k in d['kernels'] if k['kernel'] == '$KER']
print('$v'.split('/')[-2], 'value', d['value'], 'deferred', d.get('value_range_check_deferred'), '$KER ms', e)"
done; done
