#!/bin/bash
# Alternates the default bench line over several libraries inside one call:   bash tools/ab_libs.sh <reps> <lib> <lib> ... [-- bench flags]
REPS="$1"; shift; LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for r in $(seq 1 "$REPS"); do for v in "${LIBS[@]}"; do
  EGNN_HIP_LIB=$v python bench.py --no-cpu-baseline --no-train-step --no-live-traffic "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline()); e = [k['avg_ms'] for k in d['kernels'] if k['kernel'] == 'edge_fused']
print('$v'.split('/')[-2], 'value', d['value'], 'deferred', d.get('value_range_check_deferred'), 'edge_fused ms', e)"
done; done
