#!/bin/bash
# A/B of a training step: bash ab_train.sh <libA> <libB> <reps> [bench flags]
A="$1"; B="$2"; REPS="$3"; shift 3
for r in $(seq 1 $REPS); do for v in "$A" "$B"; do
  EGNN_HIP_LIB=$v python bench.py --no-cpu-baseline --no-live-traffic --train-step "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v'.split('/')[-2], d['value'], d.get('train_step'))"
done; done
