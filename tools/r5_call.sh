#!/bin/bash
# One gpurun call of round 5: A/B of the compile-time variants under build_variants/ (tools/variants.py), inside ONE box.
#   bash tools/r5_call.sh <tag> [shapes] [only]
TAG="${1:-r05_exp1}"
SHAPES="${2:-ns}"
ONLY="${3:-}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -n "$ONLY" ]; then ONLYARG="only=$ONLY"; else ONLYARG=""; fi
timeout 1500 python tools/variants.py run shapes=$SHAPES reps=10 $ONLYARG 2>&1 | tee $OUT/variants.txt | python -c "
import sys, json
for l in sys.stdin:
    try:
        shape, tag = l.split()[:2]; d = json.loads(l[l.index('{'):])
        print(f\"{shape:3s} {tag:58s} edge {d.get('edge_fused')} avg {d.get('_avg_edge')} proj {d.get('node_proj')} mlp {d.get('node_mlp0')} {d.get('node_mlp1')} knn {d.get('knn_select')} {d.get('_digest')}\")
    except Exception as e:
        print(l.rstrip()[:300])
"
