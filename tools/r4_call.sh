#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
O=gpurun_out/r04
timeout 900 python tools/variants.py run shapes=ns,c3,c5 reps=10 > $O/variants.log 2>&1
echo "variants rc=$?"; cut -c1-70 $O/variants.log | paste -d' ' - <(grep -o "\"edge_fused\": [0-9.]*, .*_avg_edge\": [0-9.]*, \"_digest\": \"[0-9a-f]*" $O/variants.log | sed 's/"node_mlp0.*"_avg/"_avg/')
timeout 900 python tools/variants.py run shapes=ns reps=10 > $O/variants_b.log 2>&1
cut -c1-70 $O/variants_b.log | paste -d' ' - <(grep -o "\"edge_fused\": [0-9.]*, .*_avg_edge\": [0-9.]*" $O/variants_b.log | sed 's/"node_mlp0.*"_avg/"_avg/')
