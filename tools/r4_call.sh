#!/bin/bash
# One gpurun call of round 4 (edit per call): correctness first, then A/B timings.
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
O=gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_kernels.py -q --tb=short -p no:cacheprovider -k "persistent or dest_lists or slot_prep" > $O/pw_test.log 2>&1
echo "pw test rc=$?"; tail -8 $O/pw_test.log
timeout 900 python tools/pw_probe.py reps=10 shapes=ns,c3,c5 > $O/pw_probe.log 2>&1
echo "probe rc=$?"; grep -o "^[a-z0-9_]* *algo=[01]\|\"edge_fused\": \[[0-9., ]*\]\|bit-identical.*" $O/pw_probe.log | paste - - | head -40
timeout 900 python tools/variants.py run shapes=ns,c3 reps=10 > $O/variants.log 2>&1
echo "variants rc=$?"; cut -c1-90 $O/variants.log | paste -d' ' - <(grep -o "\"edge_fused\": [0-9.]*, .*_avg_edge\": [0-9.]*" $O/variants.log | sed 's/"node_mlp0.*"_avg/"_avg/')
