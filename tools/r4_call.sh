#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
O=gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_kernels.py -q --tb=short -p no:cacheprovider -k "persistent or slot_prep" > $O/pw_test.log 2>&1
echo "pw test rc=$?"; tail -4 $O/pw_test.log
timeout 900 python tools/variants.py run shapes=ns,c3,c5 reps=10 > $O/variants.log 2>&1
echo "variants rc=$?"; cut -c1-70 $O/variants.log | paste -d' ' - <(grep -o "\"edge_fused\": [0-9.]*, .*_avg_edge\": [0-9.]*" $O/variants.log | sed 's/"node_mlp0.*"_avg/"_avg/')
timeout 600 bash tools/pmc_quick.sh gpurun_out/r04/pmc_ns ns
timeout 600 bash tools/pmc_quick.sh gpurun_out/r04/pmc_c3 c3
