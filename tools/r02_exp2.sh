#!/bin/bash
# Round-2 experiment 2 (GPU box): parity with the staging ring; what the hidden loop of the edge pass waits for (ablations).
mkdir -p gpurun_out/r02_exp2
OUT=gpurun_out/r02_exp2
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log
timeout 900 python tools/edge_tune.py "EDGE_RING=1" "EDGE_RING=1,EDGE_PRIO=1" "EDGE_RING=1,EDGE_PRIO=2" \
   "EDGE_RING=1,EDGE_ABL=8" "EDGE_RING=1,EDGE_ABL=16" "EDGE_RING=1,EDGE_ABL=24" "EDGE_RING=1,EDGE_ABL=32" "EDGE_RING=1,EDGE_ABL=56" \
   "EDGE_RING=1,EDGE_ABL=1" "EDGE_RING=1,EDGE_ABL=57" "EDGE_RING=1,EDGE_ABL=2" "EDGE_RING=1,EDGE_ABL=58" 2>&1 | sed 's/"knn_select.*"edge_fused"/"edge_fused"/' | tee $OUT/edge_tune.txt
