#!/bin/bash
mkdir -p gpurun_out/r02_exp3
OUT=gpurun_out/r02_exp3
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 900 python tools/variant_check.py "EDGE_RING=1,EDGE_DMA_BUILTIN=1" "EDGE_RING=1,EDGE_RING_DBG=1" 2>&1 | tee $OUT/variant_check.txt
timeout 900 python tools/edge_tune.py "EDGE_RING=0" "EDGE_RING=1" "EDGE_RING=0" "EDGE_RING=1" "EDGE_RING=1,EDGE_DMA_BUILTIN=1" "EDGE_RING=1,EDGE_PRIO=1" "EDGE_RING=1,EDGE_HC=128" "EDGE_RING=1,EDGE_HC=512" \
   "EDGE_RING=1,EDGE_ABL=32"  "EDGE_RING=1,EDGE_ABL=1" "EDGE_RING=1,EDGE_ABL=2" "EDGE_RING=1,EDGE_ABL=35" 2>&1 | sed 's/"knn_select.*"edge_fused"/"edge_fused"/' | tee $OUT/edge_tune.txt
