#!/bin/bash
mkdir -p gpurun_out/r02_exp11
OUT=gpurun_out/r02_exp11
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python tools/edge_tune.py src=linear_hl "HL_CFG=1" "HL_CFG=2" "HL_CFG=0" "HL_CFG=1,HL_ILV=0" "HL_CFG=1" 2>&1 | sed 's/"knn_select[^}]*"node_proj"/"node_proj"/; s/"spatial_order": [0-9.]*, "edge_fused": [0-9.]*, //' | tee $OUT/gemm_tune.txt
