#!/bin/bash
# A/B of the stage-wise SiLU+split ordering in the edge kernel's hidden loop (4 or 8 independent chains between dependent
# VALU instructions instead of the compiler's 2), same box, same call.
mkdir -p gpurun_out/r02_exp12
OUT=gpurun_out/r02_exp12
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 1200 python tools/edge_tune.py "EDGE_STAGEWISE=0" "EDGE_STAGEWISE=4" "EDGE_STAGEWISE=8" "EDGE_STAGEWISE=0" "EDGE_STAGEWISE=4" 2>&1 | tee $OUT/stagewise.txt
