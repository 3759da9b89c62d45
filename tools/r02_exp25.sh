#!/bin/bash
mkdir -p gpurun_out/r02_exp25
OUT=$(pwd)/gpurun_out/r02_exp25
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python tools/bwd_tune.py src=edge_bwd "BWD_CH_W2=3" "BWD_CH_W2=4" "BWD_CH_W2=4,BWD_W2_BLOCKS=2" "BWD_GROUP_SLABS=16" "BWD_GROUP_SLABS=64" "BWD_CH_W2=3" 2>&1 | cut -c1-700 | tee $OUT/tune.txt
