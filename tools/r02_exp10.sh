#!/bin/bash
mkdir -p gpurun_out/r02_exp10
OUT=gpurun_out/r02_exp10
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 900 python tools/edge_tune.py "EDGE_LO_PLAIN=1" "EDGE_LO_PLAIN=0" "EDGE_LO_PLAIN=1" "EDGE_LO_PLAIN=0" 2>&1 | sed 's/"knn_select.*"edge_fused"/"edge_fused"/' | tee $OUT/edge_tune.txt
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], [(k['kernel'], k['avg_ms']) for k in d['kernels']])"
