#!/bin/bash
mkdir -p gpurun_out/r02_exp41
OUT=$(pwd)/gpurun_out/r02_exp41
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
EGNN_POISON_ALLOC=1 timeout 600 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_autograd.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|amdgpu.ids" $OUT/pytest_autograd.log | tail -4 | cut -c1-250
timeout 300 python bench.py --no-cpu-baseline --train-step --workload c4_sparse > $OUT/bench_c4_sparse.json 2> $OUT/bench_c4.err; python -c "
import json; d=json.load(open('$OUT/bench_c4_sparse.json')); print('c4', d['value'], d['ms_per_step'], d.get('train_step'))"
