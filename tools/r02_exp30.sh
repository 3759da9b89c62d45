#!/bin/bash
mkdir -p gpurun_out/r02_exp30
OUT=$(pwd)/gpurun_out/r02_exp30
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
for wl in north_star c4_sparse c2_dense; do
  timeout 300 python bench.py --no-cpu-baseline --train-step --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; echo "$wl rc=$?"
  python -c "
import json; d=json.load(open('$OUT/bench_$wl.json')); print('$wl', d['value'], d['ms_per_step'], d.get('train_step'))"
done
