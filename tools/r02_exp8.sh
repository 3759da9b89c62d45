#!/bin/bash
mkdir -p gpurun_out/r02_exp8
OUT=gpurun_out/r02_exp8
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
for rep in 1 2; do
for ss in 1 0; do
  EGNN_SIDE_STREAM=$ss python bench.py --no-cpu-baseline --steps 40 > $OUT/bench_ss$ss.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_ss$ss.json')); print('side_stream=$ss', d['value'], d['ms_per_step'], [(k['kernel'], k['avg_ms']) for k in d['kernels']])"
done; done
EGNN_SIDE_STREAM=1 python bench.py --no-cpu-baseline --workload c3_network > $OUT/c3_ss1.json 2>> $OUT/bench.err; EGNN_SIDE_STREAM=0 python bench.py --no-cpu-baseline --workload c3_network > $OUT/c3_ss0.json 2>> $OUT/bench.err
python -c "
import json
for s in (1,0):
    d=json.load(open('$OUT/c3_ss%d.json'%s)); print('c3 side_stream=%d'%s, d['value'], d['ms_per_step'])"
