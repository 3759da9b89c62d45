#!/bin/bash
mkdir -p gpurun_out/r02_exp4
OUT=gpurun_out/r02_exp4
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log
python bench.py --reference-eager > $OUT/bench_sync.json 2> $OUT/bench_sync.err; tail -c 900 $OUT/bench_sync.json; tail -3 $OUT/bench_sync.err
EGNN_RANGE_CHECK=deferred python bench.py --no-cpu-baseline > $OUT/bench_deferred.json 2> $OUT/bench_deferred.err; head -c 400 $OUT/bench_deferred.json
EGNN_RANGE_CHECK=off python bench.py --no-cpu-baseline > $OUT/bench_off.json 2> $OUT/bench_off.err; head -c 400 $OUT/bench_off.json
