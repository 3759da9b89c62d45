#!/bin/bash
# End-of-round check on the GPU box: full GPU test suite, smoke, the default bench line (+ reference eager on the GPU, + training
# step), rocprofv3 kernel trace + stats of the default command.  (The PMC passes / other configs of tools/r02_measure.sh were run
# earlier in the round on the same forward kernels.)
TAG="${1:-r02_final2}"
REPO="$(pwd)"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --reference-eager --train-step > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'], d.get('reference_gpu_eager', {}).get('value'), d.get('train_step'))"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/rocprof -o bench --output-format csv -- python $REPO/bench.py --no-cpu-baseline > $OUT/bench_line_under_rocprof.json 2> $OUT/rocprof.err; echo "rocprof rc=$?"
cd $REPO
rm -f $OUT/rocprof/*kernel_trace.csv
head -12 $OUT/rocprof/bench_kernel_stats.csv | cut -c1-160
