#!/bin/bash
# Round-2 measurement run (GPU box): the default bench line, the other BASELINE.json configs (+ ragged mask, + the reference
# eager on the GPU), rocprofv3 kernel trace of the default command, PMC passes.  Everything lands under gpurun_out/prof_<tag>/;
# copy the summaries into profiles/<tag>/.
TAG="${1:-r02_final}"
REPO="$(pwd)"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
python bench.py --reference-eager > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench rc=$?"; head -c 300 $OUT/bench_line.json; echo
python bench.py --ragged-mask --no-cpu-baseline > $OUT/bench_ragged_mask.json 2>> $OUT/bench_line.err
for w in c2_dense c3_network c4_sparse c5_shard; do
  python bench.py --workload $w > $OUT/bench_$w.json 2>> $OUT/bench_line.err; head -c 200 $OUT/bench_$w.json; echo
done
bash tools/profile.sh $TAG 2>&1 | tail -45
