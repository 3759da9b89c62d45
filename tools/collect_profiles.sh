#!/bin/bash
# copy the summaries of a tools/measure.sh run (gpurun_out/prof_<tag>/) into profiles/<tag>/ (tracked): bash tools/collect_profiles.sh <tag>
TAG="$1"; SRC="gpurun_out/prof_$TAG"; DST="profiles/$TAG"
mkdir -p "$DST"
cp $SRC/bench_*.json $SRC/net_train_step.txt $SRC/train_step_kernels.txt $SRC/train_step_kernel_trace.txt $SRC/pmc_traffic.json "$DST"/ 2>/dev/null
tail -3 $SRC/pytest_gpu.log > "$DST/pytest_gpu_tail.txt"
for w in north_star c3_network c5_shard c2_dense c4_sparse; do
  [ -d "$SRC/$w" ] || continue
  mkdir -p "$DST/$w"
  cp "$SRC/$w/summary.txt" "$DST/$w/rocprofv3_summary.txt"
  cp "$SRC/$w/bench_line_under_rocprof.json" "$DST/$w/" 2>/dev/null
  f=$(find "$SRC/$w/trace" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$DST/$w/kernel_stats.csv"
done
cp $SRC/pmc_traffic.json profiles/pmc_traffic.json
du -sh "$DST"
