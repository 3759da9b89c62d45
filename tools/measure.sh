#!/bin/bash
# The round's measurement run on the GPU box (from the repo root):   bash tools/measure.sh <tag>
#   * full GPU test suite + smoke()
#   * the default bench line with the reference timed on the host cores, on the MI355X through PyTorch eager, and one training step
#   * ragged masks and the other BASELINE.json configs at full size
#   * one training step with per-kernel HIP-event times, the EGNN_Network configurations under autograd, rocprofv3 kernel trace of the training step
#   * tools/profile.sh: rocprofv3 kernel trace + stats of the default command and one PMC pass per counter group
# Everything lands under gpurun_out/prof_<tag>/; the summaries to keep are copied into profiles/<tag>/ afterwards.
TAG="${1:-r03_final}"
REPO="$(pwd)"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --reference-eager --train-step > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'], d.get('reference_gpu_eager', {}).get('value'), d.get('train_step'))"
python bench.py --ragged-mask --no-cpu-baseline --no-live-traffic > $OUT/bench_ragged_mask.json 2>> $OUT/bench_line.err
for w in c3_network c5_shard; do   # padded batches (the parity protocol's ragged masks): what the padded-node skips are worth
  python bench.py --workload $w --ragged-mask --no-cpu-baseline --no-live-traffic --no-train-step > $OUT/bench_ragged_mask_$w.json 2>> $OUT/bench_line.err
done
for w in c2_dense c3_network c4_sparse c5_shard; do
  HG="--hipgraph"; [ $w = c4_sparse ] && HG=""
  python bench.py --workload $w $HG > $OUT/bench_$w.json 2>> $OUT/bench_line.err; head -c 160 $OUT/bench_$w.json; echo
done
python bench.py --workload c4_sparse --no-cpu-baseline --no-live-traffic --train-step > $OUT/bench_train_step_c4_sparse.json 2>> $OUT/bench_line.err
python tools/train_step_probe.py 3 > $OUT/train_step_kernels.txt 2>&1; tail -2 $OUT/train_step_kernels.txt | cut -c1-400
EGNN_PROBE_PHASES=1 python tools/net_train_probe.py > $OUT/net_train_step.txt 2>&1; grep "^c[35]" $OUT/net_train_step.txt
bash tools/train_trace.sh $TAG > $OUT/train_trace.log 2>&1; cp gpurun_out/train_$TAG/kernels.txt $OUT/train_step_kernel_trace.txt; tail -1 $OUT/train_trace.log
HEAD_="$(cat .head 2>/dev/null)"
for w in north_star c3_network c5_shard c2_dense c4_sparse; do
  bash tools/profile.sh $TAG $w "$HEAD_" 2>&1 | tail -25
done
python - <<PY
import json, glob, os
merged = {}
for f in sorted(glob.glob("$OUT/*/pmc_traffic.json")):
    merged.update(json.load(open(f)))
json.dump(merged, open("$OUT/pmc_traffic.json", "w"), indent=1)
PY
find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
