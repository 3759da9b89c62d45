#!/bin/bash
# rocprofv3 passes over the default bench (north-star workload). Run on the GPU box from the repo root:
#   bash tools/profile.sh <tag>
# kernel-trace/stats and every PMC group are separate runs (FETCH_SIZE and WRITE_SIZE cannot share a pass;
# gpurun refuses --pmc combined with sys/hip/hsa traces).
#   bash tools/profile.sh <tag> [workload] [head]      (workload: bench.py --workload, default north_star; other workloads take the
#                                                       kernel trace and the traffic / instruction counters only; head: the commit
#                                                       hash to stamp the traffic file with -- the GPU box has no .git)
TAG="${1:-r01}"
WL="${2:-north_star}"
HEAD_="${3:-}"
REPO="$(pwd)"
OUT="$REPO/gpurun_out/prof_$TAG/$WL"
mkdir -p "$OUT"
export TMPDIR=/tmp
# (the counter passes leave the training step out: its gradient GEMMs share kernel instantiations with node_proj / node_mlp and
# would be averaged into their counters)
BENCH="python $REPO/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --no-live-traffic"
cd /tmp
# the kernel trace runs the DEFAULT bench command (what the driver runs) minus the training step that the default appends AFTER the
# timed region (--no-train-step: its gradient GEMMs share kernel instantiations with node_proj / node_mlp and would be averaged into
# their rows; tools/train_trace.sh traces that part), so its per-kernel averages are the ones the bench line's live measurement has
# to agree with; the counter passes use a shorter run
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python $REPO/bench.py --workload $WL --no-train-step --no-live-traffic $([ "$WL" = north_star ] || echo --no-cpu-baseline) > "$OUT/trace.log" 2>&1
grep "^{\"metric\"" "$OUT/trace.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"
echo "trace rc=$?"
i=0
GROUPS_ALL=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "MfmaUtil" "VALUBusy")
[ "$WL" = north_star ] || GROUPS_ALL=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "VALUBusy")
for grp in "${GROUPS_ALL[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" -o pmc --output-format csv -- $BENCH > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i [$grp] rc=$?"
done
cd "$REPO"
python tools/summarize_prof.py "$OUT" "$WL" "$HEAD_" > "$OUT/summary.txt" 2>&1
tail -60 "$OUT/summary.txt"
