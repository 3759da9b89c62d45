#!/bin/bash
# Round-2 experiment 1 (GPU box): parity of the new edge-kernel defaults, instruction costs, edge-kernel variants.
mkdir -p gpurun_out/r02_exp1
OUT=gpurun_out/r02_exp1
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-inline-asm -o /tmp/mix_rates tools/ubench/mix_rates.hip 2>/dev/null && /tmp/mix_rates | tee $OUT/mix_rates.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/silu_seq tools/ubench/silu_seq.hip 2>/dev/null && /tmp/silu_seq | tee $OUT/silu_seq.txt
timeout 900 python tools/edge_tune.py "EDGE_MIXLO=0,EDGE_RING=0" "EDGE_MIXLO=1,EDGE_RING=0" "EDGE_MIXLO=0,EDGE_RING=1" "EDGE_MIXLO=1,EDGE_RING=1" \
   "EDGE_MIXLO=1,EDGE_RING=1,EDGE_PRIO=1" "EDGE_MIXLO=1,EDGE_RING=1,EDGE_PRIO=3" "EDGE_MIXLO=1,EDGE_RING=1,EDGE_HC=128" \
   "EDGE_MIXLO=1,EDGE_RING=1,EDGE_ABL=4" 2>&1 | tee $OUT/edge_tune.txt
cd /tmp
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "VALUBusy" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d "$GRAFT_REPO_ROOT/$OUT/pmc$i" -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$OUT/pmc$i.log" 2>&1
  echo "pmc$i rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; grep -E "edge" $OUT/summary.txt | cut -c1-160 | head
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
