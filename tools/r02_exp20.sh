#!/bin/bash
mkdir -p gpurun_out/r02_exp20
OUT=$(pwd)/gpurun_out/r02_exp20
REPO=$(pwd)
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
EGNN_POISON_ALLOC=1 EGNN_TEST_VERBOSE=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line -p no:cacheprovider -W ignore::UserWarning -k "edge_bwd_pass" -s 2>&1 | grep -v amdgpu.ids | tee $OUT/pytest_kernel.log | grep "fused\|passed\|failed\|Error" | sort | uniq -c | sort -rn | head -30
timeout 600 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_autograd.log 2>&1; echo "pytest autograd rc=$?"; tail -3 $OUT/pytest_autograd.log
timeout 300 python tools/train_step_probe.py 2 | tee $OUT/steps.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bwd --output-format csv -- python $REPO/tools/train_step_probe.py 1 > $OUT/prof.log 2>&1; echo "prof rc=$?"
cd $REPO
rm -f $OUT/prof/*kernel_trace.csv
python - <<'PY'
import csv, re
rows=list(csv.DictReader(open('gpurun_out/r02_exp20/prof/bwd_kernel_stats.csv')))
for r in rows[:32]:
    n=re.sub(r'_UserArgs_(MT\d+x\d+x\d+).*', r' \1', r['Name'])[:100]
    print(f"{int(r['TotalDurationNs'])/1e6:8.2f} ms total  calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e6:7.3f}  {n}")
PY
