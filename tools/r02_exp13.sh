#!/bin/bash
# kernel-by-kernel timeline of one training step at the north-star shape
mkdir -p gpurun_out/r02_exp13
OUT=$(pwd)/gpurun_out/r02_exp13
REPO=$(pwd)
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
python tools/train_step_probe.py 1 | tee $OUT/steps.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bwd --output-format csv -- python $REPO/tools/train_step_probe.py 2 > $OUT/prof.log 2>&1; echo "prof rc=$?"
cd $REPO
f=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); echo $f; head -40 "$f" | cut -c1-200
rm -f $OUT/prof/*kernel_trace.csv
