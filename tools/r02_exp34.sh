#!/bin/bash
mkdir -p gpurun_out/r02_exp34
OUT=$(pwd)/gpurun_out/r02_exp34
REPO=$(pwd)
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 300 python tools/train_step_probe.py 2 | tail -3 | cut -c1-600 | tee $OUT/steps.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bwd --output-format csv -- python $REPO/tools/train_step_probe.py 1 > $OUT/prof.log 2>&1; echo "prof rc=$?"
cd $REPO
rm -f $OUT/prof/*kernel_trace.csv
python - <<'PY'
import csv, re
rows=list(csv.DictReader(open('gpurun_out/r02_exp34/prof/bwd_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("sum of kernel time over 3 steps (1 cold): %.1f ms" % (tot/1e6))
for r in rows[:40]:
    n=re.sub(r'_UserArgs_(MT\d+x\d+x\d+).*', r' \1', r['Name'])[:110]
    print(f"{int(r['TotalDurationNs'])/1e6:8.2f} ms  calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e6:7.3f}  {n}")
PY
