#!/bin/bash
mkdir -p gpurun_out/r02_exp39
OUT=$(pwd)/gpurun_out/r02_exp39
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o net --output-format csv -- python $REPO/tools/net_train_probe.py > $OUT/prof.log 2>&1; echo "prof rc=$?"
cd $REPO
rm -f $OUT/prof/*kernel_trace.csv
grep -v amdgpu $OUT/prof.log | tail -3
python - <<'PY'
import csv, re
rows=list(csv.DictReader(open('gpurun_out/r02_exp39/prof/net_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("sum of kernel time: %.1f ms (c3: 4 steps x 3 layers, c5: 4 steps x 6 layers)" % (tot/1e6))
for r in rows[:28]:
    n=re.sub(r'_UserArgs_(MT\d+x\d+x\d+).*', r' \1', r['Name'])[:100]
    print(f"{int(r['TotalDurationNs'])/1e6:8.2f} ms  calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e6:7.3f}  {n}")
PY
