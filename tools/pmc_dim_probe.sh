export TMPDIR=/tmp
cd /tmp
for grp in "VALUBusy" "SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pd -o pmc --output-format csv -- python /root/repo/tools/dim_probe.py > /tmp/pd.log 2>&1
  python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pd/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "edge_bwd_kernel" not in n: continue
        kind = "dest" if "Lb1ELb0" in n or "true, false" in n else "src"
        acc[(kind, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v)/len(v), 1) for c, v in acc[k].items()})
PY
  rm -rf /tmp/pd
done
