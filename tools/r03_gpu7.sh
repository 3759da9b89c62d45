#!/bin/bash
# round 3, GPU call 7: slot records + two rounds per workgroup on narrow layers: full GPU suite, bench lines (north star, c3, c5)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r03_7_pytest.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r03_7_pytest.log
for wl in north_star c3_network c5_shard; do
  timeout 600 python bench.py --steps 20 --warmup 5 --workload $wl --no-cpu-baseline > gpurun_out/r03_7_bench_$wl.json 2> gpurun_out/r03_7_bench_$wl.err
  echo "bench $wl rc=$?"; python - <<PY
import json
d = json.load(open("gpurun_out/r03_7_bench_$wl.json"))
print(d["value"], d["ms_per_step"], d.get("kernel_ms_per_step") or {k["kernel"]: k["avg_ms"] for k in d["kernels"]})
PY
done
EGNN_SLOT_PREP=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_7_bench_noslots.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r03_7_bench_noslots.json')); print('no slots:', d['value'], d['ms_per_step'], {k['kernel']: k['avg_ms'] for k in d['kernels']})"
