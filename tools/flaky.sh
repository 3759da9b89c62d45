#!/bin/bash
# GPU suite three times + the default bench line three times on one box (run-to-run variance / flakiness record)
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -W ignore 2>&1 | tail -1
done | tee gpurun_out/flaky_pytest.txt
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k['kernel']: k['avg_ms'] for k in d['kernels']})"
done | tee gpurun_out/flaky_bench.txt
