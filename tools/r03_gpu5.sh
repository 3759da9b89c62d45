#!/bin/bash
# round 3, GPU call 5: full GPU suite + default bench line with the five-workgroup edge pass
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r03_5_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r03_5_pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_5_bench.json 2> gpurun_out/r03_5_bench.err
echo "bench rc=$?"; cut -c1-1200 gpurun_out/r03_5_bench.json
