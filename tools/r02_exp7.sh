#!/bin/bash
mkdir -p gpurun_out/r02_exp7
OUT=gpurun_out/r02_exp7
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/bpermute_dma tools/ubench/bpermute_dma.hip 2>/dev/null && timeout 120 /tmp/bpermute_dma | tee $OUT/bpermute_dma.txt
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], [(k['kernel'], k['avg_ms']) for k in d['kernels']])"
python bench.py --no-cpu-baseline --ragged-mask > $OUT/bench_ragged.json 2>> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_ragged.json')); print('ragged', d['value'], d['ms_per_step'], [(k['kernel'], k['avg_ms']) for k in d['kernels']])"
python bench.py --no-cpu-baseline --workload c3_network > $OUT/bench_c3.json 2>> $OUT/bench.err; cat $OUT/bench_c3.json | head -c 700
