#!/bin/bash
mkdir -p gpurun_out/r02_exp36
OUT=$(pwd)/gpurun_out/r02_exp36
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
EGNN_POISON_ALLOC=1 timeout 600 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_autograd.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|amdgpu.ids" $OUT/pytest_autograd.log | tail -8 | cut -c1-250
for sp in 1 0 1 0; do echo "SPATIAL=$sp"; EGNN_BWD_SPATIAL_ORDER=$sp timeout 300 python tools/train_step_probe.py 2 | tail -2 | cut -c1-900; done | tee $OUT/steps.txt
