#!/bin/bash
mkdir -p gpurun_out/r02_exp40
OUT=$(pwd)/gpurun_out/r02_exp40
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
EGNN_POISON_ALLOC=1 timeout 600 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_autograd.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|amdgpu.ids" $OUT/pytest_autograd.log | tail -8 | cut -c1-250
python tools/net_train_probe.py 2>&1 | grep -v amdgpu | tee $OUT/net_train.txt
python tools/train_step_probe.py 2 | tail -2 | cut -c1-200
