#!/bin/bash
mkdir -p gpurun_out/r02_exp32
OUT=$(pwd)/gpurun_out/r02_exp32
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
EGNN_POISON_ALLOC=1 timeout 600 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning -k "tail_kernel" > $OUT/pytest_tail.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|amdgpu.ids" $OUT/pytest_tail.log | tail -25 | cut -c1-220
