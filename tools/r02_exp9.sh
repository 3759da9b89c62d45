#!/bin/bash
mkdir -p gpurun_out/r02_exp9
OUT=gpurun_out/r02_exp9
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "^$" $OUT/pytest_gpu.log | tail -12
