#!/bin/bash
mkdir -p gpurun_out/r02_exp15
OUT=gpurun_out/r02_exp15
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_autograd.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_autograd.log
python tools/train_step_probe.py 3 | tee $OUT/steps.txt
