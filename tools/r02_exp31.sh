#!/bin/bash
mkdir -p gpurun_out/r02_exp31
OUT=$(pwd)/gpurun_out/r02_exp31
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
EGNN_POISON_ALLOC=1 timeout 600 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_autograd.log 2>&1; echo "pytest autograd rc=$?"; tail -25 $OUT/pytest_autograd.log | cut -c1-220
for t in 1 0; do echo "TAIL_KERNEL=$t"; EGNN_BWD_TAIL_KERNEL=$t timeout 300 python tools/train_step_probe.py 2 | tail -2 | cut -c1-900; done | tee $OUT/steps.txt
