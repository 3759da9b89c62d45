#!/bin/bash
mkdir -p gpurun_out/r02_exp17
OUT=gpurun_out/r02_exp17
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning -k "edge_bwd_pass" > $OUT/pytest_kernel.log 2>&1; echo "pytest kernel rc=$?"; tail -25 $OUT/pytest_kernel.log
timeout 600 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_autograd.log 2>&1; echo "pytest autograd rc=$?"; tail -5 $OUT/pytest_autograd.log
timeout 300 python tools/train_step_probe.py 2 | tee $OUT/steps.txt
