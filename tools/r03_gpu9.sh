#!/bin/bash
# round 3, GPU call 9: full GPU suite with training-mode dropout
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r03_9_pytest.log 2>&1
echo "pytest rc=$?"; tail -30 gpurun_out/r03_9_pytest.log | cut -c1-300
