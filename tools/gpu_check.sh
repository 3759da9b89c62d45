#!/bin/bash
# First-contact GPU run: build, smoke, full GPU test-suite; logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "build rc=$?"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
