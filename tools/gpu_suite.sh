#!/bin/bash
# the whole -m gpu suite on the box; log under gpurun_out/<tag>/
TAG="${1:-suite}"; shift
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning "$@" > gpurun_out/$TAG/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)" gpurun_out/$TAG/pytest_gpu.log | head -40; tail -3 gpurun_out/$TAG/pytest_gpu.log
