#!/bin/bash
mkdir -p gpurun_out/r02_exp26
OUT=gpurun_out/r02_exp26
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
