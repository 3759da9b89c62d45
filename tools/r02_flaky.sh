#!/bin/bash
mkdir -p gpurun_out/r02_flaky
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r02_flaky/build.log 2>&1; echo "build rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider -W ignore > gpurun_out/r02_flaky/run$i.log 2>&1; echo "run $i rc=$? $(tail -1 gpurun_out/r02_flaky/run$i.log)"
done
