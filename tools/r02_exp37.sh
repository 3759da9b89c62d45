#!/bin/bash
mkdir -p gpurun_out/r02_exp37
OUT=$(pwd)/gpurun_out/r02_exp37
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning -k "full_size" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|amdgpu.ids" $OUT/pytest.log | tail -20 | cut -c1-250
