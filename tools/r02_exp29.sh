#!/bin/bash
mkdir -p gpurun_out/r02_exp29
OUT=$(pwd)/gpurun_out/r02_exp29
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
EGNN_POISON_ALLOC=1 EGNN_TEST_VERBOSE=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning -k "edge_bwd_pass" -s 2>&1 | grep -v amdgpu.ids > $OUT/pytest_kernel.log; grep "passed\|failed\|Error\|assert" $OUT/pytest_kernel.log | head -20; grep "fused" $OUT/pytest_kernel.log | awk '{print $2, $5}' | sort | awk '{a[$1]=a[$1]" "$2} END{for (k in a) print k, a[k]}'
timeout 600 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_autograd.log 2>&1; echo "pytest autograd rc=$?"; tail -5 $OUT/pytest_autograd.log
