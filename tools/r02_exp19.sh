#!/bin/bash
mkdir -p gpurun_out/r02_exp19
OUT=$(pwd)/gpurun_out/r02_exp19
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
EGNN_POISON_ALLOC=1 EGNN_TEST_VERBOSE=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line -p no:cacheprovider -W ignore::UserWarning -k "edge_bwd_pass" -s 2>&1 | grep -v amdgpu.ids | tee $OUT/pytest_kernel.log | grep "g_w2\|passed\|failed\|Error"
