#!/bin/bash
mkdir -p gpurun_out/r02_exp35
OUT=$(pwd)/gpurun_out/r02_exp35
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
for sp in src dest src dest; do echo "SPLIT=$sp"; EGNN_BWD_SPLIT=$sp timeout 300 python tools/train_step_probe.py 2 | tail -2 | cut -c1-900; done | tee $OUT/steps.txt
EGNN_BWD_SPLIT=src timeout 300 python -m pytest tests/test_autograd.py tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning -k "native_backward or edge_bwd_pass" 2>&1 | tail -2
