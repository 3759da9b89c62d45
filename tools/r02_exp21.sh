#!/bin/bash
mkdir -p gpurun_out/r02_exp21
OUT=$(pwd)/gpurun_out/r02_exp21
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
EGNN_POISON_ALLOC=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line -p no:cacheprovider -W ignore::UserWarning -k "edge_bwd_pass" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_autograd.log 2>&1; echo "pytest autograd rc=$?"; tail -3 $OUT/pytest_autograd.log
for r in 8 16 4 32; do echo "ROUNDS_PER_SLAB=$r"; EGNN_BWD_ROUNDS_PER_SLAB=$r timeout 300 python tools/train_step_probe.py 2 | tail -2 | cut -c1-900; done | tee $OUT/steps.txt
