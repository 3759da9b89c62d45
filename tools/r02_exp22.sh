#!/bin/bash
mkdir -p gpurun_out/r02_exp22
OUT=$(pwd)/gpurun_out/r02_exp22
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
for sp in dest both; do
EGNN_BWD_SPLIT=$sp EGNN_POISON_ALLOC=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line -p no:cacheprovider -W ignore::UserWarning -k "edge_bwd_pass" 2>&1 | grep -v amdgpu.ids | tail -2
done
timeout 600 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_autograd.log 2>&1; echo "pytest autograd rc=$?"; tail -3 $OUT/pytest_autograd.log
for sp in dest both dest; do echo "SPLIT=$sp"; EGNN_BWD_SPLIT=$sp timeout 300 python tools/train_step_probe.py 2 | tail -2 | cut -c1-900; done | tee $OUT/steps.txt
