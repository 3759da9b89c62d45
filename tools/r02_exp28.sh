#!/bin/bash
# which operand of the d/d W_2 product needs the pre-scaling (A_UP: SiLU(z), GT_UP: gU^T)?  Kernel test errors per variant.
mkdir -p gpurun_out/r02_exp28
OUT=$(pwd)/gpurun_out/r02_exp28
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
for v in "BWD_A_UP=1.f,BWD_GT_UP=1.f" "BWD_A_UP=64.f,BWD_GT_UP=1.f" "BWD_A_UP=1.f,BWD_GT_UP=64.f" "BWD_A_UP=64.f,BWD_GT_UP=64.f"; do
  lib=$(python - <<PY
import sys; sys.path.insert(0, "tools")
import edge_tune
defs = dict(kv.split("=") for kv in "$v".split(","))
print(edge_tune.build("sub_" + "$v".replace("=", "").replace(",", "_").replace(".", ""), defs, "edge_bwd", tuning=False))
PY
)
  echo "== $v"
  EGNN_HIP_LIB=$lib EGNN_TEST_VERBOSE=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line -p no:cacheprovider -W ignore::UserWarning -k "edge_bwd_pass" -s 2>&1 | grep "fused  g_w2\|passed\|failed" | tr '\n' ' '; echo
done | tee $OUT/which_operand.txt
