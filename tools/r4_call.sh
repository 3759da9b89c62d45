#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
O=gpurun_out/r04
timeout 900 python tools/variants.py run shapes=ns,c3,c5 reps=10 > $O/variants_gemm.log 2>&1
echo "variants rc=$?"; cut -c1-60 $O/variants_gemm.log | paste -d' ' - <(grep -o "\"node_proj\": [0-9.]*\|\"node_mlp0\": [0-9.]*\|\"node_mlp1\": [0-9.]*\|\"_digest\": \"[0-9a-f]*" $O/variants_gemm.log | paste - - - -)
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
