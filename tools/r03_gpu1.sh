#!/bin/bash
# round 3, GPU call 1: full GPU suite with the LDS-DMA gathers + residual-on-MFMA edge pass, then the four variants A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r03_1_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r03_1_pytest.log
timeout 600 python tools/variants.py run shapes=ns,c3 reps=10 2>&1 | tee gpurun_out/r03_1_variants.txt
