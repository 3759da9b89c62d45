#!/bin/bash
# round 3, GPU call 6: rounds per workgroup, legacy K=16 MFMAs for the second layer
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/variants.py run shapes=ns,c3 reps=10 2>&1 | tee gpurun_out/r03_6_variants.txt
