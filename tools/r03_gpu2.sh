#!/bin/bash
# round 3, GPU call 2: VALU / MFMA / transcendental port model of one SIMD; edge variants at the c3 / c5 widths
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 ./build_variants/port_model 2>&1 | tee gpurun_out/r03_2_port_model.txt
timeout 600 python tools/variants.py run shapes=c3,c5 reps=10 2>&1 | tee gpurun_out/r03_2_variants.txt
