#!/bin/bash
# round 3, GPU call 3: sensitivity map of the LDS-DMA / residual-on-MFMA edge pass (ablations) and occupancy variants
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/variants.py run shapes=ns reps=10 2>&1 | tee gpurun_out/r03_3_variants_ns.txt
timeout 600 python tools/variants.py run shapes=c3 reps=10 only=default+HC128+ABL4 2>&1 | tee gpurun_out/r03_3_variants_c3.txt
