#!/bin/bash
# round 3, GPU call 4: what the 0.2 ms of setup + epilogue (hidden loop compiled out) consist of
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/variants.py run shapes=ns reps=10 only=ABL4+ABL260+ABL516+ABL1028+ABL1540 2>&1 | tee gpurun_out/r03_4_fixed_cost.txt
