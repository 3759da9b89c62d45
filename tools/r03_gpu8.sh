#!/bin/bash
# round 3, GPU call 8: hand-scheduled K-tile of the GEMM; one training step with the native destination lists + gradient GEMMs
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/variants.py run shapes=ns,c3 reps=10 2>&1 | tee gpurun_out/r03_8_gemm_variants.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-step > gpurun_out/r03_8_train_step.json 2> gpurun_out/r03_8_train_step.err
python -c "
import json; d=json.load(open('gpurun_out/r03_8_train_step.json')); print('train step:', d['train_step'])"
EGNN_BWD_GRAD_GEMM=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-step > gpurun_out/r03_8_train_step_libgemm.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r03_8_train_step_libgemm.json')); print('train step (library GEMMs):', d['train_step'])"
timeout 300 python tools/train_step_probe.py 2>&1 | tail -30 | tee gpurun_out/r03_8_train_probe.txt
