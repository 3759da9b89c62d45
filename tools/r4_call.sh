#!/bin/bash
# One gpurun call of round 4: correctness of the persistent edge kernel first, then the A/B timings, then the whole suite.
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
O=gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_kernels.py -q --tb=short -p no:cacheprovider -k "persistent or dest_lists" > $O/pw_test.log 2>&1
echo "pw test rc=$?"; tail -15 $O/pw_test.log
timeout 900 python tools/pw_probe.py reps=10 > $O/pw_probe.log 2>&1
echo "probe rc=$?"; cat $O/pw_probe.log | cut -c1-600
timeout 900 python tools/variants.py run shapes=ns,c3 reps=10 > $O/variants.log 2>&1
echo "variants rc=$?"; cat $O/variants.log | cut -c1-400
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; cat $O/bench.json | cut -c1-1500
