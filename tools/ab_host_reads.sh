#!/bin/bash
# A/B of the backward's host reads (GPU box): EGNN_HOST_READ_EVENTS=0 (every small read-back drains the stream) vs 1 (_ops.HostRead)
export TMPDIR=/tmp
python -m pytest tests/test_autograd.py tests/test_dropout.py tests/test_gpu_fuzz.py -m gpu -q -x -p no:cacheprovider -W ignore::UserWarning 2>&1 | tail -2
for i in 1 2 3; do
for v in 0 1; do
  echo "events=$v"; EGNN_HOST_READ_EVENTS=$v python tools/train_step_probe.py 6 2>&1 | grep "^step [3-6]"
done
done
EGNN_PROBE_PHASES=0 python tools/net_train_probe.py 2>&1 | grep "^c[35]"
EGNN_HOST_READ_EVENTS=0 EGNN_PROBE_PHASES=0 python tools/net_train_probe.py 2>&1 | grep "^c[35]"
