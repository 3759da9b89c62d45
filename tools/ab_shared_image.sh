export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_c_abi.py -m gpu -q -x -p no:cacheprovider -W ignore::UserWarning 2>&1 | tail -3
for i in 1 2 3; do
for v in 0 1; do
  EGNN_SHARED_FEATS_IMAGE=$v python bench.py --no-cpu-baseline --no-live-traffic --no-train-step 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('shared=$v', d['value'], d['ms_per_step'], {k['label'] if 'label' in k else k.get('kernel'): k.get('avg_ms') for k in d.get('kernels', [])})"
done
done
