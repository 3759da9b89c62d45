#!/bin/bash
mkdir -p gpurun_out/r02_exp6
OUT=gpurun_out/r02_exp6
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_autograd.py -m gpu -q --tb=short -p no:cacheprovider -W ignore::UserWarning > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
python - <<'PY' 2>&1 | tail -12
import os, torch, time
from egnn_pytorch_amd import EGNN, phase_timer
from torch.profiler import profile, ProfilerActivity
torch.manual_seed(0)
layer = EGNN(dim=512, num_nearest_neighbors=32).cuda()
f = torch.randn(64, 1024, 512, device="cuda", requires_grad=True); c = torch.randn(64, 1024, 3, device="cuda", requires_grad=True)
mask = torch.ones(64, 1024, dtype=torch.bool, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n, co = layer(f, c, mask=mask)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    (n.sum() + co.sum()).backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"north-star B=64: forward {1e3*(t1-t0):.1f} ms, backward {1e3*(t2-t1):.1f} ms, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GB")
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    n, co = layer(f, c, mask=mask); (n.sum() + co.sum()).backward(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
PY
