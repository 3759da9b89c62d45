"""One training step (forward + backward) of the north-star layer, timed, for rocprofv3 --kernel-trace --stats:
    python tools/train_step_probe.py [steps]
PROBE_KW='{"dim": 512, "m_dim": 32, "num_nearest_neighbors": 32}' PROBE_B=16 PROBE_C=3: another layer / batch / coordinate dimension."""
import sys
import time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egnn_pytorch_amd import EGNN, _ops

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.manual_seed(0)
dev = "cuda"
import json
kw = json.loads(os.environ.get("PROBE_KW", '{"dim": 512, "num_nearest_neighbors": 32}'))
B, C = int(os.environ.get("PROBE_B", "64")), int(os.environ.get("PROBE_C", "3"))
layer = EGNN(**kw).to(dev)
feats = torch.randn(B, 1024, kw["dim"], device=dev, requires_grad=True)
coors = torch.randn(B, 1024, C, device=dev, requires_grad=True)
for it in range(steps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f, c = layer(feats, coors)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    (f.square().mean() + c.square().mean()).backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"step {it}: forward {1e3 * (t1 - t0):.2f} ms, backward {1e3 * (t2 - t1):.2f} ms, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    layer.zero_grad(); feats.grad = None; coors.grad = None
# per-kernel times of the package's own launches in one more step (HIP events; serialises the side stream)
with _ops.phase_timer() as t:
    f, c = layer(feats, coors)
    (f.square().mean() + c.square().mean()).backward()
    torch.cuda.synchronize()
print({k: [round(x, 3) for x in v] for k, v in t.summary().items()})
