"""Training step (forward + backward) of the EGNN_Network configurations of BASELINE.json (c3, c5 shard):
    python tools/net_train_probe.py"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egnn_pytorch_amd import EGNN_Network

for name, kw, b, n in (("c3_network", dict(depth=3, dim=128, num_nearest_neighbors=32), 64, 1024),
                       ("c5_shard", dict(depth=6, dim=256, num_nearest_neighbors=32, norm_coors=True), 64, 1024)):
    torch.manual_seed(0)
    net = EGNN_Network(**kw).cuda()
    feats = torch.randn(b, n, kw["dim"], device="cuda", requires_grad=True)
    coors = torch.randn(b, n, 3, device="cuda", requires_grad=True)
    mask = torch.ones(b, n, dtype=torch.bool, device="cuda")
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f, c = net(feats, coors, mask=mask)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        (f.square().mean() + c.square().mean()).backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        net.zero_grad(); feats.grad = None; coors.grad = None
    print(f"{name}: forward {1e3 * (t1 - t0):.2f} ms, backward {1e3 * (t2 - t1):.2f} ms, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    if os.environ.get("EGNN_PROBE_PHASES"):
        from egnn_pytorch_amd import _ops
        with _ops.phase_timer() as t:
            f, c = net(feats, coors, mask=mask)
            (f.square().mean() + c.square().mean()).backward()
            torch.cuda.synchronize()
        print({k: round(sum(v), 3) for k, v in t.summary().items()}, flush=True)
        net.zero_grad(); feats.grad = None; coors.grad = None
