"""What re-laying the weights costs after an optimizer step (the packed images are cached per parameter version):
    python tools/pack_cost_probe.py"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egnn_pytorch_amd import EGNN

for kw in (dict(dim=512, num_nearest_neighbors=32), dict(dim=128, num_nearest_neighbors=32, norm_feats=True)):
    layer = EGNN(**kw).cuda()
    feats = torch.randn(8, 256, kw["dim"], device="cuda", requires_grad=True)
    coors = torch.randn(8, 256, 3, device="cuda", requires_grad=True)
    opt = torch.optim.SGD(layer.parameters(), lr=1e-6)
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        layer.packed_weights()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        f, c = layer(feats, coors)
        (f.square().mean() + c.square().mean()).backward()
        opt.step(); opt.zero_grad()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if it >= 3:
            print(f"{kw}: pack {1e3 * (t1 - t0):.2f} ms, forward + backward + step (8 x 256 nodes) {1e3 * (t2 - t1):.2f} ms", flush=True)
