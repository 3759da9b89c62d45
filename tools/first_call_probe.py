import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from egnn_pytorch_amd import EGNN, layer as L
dev = torch.device("cuda", 0)
for dim in (16, 64, 128, 512):
    for c in (True, False):
        L._C_FORWARD = c
        ts = []
        for rep in range(6):
            layer = EGNN(dim=dim, num_nearest_neighbors=8, edge_dim=2).to(dev).eval()
            f, x = torch.randn(2, 64, dim, device=dev), torch.randn(2, 64, 3, device=dev)
            e = torch.randn(2, 64, 64, 2, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                layer(f, x, e)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"dim {dim:4d} one_call={c}: first forward of a fresh layer {min(ts[1:]):7.2f} ms (min of 5), first ever {ts[0]:7.2f}")
