import sys, time, torch
sys.path.insert(0, "/root/repo")
from egnn_pytorch_amd import EGNN
torch.manual_seed(0)
for kw in (dict(dim=512, num_nearest_neighbors=32, edge_dim=8), dict(dim=512, num_nearest_neighbors=32, edge_dim=4)):
    layer = EGNN(**kw).cuda()
    b, n = 16, 1024
    f = torch.randn(b, n, 512, device="cuda", requires_grad=True); c = torch.randn(b, n, 3, device="cuda", requires_grad=True)
    e = torch.randn(b, n, n, kw["edge_dim"], device="cuda")
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = layer(f, c, e)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        (o[0].square().mean() + o[1].square().mean()).backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        layer.zero_grad(); f.grad = None; c.grad = None
    print(kw, f"B=16: forward {1e3*(t1-t0):.2f} ms, backward {1e3*(t2-t1):.2f} ms", flush=True)
