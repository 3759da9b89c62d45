import sys, torch
sys.path.insert(0, "/root/repo")
from egnn_pytorch_amd import EGNN, _ops
for dim in (128, 256, 512):
    torch.manual_seed(0)
    layer = EGNN(dim=dim, num_nearest_neighbors=32).cuda()
    feats = torch.randn(64, 1024, dim, device="cuda", requires_grad=True)
    coors = torch.randn(64, 1024, 3, device="cuda", requires_grad=True)
    for it in range(3):
        f, c = layer(feats, coors); (f.square().mean() + c.square().mean()).backward()
        layer.zero_grad(); feats.grad = None; coors.grad = None
    with _ops.phase_timer() as t:
        f, c = layer(feats, coors); (f.square().mean() + c.square().mean()).backward(); torch.cuda.synchronize()
    s = t.summary()
    print(dim, {k: round(sum(v), 3) for k, v in s.items() if k.startswith(("edge_bwd", "edge_fused", "edge_tail", "rows_g", "split_sc"))}, flush=True)
