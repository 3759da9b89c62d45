"""Edge-pass time per edge for different neighbour counts K (K % 32 == 0 takes the wave-uniform fast path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)          # these tools time / check inference
from egnn_pytorch_amd import EGNN, phase_timer
B, N, dim = 64, 1024, 512
g = torch.Generator().manual_seed(1)
feats = torch.randn(B, N, dim, generator=g).cuda(); coors = torch.randn(B, N, 3, generator=g).cuda()
mask = torch.ones(B, N, dtype=torch.bool).cuda()
for k in (8, 16, 24, 32, 48, 64):
    torch.manual_seed(0)
    layer = EGNN(dim=dim, num_nearest_neighbors=k).cuda().eval()
    for _ in range(2): layer(feats, coors, mask=mask)
    with phase_timer() as pt:
        for _ in range(5): layer(feats, coors, mask=mask)
    s = {n: min(v) for n, v in pt.summary().items()}
    print(f"k={k:3d}: edge {s['edge_fused']:.3f} ms = {s['edge_fused'] * 1e6 / (B * N * k):.3f} ns/edge   knn {s['knn_select']:.3f}  step {sum(s.values()):.3f} ms")
