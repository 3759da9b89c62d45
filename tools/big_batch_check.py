"""B = 256 graphs of the north-star shape in one call (P = 4.4 GB: past 32-bit byte offsets) == the same graphs in chunks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)          # these tools time / check inference
from egnn_pytorch_amd import EGNN
torch.manual_seed(0)
layer = EGNN(dim=512, num_nearest_neighbors=32).cuda().eval()
g = torch.Generator().manual_seed(3)
B, N = 256, 1024
feats = torch.randn(B, N, 512, generator=g).cuda(); coors = torch.randn(B, N, 3, generator=g).cuda()
mask = (torch.arange(N)[None] < torch.randint(N // 2, N + 1, (B, 1), generator=g)).cuda()
big = layer(feats, coors, mask=mask)
ok = True
for c in range(0, B, 64):
    part = layer(feats[c:c + 64], coors[c:c + 64], mask=mask[c:c + 64])
    ok &= torch.equal(part[0], big[0][c:c + 64]) and torch.equal(part[1], big[1][c:c + 64])
torch.cuda.synchronize()
print("B=256 one call == 4 x B=64:", ok, " finite:", bool(torch.isfinite(big[0]).all() and torch.isfinite(big[1]).all()),
      " peak mem GB:", round(torch.cuda.max_memory_allocated() / 2**30, 2))
