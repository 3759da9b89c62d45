import os, sys
sys.path.insert(0, os.getcwd())
import torch
from egnn_pytorch_amd import EGNN, _ops
torch.manual_seed(0)
dev = torch.device("cuda", 0)
B, N, D = 16, 1024, 512
layer = EGNN(dim=D, num_nearest_neighbors=32).to(dev)
f = torch.randn(B, N, D, device=dev, requires_grad=True)
c = torch.randn(B, N, 3, device=dev, requires_grad=True)
g = torch.Generator().manual_seed(2000)
lens = torch.randint(N // 2, N + 1, (B,), generator=g)
mask = (torch.arange(N)[None, :] < lens[:, None]).to(dev)
orig = _ops.edge_bwd_pass
def spy(w, proj, idx32, gu16, gu_scale, scal, ent, b, n, k, by_dest, **kw):
    gu = gu16 if torch.is_tensor(gu16) else gu16[0]
    g2 = gu.reshape(-1, gu.shape[-1]).float()
    zero_rows = (g2 == 0).all(dim=1)
    print("by_dest" if by_dest else "by_src ", "gU", tuple(gu.shape), gu.dtype, "rows all-zero:", float(zero_rows.float().mean()),
          "expected masked edges:", float(1 - (mask.float().mean())))
    e = ent.clone()
    dead32 = zero_rows[e.clamp(min=0).long()] | (e < 0)
    print("   32-entry groups of the list all dead:", float(dead32.view(-1, 32).all(dim=1).float().mean()), " entries:", e.numel())
    return orig(w, proj, idx32, gu16, gu_scale, scal, ent, b, n, k, by_dest, **kw)
_ops.edge_bwd_pass = spy
import egnn_pytorch_amd.autograd as A
out = layer(f, c, mask=mask)
(out[0].square().mean() + out[1].square().mean()).backward()
torch.cuda.synchronize()
