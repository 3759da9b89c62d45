"""Repeat the multi-round stress configuration and report where repeated runs differ (debugging aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import egnn_oracle as O
from egnn_pytorch_amd import EGNN

kwargs = dict(dim=32)
cfg = O.EGNNConfig(**kwargs)
params = O.random_params(cfg, seed=17)
params["edge_mlp.3.weight"] = params["edge_mlp.3.weight"] * np.float32(0.1)
params["coors_mlp.3.weight"] = params["coors_mlp.3.weight"] * np.float32(0.05)
rng = np.random.default_rng(12345)
b, n = 6, int(sys.argv[1]) if len(sys.argv) > 1 else 600
feats = rng.standard_normal((b, n, 32)).astype(np.float32)
coors = rng.standard_normal((b, n, 3)).astype(np.float32)
mask = np.arange(n)[None, :] < rng.integers(n // 2, n + 1, size=b)[:, None]
net = EGNN(**kwargs)
net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
net = net.cuda().eval()
fd, cd, md = torch.from_numpy(feats).cuda(), torch.from_numpy(coors).cuda(), torch.from_numpy(mask).cuda()
first = None
for it in range(20):
    node, co = net(fd, cd, mask=md)
    if first is None:
        first = (node.clone(), co.clone())
        continue
    dn = (node != first[0]).nonzero()
    dc = (co != first[1]).nonzero()
    if len(dn) or len(dc):
        print(f"iter {it}: node diffs {len(dn)} coors diffs {len(dc)}", dc[:6].tolist(),
              (co - first[1]).abs().max().item())
rn, rc = O.egnn_forward(cfg, params, feats, coors, None, mask, None)
print("max err vs oracle: node", np.abs(first[0].cpu().numpy() - rn).max(), "coors", np.abs(first[1].cpu().numpy() - rc).max())
