"""Debug helper: run one oracle-vs-HIP case under several edge-kernel build variants and report where they differ."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASE = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from oracle import egnn_oracle as O
from egnn_pytorch_amd import EGNN
kwargs = dict(dim=32); b, n = 6, 600
rng = np.random.default_rng(abs(hash("dense_n600_multi_round")) %% (2 ** 31))
rng = np.random.default_rng(12345)
cfg = O.EGNNConfig(**kwargs); params = O.random_params(cfg, seed=17)
params["edge_mlp.3.weight"] = params["edge_mlp.3.weight"] * np.float32(0.1)
params["coors_mlp.3.weight"] = params["coors_mlp.3.weight"] * np.float32(0.05)
feats = rng.standard_normal((b, n, 32)).astype(np.float32); coors = rng.standard_normal((b, n, 3)).astype(np.float32)
lens = rng.integers(n // 2, n + 1, size=b); mask = np.arange(n)[None, :] < lens[:, None]
rn, rc = O.egnn_forward(cfg, params, feats, coors, None, mask, None)
net = EGNN(**kwargs); net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}); net = net.cuda().eval()
outs = []
for rep in range(40):
    node, co = net(torch.from_numpy(feats).cuda(), torch.from_numpy(coors).cuda(), mask=torch.from_numpy(mask).cuda())
    outs.append((node.cpu().numpy(), co.cpu().numpy()))
dn = np.abs(outs[0][0] - rn).max(-1).ravel(); dc = np.abs(outs[0][1] - rc).max(-1).ravel()
print("len", lens, "max feats diff %%.3e at node %%d | max coors diff %%.3e at node %%d" %% (dn.max(), dn.argmax(), dc.max(), dc.argmax()),
      "bad nodes", np.where(dc > 1e-4)[0][:10], "bad runs", sum(int(np.abs(o[1] - rc).max() > 1e-4 or np.abs(o[0] - rn).max() > 1e-4) for o in outs), "of", len(outs), "repeatable", all(np.array_equal(outs[0][1], o[1]) for o in outs))
''' % ROOT
from tools.edge_tune import build
for spec in sys.argv[1:]:
    defs = dict(kv.split("=") for kv in spec.split(",") if kv)
    lib = build("dbg_" + spec.replace("=", "").replace(",", "_"), defs)
    r = subprocess.run([sys.executable, "-c", CASE], env=dict(os.environ, EGNN_HIP_LIB=lib), capture_output=True, text=True)
    print(f"{spec:40s}", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
