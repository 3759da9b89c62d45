"""A/B of the two edge kernels in ONE process on one box: the general kernel (egnn_edge_args.algo = 1, csrc/edge_fused.hip) against the
persistent wave-per-node kernel (algo = 0, csrc/edge_pw.hip) on the k-NN shapes of BASELINE.json.  Prints, per shape, every kernel's
min / mean over the repetitions (HIP events on the launch stream), the step total, and whether the outputs are bit-identical.

    python tools/pw_probe.py [shapes=ns,c3,c5,ns_ragged,k64] [reps=10]
"""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from egnn_pytorch_amd import EGNN, phase_timer  # noqa: E402
from egnn_pytorch_amd import layer as L  # noqa: E402

torch.set_grad_enabled(False)
opts = dict(a.split("=", 1) for a in sys.argv[1:])
shapes = opts.get("shapes", "ns,c3,c5,ns_ragged,k64").split(",")
reps = int(opts.get("reps", "10"))


def build(shape):
    torch.manual_seed(0)
    kw = dict(num_nearest_neighbors=32)
    b, n = 64, 1024
    ragged = False
    if shape in ("ns", "ns_ragged"):
        d = 512
        ragged = shape == "ns_ragged"
    elif shape == "c3":
        d, kw = 128, dict(num_nearest_neighbors=32, norm_feats=True)
    elif shape == "c5":
        d, kw = 256, dict(num_nearest_neighbors=32, norm_feats=True, norm_coors=True)
    elif shape == "k64":
        d, kw, b = 256, dict(num_nearest_neighbors=64, soft_edges=True, m_pool_method="mean", coor_weights_clamp_value=2.0), 16
    else:
        raise SystemExit("unknown shape " + shape)
    layer = EGNN(dim=d, **kw)
    for m in layer.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.xavier_normal_(m.weight)
    layer = layer.cuda().eval()
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(b, n, d, generator=g).cuda()
    coors = torch.randn(b, n, 3, generator=g).cuda()
    if ragged:
        lens = torch.randint(n // 2, n + 1, (b,), generator=g)
        mask = (torch.arange(n)[None, :] < lens[:, None]).cuda()
    else:
        mask = torch.ones(b, n, dtype=torch.bool).cuda()
    return layer, feats, coors, mask


for shape in shapes:
    layer, feats, coors, mask = build(shape)
    outs = {}
    for algo in (1, 0, 1, 0):
        L._EDGE_ALGO = algo
        for _ in range(3):
            out = layer(feats, coors, mask=mask)
        with phase_timer() as pt:
            for _ in range(reps):
                out = layer(feats, coors, mask=mask)
        s = pt.summary()
        res = {k: [round(min(v), 4), round(sum(v) / len(v), 4)] for k, v in s.items()}
        res["step_sum_of_mins"] = round(sum(min(v) for v in s.values()), 4)
        outs[algo] = [o.clone() for o in out]
        print(f"{shape:10s} algo={algo} {json.dumps(res)}", flush=True)
    same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    diff = max(float((a - b).abs().max()) for a, b in zip(outs[0], outs[1]))
    print(f"{shape:10s} bit-identical={same} max|diff|={diff:.3e} finite={all(bool(torch.isfinite(o).all()) for o in outs[0])}", flush=True)
L._EDGE_ALGO = 0
