import torch, sys
sys.path.insert(0, ".")
torch.set_grad_enabled(False)
from egnn_pytorch_amd import EGNN, _ops, layer as L
for (b, n, k, use_mask) in [(1, 64, 32, False), (3, 40, 5, True), (2, 100, 8, True)]:
    g = torch.Generator().manual_seed(1)
    torch.manual_seed(3)
    layer = EGNN(dim=32, num_nearest_neighbors=k).cuda().eval()
    for p in layer.parameters():
        p.mul_(40.0)
    feats = torch.randn(b, n, 32, generator=g).cuda(); coors = torch.randn(b, n, 3, generator=g).cuda()
    mask = (torch.arange(n)[None] < torch.randint(n // 2, n + 1, (b, 1), generator=g)).cuda() if use_mask else None
    L._SLOT_PREP = True
    a = layer(feats, coors, mask=mask)
    L._SLOT_PREP = False
    c = layer(feats, coors, mask=mask)
    df, dc = (a[0] - c[0]).abs(), (a[1] - c[1]).abs()
    print(b, n, k, use_mask, "feats diff", float(df.max()), "coors diff", float(dc.max()), "bad nodes feats", int((df.amax(-1) > 0).sum()), "coors", int((dc.amax(-1) > 0).sum()))
    if float(dc.max()) > 0:
        bad = (dc.amax(-1) > 0).nonzero()[:10]
        print(bad.tolist())
