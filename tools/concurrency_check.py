"""Results must not depend on what else the GPU is doing: run chunks of the north-star batch concurrently on two HIP
streams (edge passes next to GEMMs of the other chunk) and compare with the same chunks run one after the other."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)          # these tools time / check inference
from egnn_pytorch_amd import EGNN

torch.manual_seed(0)
layer = EGNN(dim=512, num_nearest_neighbors=32).cuda().eval()
g = torch.Generator().manual_seed(1)
B, N = 64, 1024
feats = torch.randn(B, N, 512, generator=g).cuda(); coors = torch.randn(B, N, 3, generator=g).cuda()
mask = torch.ones(B, N, dtype=torch.bool).cuda()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
cs = B // 4
chunk = lambda c: (feats[c * cs:(c + 1) * cs], coors[c * cs:(c + 1) * cs], mask[c * cs:(c + 1) * cs])
seq = [layer(f, c, mask=m) for f, c, m in map(chunk, range(4))]
torch.cuda.synchronize()
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    main = torch.cuda.current_stream()
    for s in streams: s.wait_stream(main)
    outs = []
    for c in range(4):
        with torch.cuda.stream(streams[c % 2]):
            f, x, m = chunk(c)
            outs.append(layer(f, x, mask=m))
    for s in streams: main.wait_stream(s)
    torch.cuda.synchronize()
    dn = sum(int((o[0] != r[0]).any(-1).sum()) for o, r in zip(outs, seq))
    dc = sum(int((o[1] != r[1]).any(-1).sum()) for o, r in zip(outs, seq))
    bad += dn + dc
    print(f"iter {it}: nodes with differing feats {dn}, coors {dc}")
print("TOTAL_DIFF", bad)
# detail of the last iteration
for ci, (o, r) in enumerate(zip(outs, seq)):
    d = (o[1] != r[1]).any(-1).nonzero()
    for bb, nn in d[:12].tolist():
        print("chunk", ci, "graph", bb, "node", nn, "got", o[1][bb, nn].tolist(), "ref", r[1][bb, nn].tolist(),
              "in", chunk(ci)[1][bb, nn].tolist())
