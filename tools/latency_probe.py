"""Small-input latency: eager launches vs hipGraph replay (egnn_pytorch_amd.graphed)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)          # these tools time / check inference
from egnn_pytorch_amd import EGNN_Network, graphed

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for (b, n, dim, depth) in [(1, 256, 128, 6), (1, 1024, 128, 6), (8, 512, 256, 6), (64, 1024, 256, 6)]:
    torch.manual_seed(0)
    net = EGNN_Network(depth=depth, dim=dim, num_nearest_neighbors=32).cuda().eval()
    f, c = torch.randn(b, n, dim).cuda(), torch.randn(b, n, 3).cuda()
    m = torch.ones(b, n, dtype=torch.bool).cuda()
    run = graphed(net, f, c, mask=m)
    print(f"B={b} N={n} dim={dim} depth={depth}: eager {timeit(lambda: net(f, c, mask=m)):.3f} ms   graph {timeit(lambda: run(f, c, mask=m)):.3f} ms")
