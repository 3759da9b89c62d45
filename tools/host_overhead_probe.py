"""Host-side cost of one forward (python + ctypes + allocator), north-star layer: cProfile over N forwards in deferred mode, and the
wall time from entering forward() to the first kernel launch (what a synchronous range check exposes per step)."""
import cProfile, pstats, sys, time, io, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egnn_pytorch_amd import EGNN, _ops
_ops.RANGE_CHECK = "deferred"
torch.set_grad_enabled(False)
layer = EGNN(dim=512, num_nearest_neighbors=32).cuda().eval()
feats, coors = torch.randn(64, 1024, 512).cuda(), torch.randn(64, 1024, 3).cuda()
mask = torch.ones(64, 1024, dtype=torch.bool).cuda()
for _ in range(5):
    layer(feats, coors, mask=mask)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    layer(feats, coors, mask=mask)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue time per forward {1e6 * (t1 - t0) / n:.1f} us; GPU-bound step {1e6 * (t2 - t0) / n:.1f} us")
# first-launch latency: time from forward() entry to the return of the first C-ABI call
import egnn_pytorch_amd._abi as A
lib = A.load()
first = []
orig = lib.egnn_knn_select_f32
def hook(*a):
    first.append(time.perf_counter())
    return orig(*a)
lib.egnn_knn_select_f32 = hook
starts = []
for _ in range(50):
    torch.cuda.synchronize()
    starts.append(time.perf_counter())
    layer(feats, coors, mask=mask)
lib.egnn_knn_select_f32 = orig
d = sorted(1e6 * (f - s) for f, s in zip(first, starts))
print(f"forward() entry -> first kernel launch: median {d[len(d)//2]:.1f} us, min {d[0]:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    layer(feats, coors, mask=mask)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
