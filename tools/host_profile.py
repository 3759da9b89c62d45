"""Where the host's time goes in one `EGNN.forward` call (north-star shape): cProfile over calls issued back to back with the range check
deferred (the host never waits for the device), so the totals are pure launch-path cost; then, with the synchronous check, the wall time
per call against the device time of its kernels.     python tools/host_profile.py [calls=300] [workload=north_star|c1_tiny|c2_dense|c3_network|c5_shard]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from egnn_pytorch_amd import EGNN, EGNN_Network, _ops  # noqa: E402

opts = dict(a.split("=") for a in sys.argv[1:])
calls = int(opts.get("calls", 300))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
kw, B, N = {"north_star": (dict(dim=512, num_nearest_neighbors=32), 64, 1024), "c2_dense": (dict(dim=512), 8, 256),
            "c1_tiny": (dict(dim=32), 1, 16),
            "c3_network": (dict(depth=3, dim=128, num_nearest_neighbors=32), 64, 1024),
            "c5_shard": (dict(depth=6, dim=256, num_nearest_neighbors=32, norm_coors=True), 64, 1024)}[opts.get("workload", "north_star")]
net = "depth" in kw
_layer = (EGNN_Network(**kw) if net else EGNN(**kw)).to(dev).eval()
feats, coors = torch.randn(B, N, kw["dim"], device=dev), torch.randn(B, N, 3, device=dev)
_mask = torch.ones(B, N, dtype=torch.bool, device=dev)
layer = (lambda f, c, mask=None: _layer(f, c, mask=mask))
mask = _mask
with torch.no_grad():
    for _ in range(40):
        layer(feats, coors, mask=mask)
    torch.cuda.synchronize()
    for mode in ("sync", "deferred"):
        _ops.RANGE_CHECK = mode
        for _ in range(10):
            layer(feats, coors, mask=mask)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            layer(feats, coors, mask=mask)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{mode:9s}: host returns after {(t1 - t0) / calls * 1e6:8.1f} us per call; device done after {(t2 - t0) / calls * 1e6:8.1f} us per call "
              f"(range check as the library reports it: {_ops.RANGE_CHECK})")
    _ops.RANGE_CHECK = "deferred"
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(calls):
        layer(feats, coors, mask=mask)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
