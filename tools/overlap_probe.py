"""Probe: chunks of the batch on two HIP streams -- do the VALU-bound edge pass of one chunk and the MFMA-bound GEMMs of
another share the CUs?  (Needs workgroups of both kernels to fit the same LDS / register slots.)"""
import sys, os, time
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

def run_seq():
    return layer(feats, coors, mask=mask)

def run_chunks(nchunk):
    cs = B // nchunk
    main = torch.cuda.current_stream()
    for s in streams: s.wait_stream(main)
    outs = []
    for c in range(nchunk):
        with torch.cuda.stream(streams[c % 2]):
            sl = slice(c * cs, (c + 1) * cs)
            outs.append(layer(feats[sl], coors[sl], mask=mask[sl]))
    for s in streams: main.wait_stream(s)
    return outs

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print("one launch sequence   : %.3f ms" % timeit(run_seq))
for nc in (2, 4, 8, 16):
    print("%2d chunks, 2 streams  : %.3f ms" % (nc, timeit(lambda: run_chunks(nc))))

# staggered: the second stream starts ~0.45 ms (about one projection GEMM) late, so that edge(A) meets proj(B), mlp(A) meets edge(B)
def run_staggered(delay_cycles):
    main = torch.cuda.current_stream()
    for s in streams: s.wait_stream(main)
    h = B // 2
    with torch.cuda.stream(streams[0]):
        o0 = layer(feats[:h], coors[:h], mask=mask[:h])
    with torch.cuda.stream(streams[1]):
        torch.cuda._sleep(delay_cycles)
        o1 = layer(feats[h:], coors[h:], mask=mask[h:])
    for s in streams: main.wait_stream(s)
    return o0, o1

for d in (0, 400_000, 900_000, 1_500_000):
    print("2 halves, stream 2 delayed %7d cycles: %.3f ms" % (d, timeit(lambda: run_staggered(d))))
