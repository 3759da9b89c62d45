"""fp32 library GEMM forms for the node-level gradient products of the backward (M = B*N = 65536 rows, Hp = 2080, dim = 512):
    python tools/ubench/lib_gemm_bwd.py"""
import torch
import torch.nn.functional as F

dev = "cuda"
m, hp, dim = 65536, 2080, 512
gz = torch.randn(m, hp, device=dev)
w = torch.randn(hp, dim, device=dev)
wt = w.t().contiguous()
f = torch.randn(m, dim, device=dev)
gzt = gz.t().contiguous()


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("d feats = gz @ w              (M x Hp)(Hp x dim), w row-major     %.3f ms" % t(lambda: gz @ w))
print("d feats = F.linear(gz, w^T)   w^T (dim x Hp) row-major            %.3f ms" % t(lambda: F.linear(gz, wt)))
print("d feats = (w^T @ gz^T)^T      via transposed views                %.3f ms" % t(lambda: (wt @ gz.t()).t()))
print("d W     = gz^T @ f            (Hp x M)(M x dim), gz^T a view      %.3f ms" % t(lambda: gz.t() @ f))
print("d W     = gz^T(contig) @ f                                        %.3f ms" % t(lambda: gzt @ f))
print("d W     = (f^T @ gz)^T                                            %.3f ms" % t(lambda: (f.t() @ gz).t()))
for s in (2, 4, 8, 16):
    print("d W     = split-K bmm, %2d slabs of rows, summed                    %.3f ms" % (s, t(lambda: torch.bmm(gz.view(s, m // s, hp).transpose(1, 2), f.view(s, m // s, dim)).sum(0))))
    print("d W^T   = split-K bmm (f^T gz), %2d slabs                           %.3f ms" % (s, t(lambda: torch.bmm(f.view(s, m // s, dim).transpose(1, 2), gz.view(s, m // s, hp)).sum(0))))
