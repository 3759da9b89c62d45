"""How fast does the vendor library run the SAME matrix-core work as the split-f16 projection GEMM?
The 3-term product is one plain fp16 GEMM with K tripled: [a_hi | a_lo | a_hi] x [w_hi ; w_hi ; w_lo]^T.
Timing only (no epilogue fusion, fp16 or fp32 output) -- a yardstick for linear_hl's MFMA utilisation."""
import torch

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

for (m, n, k, name) in [(65536, 4160, 512, "node_proj"), (65536, 1024, 544, "node_mlp0"), (65536, 512, 1024, "node_mlp1")]:
    a = torch.randn(m, 3 * k, device="cuda", dtype=torch.float16)
    w = torch.randn(n, 3 * k, device="cuda", dtype=torch.float16)
    flops = 2.0 * m * n * 3 * k
    t16 = timeit(lambda: torch.mm(a, w.t()))
    line = f"{name:10s} M={m} N={n} K=3x{k}: fp16-out {t16:.3f} ms = {flops / t16 / 1e9:.0f} TF/s"
    try:
        t32 = timeit(lambda: torch.mm(a, w.t(), out_dtype=torch.float32))
        line += f" | fp32-out {t32:.3f} ms = {flops / t32 / 1e9:.0f} TF/s"
    except Exception as exc:  # noqa: BLE001
        line += f" | fp32-out unsupported ({type(exc).__name__})"
    print(line, flush=True)
