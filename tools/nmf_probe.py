"""egnn_node_mlp_fused_f32 alone at the c3 / c5 widths for every library under build_variants/ (tools/variants.py build src=node_mlp_fused ...),
one process, min of 5 x 10 launches; digest of the output."""
import ctypes, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.set_grad_enabled(False)
from egnn_pytorch_amd import _ops, _weights
c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
tags = json.load(open(os.path.join(ROOT, "build_variants", "index.json")))
M = 65536
for dim in (128, 256):
    g = torch.Generator().manual_seed(dim)
    w5 = torch.randn(2 * dim, dim + 16, generator=g) / (dim + 16) ** 0.5
    w6 = torch.randn(dim, 2 * dim, generator=g) / (2 * dim) ** 0.5
    s5 = tuple(t.cuda() if torch.is_tensor(t) else t for t in _weights.split_f16(w5))
    s6 = tuple(t.cuda() if torch.is_tensor(t) else t for t in _weights.split_f16(w6))
    b5, b6 = torch.randn(2 * dim, generator=g).cuda(), torch.randn(dim, generator=g).cuda()
    x = _ops.split_f16(torch.randn(M, dim + 16, generator=g).cuda())
    res = torch.randn(M, dim, generator=g).cuda()
    out = torch.empty(M, dim).cuda()
    status = torch.zeros(4, dtype=torch.int32).cuda()
    img = _ops.node_mlp_fused_image(s5, s6, dim, 16)
    st = torch.cuda.current_stream().cuda_stream
    for tag in tags:
        lib = ctypes.CDLL(os.path.join(ROOT, "build_variants", tag, "libegnn_hip.so"))
        f = lib.egnn_node_mlp_fused_f32
        f.argtypes = [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]
        def run():
            rc = f(x.hi.data_ptr(), x.lo.data_ptr(), img.data_ptr(), s5[2], b5.data_ptr(), s6[2], b6.data_ptr(), res.data_ptr(), out.data_ptr(), M, dim, 16,
                   status.data_ptr(), st)
            assert rc == 0, rc
        for _ in range(3): run()
        torch.cuda.synchronize()
        best = 1e9
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        flops = 3 * 2.0 * M * ((dim + 16) * 2 * dim + 2 * dim * dim)
        print(f"dim {dim:4d} {tag:40s} {best * 1e3:7.1f} us  {flops / best / 1e9 / 2500:.3f} of the f16 MFMA peak  {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:10]}", flush=True)
