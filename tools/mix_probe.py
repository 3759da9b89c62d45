"""Experiment: one launch whose workgroup slots alternate between edge groups and projection-GEMM tiles (tools/ubench/mix_probe.hip).
Times, at the north-star shape: the edge pass alone, the GEMM alone (both through the dispatcher kernel), both in one launch
back to back, and both interleaved."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.set_grad_enabled(False)          # these tools time / check inference
from egnn_pytorch_amd import EGNN, _abi, _ops

csrc = os.path.join(ROOT, "egnn_pytorch_amd", "csrc")
lib_path = "/tmp/libmix.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-shared",
                os.path.join(ROOT, "tools", "ubench", "mix_probe.hip"), os.path.join(csrc, "node_ops.hip"), "-o", lib_path], check=True)
mix = ctypes.CDLL(lib_path)
vp = ctypes.c_void_p
mix.egnn_mix_probe.argtypes = [ctypes.POINTER(_abi.EdgeArgs), vp, vp, vp, vp, ctypes.c_float, vp, vp, ctypes.c_int64, ctypes.c_int64,
                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
mix.egnn_mix_probe.restype = ctypes.c_int

torch.manual_seed(0)
layer = EGNN(dim=512, num_nearest_neighbors=32).cuda().eval()
g = torch.Generator().manual_seed(1)
B, N, dim, k = 64, 1024, 512, 32
feats = torch.randn(B, N, dim, generator=g).cuda(); coors = torch.randn(B, N, 3, generator=g).cuda()
mask = torch.ones(B, N, dtype=torch.bool).cuda()
ref_node, ref_co = layer(feats, coors, mask=mask)

# ---- the layer's prelude by hand (egnn_pytorch_amd/layer.py::_forward_hip)
w = layer.packed_weights()
hp = w["Hp"]
feats2d = feats.view(B * N, dim)
idx, rank = _ops.knn_select(coors, mask, None, k)
node_in, feats_hl = _ops.node_prep_hl(feats2d, None, w.get("gamma"), w.get("beta"), 1e-5, layer.m_dim, with_raw=True)
proj = _ops.linear_hl(feats_hl, w["Wcat_split"], 2 * hp, w["bcat"], split_cols=hp)
order = _ops.spatial_order(coors)
a = _abi.EdgeArgs()
a.B, a.N, a.K, a.dim, a.m_dim = B, N, k, dim, layer.m_dim
a.H, a.Hp, a.fourier, a.edge_dim, a.S, a.pi_split = w["H"], hp, 0, 0, w["S"], 1
a.Pi, a.Pj, a.ldp = proj.data_ptr(), proj.data_ptr() + 4 * hp, 2 * hp
a.Wst, a.W2h, a.b2 = w["Wst"].data_ptr(), w["W2h"].data_ptr(), w["b2"].data_ptr()
a.ws_inv_scale, a.wst_terms, a.w2_inv_scale = w["ws_inv_scale"], w["Wst"].shape[1], w["w2_inv_scale"]
a.W3h, a.b3, a.W4, a.b4 = (w[x].data_ptr() for x in ("W3h", "b3", "W4", "b4"))
a.w3_inv_scale = w["w3_inv_scale"]
coors_out = torch.empty_like(coors)
a.coors_out, a.coors, a.coor_dim = coors_out.data_ptr(), coors.data_ptr(), 3
a.mask = mask.view(torch.uint8).data_ptr()
a.idx, a.rank, a.order = idx.data_ptr(), rank.data_ptr(), order.data_ptr()
a.valid_radius, a.clamp, a.pool_mean = 3.0e38, -1.0, 0
a.node_hi, a.node_lo, a.node_kp = node_in.hi.data_ptr(), node_in.lo.data_ptr(), node_in.kp
whi, wlo, inv, w_rows = w["Wcat_split"]
proj2 = torch.empty_like(proj)                       # the GEMM of "another chunk": same operands, its own output

def launch(which, mode):
    rc = mix.egnn_mix_probe(ctypes.byref(a), feats_hl.hi.data_ptr(), feats_hl.lo.data_ptr(), whi.data_ptr(), wlo.data_ptr(),
                            float(inv), w["bcat"].data_ptr(), proj2.data_ptr(), 2 * hp, B * N, 2 * hp, feats_hl.kp, hp, mode, which,
                            torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

launch(3, 0); torch.cuda.synchronize()
print("mixed launch correct: coords", torch.equal(coors_out, ref_co), " P", torch.equal(proj2, proj))
te, tg = timeit(lambda: launch(1, 1)), timeit(lambda: launch(2, 1))
print(f"edge alone {te:.3f} ms   gemm (128x128 tiles) alone {tg:.3f} ms   sum {te + tg:.3f} ms")
print(f"one launch, GEMM tiles then edge groups : {timeit(lambda: launch(3, 1)):.3f} ms")
print(f"one launch, interleaved slots            : {timeit(lambda: launch(3, 0)):.3f} ms")
