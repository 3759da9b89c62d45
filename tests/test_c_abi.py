"""The C ABI without torch: tests/c_abi/layer_forward_test.c (plain C: include/egnn_hip.h + the HIP runtime API) runs golden
cases of the reference through egnn_pack_weights_host / egnn_workspace_bytes / egnn_layer_forward_f32, and the same
single-call entry is compared with the Python module's launch sequence (bit-identical: it chains the same kernels)."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from tests._util import layer_kwargs, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _block(arr, dtype):
    if arr is None:
        return struct.pack("<q", 0)
    a = np.ascontiguousarray(arr, dtype=dtype)
    return struct.pack("<q", a.size) + a.tobytes()


def _write_case(path, name):
    """tests/golden/<name>.npz -> the flat binary the C program reads (layout documented in layer_forward_test.c)."""
    from egnn_pytorch_amd import EGNN, _abi
    meta, params, d = load_golden(name)
    assert meta["kind"] == "layer"
    layer = EGNN(**layer_kwargs(meta))
    desc = _abi.layer_desc(layer)
    feats, coors = d["feats"], d["coors"]
    b, n = feats.shape[:2]
    adj = d.get("adj_mat")
    k = layer.num_nearest_neighbors
    if adj is not None and layer.only_sparse_neighbors:
        k = int(adj.astype(np.float32).sum(-1).max())                          # egnn_pytorch.py:249
    if not (layer.num_nearest_neighbors > 0 or layer.only_sparse_neighbors):
        k = n
    adj_kind = 0 if adj is None else (1 if adj.ndim == 2 else 2)
    with open(path, "wb") as f:
        f.write(b"EGNNCASE")
        f.write(bytes(desc))
        f.write(struct.pack("<7i", b, n, k, coors.shape[-1], adj_kind, 0, 0))
        for field in _abi.PARAM_FIELDS:
            key = next((kk for kk in params if kk.replace(".", "_") == field), None)
            f.write(_block(params[key] if key else None, np.float32))
        f.write(_block(feats, np.float32))
        f.write(_block(coors, np.float32))
        f.write(_block(d.get("edges"), np.float32))
        f.write(_block(d.get("mask"), np.uint8))
        f.write(_block(adj, np.uint8))
        f.write(_block(d["node_out"], np.float32))
        f.write(_block(d["coors_out"], np.float32))


@pytest.fixture(scope="module")
def c_program(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("c_abi") / "layer_forward_test")
    libdir = os.path.join(ROOT, "egnn_pytorch_amd")
    cmd = ["gcc", "-O1", "-std=c11", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "c_abi", "layer_forward_test.c"), "-L", libdir, "-legnn_hip", "-L", "/opt/rocm/lib",
           "-lamdhip64", "-lm", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


@pytest.mark.parametrize("name", ["knn8_mask", "all_flags", "c1_dense_dim32", "sparse_chain_edges_mask", "knn32_dim128_mask",
                                  "no_feats_update", "knn8_coor_dim5_normcoors_mask"])
def test_c_program_reproduces_golden(c_program, tmp_path, name):
    case = str(tmp_path / (name + ".bin"))
    _write_case(case, name)
    r = subprocess.run([c_program, case], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


@pytest.mark.parametrize("kw,n,flags", [(dict(dim=64, num_nearest_neighbors=16), 200, dict(mask=True)),
                                        (dict(dim=32, edge_dim=3, fourier_features=2, soft_edges=True, norm_coors=True,
                                              norm_feats=True, m_pool_method="mean"), 40, dict(mask=True, edges=True)),
                                        (dict(dim=48, only_sparse_neighbors=True, edge_dim=2), 64, dict(mask=True, edges=True, adj=True)),
                                        (dict(dim=32, m_dim=40, num_nearest_neighbors=16, soft_edges=True), 80, dict(mask=True)),
                                        (dict(dim=128, num_nearest_neighbors=32, norm_feats=True), 96, dict(mask=True)),
                                        (dict(dim=24, num_nearest_neighbors=8, update_feats=False), 70, dict()),
                                        (dict(dim=24, num_nearest_neighbors=8, update_coors=False, valid_radius=1.5), 70, dict(mask=True)),
                                        (dict(dim=40, num_nearest_neighbors=6, coor_weights_clamp_value=0.5), 64, dict(adj=True))])
def test_c_layer_forward_matches_module(kw, n, flags):
    """Three ways to one forward, bit-identical outputs: the Python launch sequence (`EGNN._forward_hip`: torch packer, a dozen separate
    C-ABI calls), egnn_layer_forward_f32 on the C host packer's blob (`_ops.forward_c`: what a binding without torch does), and the
    module's own inference path (`EGNN._forward_c`: egnn_layer_forward_opts_f32 with the side stream, one call per forward)."""
    from egnn_pytorch_amd import EGNN, _ops, layer as L
    torch.manual_seed(3)
    layer = EGNN(**kw).cuda().eval()
    with torch.no_grad():
        for p in layer.parameters():
            p.mul_(100.0)
    g = torch.Generator().manual_seed(1)
    b = 3
    feats = torch.randn(b, n, kw["dim"], generator=g).cuda()
    coors = torch.randn(b, n, 3, generator=g).cuda()
    mask = (torch.arange(n)[None] < torch.tensor([[n], [n - 7], [n // 2 + 3]])).cuda() if flags.get("mask") else None
    edges = torch.randn(b, n, n, kw.get("edge_dim", 0), generator=g).cuda() if flags.get("edges") else None
    adj = None
    if flags.get("adj"):
        i = torch.arange(n)
        adj = ((i[:, None] - i[None, :]).abs() <= 2).cuda()
    with torch.no_grad():
        want = layer._forward_hip(feats, coors, edges, mask, adj)[:2]          # the Python launch sequence
        assert L._C_FORWARD
        calls = []
        orig = L.EGNN._forward_c
        L.EGNN._forward_c = lambda self, *a: calls.append(1) or orig(self, *a)
        try:
            mod = layer(feats, coors, edges, mask, adj)                         # the module: one C call
        finally:
            L.EGNN._forward_c = orig
        assert calls, "the module's inference forward did not take the one-call path"
    got = _ops.forward_c(layer, feats, coors, edges, mask, adj)
    torch.cuda.synchronize()
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert torch.equal(mod[0], want[0]) and torch.equal(mod[1], want[1])


def test_one_call_forward_in_a_network_reuses_the_first_layers_order():
    """EGNN_Network on the one-call path: layer 0 writes the Morton order into a buffer of the caller, the next layers read it as a hint
    (egnn_forward_opts.order / order_is_hint) -- same outputs, bit for bit, as the Python launch sequence (EGNN_C_FORWARD=0)."""
    from egnn_pytorch_amd import EGNN_Network, layer as L
    torch.manual_seed(5)
    net = EGNN_Network(depth=3, dim=32, num_nearest_neighbors=8, norm_coors=True).cuda().eval()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(30.0)
    g = torch.Generator().manual_seed(2)
    feats, coors = torch.randn(2, 128, 32, generator=g).cuda(), torch.randn(2, 128, 3, generator=g).cuda()
    mask = (torch.arange(128)[None] < torch.tensor([[128], [77]])).cuda()
    hints = []
    orig = L.EGNN._forward_c

    def spy(self, feats, coors, edges, mask, adj_mat, order_hint):
        hints.append(order_hint)
        return orig(self, feats, coors, edges, mask, adj_mat, order_hint)

    with torch.no_grad():
        L.EGNN._forward_c = spy
        try:
            fast = net(feats, coors, mask=mask)
        finally:
            L.EGNN._forward_c = orig
        assert len(hints) == 3 and hints[0] is None and hints[1] is not None and hints[2] is hints[1]
        was = L._C_FORWARD
        L._C_FORWARD = False
        try:
            slow = net(feats, coors, mask=mask)
        finally:
            L._C_FORWARD = was
    torch.cuda.synchronize()
    assert torch.equal(fast[0], slow[0]) and torch.equal(fast[1], slow[1])
