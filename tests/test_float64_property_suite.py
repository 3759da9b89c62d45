"""float64 modules compute in float64 (VERDICT r3 missing #2 / next #8).

The reference is dtype-generic and its own tests run in float64 (/root/reference/tests/test_equivariance.py:6-102, atol 1e-6).  A
module converted with .double() runs on the float64 kernels of libegnn_hip.so (include/egnn_hip.h, "The float64 path":
egnn_knn_select_f64, egnn_linear_f64 on v_mfma_f64_16x16x4_f64, egnn_node_prep_f64, egnn_edge_exact_f64).  Checked here:
  * the reference's four property tests as upstream runs them, but at XAVIER-scale weights, where fp32-class arithmetic fails the
    1e-6 bars (with upstream's default init, std 1e-3, any arithmetic passes);
  * the layer and the network against the numpy oracle evaluated in float64 at 1e-10 of the output's scale;
  * the kernels on their own against torch's float64 operators."""
import math
import warnings

import numpy as np
import pytest
import torch

from oracle import egnn_oracle as O

pytestmark = pytest.mark.gpu


def _rotation(g):
    """a proper rotation of R^3 from three random Euler angles (what egnn_pytorch/utils.py::rot builds)"""
    a, b, c = (float(x) * 2.0 * math.pi for x in torch.rand(3, generator=g))
    f64 = torch.float64                         # (built in float64: a float32 rotation is orthogonal to 1e-7 only)
    rz = torch.tensor([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]], dtype=f64)
    ry = torch.tensor([[math.cos(b), 0.0, math.sin(b)], [0.0, 1.0, 0.0], [-math.sin(b), 0.0, math.cos(b)]], dtype=f64)
    rz2 = torch.tensor([[math.cos(c), -math.sin(c), 0.0], [math.sin(c), math.cos(c), 0.0], [0.0, 0.0, 1.0]], dtype=f64)
    return rz @ ry @ rz2


def _xavier_(module, g):
    """weights of every Linear ~ N(0, 1 / fan_in): activations of order one all the way through (upstream's default is std 1e-3)"""
    with torch.no_grad():
        for mod in module.modules():
            if type(mod) is torch.nn.Linear:
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g, dtype=torch.float64) / math.sqrt(mod.weight.shape[1]))


@pytest.mark.parametrize("kwargs,n,edge_dim", [
    (dict(dim=512, edge_dim=4), 16, 4),                                          # test_egnn_equivariance (:8-34)
    (dict(dim=512, edge_dim=1, num_nearest_neighbors=8), 256, 1),                # ..._with_nearest_neighbors (:47-73)
    (dict(dim=512, edge_dim=1, num_nearest_neighbors=8, norm_coors=True), 256, 1),   # ..._with_coord_norm (:76-102)
])
@pytest.mark.parametrize("init", ["default", "xavier"])
def test_reference_float64_assertions_hold(kwargs, n, edge_dim, init):
    from egnn_pytorch_amd import EGNN
    g = torch.Generator().manual_seed(n + edge_dim)
    torch.manual_seed(17)
    layer = EGNN(**kwargs).double()
    if init == "xavier":
        _xavier_(layer, g)
    layer = layer.cuda()
    rot, shift = _rotation(g).cuda(), torch.randn(1, 1, 3, generator=g).double().cuda()
    feats = torch.randn(1, n, 512, generator=g).double().cuda()
    coors = torch.randn(1, n, 3, generator=g).double().cuda()
    edges = torch.randn(1, n, n, edge_dim, generator=g).double().cuda()
    mask = torch.ones(1, n, dtype=torch.bool).cuda()
    swapped = feats.clone()
    swapped[:, 0], swapped[:, 1] = feats[:, 1], feats[:, 0]
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                           # (no "fp32-class arithmetic" warning any more)
        with torch.no_grad():
            f_moved, c_moved = layer(feats, coors @ rot + shift, edges, mask=mask)
            f_plain, c_plain = layer(feats, coors, edges, mask=mask)
            f_swapped, _ = layer(swapped, coors, edges, mask=mask)
    assert f_plain.dtype == torch.float64 and c_plain.dtype == torch.float64
    assert torch.allclose(f_moved, f_plain, atol=1e-6), "type 0 features are invariant"
    assert torch.allclose(c_moved, c_plain @ rot + shift, atol=1e-6), "type 1 features are equivariant"
    assert not torch.allclose(f_moved, f_swapped, atol=1e-6), "the layer must see a permutation of the node features"
    if init == "xavier":
        # far beyond what 22-bit products give at this scale: the invariance holds to float64 rounding
        assert float((f_moved - f_plain).abs().max()) <= 1e-10 * max(1.0, float(f_plain.abs().max()))


def test_five_dimensional_coordinates_run_in_float64():
    """test_higher_dimension (:36-45): runs, shapes and dtype preserved."""
    from egnn_pytorch_amd import EGNN
    layer = EGNN(dim=512, edge_dim=4).double().cuda()
    g = torch.Generator().manual_seed(5)
    feats, coors = torch.randn(1, 16, 512, generator=g).double().cuda(), torch.randn(1, 16, 5, generator=g).double().cuda()
    edges = torch.randn(1, 16, 16, 4, generator=g).double().cuda()
    with torch.no_grad():
        f, c = layer(feats, coors, edges, mask=torch.ones(1, 16, dtype=torch.bool).cuda())
    assert f.shape == feats.shape and c.shape == coors.shape and f.dtype == torch.float64 and torch.isfinite(f).all()


def test_float64_inputs_of_a_float32_module_are_told_about_the_precision_once():
    """float64 INPUTS to an fp32 module (the reference raises a dtype mismatch there) are converted at the boundary, with a warning;
    a float64 MODULE does not warn: it computes in float64."""
    from egnn_pytorch_amd import EGNN, layer as L
    L._FP64_WARNED = False
    mod = EGNN(dim=16, num_nearest_neighbors=4).cuda()
    f, c = torch.randn(1, 12, 16).double().cuda(), torch.randn(1, 12, 3).double().cuda()
    with torch.no_grad():
        with pytest.warns(RuntimeWarning, match="fp32-class"):
            mod(f, c)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            mod(f, c)                                                            # once per process
            mod.double()(f, c)                                                   # the float64 kernels: nothing to warn about


def _dev(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


LAYER_CASES = [
    ("knn_fourier_edges", dict(dim=48, num_nearest_neighbors=11, fourier_features=2, edge_dim=3), 3, 77, dict(mask=True, edges=True)),
    ("dense_mask", dict(dim=40, edge_dim=1), 2, 37, dict(mask=True, edges=True)),
    ("normcoors_normfeats_clamp", dict(dim=64, num_nearest_neighbors=32, norm_feats=True, norm_coors=True, coor_weights_clamp_value=0.3),
     2, 96, dict(mask=True)),
    ("soft_edges_mean_radius", dict(dim=32, num_nearest_neighbors=8, soft_edges=True, m_pool_method="mean", valid_radius=1.5), 2, 50,
     dict(mask=True)),
    ("five_dims_m32", dict(dim=32, m_dim=32, num_nearest_neighbors=6), 2, 40, dict(cdim=5)),
    ("eleven_dims_normcoors", dict(dim=32, num_nearest_neighbors=8, norm_coors=True), 2, 40, dict(cdim=11, mask=True)),
    ("sparse_adjacency", dict(dim=32, edge_dim=2, only_sparse_neighbors=True), 2, 48, dict(mask=True, edges=True, adj=True)),
    ("no_coors_update", dict(dim=32, update_coors=False, num_nearest_neighbors=5), 1, 30, dict()),
    ("ns_width", dict(dim=512, num_nearest_neighbors=32), 1, 128, dict(mask=True)),
]


@pytest.mark.parametrize("name,kwargs,b,n,opt", LAYER_CASES, ids=[c[0] for c in LAYER_CASES])
def test_float64_layer_matches_the_float64_oracle(name, kwargs, b, n, opt):
    from egnn_pytorch_amd import EGNN
    cfg = O.EGNNConfig(**kwargs)
    params = O.random_params(cfg, seed=11, dtype=np.float64)
    rng = np.random.default_rng(len(name))
    cdim = opt.get("cdim", 3)
    feats, coors = rng.standard_normal((b, n, kwargs["dim"])), rng.standard_normal((b, n, cdim))
    edges = rng.standard_normal((b, n, n, kwargs["edge_dim"])) if opt.get("edges") else None
    mask = (np.arange(n)[None] < rng.integers(n // 2, n + 1, (b, 1))) if opt.get("mask") else None
    i = np.arange(n)
    adj = (np.abs(i[:, None] - i[None, :]) <= 1) if opt.get("adj") else None
    want_n, want_c = O.egnn_forward(cfg, params, feats, coors, edges=edges, mask=mask, adj_mat=adj)
    assert want_n.dtype == np.float64
    net = EGNN(**kwargs).double()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        node, co = net(_dev(feats), _dev(coors), _dev(edges), _dev(mask), _dev(adj))
    assert node.dtype == torch.float64 and co.dtype == torch.float64
    for got, want, what in ((node, want_n, "feats"), (co, want_c, "coors")):
        scale = max(1.0, float(np.abs(want).max()))
        err = float(np.abs(got.cpu().numpy() - want).max())
        assert err <= 1e-10 * scale, (name, what, err, scale)


def test_float64_network_matches_the_float64_oracle():
    """EGNN_Network in float64: embeddings, N-degree adjacency, the induced-set attention block (ATen in float64) and three layers."""
    from egnn_pytorch_amd import EGNN_Network
    kw = dict(depth=3, dim=32, num_tokens=12, num_edge_tokens=5, edge_dim=4, num_adj_degrees=2, adj_dim=3, num_nearest_neighbors=8,
              global_linear_attn_every=2, global_linear_attn_heads=2, global_linear_attn_dim_head=8, norm_coors=True)
    torch.manual_seed(9)
    net = EGNN_Network(**kw).double()
    g = torch.Generator().manual_seed(2)
    _xavier_(net, g)
    net = net.cuda().eval()
    b, n = 2, 40
    seq = torch.randint(0, 12, (b, n), generator=g)
    coors = torch.randn(b, n, 3, generator=g, dtype=torch.float64)
    etok = torch.randint(0, 5, (b, n, n), generator=g)
    i = torch.arange(n)
    adj = (i[:, None] - i[None, :]).abs() <= 1
    mask = torch.arange(n)[None] < torch.tensor([[n], [n - 9]])
    params = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    cfg = O.EGNNConfig(dim=32, edge_dim=4 + 3, num_nearest_neighbors=8, norm_feats=True, norm_coors=True)
    want_n, want_c = O.egnn_network_forward(3, cfg, params, seq.numpy(), coors.numpy(), adj_mat=adj.numpy(), edges=etok.numpy(),
                                            mask=mask.numpy(), num_adj_degrees=2, global_linear_attn_every=2, global_linear_attn_heads=2)[:2]
    with torch.no_grad():
        node, co = net(seq.cuda(), coors.cuda(), adj_mat=adj.cuda(), edges=etok.cuda(), mask=mask.cuda())
    assert node.dtype == torch.float64
    for got, want in ((node, want_n), (co, want_c)):
        scale = max(1.0, float(np.abs(want).max()))
        assert float(np.abs(got.cpu().numpy() - want).max()) <= 1e-9 * scale


@pytest.mark.parametrize("kw,n,cdim,use_mask", [
    (dict(dim=24, num_nearest_neighbors=6, norm_feats=True), 30, 3, False),
    (dict(dim=16, edge_dim=3, fourier_features=2, soft_edges=True, norm_coors=True, m_pool_method="mean", coor_weights_clamp_value=1.5), 20, 3, True),
    (dict(dim=16, m_dim=80, num_nearest_neighbors=9, fourier_features=1), 25, 5, True),
    (dict(dim=12, num_nearest_neighbors=4, update_coors=False), 18, 2, False),
])
def test_float64_module_trains_on_the_float64_backward_kernels(kw, n, cdim, use_mask, monkeypatch):
    """Under autograd a float64 module's forward is the float64 kernels and -- round 5 -- its backward `_backward_exact`: the E x H work on
    csrc/edge_exact_bwd.hip, every contraction on egnn_linear_f64, the ATen recompute never called.  Gradients equal float64 autograd of
    the restated layer over the same neighbour list at 1e-9, and the recompute path (EGNN_NATIVE_BACKWARD_EXACT=0) gives the same."""
    from egnn_pytorch_amd import EGNN, _ops
    from egnn_pytorch_amd import autograd as A
    g = torch.Generator().manual_seed(4)
    layer = EGNN(**kw).double()
    _xavier_(layer, g)
    layer = layer.cuda()
    b = 2
    feats = torch.randn(b, n, kw["dim"], generator=g, dtype=torch.float64).cuda().requires_grad_(True)
    coors = torch.randn(b, n, cdim, generator=g, dtype=torch.float64).cuda().requires_grad_(True)
    edges = torch.randn(b, n, n, kw["edge_dim"], generator=g, dtype=torch.float64).cuda().requires_grad_(True) if kw.get("edge_dim") else None
    mask = (torch.arange(n)[None] < torch.tensor([[n], [n - 4]])).cuda() if use_mask else None
    leaves = [feats, coors] + ([edges] if edges is not None else [])

    def run(fn):
        for t in leaves:
            t.grad = None
        layer.zero_grad()
        with torch.enable_grad():
            f, c = fn()
            (f.square().sum() + c.square().sum()).backward()
        return [t.grad.clone() for t in leaves] + [None if p.grad is None else p.grad.clone() for p in layer.parameters()]

    real = A._backward_recompute

    def no_recompute(*a, **k):
        raise AssertionError("the ATen recompute backward ran")
    monkeypatch.setattr(A, "_backward_recompute", no_recompute)
    with _ops.phase_timer() as pt:
        got = run(lambda: layer(feats, coors, edges, mask))
    assert {"edge_exact", "edge_exact_bwd", "edge_exact_node_sums", "bwd_exact_dw2", "bwd_exact_dw1", "bwd_exact_dfeats"} <= set(pt.summary())
    monkeypatch.setattr(A, "_backward_recompute", real)
    monkeypatch.setattr(A, "_NATIVE_EXACT", False)
    alt = run(lambda: layer(feats, coors, edges, mask))
    with torch.no_grad():
        idx, rank, radius = layer._forward_with_hint(feats.detach(), coors.detach(), None if edges is None else edges.detach(), mask, None, None)[3:6]
    want = run(lambda: A.layer_given_neighbors(layer, feats, coors, edges, mask, None if idx is None else idx.long(), rank, radius))
    for a, b_, c_ in zip(got, want, alt):
        assert (a is None) == (b_ is None)
        if a is not None:
            assert float((a - b_).abs().max()) <= 1e-9 * max(1.0, float(b_.abs().max())), (float((a - b_).abs().max()), float(b_.abs().max()))
            assert float((c_ - b_).abs().max()) <= 1e-9 * max(1.0, float(b_.abs().max()))


# ------------------------------------------------------------------------------------------------ the kernels on their own
@pytest.mark.parametrize("m,n,k,act,res", [(300, 70, 33, 0, False), (64, 64, 16, 1, True), (1000, 130, 515, 0, True), (5, 3, 2, 1, False)])
def test_linear_f64_kernel(m, n, k, act, res):
    from egnn_pytorch_amd import _ops
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k + 3, generator=g, dtype=torch.float64).cuda()[:, :k]            # (a column slice of a wider matrix)
    w = torch.randn(n, k, generator=g, dtype=torch.float64).cuda()
    bias = torch.randn(n, generator=g, dtype=torch.float64).cuda()
    r = torch.randn(m, n, generator=g, dtype=torch.float64).cuda() if res else None
    out = _ops.linear_f32(a, w, n, k, bias=bias, residual=r, act=act)
    want = a @ w.t() + bias
    if act:
        want = torch.nn.functional.silu(want)
    if res:
        want = want + r
    assert out.dtype == torch.float64
    assert float((out - want).abs().max()) <= 1e-12 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("n,k,cdim,use_mask,use_adj", [(64, 8, 3, True, False), (300, 32, 3, True, True), (50, 50, 5, False, False),
                                                       (1024, 32, 3, True, False), (5000, 17, 2, False, False)])
def test_knn_select_f64_kernel(n, k, cdim, use_mask, use_adj):
    from egnn_pytorch_amd import _ops
    g = torch.Generator().manual_seed(n + k)
    b = 2
    coors = torch.randn(b, n, cdim, generator=g, dtype=torch.float64).cuda()
    coors[:, 3] = coors[:, 2]                                                     # exact ties: lowest index first
    mask = (torch.arange(n)[None] < torch.tensor([[n], [max(k, n - 5)]])).cuda() if use_mask else None
    i = torch.arange(n)
    adj = ((i[:, None] - i[None, :]).abs() <= 2).cuda() if use_adj else None
    idx, rank = _ops.knn_select(coors, mask, adj, k)
    rel = coors[:, :, None, :] - coors[:, None, :, :]
    sq = rel * rel
    order = list(range(cdim)) if cdim not in (5, 6, 7) else [0] + list(range(4, cdim)) + [1, 2, 3]
    dist = sq[..., order[0]].clone()
    for c in order[1:]:
        dist = dist + sq[..., c]
    ranking = dist.clone()
    if mask is not None:
        ranking.masked_fill_(~(mask[:, :, None] & mask[:, None, :]), 1e5)
    if adj is not None:
        eye = torch.eye(n, dtype=torch.bool, device="cuda")
        ranking.masked_fill_(eye[None], -1.0)
        ranking.masked_fill_((adj & ~eye)[None].expand(b, -1, -1), 0.0)
    # (value, index) order = a stable sort by value
    want_rank, want_idx = torch.sort(ranking, dim=-1, stable=True)
    assert rank.dtype == torch.float64
    assert torch.equal(rank, want_rank[..., :k])
    assert torch.equal(idx.long(), want_idx[..., :k])


def test_node_prep_f64_kernel():
    from egnn_pytorch_amd import _ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(77, 130, generator=g, dtype=torch.float64).cuda() * 3 + 1
    m = torch.randn(77, 16, generator=g, dtype=torch.float64).cuda()
    gam, bet = torch.randn(130, generator=g, dtype=torch.float64).cuda(), torch.randn(130, generator=g, dtype=torch.float64).cuda()
    out = _ops.node_prep_f32(x, m, gam, bet, 1e-5, 16)
    want = torch.cat((torch.nn.functional.layer_norm(x, (130,), gam, bet, 1e-5), m), dim=-1)
    assert float((out - want).abs().max()) <= 1e-13 * float(want.abs().max())
    out = _ops.node_prep_f32(x, None, None, None, 1e-5, 16)
    assert torch.equal(out[:, :130], x) and not out[:, 130:].any()


# ------------------------------------------------------------------------------------------------ the backward kernels on their own
@pytest.mark.parametrize("kw,n,k,cdim,use_mask", [
    (dict(dim=8, m_dim=16, num_nearest_neighbors=6), 20, 6, 3, False),
    (dict(dim=8, m_dim=24, num_nearest_neighbors=5, soft_edges=True, norm_coors=True, coor_weights_clamp_value=0.7), 18, 5, 5, True),
    (dict(dim=8, m_dim=40), 12, 12, 2, True),                                     # dense (idx = NULL)
    (dict(dim=8, m_dim=7, num_nearest_neighbors=4, update_coors=False), 15, 4, 3, False),
])
def test_closed_form_tail_kernel_equals_its_specification(kw, n, k, cdim, use_mask):
    """egnn_edge_tail_exact_bwd_f64 against egnn_pytorch_amd.autograd.tail_edge_backward (the torch specification, itself equal to
    autograd of the restated layer): d/d u, the operands of the parameter gradients and -- through the per-node sums -- d/d coors, at
    1e-12 of each tensor's scale.  (d/d rel is compared through its per-node sums: the kernel writes a self pair's as the exact zero it
    sums to.)"""
    from egnn_pytorch_amd import EGNN, _ops, autograd as A
    g = torch.Generator().manual_seed(n * 31 + k)
    layer = EGNN(**kw).double()
    _xavier_(layer, g)
    layer = layer.cuda()
    b, m = 2, kw["m_dim"]
    dense = "num_nearest_neighbors" not in kw
    u = torch.randn(b, n, k, m, generator=g, dtype=torch.float64).cuda()
    coors = torch.randn(b, n, cdim, generator=g, dtype=torch.float64).cuda()
    idx = None if dense else torch.stack([torch.stack([torch.randperm(n, generator=g)[:k] for _ in range(n)]) for _ in range(b)]).cuda()
    if idx is not None:
        idx[:, :, 0] = torch.arange(n, device="cuda")[None]                        # every node has its self pair, as the k-NN lists do
    pm = (torch.rand(b, n, k, generator=g) > 0.2).cuda() if use_mask else None
    g_co = torch.randn(b, n, cdim, generator=g, dtype=torch.float64).cuda()
    g_ms = torch.randn(b, n, m, generator=g, dtype=torch.float64).cuda()
    with torch.no_grad():
        want = A.tail_edge_backward(layer, u, coors, idx, pm, g_co, g_ms) if layer.coors_mlp is not None else None
    i32 = None if idx is None else idx.int().contiguous()
    dl = _ops.dest_lists(i32, b, n, k, u.device)
    grads = {id(p): torch.zeros_like(p) for p in layer.parameters()}
    with torch.no_grad():
        g_u, g_c = A._tail_closed_form(layer, u.reshape(b * n * k, m).contiguous(), coors, i32, pm, g_co, g_ms.reshape(b * n, m).contiguous(),
                                       grads, dl, b, n, k)
    close = lambda a, w: float((a - w).abs().max()) <= 1e-12 * max(1.0, float(w.abs().max()))     # noqa: E731
    if want is None:                                                                # update_coors=False: only the pooling / SiLU chain
        sg = torch.sigmoid(u)
        gm = g_ms[:, :, None, :].expand(b, n, k, m)
        assert close(g_u.view(b, n, k, m), gm * (sg * (1 + u * (1 - sg))))
        assert float(g_c.abs().max()) == 0.0
        return
    assert close(g_u.view(b, n, k, m), want["g_u"])
    # d/d coors of this part = sum over k of g_rel at the source, minus the sum over the edges arriving at the node
    g_rel = want["g_rel"].clone()
    ar = torch.arange(n, device="cuda")
    self_pair = (ar[None, :, None] == (ar[None, None, :] if idx is None else idx)).expand(b, n, k)
    g_rel[self_pair] = 0.0                        # x_i - x_i: contributes +g and -g to the same coordinate (1 / eps-sized under CoorsNorm)
    src = g_rel.sum(dim=2)
    dst = torch.zeros_like(src)
    if idx is None:
        dst = g_rel.sum(dim=1)
    else:
        bi = torch.arange(b, device="cuda")[:, None, None].expand(b, n, k)
        dst.index_put_((bi.reshape(-1), idx.reshape(-1)), g_rel.reshape(-1, cdim), accumulate=True)
    assert close(g_c, src - dst)
    cm = layer.coors_mlp
    e = b * n * k
    assert close(grads[id(cm[0].weight)], want["g_hid"].t() @ want["m"])
    assert close(grads[id(cm[0].bias)], want["g_hid"].sum(0))
    assert close(grads[id(cm[3].weight)], (want["g_w"][None] @ want["a3"]))
    assert close(grads[id(cm[3].bias)], want["g_w"].sum().reshape(1))
    if layer.norm_coors:
        assert close(grads[id(layer.coors_norm.scale)], want["g_scale"].sum().reshape(1))
    if layer.edge_gate is not None:
        assert close(grads[id(layer.edge_gate[0].weight)], want["g_gate"][None] @ want["m0"])
        assert close(grads[id(layer.edge_gate[0].bias)], want["g_gate"].sum().reshape(1))


@pytest.mark.parametrize("dim,m,n,k,four,edim,cdim", [(8, 16, 20, 6, 0, 0, 3), (6, 24, 16, 16, 2, 3, 5), (4, 80, 10, 4, 1, 0, 2)])
def test_exact_backward_kernels_equal_autograd_of_the_first_linear(dim, m, n, k, four, edim, cdim):
    """egnn_edge_exact_bwd_f64 + egnn_edge_exact_node_sums_f64 against torch autograd of u = W2 SiLU(P_i[i] + P_j[j] + W_s s) + b2 in
    float64: a^T, dz^T, d/d scalars and the per-node sums of dz (d/d P_i, d/d P_j), 1e-12."""
    from egnn_pytorch_amd import _abi, _ops, autograd as A
    g = torch.Generator().manual_seed(dim + m + n)
    b, dense = 2, k == n
    h = 2 * (2 * dim + 2 * four + 1 + edim)
    s_in = 2 * four + 1 + edim
    dev = torch.device("cuda")
    r = lambda *sh: torch.randn(*sh, generator=g, dtype=torch.float64).to(dev)          # noqa: E731
    proj = r(b * n, 2 * h)                                                             # [P_i | P_j]
    ws, w2, coors, gu = r(h, s_in) * 0.3, r(m, h) * 0.3, r(b, n, cdim), r(b * n * k, m)
    edges = r(b, n, n, edim) if edim else None
    idx = None if dense else torch.stack([torch.stack([torch.randperm(n, generator=g)[:k] for _ in range(n)]) for _ in range(b)]).to(dev)
    i32 = None if idx is None else idx.int().contiguous()

    class L:                                                                           # what edge_scalars reads of a layer
        fourier_features = four
    pr = proj.clone().requires_grad_(True)
    rel, scal = A.edge_scalars(L, coors, edges, idx)
    sc = scal.detach().clone().requires_grad_(True)
    pi, pj = pr[:, :h].view(b, n, h), pr[:, h:].view(b, n, h)
    bi = torch.arange(b, device=dev)[:, None, None]
    z = pi[:, :, None, :] + (pj[:, None, :, :] if dense else pj[bi, idx]) + sc @ ws.t()
    a_ref = torch.nn.functional.silu(z)
    u = a_ref @ w2.t()
    z.retain_grad()
    (u.reshape(-1, m) * gu).sum().backward()
    e = b * n * k
    a_t = torch.empty(h, e, dtype=torch.float64, device=dev)
    dz_t = torch.empty(h, e, dtype=torch.float64, device=dev)
    g_scal = torch.empty(e, s_in, dtype=torch.float64, device=dev)
    args = _abi.EdgeExactBwdArgs()
    args.B, args.N, args.K, args.m_dim, args.H = b, n, k, m, h
    args.fourier, args.edge_dim, args.coor_dim, args.edges_by_k = four, edim, cdim, 0
    args.Pi, args.Pj, args.ldp = proj.data_ptr(), proj.data_ptr() + 8 * h, 2 * h
    args.Ws, args.ldws, args.W2 = ws.data_ptr(), s_in, w2.data_ptr()
    args.coors, args.edges, args.idx = coors.data_ptr(), _ops._ptr(edges), _ops._ptr(i32)
    args.gU, args.A_T, args.DZ_T, args.g_scal = gu.data_ptr(), a_t.data_ptr(), dz_t.data_ptr(), g_scal.data_ptr()
    _ops.edge_exact_bwd(args, torch.float64)
    dl = _ops.dest_lists(i32, b, n, k, dev)
    gpi, gpi_t, gpj, gpj_t = _ops.edge_exact_node_sums(dz_t, b * n, k, dl.order, dl.seg)
    close = lambda x, w: float((x - w).abs().max()) <= 1e-12 * max(1.0, float(w.abs().max()))     # noqa: E731
    assert close(a_t.t().reshape(b, n, k, h), a_ref.detach())
    assert close(dz_t.t().reshape(b, n, k, h), z.grad)
    assert close(g_scal.view_as(sc), sc.grad)
    assert close(gpi, pr.grad[:, :h]) and close(gpj, pr.grad[:, h:])
    assert torch.equal(gpi_t, gpi.t()) and torch.equal(gpj_t, gpj.t())
