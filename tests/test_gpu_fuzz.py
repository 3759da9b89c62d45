"""Seeded random walks over the constructor / input space of EGNN.forward (egnn_pytorch.py:149-168, 224-341): the HIP path against the
numpy oracle, one small case per seed.  The hand-picked cases of tests/test_gpu_parity.py cover each option on its own and the
combinations somebody thought of; this covers the ones nobody did (kernel variants are chosen by K, m_dim, the number of per-edge
scalars, the coordinate dimension and the flags *together*).  1e-4 of the output's scale, as everywhere."""
import os

import numpy as np
import pytest
import torch

from oracle import egnn_oracle as O

pytestmark = pytest.mark.gpu


def _dev(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _draw(seed):
    rng = np.random.default_rng(1000 + seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]                                      # noqa: E731
    # (dim 32 / 64 / 128 with m_dim 16: node_mlp in one launch, csrc/node_mlp_fused.hip)
    kw = dict(dim=pick([8, 16, 24, 32, 40, 64, 128]), m_dim=pick([4, 16, 16, 16, 20, 32, 48]), edge_dim=pick([0, 0, 1, 3]),
              fourier_features=pick([0, 0, 1, 2]), norm_feats=bool(rng.integers(2)), norm_coors=bool(rng.integers(2)),
              m_pool_method=pick(["sum", "mean"]), soft_edges=bool(rng.integers(2)),
              coor_weights_clamp_value=pick([None, None, 0.5, 2.0]), valid_radius=pick([float("inf"), float("inf"), 2.5]))
    upd = pick(["both", "both", "feats", "coors"])
    kw["update_feats"], kw["update_coors"] = upd != "coors", upd != "feats"
    mode = pick(["dense", "knn", "knn", "knn", "sparse", "knn_adj"])
    n = int(rng.integers(6, 90))
    if mode in ("sparse", "knn_adj") and rng.integers(3) == 0:
        n = pick([144, 160, 256, 272])                       # (N % 16 == 0 beyond 128 nodes: rows the adjacency decides, csrc/knn_select.hip)
    if mode in ("knn", "knn_adj"):
        kw["num_nearest_neighbors"] = min(n, pick([3, 5, 8, 16, 32, 32, 64]))
    if mode == "sparse":
        kw["only_sparse_neighbors"] = True
    cdim = pick([3, 3, 3, 1, 2, 5, 8, 11])
    b = int(rng.integers(1, 4))
    use_mask = bool(rng.integers(3))
    return kw, mode, b, n, cdim, use_mask, rng


@pytest.mark.parametrize("seed", range(int(os.environ.get("EGNN_FUZZ_SEEDS", "64"))))      # (EGNN_FUZZ_SEEDS=1000: a longer walk)
def test_random_configuration_against_the_oracle(seed):
    from egnn_pytorch_amd import EGNN
    kw, mode, b, n, cdim, use_mask, rng = _draw(seed)
    cfg = O.EGNNConfig(**kw)
    params = O.random_params(cfg, seed=seed)
    # dense graphs and wide neighbourhoods sum many messages: keep the outputs of order one to ten (the 1e-4 bar is absolute up to 256)
    k_eff = kw.get("num_nearest_neighbors", 0) or (3 if mode == "sparse" else n)
    if "coors_mlp.3.weight" in params:
        params["coors_mlp.3.weight"] = params["coors_mlp.3.weight"] * np.float32(min(1.0, 8.0 / k_eff))
    params["edge_mlp.3.weight"] = params["edge_mlp.3.weight"] * np.float32(min(1.0, 4.0 / np.sqrt(k_eff)))
    feats = rng.standard_normal((b, n, kw["dim"])).astype(np.float32)
    coors = rng.standard_normal((b, n, cdim)).astype(np.float32)
    edges = rng.standard_normal((b, n, n, kw["edge_dim"])).astype(np.float32) if kw["edge_dim"] else None
    mask = None
    if use_mask:
        lo = max(1, kw.get("num_nearest_neighbors", 1))
        lens = rng.integers(min(n, max(lo, n // 2)), n + 1, size=b)
        mask = np.arange(n)[None, :] < lens[:, None]
        if rng.integers(2):
            # padding anywhere, not only behind the real nodes (the edge pass skips padded nodes by group and by wave: csrc/edge_pw.hip)
            mask = np.stack([rng.permutation(row) for row in mask])
    adj = None
    if mode in ("sparse", "knn_adj"):
        i = np.arange(n)
        adj = (np.abs(i[:, None] - i[None, :]) <= 1) | (rng.random((n, n)) < 0.02)
        adj = adj | adj.T
    want_n, want_c = O.egnn_forward(cfg, params, feats, coors, edges, mask, adj)
    net = EGNN(**kw)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        node, co = net(_dev(feats), _dev(coors), _dev(edges), _dev(mask), _dev(adj))
    what = (seed, kw, mode, b, n, cdim, use_mask)
    for got, want in ((node, want_n), (co, want_c)):
        assert np.isfinite(want).all(), what
        tol = 1e-4 * max(1.0, float(np.abs(want).max()) / 256.0)
        err = float(np.abs(got.cpu().numpy() - want).max())
        assert err <= tol, (what, err, tol)


@pytest.fixture(scope="module")
def ref():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import build_ref
    if not build_ref.build():
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return build_ref.import_reference()


@pytest.mark.parametrize("seed", range(int(os.environ.get("EGNN_FUZZ_GRAD_SEEDS", "32"))))
def test_random_configuration_gradients_against_the_reference_autograd(ref, seed):
    """The same walk under autograd: gradients of the inputs and of every parameter through the HIP forward + backward against the
    REFERENCE module's own autograd in float64 on the same device (oracle/_ref), 2e-4 of each gradient's scale.  Without CoorsNorm:
    its self pair makes any fp32 autograd -- the reference's too -- carry O(1) noise in the coordinate gradient (DESIGN.md section 10;
    tests/test_autograd.py pins that case in float64)."""
    from egnn_pytorch_amd import EGNN
    kw, mode, b, n, cdim, use_mask, rng = _draw(5000 + seed)
    kw["norm_coors"] = False
    n = min(n, 48)
    if "num_nearest_neighbors" in kw:
        kw["num_nearest_neighbors"] = min(kw["num_nearest_neighbors"], n)
    torch.manual_seed(seed)
    rlayer = ref.EGNN(**kw)
    k_eff = kw.get("num_nearest_neighbors", 0) or (3 if mode == "sparse" else n)
    with torch.no_grad():
        for mod in rlayer.modules():
            if isinstance(mod, torch.nn.Linear):
                torch.nn.init.xavier_normal_(mod.weight)
        rlayer.edge_mlp[3].weight.mul_(min(1.0, 4.0 / k_eff ** 0.5))
        if rlayer.coors_mlp is not None:
            rlayer.coors_mlp[3].weight.mul_(min(1.0, 8.0 / k_eff))
    layer = EGNN(**kw)
    layer.load_state_dict(rlayer.state_dict(), strict=True)
    layer, rlayer = layer.cuda(), rlayer.double().cuda()
    feats = torch.from_numpy(rng.standard_normal((b, n, kw["dim"])).astype(np.float32)).cuda()
    coors = torch.from_numpy(rng.standard_normal((b, n, cdim)).astype(np.float32)).cuda()
    edges = torch.from_numpy(rng.standard_normal((b, n, n, kw["edge_dim"])).astype(np.float32)).cuda() if kw["edge_dim"] else None
    mask = None
    if use_mask:
        lo = max(1, kw.get("num_nearest_neighbors", 1))
        lens = rng.integers(min(n, max(lo, n // 2)), n + 1, size=b)
        mask = torch.from_numpy(np.arange(n)[None, :] < lens[:, None]).cuda()
    adj = None
    if mode in ("sparse", "knn_adj"):
        i = np.arange(n)
        a = (np.abs(i[:, None] - i[None, :]) <= 1) | (rng.random((n, n)) < 0.02)
        adj = torch.from_numpy(a | a.T).cuda()
    g = torch.Generator().manual_seed(seed)
    rn, rc = torch.randn(b, n, kw["dim"], generator=g).cuda(), torch.randn(b, n, cdim, generator=g).cuda()

    def grads(mod, dt):
        f, c = feats.to(dt).requires_grad_(True), coors.to(dt).requires_grad_(True)
        e = None if edges is None else edges.to(dt).requires_grad_(True)
        with torch.enable_grad():
            node, co = mod(f, c, e, mask, adj)
            wrt = [t for t in (f, c, e) if t is not None] + [p for p in mod.parameters()]
            return torch.autograd.grad((node * rn.to(dt)).sum() + (co * rc.to(dt)).sum(), wrt, allow_unused=True)

    got, want = grads(layer, torch.float32), grads(rlayer, torch.float64)
    import copy
    ref32 = grads(copy.deepcopy(rlayer).float(), torch.float32)            # the reference's own fp32 run: the yardstick for cancelling sums
    what = (seed, kw, mode, b, n, cdim, use_mask)
    for i, (gg, ww, rr) in enumerate(zip(got, want, ref32)):
        assert (gg is None) == (ww is None), (what, i)
        if gg is None:
            continue
        scale = float(ww.abs().max())
        if scale == 0.0:
            assert float(gg.abs().max()) == 0.0, (what, i)
            continue
        err = float((gg.double() - ww).abs().max())
        # 2e-4 of the gradient's scale -- or, where the gradient is a heavily cancelling sum over all edges (a bias of 1e-3 made of
        # terms of 1e-1), no further from float64 than four times the reference's own fp32 autograd is
        assert err <= max(2e-4 * scale, 4.0 * float((rr.double() - ww).abs().max())), (what, i, err / scale)


@pytest.mark.parametrize("seed", range(int(os.environ.get("EGNN_FUZZ_NET_SEEDS", "32"))))
def test_random_network_against_the_oracle(seed):
    """EGNN_Network (egnn_pytorch.py:343-454): token / position / edge-token / adjacency-degree front-end, induced-set attention blocks and
    the layer loop, random combinations, against the oracle's restatement."""
    from egnn_pytorch_amd import EGNN_Network
    rng = np.random.default_rng(9000 + seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]                                      # noqa: E731
    n, b, depth, dim = int(rng.integers(8, 48)), int(rng.integers(1, 3)), int(rng.integers(1, 4)), pick([16, 24, 32])
    kw = dict(depth=depth, dim=dim)
    num_tokens = pick([None, 11])
    if num_tokens:
        kw["num_tokens"] = num_tokens
    if rng.integers(2):
        kw["num_positions"] = n + int(rng.integers(0, 5))
    edge_mode = pick(["none", "float", "tokens"])
    edge_dim = 0
    if edge_mode != "none":
        edge_dim = pick([1, 3, 4])
        kw["edge_dim"] = edge_dim
        if edge_mode == "tokens":
            kw["num_edge_tokens"] = 6
    adj_degrees = pick([None, None, 1, 2, 3])
    adj_dim = 0
    if adj_degrees:
        kw["num_adj_degrees"] = adj_degrees
        adj_dim = pick([0, 2, 3])
        kw["adj_dim"] = adj_dim
    attn_every = pick([0, 0, 1, 2])
    if attn_every:
        kw.update(global_linear_attn_every=attn_every, global_linear_attn_heads=2, global_linear_attn_dim_head=8, num_global_tokens=int(rng.integers(1, 6)))
    layer_kw = dict(m_dim=pick([8, 16, 16, 32]), fourier_features=pick([0, 0, 1]), norm_coors=bool(rng.integers(2)),
                    m_pool_method=pick(["sum", "mean"]), soft_edges=bool(rng.integers(2)), coor_weights_clamp_value=pick([None, 1.0]))
    mode = pick(["dense", "knn", "knn", "sparse"]) if adj_degrees else pick(["dense", "knn", "knn"])
    if mode == "knn":
        layer_kw["num_nearest_neighbors"] = min(n, pick([4, 8, 16, 32]))
    elif mode == "sparse":
        layer_kw["only_sparse_neighbors"] = True
    kw.update(layer_kw)
    torch.manual_seed(seed)
    net = EGNN_Network(**kw)
    k_eff = layer_kw.get("num_nearest_neighbors", 0) or (7 if mode == "sparse" else n)
    with torch.no_grad():
        for name, mod in net.named_modules():
            if type(mod) is torch.nn.Linear:
                torch.nn.init.xavier_normal_(mod.weight)
                if name.endswith("coors_mlp.3"):                      # (stacked layers: keep the coordinate updates small)
                    mod.weight.mul_(min(0.25, 1.0 / k_eff) * (1.0 if layer_kw["norm_coors"] else 0.05))
                if name.endswith("edge_mlp.3"):
                    mod.weight.mul_(min(1.0, 2.0 / k_eff ** 0.5))
                if name.endswith("node_mlp.3"):
                    mod.weight.mul_(0.5)
    net = net.cuda().eval()
    feats = rng.integers(0, 11, (b, n)) if num_tokens else rng.standard_normal((b, n, dim)).astype(np.float32)
    coors = rng.standard_normal((b, n, 3)).astype(np.float32)
    edges = None
    if edge_mode == "float":
        edges = rng.standard_normal((b, n, n, edge_dim)).astype(np.float32)
    elif edge_mode == "tokens":
        edges = rng.integers(0, 6, (b, n, n))
    i = np.arange(n)
    adj = ((np.abs(i[:, None] - i[None, :]) <= 1) | (rng.random((n, n)) < 0.03)) if (adj_degrees or rng.integers(2)) else None
    if adj is not None:
        adj = adj | adj.T
    mask = (np.arange(n)[None, :] < rng.integers(max(layer_kw.get("num_nearest_neighbors", 1), n // 2), n + 1, size=(b, 1))) if rng.integers(3) else None
    params = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    cfg = O.EGNNConfig(dim=dim, edge_dim=edge_dim + adj_dim, norm_feats=True, **layer_kw)
    want_n, want_c = O.egnn_network_forward(depth, cfg, params, feats, coors, adj_mat=adj, edges=edges, mask=mask, num_adj_degrees=adj_degrees,
                                            global_linear_attn_every=attn_every, global_linear_attn_heads=2)[:2]
    with torch.no_grad():
        node, co = net(_dev(feats), _dev(coors), adj_mat=_dev(adj), edges=_dev(edges), mask=_dev(mask))
    what = (seed, kw, mode, b, n)
    for got, want in ((node, want_n), (co, want_c)):
        assert np.isfinite(want).all(), what
        # (stacked layers amplify rounding differences: DESIGN.md section 6 -- 1e-4 of the output's scale, absolute up to 16)
        tol = 1e-4 * max(1.0, float(np.abs(want).max()) / 16.0) * depth
        err = float(np.abs(got.cpu().numpy() - want).max())
        assert err <= tol, (what, err, tol)
