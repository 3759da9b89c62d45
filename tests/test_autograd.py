"""Autograd support: `egnn_pytorch_amd.autograd`.

CPU part: the differentiable restatement `layer_given_neighbors` -- what the backward recomputes -- against the reference
itself (oracle/_ref): outputs for the neighbour lists the reference's own topk produced (golden files) and gradients of every
input and parameter against the reference's autograd.
GPU part: HIP forward + recompute backward of the drop-in modules against the reference's autograd on the same inputs, and a
denoise_sparse.py-style training loop (denoise_sparse.py:45-78)."""
import os
import sys

import numpy as np
import pytest
import torch

from tests._util import golden_names, layer_kwargs, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYER_CASES = [n for n in golden_names() if not n.startswith("net")]


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, ROOT)
    from oracle import build_ref
    if not build_ref.build():
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return build_ref.import_reference()


def _t(d, k, dtype=None):
    if k not in d:
        return None
    t = torch.from_numpy(np.ascontiguousarray(d[k]))
    return t if dtype is None else t.to(dtype)


def _case(name):
    from egnn_pytorch_amd import EGNN
    meta, params, d = load_golden(name)
    if meta["kind"] != "layer":
        pytest.skip("network case")
    kw = layer_kwargs(meta)
    layer = EGNN(**kw)
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return meta, kw, layer, params, d


@pytest.mark.parametrize("name", LAYER_CASES)
def test_restatement_reproduces_reference_outputs(name):
    """forward values for the neighbour list the reference's own topk returned (stored in the golden file)."""
    from egnn_pytorch_amd.autograd import layer_given_neighbors
    meta, kw, layer, params, d = _case(name)
    feats, coors = _t(d, "feats"), _t(d, "coors")
    if feats.shape[-1] != layer.dim:
        pytest.skip("token inputs")
    idx, rank = _t(d, "topk_indices.0", torch.int64), _t(d, "topk_values.0")
    radius = 0.0 if (layer.only_sparse_neighbors and "adj_mat" in d) else layer.valid_radius
    for factorised in (True, False):
        with torch.no_grad():
            node, co = layer_given_neighbors(layer.eval(), feats, coors, _t(d, "edges"), _t(d, "mask"), idx, rank, radius, factorised)
        np.testing.assert_allclose(node.numpy(), d["node_out"], atol=3e-5, rtol=0)
        np.testing.assert_allclose(co.numpy(), d["coors_out"], atol=3e-5, rtol=0)


def _grads(module, call, inputs, seed=0):
    """d(sum(node * Rn) + sum(coors * Rc)) / d(inputs, parameters) with fixed random cotangents."""
    node, co = call()
    g = torch.Generator().manual_seed(seed)
    rn = torch.randn(node.shape, generator=g).to(node.device)
    rc = torch.randn(co.shape, generator=g).to(co.device)
    loss = (node * rn).sum() + (co * rc).sum()
    wrt = [t for t in inputs if t is not None] + list(module.parameters())
    grads = torch.autograd.grad(loss, wrt, allow_unused=True)
    return [None if gr is None else gr.detach().cpu() for gr in grads], [tuple(w.shape) for w in wrt]


@pytest.mark.parametrize("name", ["knn8_mask", "all_flags", "c1_dense_dim32", "dense_edges_mask", "fourier2_knn",
                                  "mean_pool_nomask", "sparse_chain_edges_mask", "knn8_coor_dim5_normcoors_mask"])
def test_restatement_gradients_match_reference_autograd(ref, name):
    from egnn_pytorch_amd.autograd import layer_given_neighbors
    # In float64: with norm_coors the self pair (rel = 0, a neighbour of every node) goes through x / clamp(|x|, 1e-8), whose
    # Jacobian is scale / 1e-8 = O(1e7); the +/- contributions cancel exactly in exact arithmetic but leave O(1) rounding
    # noise in fp32 -- in the reference's own gradient as much as in ours.  fp64 checks the mathematics.
    meta, kw, layer, params, d = _case(name)
    layer = layer.double()
    rlayer = ref.EGNN(**meta["kwargs"])
    rlayer.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    rlayer = rlayer.double()
    idx, rank = _t(d, "topk_indices.0", torch.int64), _t(d, "topk_values.0")
    radius = 0.0 if (layer.only_sparse_neighbors and "adj_mat" in d) else layer.valid_radius
    mk = lambda k: None if k not in d else _t(d, k).double().clone().requires_grad_(True)
    f1, c1, e1 = mk("feats"), mk("coors"), mk("edges")
    f2, c2, e2 = mk("feats"), mk("coors"), mk("edges")
    mask, adj = _t(d, "mask"), _t(d, "adj_mat")
    want, shapes2 = _grads(rlayer, lambda: rlayer(f2, c2, e2, mask, adj), (f2, c2, e2))
    for factorised in (True, False):                     # node-level projections (what the backward runs) / literal cat + Linear
        got, shapes = _grads(layer, lambda: layer_given_neighbors(layer, f1, c1, e1, mask, idx, rank, radius, factorised),
                             (f1, c1, e1))
        assert shapes == shapes2
        for g, w in zip(got, want):
            assert (g is None) == (w is None)
            if g is not None:
                scale = max(1.0, float(w.abs().max()))
                np.testing.assert_allclose(g.numpy(), w.numpy(), atol=1e-7 * scale, rtol=0)


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("kw,n,flags", [
    (dict(dim=32, num_nearest_neighbors=8), 48, dict(mask=True)),
    (dict(dim=16, edge_dim=3, fourier_features=2, soft_edges=True, norm_coors=True, norm_feats=True, m_pool_method="mean",
          coor_weights_clamp_value=2.0), 20, dict(mask=True, edges=True)),
    (dict(dim=64, num_nearest_neighbors=16, norm_feats=True), 200, dict(mask=False)),
    (dict(dim=24, only_sparse_neighbors=True, edge_dim=2), 40, dict(mask=True, edges=True, adj=True)),
    (dict(dim=16, update_feats=False, num_nearest_neighbors=6), 30, dict(mask=True)),
])
def test_hip_forward_with_recompute_backward_matches_reference(ref, kw, n, flags):
    """loss.backward() through the drop-in layer on the MI355X == the reference's autograd on the CPU (1e-4 of each gradient's scale)."""
    from egnn_pytorch_amd import EGNN
    torch.manual_seed(5)
    rlayer = ref.EGNN(**kw)
    with torch.no_grad():
        for p in rlayer.parameters():
            p.mul_(60.0)                                    # away from the vacuous default init (std 1e-3)
    layer = EGNN(**kw)
    layer.load_state_dict(rlayer.state_dict(), strict=True)
    layer = layer.cuda()
    g = torch.Generator().manual_seed(2)
    b = 3
    feats, coors = torch.randn(b, n, kw["dim"], generator=g), torch.randn(b, n, 3, generator=g)
    mask = (torch.arange(n)[None] < torch.tensor([[n], [n - 5], [n // 2 + 4]])) if flags.get("mask") else None
    edges = torch.randn(b, n, n, kw.get("edge_dim", 0), generator=g) if flags.get("edges") else None
    adj = None
    if flags.get("adj"):
        i = torch.arange(n)
        adj = (i[:, None] - i[None, :]).abs() <= 2
    mk = lambda t, dev: None if t is None else t.clone().to(dev).requires_grad_(True)
    dv = lambda t: None if t is None else t.cuda()
    f1, c1, e1 = mk(feats, "cuda"), mk(coors, "cuda"), mk(edges, "cuda")
    f2, c2, e2 = mk(feats, "cpu"), mk(coors, "cpu"), mk(edges, "cpu")
    got, s1 = _grads(layer, lambda: layer(f1, c1, e1, dv(mask), dv(adj)), (f1, c1, e1))
    want, s2 = _grads(rlayer, lambda: rlayer(f2, c2, e2, mask, adj), (f2, c2, e2))
    assert s1 == s2
    for pos, (gg, ww) in enumerate(zip(got, want)):
        assert (gg is None) == (ww is None)
        if gg is not None:
            scale = max(1.0, float(ww.abs().max()))
            # d/d coors with norm_coors is cancellation noise of the self pair (O(scale / 1e-8 * 2^-24) per term) in BOTH
            # implementations in fp32 (see the fp64 CPU test, which pins the mathematics): nothing to compare there
            if kw.get("norm_coors") and pos == 1:
                assert torch.isfinite(gg).all()
                continue
            np.testing.assert_allclose(gg.numpy(), ww.numpy(), atol=1e-4 * scale, rtol=0)


@pytest.mark.gpu
def test_network_training_loop_denoise_style():
    """denoise_sparse.py:45-78 in miniature: tokens + chain adjacency, noise the coordinates, regress them back with Adam.
    The loss must go down and every parameter must receive a finite gradient."""
    from egnn_pytorch_amd import EGNN_Network
    torch.manual_seed(0)
    net = EGNN_Network(num_tokens=21, dim=16, depth=2, num_nearest_neighbors=6, norm_coors=True,
                       coor_weights_clamp_value=2.0, num_adj_degrees=2, adj_dim=4).cuda()
    opt = torch.optim.Adam(net.parameters(), lr=2e-2)          # (the reference itself goes 0.095 -> 0.026 in 60 such steps)
    g = torch.Generator().manual_seed(1)
    n = 48
    seq = torch.randint(0, 21, (2, n), generator=g).cuda()
    coords = torch.randn(2, n, 3, generator=g).cumsum(dim=1).cuda()
    coords = coords - coords.mean(dim=1, keepdim=True)
    mask = torch.ones(2, n, dtype=torch.bool).cuda()
    i = torch.arange(n)
    adj = ((i[:, None] - i[None, :]).abs() <= 1).cuda()
    losses = []
    noised = coords + torch.randn(coords.shape, generator=g).cuda() * 0.3      # one fixed noise sample: a monotone objective
    for step in range(60):
        feats, denoised = net(seq, noised, adj_mat=adj, mask=mask)
        loss = ((denoised - coords) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        for name, p in net.named_parameters():
            assert p.grad is None or torch.isfinite(p.grad).all(), name
        opt.step()
        losses.append(float(loss))
    assert any(p.grad is not None and float(p.grad.abs().max()) > 0 for p in net.layers[0][1].edge_mlp.parameters())
    assert np.mean(losses[-5:]) < 0.5 * np.mean(losses[:5]), losses


@pytest.mark.gpu
def test_inference_paths_record_nothing():
    from egnn_pytorch_amd import EGNN
    layer = EGNN(dim=16, num_nearest_neighbors=4).cuda()
    f, c = torch.randn(1, 12, 16).cuda(), torch.randn(1, 12, 3).cuda()
    with torch.no_grad():
        n1, c1 = layer(f, c)
    assert not n1.requires_grad and n1.grad_fn is None
    n2, c2 = layer(f, c)                                    # grad mode on, parameters require grad -> a graph, as upstream
    assert n2.requires_grad and n2.grad_fn is not None
    assert torch.equal(n1, n2.detach()) and torch.equal(c1, c2.detach())


@pytest.mark.gpu
@pytest.mark.parametrize("kw,n,flags", [
    (dict(dim=64, num_nearest_neighbors=32, norm_feats=True), 128, dict(mask=True)),                      # P_i shared by the wave
    (dict(dim=32, num_nearest_neighbors=8, edge_dim=3, fourier_features=2, soft_edges=True), 48, dict(mask=True, edges=True)),
    (dict(dim=32), 20, dict(mask=True)),                                                                   # dense all-pairs
    (dict(dim=16, num_nearest_neighbors=5, m_dim=8, m_pool_method="mean", coor_weights_clamp_value=1.0), 30, dict(mask=False)),
    (dict(dim=24, num_nearest_neighbors=48), 96, dict(mask=True)),                                         # multi-round node groups
    (dict(dim=32, edge_dim=4, only_sparse_neighbors=True, norm_coors=True), 40, dict(mask=True, edges=True, adj=True)),   # README-style: 5 scalars
    (dict(dim=32, num_nearest_neighbors=16, fourier_features=1, coor_weights_clamp_value=2.0), 64, dict(mask=False)),     # 3 scalars, one node per tile
    # round 3: two tiles per source node, the second padded (summed in the kernel); the gate, CoorsNorm, mean pooling and ragged masks through
    # the matrix-core tail kernel and its in-kernel sums; LayerNorm around the native node_mlp backward
    (dict(dim=48, num_nearest_neighbors=24, soft_edges=True, norm_coors=True, m_pool_method="mean", norm_feats=True,
          coor_weights_clamp_value=1.5), 70, dict(mask=True)),
    (dict(dim=40, num_nearest_neighbors=20, m_dim=12, soft_edges=True), 50, dict(mask=True)),
    # more than five per-edge scalars on the register-contraction kernels (d/d s on the matrix cores): 8 and 13 scalars
    (dict(dim=32, num_nearest_neighbors=24, edge_dim=3, fourier_features=2, norm_coors=True), 48, dict(mask=True, edges=True)),
    (dict(dim=32, num_nearest_neighbors=8, edge_dim=12), 40, dict(mask=False, edges=True)),
])
def test_native_backward_kernel_matches_the_aten_recompute(kw, n, flags):
    """The native backwards (E x H work on the HIP kernels, small tail and node-level GEMMs around them) against the pure-ATen
    recompute backward of the same Function: every gradient within 1e-4 of its scale.  (The recompute itself is pinned to the reference's
    autograd in float64 on the CPU.)"""
    from egnn_pytorch_amd import EGNN, autograd
    torch.manual_seed(9)
    layer = EGNN(**kw).cuda()
    with torch.no_grad():
        for p in layer.parameters():
            p.mul_(60.0)
    g = torch.Generator().manual_seed(4)
    b = 3
    feats, coors = torch.randn(b, n, kw["dim"], generator=g).cuda(), torch.randn(b, n, 3, generator=g).cuda()
    mask = (torch.arange(n)[None] < torch.tensor([[n], [n - 5], [n // 2 + 4]])).cuda() if flags.get("mask") else None
    edges = torch.randn(b, n, n, kw.get("edge_dim", 0), generator=g).cuda() if flags.get("edges") else None
    adj = None
    if flags.get("adj"):
        i = torch.arange(n)
        adj = ((i[:, None] - i[None, :]).abs() <= 2).cuda()                      # a band: every node has up to 5 neighbours incl. itself
    # (native, mode, graphs per chunk): the register-contraction backward (egnn_edge_bwd_pass_f32; where it applies: one per-edge
    # scalar), the same with the batch cut into chunks (what batches beyond the kernels' 2 GB tables get), and the pure-ATen
    # recompute they are compared with
    results = {}
    for name, native, mode, max_graphs in (("fused", True, "1", 0), ("fused, chunked", True, "1", 2), ("recompute", False, "1", 0)):
        old = autograd._NATIVE, autograd._NATIVE_MODE, autograd._FUSED_MAX_GRAPHS
        autograd._NATIVE, autograd._NATIVE_MODE, autograd._FUSED_MAX_GRAPHS = native, mode, max_graphs
        try:
            f, c = feats.clone().requires_grad_(True), coors.clone().requires_grad_(True)
            e = None if edges is None else edges.clone().requires_grad_(True)
            results[name] = _grads(layer, lambda: layer(f, c, e, mask, adj), (f, c, e))[0]
        finally:
            autograd._NATIVE, autograd._NATIVE_MODE, autograd._FUSED_MAX_GRAPHS = old
    ref = results.pop("recompute")
    if kw.get("norm_coors"):
        # With CoorsNorm a self pair (rel = 0: the diagonal of the adjacency, every dense layer) sends +/- 1 / eps = 1e8-sized terms to
        # x_i that cancel identically; in fp32 autograd -- the recompute here, the reference alike -- they leave O(1) rounding noise
        # in the coordinate gradient.  The native paths are compared with float64 autograd over the same neighbour list instead.
        import copy
        with torch.no_grad():
            idx, rank, radius = layer._forward_hip_checked(feats, coors, edges, mask, adj, None)[3:6]
        l64 = copy.deepcopy(layer).double()
        f, c = feats.double().requires_grad_(True), coors.double().requires_grad_(True)
        e = None if edges is None else edges.double().requires_grad_(True)
        ref = _grads(l64, lambda: autograd.layer_given_neighbors(l64, f, c, e, mask, None if idx is None else idx.long(), None if rank is None else rank.double(), radius), (f, c, e))[0]
        ref = [None if r is None else r.float() for r in ref]
    for name, got in results.items():
        for pos, (a, r) in enumerate(zip(got, ref)):
            assert (a is None) == (r is None)
            if a is not None:
                scale = max(1.0, float(r.abs().max()))
                np.testing.assert_allclose(a.numpy(), r.numpy(), atol=1e-4 * scale, rtol=0, err_msg=f"{name}: gradient #{pos}")


def test_entry_list_pads_every_node_to_whole_tiles():
    """autograd.entry_list (the list layout egnn_edge_bwd_pass_f32 reads, include/egnn_hip.h): each key's entries are consecutive,
    in their given order, padded with -1 to whole 16-entry tiles; the list to a multiple of 128; seg = the tiles of each key."""
    from egnn_pytorch_amd.autograd import entry_list
    g = torch.Generator().manual_seed(3)
    n_keys = 9
    keys = torch.sort(torch.cat([torch.randint(0, n_keys, (70,), generator=g), torch.full((40,), 5)])).values
    keys = keys[keys != 2]                                                 # a node nobody points to
    eids = torch.randperm(keys.numel(), generator=g)
    ent, seg = entry_list(eids, keys, n_keys)
    assert ent.dtype == torch.int32 and ent.numel() % 128 == 0 and seg.tolist()[0] == 0
    assert int((ent >= 0).sum()) == eids.numel() and int(seg[-1]) * 16 <= ent.numel()
    for k in range(n_keys):
        tile0, tile1 = int(seg[k]), int(seg[k + 1])
        chunk = ent[tile0 * 16:tile1 * 16]
        assert torch.equal(chunk[chunk >= 0], eids[keys == k].to(torch.int32))         # same entries, same order
        assert tile1 - tile0 == (int((keys == k).sum()) + 15) // 16
        valid = (chunk >= 0).int()
        assert bool((valid[1:] <= valid[:-1]).all())                                   # padding only behind a node's entries
    assert bool((ent[int(seg[-1]) * 16:] == -1).all())
    # uniform groups (the by-source list: K consecutive entries per node) without the keys
    for per in (5, 16, 40):
        e2 = torch.randperm(n_keys * per, generator=g)
        a, sa = entry_list(e2, None, n_keys)
        b_, sb = entry_list(e2, torch.arange(n_keys).repeat_interleave(per), n_keys)
        assert torch.equal(a, b_) and torch.equal(sa, sb)


def test_split_k_transposed_product_equals_plain_product():
    """autograd._tn (a^T b over tall operands; split-K batched GEMM on the GPU, plain on the CPU) and the per-edge heads routed
    through it give the gradients of the plain modules."""
    from egnn_pytorch_amd import autograd as A
    g = torch.Generator().manual_seed(4)
    a, b = torch.randn(4096, 24, generator=g), torch.randn(4096, 8, generator=g)
    torch.testing.assert_close(A._tn(a, b), a.t() @ b)
    head = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Dropout(0.0), torch.nn.SiLU(), torch.nn.Linear(64, 1)).double()
    x = torch.randn(500, 16, dtype=torch.float64, generator=g, requires_grad=True)
    y_ref = head(x)
    gx_ref, *gp_ref = torch.autograd.grad(y_ref.square().sum(), [x] + list(head.parameters()))
    y = x
    for sub in head:
        y = A._TallLinear.apply(y, sub.weight, sub.bias) if isinstance(sub, torch.nn.Linear) else sub(y)
    gx, *gp = torch.autograd.grad(y.square().sum(), [x] + list(head.parameters()))
    torch.testing.assert_close(y, y_ref)
    torch.testing.assert_close(gx, gx_ref)
    for u, v in zip(gp, gp_ref):
        torch.testing.assert_close(u, v)


def _tail_case(kw, use_mask, dense, dtype, device, n=9, b=2):
    """Inputs of the per-edge tail for one configuration + what autograd of `layer_tail` says (tests below)."""
    from egnn_pytorch_amd import EGNN, autograd as A
    torch.manual_seed(0)
    layer = EGNN(**kw).to(device=device, dtype=dtype)
    with torch.no_grad():
        for nme, p in layer.named_parameters():
            p.mul_(8.0 if nme.startswith("edge_gate") else 40.0)      # (the gate: large enough to matter, not saturated everywhere)
    m = layer.m_dim
    k = n if dense else kw["num_nearest_neighbors"]
    g = torch.Generator().manual_seed(1)
    rnd = lambda *shape: torch.randn(*shape, dtype=torch.float64, generator=g).to(device=device, dtype=dtype)
    feats, coors = rnd(b, n, kw["dim"]), rnd(b, n, 3)
    idx = None if dense else torch.randint(0, n, (b, n, k), generator=g).to(device)
    rank = None if dense else torch.rand(b, n, k, generator=g, dtype=torch.float64).to(device=device, dtype=dtype)
    radius = float("inf") if dense else 0.7
    mask = (torch.arange(n)[None] < torch.tensor([n, n - 3, n // 2 + 1][:b])[:, None]).to(device) if use_mask else None
    u = rnd(b, n, k, m).requires_grad_(True)
    rel_leaf = A.edge_scalars(layer, coors, None, idx)[0].detach().requires_grad_(True)
    out_n, out_c = A.layer_tail(layer, feats, coors, u, rel_leaf, mask, idx, rank, radius)
    gn, gc = rnd(*out_n.shape), rnd(*out_c.shape)
    names = [nme for nme, _ in layer.named_parameters() if nme.startswith(("coors_mlp", "coors_norm", "edge_gate"))]
    tparams = [p for nme, p in layer.named_parameters() if nme in names]
    grads = torch.autograd.grad([out_n, out_c], [u, rel_leaf] + tparams, [gn, gc], allow_unused=True)
    ref = dict(g_u=grads[0], g_rel=grads[1], **dict(zip(names, grads[2:])))
    # what the closed form takes: the pair mask and d loss / d (masked sum over k of m_ij) from the node-level graph
    pm = None
    if mask is not None:
        bi = torch.arange(b, device=device)[:, None, None]
        pm = mask[:, :, None] & mask[:, None, :] if dense else mask[:, :, None] & mask[bi, idx] & (rank <= radius)
    mm = torch.nn.functional.silu(u.detach())
    if layer.edge_gate is not None:
        mm = mm * layer.edge_gate(mm).detach()
    mmask = mm if pm is None else mm.masked_fill(~pm[..., None], 0.0)
    cnt = None
    if layer.m_pool_method == "mean" and pm is not None:
        cnt = pm.sum(-1, keepdim=True).to(dtype)
        m_i = (mmask.sum(2) / cnt.clamp(min=1e-8)).masked_fill(cnt == 0, 0.0)
    else:
        m_i = mmask.mean(2) if layer.m_pool_method == "mean" else mmask.sum(2)
    m_leaf = m_i.clone().requires_grad_(True)
    g_mi = torch.autograd.grad(layer.node_mlp(torch.cat((layer.node_norm(feats), m_leaf), -1)) + feats, m_leaf, gn)[0]
    if layer.m_pool_method == "mean":
        g_msum = (g_mi / cnt.clamp(min=1e-8)).masked_fill(cnt == 0, 0.0) if cnt is not None else g_mi / k
    else:
        g_msum = g_mi
    return layer, u.detach(), coors, idx, pm, gc, g_msum, ref


TAIL_CASES = [(dict(dim=8, num_nearest_neighbors=5), False, False),
              (dict(dim=8, num_nearest_neighbors=6, norm_coors=True, coor_weights_clamp_value=0.6), True, False),
              (dict(dim=8, m_dim=8, m_pool_method="mean", norm_coors=True), True, True),
              (dict(dim=8, num_nearest_neighbors=4, m_pool_method="mean", coor_weights_clamp_value=1.5), True, False),
              (dict(dim=8, num_nearest_neighbors=5, soft_edges=True, norm_coors=True), True, False),
              (dict(dim=8, m_dim=12, soft_edges=True, m_pool_method="mean"), False, True)]


def _tail_param_grads(r):
    return {"coors_mlp.0.weight": r["g_hid"].t() @ r["m"], "coors_mlp.0.bias": r["g_hid"].sum(0),
            "coors_mlp.3.weight": r["g_w"][None, :] @ r["a3"], "coors_mlp.3.bias": r["g_w"].sum()[None],
            "coors_norm.scale": None if r["g_scale"] is None else r["g_scale"].sum()[None],
            "edge_gate.0.weight": None if r["g_gate"] is None else r["g_gate"][None, :] @ r["m0"],
            "edge_gate.0.bias": None if r["g_gate"] is None else r["g_gate"].sum()[None]}


@pytest.mark.parametrize("kw,use_mask,dense", TAIL_CASES)
def test_closed_form_tail_backward_equals_autograd(kw, use_mask, dense):
    """autograd.tail_edge_backward -- the specification of egnn_edge_tail_bwd_f32: second SiLU, pair mask, coors_mlp, CoorsNorm,
    clamp, coordinate update and pooling differentiated by hand -- against autograd of `layer_tail` in float64."""
    from egnn_pytorch_amd import autograd as A
    layer, u, coors, idx, pm, gc, g_msum, ref = _tail_case(kw, use_mask, dense, torch.float64, "cpu")
    r = A.tail_edge_backward(layer, u, coors, idx, pm, gc, g_msum)
    got = dict(g_u=r["g_u"], g_rel=r["g_rel"], **_tail_param_grads(r))
    assert float(ref["coors_mlp.3.weight"].abs().max()) > 0                       # (the clamp leaves some weights active)
    for key, want in ref.items():
        torch.testing.assert_close(got[key].reshape(want.shape), want, rtol=1e-9, atol=1e-9 * max(1.0, float(want.abs().max())), msg=key)


@pytest.mark.gpu
@pytest.mark.parametrize("kw,use_mask,dense", TAIL_CASES)
def test_tail_kernel_matches_closed_form(kw, use_mask, dense):
    """egnn_edge_tail_bwd_f32 (one edge per lane) against its torch specification evaluated in float64 on the same inputs."""
    from egnn_pytorch_amd import _ops, autograd as A
    layer, u, coors, idx, pm, gc, g_msum, _ = _tail_case(kw, use_mask, dense, torch.float32, "cuda", n=40, b=3)
    b, n, k, m = u.shape
    layer64 = __import__("copy").deepcopy(layer).double()
    want = A.tail_edge_backward(layer64, u.double(), coors.double(), idx, pm, gc.double(), g_msum.double())
    e = b * n * k
    u16 = torch.zeros(b, n, k, 16, device="cuda"); u16[..., :m] = u
    gm16 = torch.zeros(b, n, 16, device="cuda"); gm16[..., :m] = g_msum
    la, lb = layer.coors_mlp[0], layer.coors_mlp[3]
    hid3 = la.weight.shape[0]
    w3p = torch.zeros(64, 16, device="cuda"); w3p[:hid3, :m] = la.weight.detach()
    b3p = torch.zeros(64, device="cuda"); b3p[:hid3] = la.bias.detach()
    w4p = torch.zeros(64, device="cuda"); w4p[:hid3] = lb.weight.detach()[0]
    norm = layer.norm_coors
    gate = None
    if layer.edge_gate is not None:
        gw16 = torch.zeros(16, device="cuda"); gw16[:m] = layer.edge_gate[0].weight.detach()[0]
        gate = (gw16, layer.edge_gate[0].bias.detach().contiguous())
    out = _ops.edge_tail_bwd(
        u16, coors.contiguous(), None if idx is None else idx.to(torch.int32).contiguous(), None if pm is None else pm.contiguous().view(torch.uint8),
        gc.contiguous(), gm16, w3p, b3p, w4p, lb.bias.detach().contiguous(), layer.coors_norm.scale.detach() if norm else None,
        layer.coors_norm.eps if norm else 0.0, layer.coor_weights_clamp_value, b, n, k, gate=gate)
    gu, g_rel, g_hid, a3, g_w, g_sc = out[:6]
    got = dict(g_u=gu[:, :m], g_rel=g_rel[:, :3], g_hid=g_hid[:, :hid3], a3=a3[:, :hid3], g_w=g_w, g_scale=g_sc)
    if gate is not None:
        got["g_gate"] = out[6]
    # (self pairs: the kernel writes their d/d rel as zero -- the two signed copies cancel identically at x_i)
    j = torch.arange(n, device="cuda")[None, None, :].expand(b, n, n) if idx is None else idx
    self_pair = (j == torch.arange(n, device="cuda")[None, :, None])
    want["g_rel"] = want["g_rel"].masked_fill(self_pair[..., None], 0.0)
    for key, t in got.items():
        if t is None:
            assert want[key] is None
            continue
        ref = want[key].reshape(t.shape)
        scale = max(1e-30, float(ref.abs().max()))
        err = float((t.double() - ref).abs().max())
        assert err <= 2e-6 * scale, (key, err, scale)
    assert float(gu[:, m:].abs().max() if m < 16 else 0.0) == 0.0 and float(g_rel[:, 3].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("kw,use_mask,dense", TAIL_CASES)
def test_tail_kernel_reduce_mode_and_pool_match_closed_form(kw, use_mask, dense):
    """egnn_edge_tail_bwd_f32 with `part` (every wave sums its edges' parameter-gradient terms; egnn_sum_parts_f32 adds the rows up)
    against the float64 specification: gU / g_rel as in the plain mode, the summed terms against the spec's tall products, the
    by-products rel / dist, two runs bit-identical; and egnn_edge_pool_f32 against the masked, gated message sum."""
    from egnn_pytorch_amd import _ops, autograd as A
    layer, u, coors, idx, pm, gc, g_msum, _ = _tail_case(kw, use_mask, dense, torch.float32, "cuda", n=40, b=3)
    b, n, k, m = u.shape
    layer64 = __import__("copy").deepcopy(layer).double()
    want = A.tail_edge_backward(layer64, u.double(), coors.double(), idx, pm, gc.double(), g_msum.double())
    e = b * n * k
    u16 = torch.zeros(b, n, k, 16, device="cuda"); u16[..., :m] = u
    gm16 = torch.zeros(b, n, 16, device="cuda"); gm16[..., :m] = g_msum
    la, lb = layer.coors_mlp[0], layer.coors_mlp[3]
    hid3 = la.weight.shape[0]
    w3p = torch.zeros(64, 16, device="cuda"); w3p[:hid3, :m] = la.weight.detach()
    b3p = torch.zeros(64, device="cuda"); b3p[:hid3] = la.bias.detach()
    w4p = torch.zeros(64, device="cuda"); w4p[:hid3] = lb.weight.detach()[0]
    norm = layer.norm_coors
    gate = None
    if layer.edge_gate is not None:
        gw16 = torch.zeros(16, device="cuda"); gw16[:m] = layer.edge_gate[0].weight.detach()[0]
        gate = (gw16, layer.edge_gate[0].bias.detach().contiguous())
    pm8 = None if pm is None else pm.contiguous().view(torch.uint8)
    args = (u16, coors.contiguous(), None if idx is None else idx.to(torch.int32).contiguous(), pm8, gc.contiguous(), gm16, w3p, b3p, w4p,
            lb.bias.detach().contiguous(), layer.coors_norm.scale.detach() if norm else None, layer.coors_norm.eps if norm else 0.0,
            layer.coor_weights_clamp_value, b, n, k)
    gu, g_rel, sums, rel, dist, gu_bits = _ops.edge_tail_bwd(*args, gate=gate, reduce=True, want_rel=True)
    assert _ops.bits_to_floats(gu_bits)[0] == float(gu.abs().max())
    again = _ops.edge_tail_bwd(*args, gate=gate, reduce=True, want_rel=True)
    assert torch.equal(sums, again[2]) and torch.equal(gu, again[0])
    j_ = torch.arange(n, device="cuda")[None, None, :].expand(b, n, n) if idx is None else idx
    self_pair = (j_ == torch.arange(n, device="cuda")[None, :, None])
    for got_t, ref_t in ((gu[:, :m], want["g_u"].reshape(e, m)), (g_rel[:, :3], want["g_rel"].masked_fill(self_pair[..., None], 0.0).reshape(e, 3))):
        assert float((got_t.double() - ref_t).abs().max()) <= 2e-6 * max(1e-30, float(ref_t.abs().max()))      # coors_mlp on the matrix cores (split f16 x 3)
    pg = _tail_param_grads(want)
    got = {"coors_mlp.0.weight": sums[:1024].view(64, 16)[:hid3, :m], "coors_mlp.0.bias": sums[1024:1024 + hid3],
           "coors_mlp.3.weight": sums[1088:1088 + hid3][None], "coors_mlp.3.bias": sums[1184:1185],
           "coors_norm.scale": sums[1185:1186] if norm else None,
           "edge_gate.0.weight": sums[1168:1168 + m][None] if gate is not None else None,
           "edge_gate.0.bias": sums[1186:1187] if gate is not None else None}
    l1 = {"coors_mlp.3.bias": "g_w", "coors_norm.scale": "g_scale", "edge_gate.0.bias": "g_gate"}
    for key, t in got.items():
        if t is None:
            continue
        ref = pg[key].reshape(t.shape)
        scale = max(1e-30, float(ref.abs().max()))
        if key in l1:                                   # a scalar: a sum of E signed terms that may cancel -- measured against their L1 norm
            scale = max(scale, float(want[l1[key]].abs().sum()))
        assert float((t.double() - ref).abs().max()) <= 2e-5 * scale, key
    col = want["g_u"].reshape(e, m).sum(dim=0)
    assert float((sums[1152:1152 + m].double() - col).abs().max()) <= 2e-5 * max(1e-30, float(col.abs().max()))
    assert float(sums[1187:].abs().max()) == 0.0
    j = torch.arange(n, device="cuda")[None, None, :].expand(b, n, n) if idx is None else idx
    bi = torch.arange(b, device="cuda")[:, None, None]
    rel_ref = (coors[:, :, None, :] - coors[bi, j]).reshape(e, 3)
    assert torch.equal(rel[:, :3], rel_ref) and float(rel[:, 3].abs().max()) == 0.0
    torch.testing.assert_close(dist, (rel_ref.double() ** 2).sum(-1).float(), rtol=1e-6, atol=1e-6)
    # pooled messages
    mm = torch.nn.functional.silu(u16.double())
    if gate is not None:
        mm = mm * torch.sigmoid(mm @ gate[0].double() + gate[1].double())[..., None]
    if pm is not None:
        mm = mm.masked_fill(~pm[..., None], 0.0)
    pooled = _ops.edge_pool(u16, gate, pm8, b, n, k)
    assert float((pooled.double() - mm.sum(dim=2)).abs().max()) <= 2e-6 * float(mm.sum(dim=2).abs().max())


@pytest.mark.gpu
def test_backward_full_size_properties():
    """The native backward at the north-star size (B=64, N=1024, dim=512, k=32, ragged masks) through size-independent properties:
    two runs bit-identical (every sum over edges has a fixed order, no float atomics), the backward is linear in the cotangent
    (2 g -> exactly 2 x every gradient: powers of two pass through all the scaled fp16 splits unchanged), a sub-batch reproduces its
    graphs' input gradients, and the input gradients are equivariant: translating the coordinates leaves them unchanged, rotating
    the coordinates rotates d/d coors and leaves d/d feats alone."""
    from egnn_pytorch_amd import EGNN
    torch.manual_seed(11)
    layer = EGNN(dim=512, num_nearest_neighbors=32).cuda()
    with torch.no_grad():
        for p in layer.parameters():
            p.mul_(30.0)                                                    # (N(0, 1e-3) init: make the hidden activations O(1))
    b, n = 64, 1024
    g = torch.Generator().manual_seed(12)
    feats, coors = torch.randn(b, n, 512, generator=g).cuda(), torch.randn(b, n, 3, generator=g).cuda()
    lens = torch.randint(n // 2, n + 1, (b,), generator=g)
    mask = (torch.arange(n)[None] < lens[:, None]).cuda()
    rn, rc = torch.randn(b, n, 512, generator=g).cuda(), torch.randn(b, n, 3, generator=g).cuda()

    def grads(f, c, m, gn, gc):
        f, c = f.clone().requires_grad_(True), c.clone().requires_grad_(True)
        node, co = layer(f, c, mask=m)
        return torch.autograd.grad([node, co], [f, c] + list(layer.parameters()), [gn, gc])

    base = grads(feats, coors, mask, rn, rc)
    again = grads(feats, coors, mask, rn, rc)
    assert all(torch.equal(x, y) for x, y in zip(base, again))
    twice = grads(feats, coors, mask, 2 * rn, 2 * rc)
    assert all(torch.equal(2 * x, y) for x, y in zip(base, twice))
    sub = grads(feats[:5], coors[:5], mask[:5], rn[:5], rc[:5])
    for x, y in zip(sub[:2], base[:2]):
        scale = float(y[:5].abs().max())
        assert float((x - y[:5]).abs().max()) <= 1e-5 * scale            # (a chunk's workgroup partition differs: order of the partial sums)
    valid = mask[..., None]
    shift = torch.tensor([0.5, -0.25, 1.0], device="cuda")
    moved = grads(feats, coors + shift, mask, rn, rc)
    for x, y in zip(moved[:2], base[:2]):
        scale = float(y.abs().max())
        assert float(((x - y) * valid).abs().max()) <= 2e-4 * scale
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g).double())
    q = q.float().cuda()
    turned = grads(feats, coors @ q, mask, rn, rc @ q)                       # loss = <node, rn> + <coors_out, rc>: rotate rc with the frame
    scale = float(base[1].abs().max())
    assert float(((turned[1] - base[1] @ q) * valid).abs().max()) <= 2e-4 * scale
    assert float(((turned[0] - base[0]) * valid).abs().max()) <= 2e-4 * float(base[0].abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,b,n", [
    ("north_star", dict(dim=512, num_nearest_neighbors=32), 4, 1024),
    ("c3_layer", dict(dim=128, num_nearest_neighbors=32, norm_feats=True), 4, 1024),
    ("c5_layer", dict(dim=256, num_nearest_neighbors=32, norm_feats=True, norm_coors=True), 2, 1024),
    ("c4_layer", dict(dim=512, edge_dim=4, only_sparse_neighbors=True), 2, 256),
])
def test_backward_at_baseline_widths_matches_the_reference_autograd(ref, name, kw, b, n):
    """the backward pinned at the BASELINE widths.  Gradients of feats, coors and EVERY parameter of
    the north-star layer (dim 512: Hp = 2080 = 17 persistent column chunks, split-K partials), the c3 layer (dim 128), the c5
    layer (dim 256, CoorsNorm) and the c4 layer (dim 512, 4 edge features, adjacency) through the HIP forward + native backward,
    against the REFERENCE module's own autograd run on the MI355X in float64 (oracle/_ref .cuda().double(): float64 so that the
    comparison measures our fp32-class arithmetic, not the eager fp32 run's own rounding -- with CoorsNorm the self pair's
    1e8-sized cancelling terms leave O(1) noise in any fp32 autograd), xavier-scale weights, ragged masks: 1e-4 of each
    gradient's scale."""
    import zlib
    from egnn_pytorch_amd import EGNN
    torch.manual_seed(zlib.crc32(name.encode()))
    rlayer = ref.EGNN(**kw)
    for mod in rlayer.modules():
        if isinstance(mod, torch.nn.Linear):
            torch.nn.init.xavier_normal_(mod.weight)
    layer = EGNN(**kw)
    layer.load_state_dict(rlayer.state_dict(), strict=True)
    layer = layer.cuda()
    rlayer = rlayer.double().cuda()
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) + 1)
    feats, coors = torch.randn(b, n, kw["dim"], generator=g).cuda(), torch.randn(b, n, 3, generator=g).cuda()
    lens = torch.randint(3 * n // 4, n + 1, (b,), generator=g)
    mask = (torch.arange(n)[None] < lens[:, None]).cuda()
    edges = adj = None
    if kw.get("edge_dim"):
        edges = torch.randn(b, n, n, kw["edge_dim"], generator=g).cuda()
    if kw.get("only_sparse_neighbors"):
        i = torch.arange(n)
        adj = ((i[:, None] - i[None, :]).abs() <= 1).cuda()                      # the README chain (diagonal included): K = 3
    mk = lambda t, dt: None if t is None else t.clone().to(dt).requires_grad_(True)
    f1, c1, e1 = mk(feats, torch.float32), mk(coors, torch.float32), mk(edges, torch.float32)
    f2, c2, e2 = mk(feats, torch.float64), mk(coors, torch.float64), mk(edges, torch.float64)
    got, s1 = _grads(layer, lambda: layer(f1, c1, e1, mask, adj), (f1, c1, e1))
    want, s2 = _grads(rlayer, lambda: rlayer(f2, c2, e2, mask, adj), (f2, c2, e2))
    assert s1 == s2
    names = ["feats", "coors"] + (["edges"] if edges is not None else []) + [k for k, _ in layer.named_parameters()]
    worst = {}
    for nm, gg, ww in zip(names, got, want):
        assert (gg is None) == (ww is None), nm
        if gg is None:
            continue
        scale = float(ww.abs().max())
        assert scale > 0, nm                                                      # (a vanishing gradient would make the test vacuous)
        worst[nm] = float((gg.double() - ww).abs().max()) / scale
    if os.environ.get("EGNN_TEST_VERBOSE"):
        print(name, {k: f"{v:.1e}" for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if not v <= 1e-4}
    assert not bad, (name, bad)


@pytest.mark.gpu
@pytest.mark.parametrize("kw,cdim", [(dict(dim=24, num_nearest_neighbors=6), 12), (dict(dim=24, m_dim=80, num_nearest_neighbors=6), 3),
                                     (dict(dim=16, fourier_features=10, num_nearest_neighbors=5), 3)])
def test_shapes_of_the_plain_kernels_train_on_their_own_backward_kernels(ref, kw, cdim, monkeypatch):
    """More than 8 coordinates, heads wider than 64 channels, more than 16 per-edge scalars: the forward runs on the plain kernels and
    -- round 5 -- the backward's E x H work on csrc/edge_exact_bwd.hip with the exact-fp32 GEMMs (`_backward_exact`), never on the ATen
    recompute: gradients against the reference's float64 autograd."""
    from egnn_pytorch_amd import EGNN, autograd as A, _ops

    def no_recompute(*a, **k):
        raise AssertionError("the ATen recompute backward ran")
    monkeypatch.setattr(A, "_backward_recompute", no_recompute)
    torch.manual_seed(12)
    rlayer = ref.EGNN(**kw)
    for mod in rlayer.modules():
        if isinstance(mod, torch.nn.Linear):
            torch.nn.init.xavier_normal_(mod.weight)
    layer = EGNN(**kw)
    layer.load_state_dict(rlayer.state_dict(), strict=True)
    layer, rlayer = layer.cuda(), rlayer.double().cuda()
    g = torch.Generator().manual_seed(13)
    feats, coors = torch.randn(2, 30, kw["dim"], generator=g).cuda(), torch.randn(2, 30, cdim, generator=g).cuda()
    f1, c1 = feats.clone().requires_grad_(True), coors.clone().requires_grad_(True)
    f2, c2 = feats.double().requires_grad_(True), coors.double().requires_grad_(True)
    with _ops.phase_timer() as pt:
        got, _ = _grads(layer, lambda: layer(f1, c1), (f1, c1))
    assert {"edge_exact", "edge_exact_bwd", "edge_exact_node_sums", "bwd_exact_dw2", "bwd_exact_dfeats"} <= set(pt.summary()), set(pt.summary())
    want, _ = _grads(rlayer, lambda: rlayer(f2, c2), (f2, c2))
    for gg, ww in zip(got, want):
        assert (gg is None) == (ww is None)
        if gg is not None:
            assert float((gg.double() - ww).abs().max()) <= 1e-4 * float(ww.abs().max())


@pytest.mark.gpu
def test_the_reference_training_recipe_trains_on_the_float64_kernels(ref, monkeypatch):
    """denoise_sparse.py:11, 23-32, 45-78 -- what SURVEY.md section 8f-2 cites as the reason for a backward: a FLOAT64
    EGNN_Network(num_tokens=21, num_positions=600, depth=5, dim=8, num_nearest_neighbors=16, fourier_features=2, norm_coors,
    coor_weights_clamp_value=2) on a chain adjacency with a mask, MSE of the denoised coordinates, Adam.  Forward: the float64
    kernels; backward: `_backward_exact` -- csrc/edge_exact_bwd.hip + egnn_linear_f64 -- with the ATen recompute disabled.  Gradients of
    every parameter against the REFERENCE network's float64 autograd at 1e-7 of each gradient's scale, and three Adam steps side by
    side with the reference (same losses)."""
    from egnn_pytorch_amd import EGNN_Network, autograd as A, _ops

    def no_recompute(*a, **k):
        raise AssertionError("the ATen recompute backward ran")
    monkeypatch.setattr(A, "_backward_recompute", no_recompute)
    kw = dict(num_tokens=21, num_positions=600, depth=5, dim=8, num_nearest_neighbors=16, fourier_features=2, norm_coors=True,
              coor_weights_clamp_value=2.0)
    torch.manual_seed(5)
    rnet = ref.EGNN_Network(**kw).double()
    net = EGNN_Network(**kw).double()
    net.load_state_dict(rnet.state_dict(), strict=True)
    net, rnet = net.cuda(), rnet.cuda()
    g = torch.Generator().manual_seed(6)
    n = 3 * 40                                               # 40 residues x 3 backbone atoms (the script's repeat(... c = 3))
    seq = torch.randint(0, 21, (1, n // 3), generator=g).repeat_interleave(3, dim=1).cuda()
    coords = (torch.randn(1, n, 3, generator=g, dtype=torch.float64) * 3.0).cuda()
    masks = (torch.arange(n)[None] < n - 6).cuda()
    i = torch.arange(n)
    adj = ((i[:, None] >= (i[None, :] - 1)) & (i[:, None] <= (i[None, :] + 1))).cuda()
    noise = torch.randn(1, n, 3, generator=g, dtype=torch.float64).cuda()

    def loss_of(model):
        feats, den = model(seq, coords + noise, adj_mat=adj, mask=masks)
        return torch.nn.functional.mse_loss(den[masks], coords[masks])

    with _ops.phase_timer() as pt:
        loss = loss_of(net)
        got = torch.autograd.grad(loss, list(net.parameters()), allow_unused=True)
    launched = set(pt.summary())
    assert {"edge_exact", "edge_exact_bwd", "edge_exact_node_sums", "bwd_exact_dw2", "bwd_exact_dws", "bwd_exact_dw1"} <= launched, launched
    rloss = loss_of(rnet)
    want = torch.autograd.grad(rloss, list(rnet.parameters()), allow_unused=True)
    assert abs(float(loss) - float(rloss)) <= 1e-9 * max(1.0, abs(float(rloss)))
    for (nm, _), gg, ww in zip(net.named_parameters(), got, want):
        assert (gg is None) == (ww is None), nm
        if gg is not None and float(ww.abs().max()) > 0:
            assert float((gg - ww).abs().max()) <= 1e-7 * float(ww.abs().max()), (nm, float((gg - ww).abs().max()), float(ww.abs().max()))
    opt, ropt = torch.optim.Adam(net.parameters(), lr=1e-3), torch.optim.Adam(rnet.parameters(), lr=1e-3)
    for _ in range(3):
        for model, o in ((net, opt), (rnet, ropt)):
            o.zero_grad()
            l_ = loss_of(model)
            l_.backward()
            o.step()
            model.last_loss = float(l_)
        assert abs(net.last_loss - rnet.last_loss) <= 1e-7 * max(1.0, abs(rnet.last_loss))


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,b,n,cdim", [
    ("m32_knn8", dict(dim=64, m_dim=32, num_nearest_neighbors=8), 3, 96, 3),
    ("m32_knn32_normfeats", dict(dim=128, m_dim=32, num_nearest_neighbors=32, norm_feats=True), 2, 256, 3),
    ("m64_dense_gate", dict(dim=32, m_dim=64, soft_edges=True), 2, 40, 3),
    ("m20_edges_fourier", dict(dim=48, m_dim=20, num_nearest_neighbors=11, edge_dim=3, fourier_features=2), 2, 64, 3),
    ("c5_knn8", dict(dim=64, num_nearest_neighbors=8), 3, 96, 5),
    ("c5_dense_edges", dict(dim=512, edge_dim=4), 1, 16, 5),                       # the reference's test_higher_dimension shape (:36-45)
    ("c2_m32_mean", dict(dim=32, m_dim=32, num_nearest_neighbors=6, m_pool_method="mean"), 2, 50, 2),
    ("c8_knn32", dict(dim=64, num_nearest_neighbors=32), 2, 128, 8),
])
def test_native_backward_for_wide_heads_and_other_coordinate_dimensions(ref, name, kw, b, n, cdim, monkeypatch):
    """VERDICT r3 missing #3 / next #5: m_dim > 16 and coordinate dimensions other than 3 used to fall to the chunked ATen recompute of
    the whole layer.  Now the forward kernels write u for them too and the E x H work runs on egnn_edge_bwd_pass_f32 (once per block
    of 16 message channels -- the pass is linear in gU); the per-edge chain behind u goes through autograd on E x m tensors.  Gradients
    of the inputs and of every parameter against the REFERENCE module's autograd in float64, xavier-scale weights, ragged masks, 1e-4
    of each gradient's scale; the recompute path must not be entered."""
    import zlib
    from egnn_pytorch_amd import EGNN, autograd as A

    def no_recompute(*a, **k):
        raise AssertionError("the ATen recompute backward was entered")
    monkeypatch.setattr(A, "_backward_recompute", no_recompute)
    torch.manual_seed(zlib.crc32(name.encode()))
    rlayer = ref.EGNN(**kw)
    for mod in rlayer.modules():
        if isinstance(mod, torch.nn.Linear):
            torch.nn.init.xavier_normal_(mod.weight)
    layer = EGNN(**kw)
    layer.load_state_dict(rlayer.state_dict(), strict=True)
    layer = layer.cuda()
    rlayer = rlayer.double().cuda()
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) + 1)
    feats, coors = torch.randn(b, n, kw["dim"], generator=g).cuda(), torch.randn(b, n, cdim, generator=g).cuda()
    lens = torch.randint(3 * n // 4, n + 1, (b,), generator=g)
    mask = (torch.arange(n)[None] < lens[:, None]).cuda()
    edges = torch.randn(b, n, n, kw["edge_dim"], generator=g).cuda() if kw.get("edge_dim") else None
    mk = lambda t, dt: None if t is None else t.clone().to(dt).requires_grad_(True)
    f1, c1, e1 = mk(feats, torch.float32), mk(coors, torch.float32), mk(edges, torch.float32)
    f2, c2, e2 = mk(feats, torch.float64), mk(coors, torch.float64), mk(edges, torch.float64)
    got, s1 = _grads(layer, lambda: layer(f1, c1, e1, mask), (f1, c1, e1))
    want, s2 = _grads(rlayer, lambda: rlayer(f2, c2, e2, mask), (f2, c2, e2))
    assert s1 == s2
    names = ["feats", "coors"] + (["edges"] if edges is not None else []) + [k for k, _ in layer.named_parameters()]
    worst = {}
    for nm, gg, ww in zip(names, got, want):
        assert (gg is None) == (ww is None), nm
        if gg is None:
            continue
        scale = float(ww.abs().max())
        assert scale > 0, nm
        worst[nm] = float((gg.double() - ww).abs().max()) / scale
    bad = {k: v for k, v in worst.items() if not v <= 1e-4}
    assert not bad, (name, bad)


# ---------------------------------------------------------------------------------------------------------------------------
# The host side of the native backward on the CPU: the three kernels replaced by torch emulations of their contracts
# (include/egnn_hip.h), everything else -- entry lists, partial rows, fixed-order sums, chunking over graphs, node-level products,
# parameter bookkeeping -- is the shipped code of autograd._backward_native.
def _emulated_backward(layer, feats, coors, mask, idx, rank, radius, g_node, g_coors, max_graphs=0):
    import types
    from egnn_pytorch_amd import _ops, _weights, autograd as A
    b, n, dim = feats.shape
    k = idx.shape[-1]
    m = layer.m_dim
    w = layer.packed_weights()
    hp, s_in = w["Hp"], w["S"]
    lin0, lin3 = layer.edge_mlp[0], layer.edge_mlp[3]
    mp = 16 * _weights.m_blocks(m)
    w2all = torch.zeros(mp, hp); w2all[:m, :w["H"]] = lin3.weight.detach().float()
    nl2e = _weights.NEG_LOG2E

    def tables(layer_, w_, f2d, pi_split):
        return f2d.float() @ w_["Wcat"].t() + w_["bcat"]                             # fp32 P_i | P_j rows in the forward's units

    def z_of(proj, idx32, scal, b_, n_, k_):
        src = torch.arange(b_ * n_).repeat_interleave(k_)
        dst = (idx32.long() + (torch.arange(b_) * n_)[:, None, None]).reshape(-1)
        return (proj[src, :hp] + proj[dst, hp:] + scal @ w["Ws"]) / nl2e

    def bwd_pass(w_, proj, idx32, gu16, gu_scale, scal, ent, b_, n_, k_, by_dest, ws_nat=None, want_w2=False, n_slabs=None, row_pairs=False,
                 drop=None, eid0=0, want_amax=False):
        assert drop is None
        blk = w_.get("w2_block", 0)                                                 # (m_dim > 16: one call per block of 16 channels)
        w2n = w2all[16 * blk:16 * blk + 16]
        z = z_of(proj, idx32, scal, b_, n_, k_)
        sg = torch.sigmoid(z)
        a = z * sg
        dz = (gu16 @ w2n) * (sg * (1 + z * (1 - sg)))
        tiles = ent.view(-1, 16).long()
        rows = (dz[tiles.clamp(min=0)] * (tiles >= 0)[..., None]).sum(dim=1)        # one partial row per tile
        if row_pairs:                                                               # ... per node: its two tiles summed in the kernel
            rows = rows.view(-1, 2, rows.shape[1]).sum(dim=1)
        out = {"rows": rows}
        if want_w2:
            out["w2"] = gu16.t() @ a
        if ws_nat is not None:
            out["ws"] = dz.t() @ scal
            out["scal"] = dz @ ws_nat
        return out

    def gather_sum(rows, order, seg, n_out):
        out = torch.zeros(n_out, rows.shape[1])
        for r in range(n_out):
            for p in range(int(seg[r]), int(seg[r + 1])):
                out[r] = out[r] + rows[order[p]]
        return out

    def tail(u16, coors_, idx32, pair_mask, g_co, g_msum16, w3p, b3p, w4p, b4, scale, eps, clamp, b_, n_, k_, gate=None):
        r = A.tail_edge_backward(A._f32_shadow(layer), u16[..., :m], coors_, None if idx32 is None else idx32.long(),
                                 None if pair_mask is None else pair_mask.view(torch.bool).view(b_, n_, k_), g_co, g_msum16[..., :m])
        e = b_ * n_ * k_
        gu = torch.zeros(e, 16); gu[:, :m] = r["g_u"].reshape(e, m)
        g_rel = torch.zeros(e, 4); g_rel[:, :3] = r["g_rel"].reshape(e, 3)
        self_pair = (idx32.long() == torch.arange(n_)[None, :, None]).reshape(-1)
        g_rel[self_pair] = 0.0
        gh = torch.zeros(e, 64); gh[:, :r["g_hid"].shape[1]] = r["g_hid"]
        a3 = torch.zeros(e, 64); a3[:, :r["a3"].shape[1]] = r["a3"]
        if gate is not None:
            return gu, g_rel, gh, a3, r["g_w"], r["g_scale"], r["g_gate"]
        return gu, g_rel, gh, a3, r["g_w"], r["g_scale"]

    # u = the second Linear's output, as the forward kernel leaves it (E, 16)
    with torch.no_grad():
        _, scal = A.edge_scalars(layer, coors.float(), None, idx.long())
        proj = tables(layer, w, feats.reshape(b * n, dim), False)
        z = z_of(proj, idx, scal.reshape(-1, s_in), b, n, k)
        u = torch.zeros(b * n * k, mp)
        u[:, :m] = torch.nn.functional.silu(z[:, :w["H"]]) @ lin3.weight.float().t() + lin3.bias.float()
    params = list(layer.parameters())
    ctx = types.SimpleNamespace(layer=layer, has_u=True, valid_radius=radius, has_edges=False, order=None,
                                saved_tensors=(feats, coors, feats.new_empty(0), mask if mask is not None else feats.new_empty(0), idx, rank, u),
                                flags=(mask is not None, True),
                                needs_input_grad=(False, False, False, False, True, True, False) + tuple(p.requires_grad for p in params))
    def dest_lists(idx32, b_, n_, k_, device):
        """egnn_dest_lists_i32's contract (include/egnn_hip.h) from a stable sort"""
        dest = (idx32.long() + (torch.arange(b_) * n_)[:, None, None]).reshape(-1)
        dest_sorted, by_dest = torch.sort(dest, stable=True)
        ent, tile_seg = A.entry_list(by_dest, dest_sorted, b_ * n_)
        seg = torch.searchsorted(dest_sorted, torch.arange(b_ * n_ + 1))
        return _ops.DestLists(ent, tile_seg, by_dest, seg)

    saved = (_ops.edge_bwd_pass, _ops.rows_gather_sum, _ops.edge_tail_bwd, A._edge_tables, A._FUSED_MAX_GRAPHS, _ops.dest_lists)
    _ops.edge_bwd_pass, _ops.rows_gather_sum, _ops.edge_tail_bwd, A._edge_tables, A._FUSED_MAX_GRAPHS, _ops.dest_lists = \
        bwd_pass, gather_sum, tail, tables, max_graphs, dest_lists
    try:
        out = A._backward_native(ctx, g_node, g_coors)
    finally:
        _ops.edge_bwd_pass, _ops.rows_gather_sum, _ops.edge_tail_bwd, A._edge_tables, A._FUSED_MAX_GRAPHS, _ops.dest_lists = saved
    return [out[4], out[5]] + list(out[7:])


@pytest.mark.parametrize("kw,use_mask,max_graphs", [(dict(dim=8, num_nearest_neighbors=5), False, 0),
                                                    (dict(dim=8, num_nearest_neighbors=6, norm_coors=True, coor_weights_clamp_value=0.6), True, 0),
                                                    (dict(dim=8, num_nearest_neighbors=20, m_pool_method="mean", norm_feats=True), True, 2),
                                                    (dict(dim=8, num_nearest_neighbors=7, soft_edges=True, m_dim=12), True, 0),
                                                    (dict(dim=8, num_nearest_neighbors=7, m_dim=20), True, 0),          # two blocks of 16 channels
                                                    (dict(dim=8, num_nearest_neighbors=9, m_dim=40, soft_edges=True), False, 2)])
def test_native_backward_host_logic_with_emulated_kernels(kw, use_mask, max_graphs):
    """autograd._backward_native on the CPU with its three kernels emulated in torch from their header contracts: what remains
    under test is the host side -- entry lists, partial rows, fixed-order sums, chunking over graphs, the node-level products,
    the parameter bookkeeping -- against autograd of the restated layer."""
    from egnn_pytorch_amd import EGNN, autograd as A
    torch.manual_seed(5)
    layer = EGNN(**kw)
    with torch.no_grad():
        for p in layer.parameters():
            p.mul_(40.0)
    b, n = 3, 24
    k = kw["num_nearest_neighbors"]
    g = torch.Generator().manual_seed(6)
    feats, coors = torch.randn(b, n, kw["dim"], generator=g), torch.randn(b, n, 3, generator=g)
    idx = torch.randint(0, n, (b, n, k), generator=g).to(torch.int32)
    idx[:, :, 0] = torch.arange(n)[None, :]                                          # the self pair, as the selection always has it
    rank = torch.rand(b, n, k, generator=g)
    mask = (torch.arange(n)[None] < torch.tensor([n, n - 5, n // 2])[:, None]) if use_mask else None
    gn, gc = torch.randn(b, n, kw["dim"], generator=g), torch.randn(b, n, 3, generator=g)
    got = _emulated_backward(layer, feats, coors, mask, idx, rank, 0.8, gn, gc, max_graphs)
    l64 = __import__("copy").deepcopy(layer).double()
    f, c = feats.double().requires_grad_(True), coors.double().requires_grad_(True)
    node, co = A.layer_given_neighbors(l64, f, c, None, mask, idx.long(), rank.double(), 0.8)
    want = torch.autograd.grad([node, co], [f, c] + list(l64.parameters()), [gn.double(), gc.double()], allow_unused=True)
    for pos, (a, r) in enumerate(zip(got, want)):
        if r is None:
            assert a is None or float(a.abs().max()) == 0.0
            continue
        scale = max(1e-12, float(r.abs().max()))
        assert float((a.double() - r).abs().max()) <= 2e-4 * scale, (pos, float((a.double() - r).abs().max()), scale)


def _emulated_case(kw, dtype=torch.float32, frozen=()):
    from egnn_pytorch_amd import EGNN
    torch.manual_seed(5)
    layer = EGNN(**kw)
    with torch.no_grad():
        for p in layer.parameters():
            p.mul_(40.0)
    layer = layer.to(dtype)
    for name, p in layer.named_parameters():
        if any(name.startswith(f) for f in frozen):
            p.requires_grad_(False)
    b, n, k = 3, 24, kw["num_nearest_neighbors"]
    g = torch.Generator().manual_seed(6)
    feats, coors = torch.randn(b, n, kw["dim"], generator=g), torch.randn(b, n, 3, generator=g)
    idx = torch.randint(0, n, (b, n, k), generator=g).to(torch.int32)
    idx[:, :, 0] = torch.arange(n)[None, :]
    rank = torch.rand(b, n, k, generator=g)
    gn, gc = torch.randn(b, n, kw["dim"], generator=g), torch.randn(b, n, 3, generator=g)
    return layer, feats, coors, idx, rank, gn, gc


@pytest.mark.parametrize("frozen", [("node_mlp",), ("node_mlp", "edge_mlp", "coors_mlp", "node_norm", "coors_norm"),
                                    ("coors_mlp.0", "edge_mlp.3")])
def test_native_backward_with_frozen_parameters(frozen):
    """the native backward handed frozen parameters to torch.autograd.grad, which raises.  Forces through
    a frozen model (every parameter frozen) and partial fine-tuning must give the input gradients and None for what is frozen."""
    from egnn_pytorch_amd import autograd as A
    kw = dict(dim=8, num_nearest_neighbors=6, norm_coors=True, norm_feats=True)
    layer, feats, coors, idx, rank, gn, gc = _emulated_case(kw, frozen=frozen)
    got = _emulated_backward(layer, feats, coors, None, idx, rank, 0.8, gn, gc)
    l64 = __import__("copy").deepcopy(layer).double()
    f, c = feats.double().requires_grad_(True), coors.double().requires_grad_(True)
    node, co = A.layer_given_neighbors(l64, f, c, None, None, idx.long(), rank.double(), 0.8)
    live = [p for p in l64.parameters() if p.requires_grad]
    want = torch.autograd.grad([node, co], [f, c] + live, [gn.double(), gc.double()], allow_unused=True)
    want_by_param = dict(zip([id(p) for p in live], want[2:]))
    for a, r in zip(got[:2], want[:2]):
        assert float((a.double() - r).abs().max()) <= 2e-4 * float(r.abs().max())
    for a, p in zip(got[2:], l64.parameters()):
        if not p.requires_grad:
            assert a is None
        else:
            r = want_by_param[id(p)]
            assert float((a.double() - r).abs().max()) <= 2e-4 * max(1e-12, float(r.abs().max()))


def test_native_backward_of_a_float64_module_goes_through_the_fp32_boundary():
    """The native backward of a module that is not fp32 differentiates an fp32 shadow of it and returns every gradient in the
    module's dtype: half / bfloat16 modules, and a float64 module with training-mode dropout (which keeps the boundary conversion;
    without dropout a float64 module runs on the float64 kernels and the float64 recompute, tests/test_float64_property_suite.py)."""
    from egnn_pytorch_amd import autograd as A
    kw = dict(dim=8, num_nearest_neighbors=6, norm_feats=True)
    layer, feats, coors, idx, rank, gn, gc = _emulated_case(kw, dtype=torch.float64)
    got = _emulated_backward(layer, feats.double(), coors.double(), None, idx, rank, 0.8, gn.double(), gc.double())
    assert all(g.dtype == torch.float64 for g in got)
    f, c = feats.double().requires_grad_(True), coors.double().requires_grad_(True)
    node, co = A.layer_given_neighbors(layer, f, c, None, None, idx.long(), rank.double(), 0.8)
    want = torch.autograd.grad([node, co], [f, c] + list(layer.parameters()), [gn.double(), gc.double()], allow_unused=True)
    for a, r in zip(got, want):
        assert float((a - r).abs().max()) <= 2e-4 * max(1e-12, float(r.abs().max()))
    assert layer.__dict__["_shadow32"][1] is A._f32_shadow(layer)                      # cached per parameter version
    assert not any(n.startswith("_shadow32") for n, _ in layer.named_parameters())       # ... and not registered as a sub-module


def test_recompute_backward_with_inputs_of_another_dtype_and_with_no_neighbours():
    """fp64 inputs into an fp32 module made backward() raise a dtype mismatch; K == 0 on the neighbour path
    (only_sparse_neighbors with an empty adjacency) was differentiated as the dense graph."""
    import types
    from egnn_pytorch_amd import EGNN, autograd as A
    torch.manual_seed(1)
    layer = EGNN(dim=8, num_nearest_neighbors=4)
    b, n = 2, 10
    g = torch.Generator().manual_seed(2)
    feats, coors = torch.randn(b, n, 8, generator=g).double(), torch.randn(b, n, 3, generator=g).double()
    idx = torch.randint(0, n, (b, n, 4), generator=g).to(torch.int32)
    rank = torch.rand(b, n, 4, generator=g)
    none = feats.new_empty(0)
    params = list(layer.parameters())

    def run(idx_, rank_):
        ctx = types.SimpleNamespace(layer=layer, has_u=False, valid_radius=1e9, has_edges=False, order=None,
                                    saved_tensors=(feats, coors, none, none, idx_, rank_, none), flags=(False, True),
                                    needs_input_grad=(False,) * 4 + (True, True, False) + (True,) * len(params))
        return A._backward_recompute(ctx, torch.ones(b, n, 8).double(), torch.ones(b, n, 3).double())

    out = run(idx, rank)
    assert out[4].dtype == torch.float64 and out[5].dtype == torch.float64 and out[7].dtype == torch.float32
    f, c = feats.float().requires_grad_(True), coors.float().requires_grad_(True)
    node, co = A.layer_given_neighbors(layer, f, c, None, None, idx.long(), rank, 1e9)
    want = torch.autograd.grad([node.sum() + co.sum()], [f, c])
    assert float((out[4].float() - want[0]).abs().max()) <= 1e-5 * float(want[0].abs().max())
    # K == 0: no messages -- coordinates pass through (gradient = cotangent), features see node_mlp([h, 0]) + h only
    out0 = run(torch.empty(b, n, 0, dtype=torch.int32), torch.empty(b, n, 0))
    assert torch.equal(out0[5], torch.ones(b, n, 3).double())
    f = feats.float().requires_grad_(True)
    node = layer.node_mlp(torch.cat((layer.node_norm(f), torch.zeros(b, n, layer.m_dim)), dim=-1)) + f
    want0 = torch.autograd.grad([node.sum()], [f])[0]
    assert float((out0[4].float() - want0).abs().max()) <= 1e-5 * float(want0.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("frozen", [("node_mlp",), ("node_mlp", "edge_mlp", "coors_mlp", "node_norm", "coors_norm", "edge_gate"), ("edge_mlp.0", "coors_mlp.3")])
def test_native_backward_on_the_device_with_frozen_parameters(frozen):
    """The device-side native backward (node_mlp on the split-f16 GEMMs, the tail kernel's in-kernel sums) with frozen parameters:
    the input gradients equal those of the unfrozen module, frozen parameters get no gradient, the others the same ones."""
    from egnn_pytorch_amd import EGNN
    kw = dict(dim=32, num_nearest_neighbors=20, soft_edges=True, norm_coors=True, norm_feats=True)
    torch.manual_seed(9)
    layer = EGNN(**kw).cuda()
    g = torch.Generator().manual_seed(2)
    feats, coors = torch.randn(2, 50, 32, generator=g).cuda(), torch.randn(2, 50, 3, generator=g).cuda()
    gn, gc = torch.randn(2, 50, 32, generator=g).cuda(), torch.randn(2, 50, 3, generator=g).cuda()

    def run():
        f, c = feats.clone().requires_grad_(True), coors.clone().requires_grad_(True)
        node, co = layer(f, c)
        layer.zero_grad()
        ((node * gn).sum() + (co * gc).sum()).backward()
        return f.grad.clone(), c.grad.clone(), {k: (None if p.grad is None else p.grad.clone()) for k, p in layer.named_parameters()}

    f0, c0, p0 = run()
    for name, p in layer.named_parameters():
        if any(name.startswith(fr) for fr in frozen):
            p.requires_grad_(False)
    for p in layer.parameters():
        p.grad = None
    f1, c1, p1 = run()
    assert torch.equal(f0, f1) and torch.equal(c0, c1)
    for name, p in layer.named_parameters():
        if p.requires_grad:
            assert torch.equal(p0[name], p1[name]), name
        else:
            assert p1[name] is None, name


@pytest.mark.gpu
def test_backward_twice_over_a_retained_graph_gives_the_same_gradients():
    """The backward decodes the forward's projection table in place, once: a second backward over the retained graph must find it
    decoded (same gradients, bit for bit), and so must torch.autograd.grad called for different inputs."""
    from egnn_pytorch_amd import EGNN
    torch.manual_seed(3)
    layer = EGNN(dim=64, num_nearest_neighbors=32).cuda()
    g = torch.Generator().manual_seed(8)
    f = torch.randn(2, 96, 64, generator=g).cuda().requires_grad_(True)
    c = torch.randn(2, 96, 3, generator=g).cuda().requires_grad_(True)
    node, co = layer(f, c)
    loss = node.square().mean() + co.square().mean()
    g1 = torch.autograd.grad(loss, [f, c] + list(layer.parameters()), retain_graph=True)
    g2 = torch.autograd.grad(loss, [f, c] + list(layer.parameters()), retain_graph=True)
    g3 = torch.autograd.grad(loss, [c])
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))
    assert torch.equal(g1[1], g3[0])
