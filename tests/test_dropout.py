"""Training-mode dropout (egnn_pytorch.py:176: one nn.Dropout shared by edge_mlp, node_mlp and coors_mlp, each time behind the
first Linear) on the gfx950 path -- VERDICT r2 missing #3.  The masks are a counter-based hash (csrc/egnn_common.h) whose torch
twin is egnn_pytorch_amd/_dropout.py: no two dropout implementations share a random stream, so parity with the reference is
(a) identical semantics -- Bernoulli(1 - p) keep, 1 / (1 - p) rescale, behind the first Linear of each MLP, fresh per call,
training mode only -- checked against a float64 restatement that applies the SAME masks, and (b) the mask's statistics."""
import copy

import numpy as np
import pytest
import torch


def _py_hash(seed, site, row, col):
    m = 0xFFFFFFFF
    x = (row * 0x9E3779B1 + seed + site * 0x27D4EB2F + col * 0x85EBCA77) & m
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & m
    x ^= x >> 12
    x = (x * 0x297A2D39) & m
    x ^= x >> 15
    return x


def test_hash_twin_equals_the_c_definition_and_is_bernoulli():
    from egnn_pytorch_amd import _dropout as D
    rows = torch.tensor([0, 1, 5, 2 ** 31 - 5, 123456789, 4294967295])
    cols = torch.tensor([0, 1, 2049, 63])
    h = D.hash32(12345, 2, rows, cols)
    for i in range(rows.numel()):
        for j in range(cols.numel()):
            assert int(h[i, j]) == _py_hash(12345, 2, int(rows[i]), int(cols[j]))
    for p in (0.1, 0.5, 0.9):
        keep = D.keep_mask(99, D.SITE_EDGE, torch.arange(3000), torch.arange(700), p).float()
        assert abs(float(keep.mean()) - (1 - p)) < 2e-3
        assert abs(float(keep.mean(dim=0).std())) < 0.03 and abs(float(keep.mean(dim=1).std())) < 0.05      # no dead rows / columns
        # neighbouring rows / columns are uncorrelated
        a, b = keep[:-1] - (1 - p), keep[1:] - (1 - p)
        assert abs(float((a * b).mean())) < 2e-3
        a, b = keep[:, :-1] - (1 - p), keep[:, 1:] - (1 - p)
        assert abs(float((a * b).mean())) < 2e-3
    assert not torch.equal(D.keep_mask(1, 0, torch.arange(100), torch.arange(64), 0.5), D.keep_mask(2, 0, torch.arange(100), torch.arange(64), 0.5))
    assert not torch.equal(D.keep_mask(1, 0, torch.arange(100), torch.arange(64), 0.5), D.keep_mask(1, 1, torch.arange(100), torch.arange(64), 0.5))


def test_restatement_with_masks_reduces_to_the_plain_layer_at_tiny_p():
    """CPU: layer_given_neighbors(drop=(p, seed)) keeps everything for p -> 0 and then equals the eval layer (times 1 / (1 - p) ~ 1)."""
    from egnn_pytorch_amd import EGNN, autograd as A
    torch.manual_seed(0)
    layer = EGNN(dim=8, num_nearest_neighbors=4, dropout=0.5).double()
    with torch.no_grad():
        for prm in layer.parameters():
            prm.mul_(40.0)
    g = torch.Generator().manual_seed(1)
    feats, coors = torch.randn(2, 10, 8, generator=g).double(), torch.randn(2, 10, 3, generator=g).double()
    idx = torch.randint(0, 10, (2, 10, 4), generator=g)
    rank = torch.rand(2, 10, 4, generator=g).double()
    layer.eval()
    with torch.no_grad():
        plain = A.layer_given_neighbors(layer, feats, coors, None, None, idx, rank, 1e9)
        tiny = A.layer_given_neighbors(layer, feats, coors, None, None, idx, rank, 1e9, drop=(1e-12, 5))
        half = A.layer_given_neighbors(layer, feats, coors, None, None, idx, rank, 1e9, drop=(0.5, 5))
        half2 = A.layer_given_neighbors(layer, feats, coors, None, None, idx, rank, 1e9, drop=(0.5, 5))
        # a chunk that starts at graph 1 sees the same masks as graph 1 inside the full batch
        part = A.layer_given_neighbors(layer, feats[1:], coors[1:], None, None, idx[1:], rank[1:], 1e9, drop=(0.5, 5), graph_offset=1)
    assert torch.allclose(plain[0], tiny[0], atol=1e-9) and torch.allclose(plain[1], tiny[1], atol=1e-9)
    assert not torch.allclose(plain[0], half[0], atol=1e-3)
    assert torch.equal(half[0], half2[0])
    assert torch.allclose(part[0], half[0][1:], atol=1e-12) and torch.allclose(part[1], half[1][1:], atol=1e-12)


CASES = [
    (dict(dim=64, num_nearest_neighbors=32, dropout=0.25, norm_feats=True), 96, True),            # one node per wave
    (dict(dim=32, num_nearest_neighbors=8, dropout=0.1, norm_coors=True, soft_edges=True), 40, True),   # four nodes per tile
    (dict(dim=32, dropout=0.5, m_pool_method="mean"), 20, False),                                 # dense all-pairs
    (dict(dim=24, num_nearest_neighbors=5, dropout=0.3, m_dim=8, coor_weights_clamp_value=1.0), 30, True),   # P_i on the VALU
    (dict(dim=32, num_nearest_neighbors=16, dropout=0.2, edge_dim=3, fourier_features=1), 48, True),
    # round 5: the shapes beyond the standard layer's (their own translation units of csrc/edge_fused.hip, -DEGNN_EDGE_DROP_TU):
    # two and four accumulator tiles per edge tile, the other coordinate dimensions (the reference's own test uses 5)
    (dict(dim=64, num_nearest_neighbors=32, dropout=0.25, m_dim=32), 80, True),
    (dict(dim=32, dropout=0.3, m_dim=48, soft_edges=True), 24, False),
    (dict(dim=32, num_nearest_neighbors=8, dropout=0.2, norm_coors=True, cdim=5), 40, True),
    (dict(dim=24, num_nearest_neighbors=12, dropout=0.15, m_dim=24, edge_dim=2, cdim=2), 36, False),
    # ... and the shapes of the plain kernels (csrc/edge_exact.hip: the same hash masks; egnn_drop_silu_f32 for node_mlp): more than 8
    # coordinates, a head wider than 64 channels (the blocked kernel), more than 16 per-edge scalars
    (dict(dim=24, num_nearest_neighbors=8, dropout=0.2, norm_coors=True, cdim=9), 30, True),
    (dict(dim=16, num_nearest_neighbors=6, dropout=0.25, m_dim=80, soft_edges=True), 24, False),
    (dict(dim=16, dropout=0.3, fourier_features=8, edge_dim=2, norm_feats=True), 14, True),
]


def _layer_kw(kw):
    return {k: v for k, v in kw.items() if k != "cdim"}


@pytest.mark.gpu
@pytest.mark.parametrize("kw,n,use_mask", CASES)
def test_training_mode_forward_applies_exactly_the_hash_masks(kw, n, use_mask):
    """HIP forward in training mode with a given seed == the float64 restatement of the layer with the SAME masks at the three
    dropout sites (1e-4, like every parity test): pins where the masks sit, their row / unit indexing and the rescaling."""
    from egnn_pytorch_amd import EGNN, autograd as A
    torch.manual_seed(11)
    layer = EGNN(**_layer_kw(kw))
    with torch.no_grad():
        for prm in layer.parameters():
            prm.mul_(40.0)
    layer = layer.cuda().train()
    g = torch.Generator().manual_seed(3)
    b = 3
    feats, coors = torch.randn(b, n, kw["dim"], generator=g).cuda(), torch.randn(b, n, kw.get("cdim", 3), generator=g).cuda()
    mask = (torch.arange(n)[None] < torch.tensor([[n], [n - 3], [n // 2 + 2]])).cuda() if use_mask else None
    edges = torch.randn(b, n, n, kw["edge_dim"], generator=g).cuda() if kw.get("edge_dim") else None
    seed = 424242
    with torch.no_grad():
        node, co, _, idx, rank, radius = layer._forward_hip_checked(feats, coors, edges, mask, None, None, drop_seed=seed)[:6]
        node2, co2 = layer._forward_hip_checked(feats, coors, edges, mask, None, None, drop_seed=seed)[:2]
        node3 = layer._forward_hip_checked(feats, coors, edges, mask, None, None, drop_seed=seed + 1)[0]
    assert torch.equal(node, node2) and torch.equal(co, co2)                      # same seed: same masks
    assert not torch.allclose(node, node3, atol=1e-3)                             # another seed: other masks
    l64 = copy.deepcopy(layer).double()
    with torch.no_grad():
        want = A.layer_given_neighbors(l64, feats.double(), coors.double(), None if edges is None else edges.double(), mask,
                                       None if idx is None else idx.long(), None if rank is None else rank.double(), radius,
                                       drop=(kw["dropout"], seed))
    np.testing.assert_allclose(node.cpu().numpy(), want[0].cpu().numpy(), atol=1e-4, rtol=0)
    np.testing.assert_allclose(co.cpu().numpy(), want[1].cpu().numpy(), atol=1e-4, rtol=0)
    # eval mode: no dropout, no seed drawn
    layer.eval()
    st = torch.random.get_rng_state()
    with torch.no_grad():
        e1 = layer(feats, coors, edges, mask)
    assert torch.equal(st, torch.random.get_rng_state())
    with torch.no_grad():
        plain = A.layer_given_neighbors(l64.eval(), feats.double(), coors.double(), None if edges is None else edges.double(), mask,
                                        None if idx is None else idx.long(), None if rank is None else rank.double(), radius)
    np.testing.assert_allclose(e1[0].cpu().numpy(), plain[0].cpu().numpy(), atol=1e-4, rtol=0)


BWD_CASES = [
    # (the first three take the NATIVE backward -- the hash masks re-evaluated inside egnn_edge_bwd_pass_f32, the matrix-core tail kernel
    # and egnn_silu_bwd_drop_f32: one tile per node, two tiles per node summed in the kernel, the gate + CoorsNorm + mean pooling + masks;
    # per-edge features + a fourier pair = five scalars; nine scalars)
    (dict(dim=32, num_nearest_neighbors=8, dropout=0.2, norm_feats=True), 40, False, True),
    (dict(dim=64, num_nearest_neighbors=32, dropout=0.25), 96, True, True),
    (dict(dim=32, num_nearest_neighbors=20, dropout=0.1, norm_coors=True, soft_edges=True, m_pool_method="mean", coor_weights_clamp_value=2.0), 50, True, True),
    (dict(dim=32, num_nearest_neighbors=16, dropout=0.2, edge_dim=2, fourier_features=1), 48, True, True),
    (dict(dim=24, num_nearest_neighbors=8, dropout=0.3, fourier_features=2, edge_dim=4, m_dim=8), 30, False, True),
    # round 5: wide heads / other coordinate dimensions -- the forward's masks in the kernels; the backward native as well: the E x H
    # passes once per block of 16 channels with the same mask of z, coors_mlp's mask re-evaluated by the generic tail kernel
    # (egnn_edge_tail_exact_bwd_f32 with drop_thr)
    (dict(dim=32, num_nearest_neighbors=16, dropout=0.2, m_dim=32), 40, True, True),
    (dict(dim=32, num_nearest_neighbors=8, dropout=0.25, soft_edges=True, cdim=5), 30, False, True),
    (dict(dim=32, num_nearest_neighbors=12, dropout=0.2, norm_coors=True, m_dim=40, m_pool_method="mean", cdim=2, coor_weights_clamp_value=1.5), 36, True, True),
    (dict(dim=24, dropout=0.3, m_dim=24, edge_dim=2), 20, True, True),                                   # dense, three scalars
    # ... more than five per-edge scalars (d/d W_s and d/d s on the matrix cores, their DROP instantiations): 7 with a wide head, 9, 13
    (dict(dim=24, num_nearest_neighbors=8, dropout=0.2, m_dim=32, fourier_features=3), 30, False, True),
    (dict(dim=32, num_nearest_neighbors=20, dropout=0.25, fourier_features=4, norm_coors=True), 40, True, True),
    (dict(dim=24, num_nearest_neighbors=6, dropout=0.15, fourier_features=4, edge_dim=4, cdim=4), 24, True, True),
    # round 6: layers without coors_mlp / without node_mlp and an odd `dim` on the native backward too (the generic tail kernel skips the
    # absent module and its mask site; node_mlp's hidden width 2 dim is no multiple of 4 when dim is odd: egnn_silu_bwd_drop_f32 walks
    # across row ends)
    (dict(dim=24, num_nearest_neighbors=8, dropout=0.2, update_coors=False), 30, False, True),
    (dict(dim=24, num_nearest_neighbors=8, dropout=0.2, update_feats=False, norm_coors=True), 30, True, True),
    (dict(dim=32, num_nearest_neighbors=32, dropout=0.25, update_feats=False), 64, False, True),
    (dict(dim=33, num_nearest_neighbors=8, dropout=0.2), 30, True, True),
    (dict(dim=17, num_nearest_neighbors=6, dropout=0.3, update_coors=False, m_dim=20), 26, False, True),
    # the plain kernels' shapes: `_backward_exact` with the masks re-evaluated (csrc/edge_exact_bwd.hip, the generic tail kernel; a head
    # wider than 64 channels keeps the autograd tail with the hash's torch twin)
    (dict(dim=24, num_nearest_neighbors=8, dropout=0.2, cdim=9, coor_weights_clamp_value=2.0), 30, True, "exact"),
    (dict(dim=16, num_nearest_neighbors=6, dropout=0.25, m_dim=80, soft_edges=True), 24, False, "exact"),
    (dict(dim=16, dropout=0.3, fourier_features=8, edge_dim=2, norm_feats=True, m_pool_method="mean"), 14, True, "exact"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("kw,n,use_mask,native", BWD_CASES)
def test_training_mode_backward_differentiates_the_masked_layer(kw, n, use_mask, native):
    """loss.backward() through the drop-in layer in training mode: every gradient equals float64 autograd of the restatement with
    the masks of THAT forward call (the seed comes from torch's CPU generator: torch.manual_seed reproduces it)."""
    from egnn_pytorch_amd import EGNN, _dropout, autograd as A
    torch.manual_seed(21)
    layer = EGNN(**_layer_kw(kw))
    cdim = kw.get("cdim", 3)
    if native == "exact":                                   # the plain kernels: EGNNFunction takes `_backward_exact`
        assert cdim > 8 or layer.m_dim > 64 or 2 * layer.fourier_features + 1 + layer.edge_dim > 16
    else:
        assert (A._dropout_native_ok(layer) and cdim <= 8) == native
    with torch.no_grad():
        for prm in layer.parameters():
            prm.mul_(40.0)
    layer = layer.cuda().train()
    g = torch.Generator().manual_seed(4)
    b = 3
    dim, p = kw["dim"], kw["dropout"]
    feats, coors = torch.randn(b, n, dim, generator=g).cuda(), torch.randn(b, n, cdim, generator=g).cuda()
    edges = torch.randn(b, n, n, kw["edge_dim"], generator=g).cuda() if kw.get("edge_dim") else None
    mask = (torch.arange(n)[None] < torch.tensor([[n], [n - 3], [n // 2 + 2]])).cuda() if use_mask else None
    rn, rc = torch.randn(b, n, dim, generator=g).cuda(), torch.randn(b, n, cdim, generator=g).cuda()
    torch.manual_seed(77)
    seed = _dropout.draw_seed()
    torch.manual_seed(77)
    f, c = feats.clone().requires_grad_(True), coors.clone().requires_grad_(True)
    saved, saved_rc = A._FUSED_MAX_GRAPHS, A._backward_recompute
    A._FUSED_MAX_GRAPHS = 2                                  # two chunks of graphs: the masks' rows are global edge / node ids
    if native:                                               # a native case must not fall back to the ATen recompute, silently
        def _no_recompute(*a_, **k_):
            raise AssertionError("this configuration is expected on a native backward, not on _backward_recompute")
        A._backward_recompute = _no_recompute
    try:
        node, co = layer(f, c, edges, mask)
        got = torch.autograd.grad((node * rn).sum() + (co * rc).sum(), [f, c] + list(layer.parameters()), allow_unused=True)
    finally:
        A._FUSED_MAX_GRAPHS, A._backward_recompute = saved, saved_rc
    with torch.no_grad():
        idx, rank, radius = layer._forward_hip_checked(feats, coors, edges, mask, None, None, drop_seed=seed)[3:6]
    l64 = copy.deepcopy(layer).double()
    f2, c2 = feats.double().requires_grad_(True), coors.double().requires_grad_(True)
    n2, co2 = A.layer_given_neighbors(l64, f2, c2, None if edges is None else edges.double(), mask, None if idx is None else idx.long(),
                                      None if rank is None else rank.double(), radius, drop=(p, seed))
    np.testing.assert_allclose(node.detach().cpu().numpy(), n2.detach().cpu().numpy(), atol=1e-4, rtol=0)
    want = torch.autograd.grad((n2 * rn.double()).sum() + (co2 * rc.double()).sum(), [f2, c2] + list(l64.parameters()), allow_unused=True)
    for a, r in zip(got, want):
        assert (a is None) == (r is None)
        if a is not None:
            assert float((a.double() - r).abs().max()) <= 1e-4 * max(1.0, float(r.abs().max()))
    # two training-mode calls draw different masks
    o = 0 if kw.get("update_feats", True) else 1             # (update_feats=False: the features pass through, the coordinates carry the masks)
    with torch.no_grad():
        a1, a2 = layer(feats, coors, edges, mask)[o], layer(feats, coors, edges, mask)[o]
    assert not torch.allclose(a1, a2, atol=1e-3)


@pytest.mark.gpu
def test_dropout_runs_on_every_path_and_a_float64_module_trains_in_float64():
    """Training-mode dropout has no shape limit left (round 5: the plain kernels evaluate the same hash masks), and a float64 module
    with dropout computes -- and differentiates -- in float64: forward against the float64 restatement with the same masks at 1e-9."""
    from egnn_pytorch_amd import EGNN, _dropout, autograd as A
    f = torch.randn(1, 12, 16).cuda()
    EGNN(dim=16, dropout=0.1).cuda().train()(f, torch.randn(1, 12, 9).cuda())
    EGNN(dim=16, dropout=0.1, m_dim=80).cuda().train()(f, torch.randn(1, 12, 3).cuda())
    EGNN(dim=16, dropout=0.1, m_dim=80).cuda().eval()(f, torch.randn(1, 12, 3).cuda())      # eval: dropout is the identity
    EGNN(dim=16, dropout=0.1, m_dim=32).cuda().train()(f, torch.randn(1, 12, 5).cuda())
    torch.manual_seed(5)
    layer = EGNN(dim=16, num_nearest_neighbors=6, dropout=0.2, norm_coors=True, fourier_features=1).double().cuda().train()
    g = torch.Generator().manual_seed(8)
    feats, coors = torch.randn(2, 20, 16, generator=g, dtype=torch.float64).cuda(), torch.randn(2, 20, 3, generator=g, dtype=torch.float64).cuda()
    torch.manual_seed(31)
    seed = _dropout.draw_seed()
    torch.manual_seed(31)
    fq, cq = feats.clone().requires_grad_(True), coors.clone().requires_grad_(True)
    node, co = layer(fq, cq)
    assert node.dtype == torch.float64
    got = torch.autograd.grad(node.sum() + (co * co).sum(), [fq, cq] + list(layer.parameters()), allow_unused=True)
    with torch.no_grad():
        idx, rank, radius = layer._forward_with_hint(feats, coors, None, None, None, None, drop_seed=seed)[3:6]
    f2, c2 = feats.clone().requires_grad_(True), coors.clone().requires_grad_(True)
    n2, co2 = A.layer_given_neighbors(layer, f2, c2, None, None, idx.long(), rank.double(), radius, drop=(0.2, seed))
    assert float((node - n2).abs().max()) <= 1e-9 and float((co - co2).abs().max()) <= 1e-9
    want = torch.autograd.grad(n2.sum() + (co2 * co2).sum(), [f2, c2] + list(layer.parameters()), allow_unused=True)
    for a, r in zip(got, want):
        assert (a is None) == (r is None)
        if a is not None:
            assert float((a - r).abs().max()) <= 1e-8 * max(1.0, float(r.abs().max()))
