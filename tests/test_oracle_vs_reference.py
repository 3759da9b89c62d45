"""Oracle vs the LIVE reference (skipped where /root/reference is absent, e.g. on the GPU box).
This is the differential pin of SURVEY.md §8c at BASELINE.json-like layer widths."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import egnn_oracle as O
from tests._util import check_neighbors

REF = os.environ.get("EGNN_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "egnn_pytorch")),
                                reason="live reference not present")


def _ref():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import egnn_pytorch
    return egnn_pytorch


def _xavier(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            if p.ndim == 2:
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / (p.shape[0] + p.shape[1])) ** 0.5)


@pytest.mark.parametrize("kwargs,b,n,use_mask", [
    (dict(dim=512, num_nearest_neighbors=32), 1, 192, True),      # north-star layer, reduced N
    (dict(dim=512), 1, 48, False),                                # config 2 layer, reduced N
    (dict(dim=128, num_nearest_neighbors=32, norm_feats=True), 2, 128, True),   # config 3 layer
    (dict(dim=256, num_nearest_neighbors=32, norm_feats=True, norm_coors=True), 1, 128, True),  # config 5 layer
])
def test_layer(kwargs, b, n, use_mask):
    ref = _ref()
    torch.manual_seed(0)
    layer = ref.EGNN(**kwargs).eval()
    _xavier(layer, 11)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(b, n, kwargs["dim"], generator=g)
    coors = torch.randn(b, n, 3, generator=g)
    mask = None
    if use_mask:
        lens = torch.randint(n // 2, n + 1, (b,), generator=g)
        mask = torch.arange(n)[None] < lens[:, None]
    captured = []
    orig = torch.Tensor.topk

    def spy(t, *a, **kw):
        out = orig(t, *a, **kw)
        captured.append(out)
        return out

    torch.Tensor.topk = spy
    try:
        with torch.no_grad():
            rn, rc = layer(feats, coors, mask=mask)
    finally:
        torch.Tensor.topk = orig
    params = {k: v.numpy() for k, v in layer.state_dict().items()}
    cfg = O.EGNNConfig(**kwargs)
    node, co, nr, ni = O.egnn_forward(cfg, params, feats.numpy(), coors.numpy(),
                                      mask=None if mask is None else mask.numpy(), return_neighbors=True)
    if captured:
        check_neighbors(captured[0][0].numpy(), captured[0][1].numpy().astype(np.int32),
                        nr.astype(np.float32), ni.astype(np.int32))
    np.testing.assert_allclose(node, rn.numpy(), atol=3e-5, rtol=0)
    np.testing.assert_allclose(co, rc.numpy(), atol=3e-5, rtol=0)


def test_sparse_chain_config4_layer():
    ref = _ref()
    n, b = 64, 2
    layer = ref.EGNN(dim=512, edge_dim=4, only_sparse_neighbors=True).eval()
    _xavier(layer, 3)
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(b, n, 512, generator=g)
    coors = torch.randn(b, n, 3, generator=g)
    edges = torch.randn(b, n, n, 4, generator=g)
    mask = torch.ones(b, n, dtype=torch.bool)
    i = torch.arange(n)
    adj = (i[:, None] - i[None, :]).abs() <= 1
    with torch.no_grad():
        rn, rc = layer(feats, coors, edges, mask, adj)
    params = {k: v.numpy() for k, v in layer.state_dict().items()}
    cfg = O.EGNNConfig(dim=512, edge_dim=4, only_sparse_neighbors=True)
    node, co = O.egnn_forward(cfg, params, feats.numpy(), coors.numpy(), edges.numpy(), mask.numpy(), adj.numpy())
    np.testing.assert_allclose(node, rn.numpy(), atol=3e-5, rtol=0)
    np.testing.assert_allclose(co, rc.numpy(), atol=3e-5, rtol=0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_squared_distance_summation_tree_matches_torch_bit_for_bit(dtype):
    """`(rel_coors ** 2).sum(dim=-1)` (egnn_pytorch.py:233) for every coordinate dimension the kernels take: the oracle's restatement of
    ATen's inner-dimension sum (egnn_oracle.inner_sum; the kernels' egnn_sqdist_any follows the same tree) against torch itself."""
    rng = np.random.default_rng(0)
    for c in list(range(1, 41)) + [48, 63, 64, 65, 100, 128, 200]:
        a = rng.standard_normal((2, 24, c)).astype(dtype)
        t = torch.from_numpy(a)
        rel = t[:, :, None, :] - t[:, None, :, :]
        want = (rel ** 2).sum(dim=-1).numpy()
        got_rel, got = O.pairwise(a)
        assert np.array_equal(got_rel, rel.numpy()), c
        assert np.array_equal(got, want), (c, dtype)


@pytest.mark.parametrize("kwargs,cdim", [(dict(dim=32, num_nearest_neighbors=8), 11), (dict(dim=24, num_nearest_neighbors=8, norm_coors=True), 33),
                                         (dict(dim=16), 9)])
def test_layer_with_more_than_eight_coordinates(kwargs, cdim):
    ref = _ref()
    torch.manual_seed(cdim)
    layer = ref.EGNN(**kwargs).eval()
    _xavier(layer, 3)
    g = torch.Generator().manual_seed(cdim + 1)
    feats, coors = torch.randn(2, 36, kwargs["dim"], generator=g), torch.randn(2, 36, cdim, generator=g)
    mask = torch.arange(36)[None] < torch.tensor([[36], [29]])
    with torch.no_grad():
        want_n, want_c = layer(feats, coors, mask=mask)
    params = {k: v.detach().numpy() for k, v in layer.state_dict().items()}
    got_n, got_c = O.egnn_forward(O.EGNNConfig(**kwargs), params, feats.numpy(), coors.numpy(), mask=mask.numpy())
    np.testing.assert_allclose(got_n, want_n.numpy(), atol=1e-5 * max(1.0, float(want_n.abs().max())), rtol=0)
    np.testing.assert_allclose(got_c, want_c.numpy(), atol=1e-5 * max(1.0, float(want_c.abs().max())), rtol=0)
