"""Layer / network parity on the MI355X: HIP path (through the drop-in nn.Module and the C ABI)
against the committed golden vectors of the live reference and against the CPU oracle."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import egnn_oracle as O
from tests._util import ATOL, golden_names, layer_kwargs, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _inference_mode():
    """These are forward-parity tests: run them the way inference code does (under autograd the modules would record a
    graph, as the reference does -- that path is tests/test_autograd.py)."""
    with torch.no_grad():
        yield


def _dev(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _module(kind, kwargs, params):
    from egnn_pytorch_amd import EGNN, EGNN_Network
    net = EGNN(**kwargs) if kind == "layer" else EGNN_Network(**kwargs)
    missing = net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net.cuda().eval()


@pytest.mark.parametrize("name", golden_names())
def test_golden(name):
    """Outputs of the reference itself (fp32 CPU), replayed from tests/golden/*.npz."""
    meta, params, d = load_golden(name)
    net = _module(meta["kind"], meta["kwargs"], params)
    feats, coors = _dev(d["feats"]), _dev(d["coors"])
    edges, mask, adj = _dev(d.get("edges")), _dev(d.get("mask")), _dev(d.get("adj_mat"))
    if meta["kind"] == "layer":
        node, co = net(feats, coors, edges, mask, adj)
    else:
        node, co, changes = net(feats, coors, adj_mat=adj, edges=edges, mask=mask, return_coor_changes=True)
        for i, c in enumerate(changes):
            np.testing.assert_allclose(c.cpu().numpy(), d[f"coor_change.{i}"], atol=ATOL, rtol=0)
    assert node.dtype == torch.float32 and node.is_cuda
    np.testing.assert_allclose(node.cpu().numpy(), d["node_out"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(co.cpu().numpy(), d["coors_out"], atol=ATOL, rtol=0)


CONFIGS = [
    # BASELINE.json configs at sizes the oracle finishes in seconds (full widths, reduced B / N)
    ("c2_dense_dim512", dict(dim=512), 2, 64, dict()),
    ("c2_dense_dim512_n256", dict(dim=512), 1, 256, dict(scale={"coors_mlp.3.weight": 0.1})),   # 256 neighbours: keep |x'| O(10)
    ("ns_dim512_k32", dict(dim=512, num_nearest_neighbors=32), 2, 256, dict(mask=True)),
    ("c3_layer_dim128_k32", dict(dim=128, num_nearest_neighbors=32, norm_feats=True), 3, 320, dict(mask=True)),
    ("c4_sparse_dim512_edges", dict(dim=512, edge_dim=4, only_sparse_neighbors=True), 2, 128,
     dict(mask=True, edges=True, adj="chain")),
    ("c4_knn32_adj", dict(dim=512, edge_dim=4, num_nearest_neighbors=32), 1, 128,
     dict(mask=True, edges=True, adj="chain")),
    ("c5_layer_dim256_normcoors", dict(dim=256, num_nearest_neighbors=32, norm_feats=True, norm_coors=True), 2, 256,
     dict(mask=True)),
    ("ragged_k_not_16", dict(dim=48, num_nearest_neighbors=11), 3, 77, dict(mask=True)),
    ("dense_odd_n", dict(dim=40, edge_dim=1), 2, 37, dict(mask=True, edges=True)),
    ("k_eq_n", dict(dim=32, num_nearest_neighbors=24), 1, 24, dict()),
    # hundreds of neighbours per node: damp the message scale so outputs stay O(1..10); at |out| ~ 600 the fp32
    # ORACLE itself sits 3e-4 from an fp64 run, i.e. the 1e-4 bar would measure summation order, not parity
    ("dense_n600_multi_round", dict(dim=32), 1, 600, dict(mask=True, scale={"edge_mlp.3.weight": 0.1, "coors_mlp.3.weight": 0.05})),
    ("knn_k300_multi_round", dict(dim=32, num_nearest_neighbors=300), 1, 400,
     dict(scale={"edge_mlp.3.weight": 0.1, "coors_mlp.3.weight": 0.05})),
    # first-layer MFMA variants: K % 32 == 0 (P_i in the MFMA) with 3 / 6 / 12 chained MFMAs of split scalars, a node
    # spanning two waves (k = 64), and the per-lane P_i variant with many scalars
    ("k32_fourier1_s3", dict(dim=32, num_nearest_neighbors=32, fourier_features=1), 2, 64, dict(mask=True)),
    ("k32_fourier2_edges3_s8", dict(dim=32, num_nearest_neighbors=32, fourier_features=2, edge_dim=3), 2, 64,
     dict(mask=True, edges=True)),
    ("k64_fourier4_edges7_s16", dict(dim=32, num_nearest_neighbors=64, fourier_features=4, edge_dim=7), 1, 96,
     dict(mask=True, edges=True, scale={"edge_mlp.3.weight": 0.3, "coors_mlp.3.weight": 0.3})),
    ("k20_fourier4_edges7_s16", dict(dim=32, num_nearest_neighbors=20, fourier_features=4, edge_dim=7), 2, 50,
     dict(mask=True, edges=True)),
    # node groups spanning several rounds of 128 slots (K = 48: 8 nodes = 3 rounds; K = 24: 16 nodes; K = 40: 16 nodes = 5 rounds)
    ("k48_groups_of_8", dict(dim=32, num_nearest_neighbors=48), 2, 100, dict(mask=True, scale={"edge_mlp.3.weight": 0.3, "coors_mlp.3.weight": 0.3})),
    ("k24_groups_of_16", dict(dim=32, num_nearest_neighbors=24, norm_coors=True), 2, 70, dict(mask=True)),
    ("k40_groups_of_16", dict(dim=32, num_nearest_neighbors=40, edge_dim=2), 1, 90, dict(edges=True, scale={"edge_mlp.3.weight": 0.3, "coors_mlp.3.weight": 0.3})),
    ("k96_groups_of_4", dict(dim=32, num_nearest_neighbors=96), 1, 130, dict(mask=True, scale={"edge_mlp.3.weight": 0.2, "coors_mlp.3.weight": 0.2})),
    # smallest K whose tiles still carry P_i in the MFMA (a tile of 16 slots touches up to 4 nodes), and the VALU path below it
    ("k6_four_nodes_per_tile", dict(dim=32, num_nearest_neighbors=6), 2, 45, dict(mask=True)),
    ("k7_four_nodes_per_tile", dict(dim=32, num_nearest_neighbors=7, edge_dim=1), 2, 45, dict(mask=True, edges=True)),
    ("k5_valu_pi", dict(dim=32, num_nearest_neighbors=5), 2, 45, dict(mask=True)),
    # degenerate sizes: a single node (dense: only the self edge), two nodes, k = 1 (self only), N < one MFMA tile
    ("tiny_n1_dense", dict(dim=16), 2, 1, dict()),
    ("tiny_n2_dense_mask", dict(dim=16, edge_dim=1), 3, 2, dict(mask_exact=[2, 1, 2], edges=True)),
    ("tiny_n5_k1", dict(dim=16, num_nearest_neighbors=1), 2, 5, dict(mask=True)),
    ("tiny_n3_k3_mean_gate", dict(dim=8, num_nearest_neighbors=3, m_pool_method="mean", soft_edges=True, m_dim=4), 1, 3, dict()),
    # coordinate dimension other than 3 (generic-C compilation of the select and edge kernels)
    ("coor_dim5_k32_normcoors", dict(dim=32, num_nearest_neighbors=32, norm_coors=True), 2, 64, dict(mask=True, coor_dim=5)),
    ("coor_dim7_dense_edges", dict(dim=32, edge_dim=2), 1, 40, dict(edges=True, coor_dim=7)),
    ("coor_dim8_k20_mean", dict(dim=32, num_nearest_neighbors=20, m_pool_method="mean"), 2, 50, dict(mask=True, coor_dim=8)),
    ("coor_dim1_k8", dict(dim=32, num_nearest_neighbors=8), 1, 30, dict(coor_dim=1)),
    # more than 8 coordinates: the plain kernels (the fused ones keep x_i - x_j in registers up to 8)
    ("coor_dim11_k8_edges", dict(dim=32, num_nearest_neighbors=8, edge_dim=2), 2, 48, dict(mask=True, edges=True, coor_dim=11)),
    ("coor_dim20_dense_normcoors", dict(dim=24, norm_coors=True, m_pool_method="mean"), 2, 30, dict(mask=True, coor_dim=20)),
    ("coor_dim64_k16_gate", dict(dim=32, num_nearest_neighbors=16, soft_edges=True), 1, 64, dict(coor_dim=64)),
    # heads wider than 64 channels and more than 64 per-edge scalars: the plain kernels (the reference has no such limits, :149-168)
    ("m96_k8_gate", dict(dim=32, m_dim=96, num_nearest_neighbors=8, soft_edges=True, norm_coors=True), 2, 40, dict(mask=True)),
    ("m130_dense_mean", dict(dim=16, m_dim=130, m_pool_method="mean"), 1, 20, dict(mask=True, scale={"edge_mlp.3.weight": 0.5})),
    ("scalars_101_k6", dict(dim=16, fourier_features=20, edge_dim=60, num_nearest_neighbors=6), 1, 24, dict(edges=True, scalar_cols=0.2)),
    # m_dim beyond one 16-channel MFMA tile: two / four accumulator tiles per edge tile (the reference has no limit, :153)
    ("m32_k32", dict(dim=64, m_dim=32, num_nearest_neighbors=32, soft_edges=True, norm_coors=True), 2, 96,
     dict(mask=True, scale={"edge_mlp.3.weight": 0.5})),
    ("m32_k8_edges", dict(dim=32, m_dim=32, num_nearest_neighbors=8, edge_dim=3, m_pool_method="mean"), 2, 50, dict(mask=True, edges=True)),
    ("m24_dense", dict(dim=32, m_dim=24), 2, 20, dict(mask=True)),
    ("m64_k32", dict(dim=48, m_dim=64, num_nearest_neighbors=32, soft_edges=True), 2, 64, dict(mask=True, scale={"edge_mlp.3.weight": 0.5})),
    ("m48_k5_fourier", dict(dim=32, m_dim=48, num_nearest_neighbors=5, fourier_features=2), 2, 40, dict(mask=True)),
    # wide dynamic range of the per-edge scalars (three-part fp16 split): coordinates x 60 -> dist^2 up to ~1e5 with a
    # distance weight damped to keep the pre-activations O(1); edge features up to ~1e3
    ("k32_large_distances", dict(dim=32, num_nearest_neighbors=32, edge_dim=2), 2, 64,
     dict(mask=True, edges=True, coors_mul=60.0, edges_mul=300.0, scalar_cols=1e-4,
          scale={"coors_mlp.3.weight": 0.01})),                      # |rel| ~ 100: keep the coordinate update O(1)
]


@pytest.mark.parametrize("name,kwargs,b,n,flags", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_layer_vs_oracle(name, kwargs, b, n, flags):
    rng = np.random.default_rng(zlib.crc32(name.encode()))      # stable across processes (hash() is salted)
    cfg = O.EGNNConfig(**kwargs)
    params = O.random_params(cfg, seed=17)
    for key, sc in flags.get("scale", {}).items():
        params[key] = params[key] * np.float32(sc)
    if "scalar_cols" in flags:                       # damp the [d | e] columns of edge_mlp.0.weight (egnn_pytorch.py:282-285)
        w1 = params["edge_mlp.0.weight"].copy()
        w1[:, 2 * kwargs["dim"]:] *= np.float32(flags["scalar_cols"])
        params["edge_mlp.0.weight"] = w1
    feats = rng.standard_normal((b, n, kwargs["dim"])).astype(np.float32)
    coors = (rng.standard_normal((b, n, flags.get("coor_dim", 3))) * flags.get("coors_mul", 1.0)).astype(np.float32)
    mask = edges = adj = None
    if flags.get("mask"):
        lens = rng.integers(n // 2, n + 1, size=b)
        mask = np.arange(n)[None, :] < lens[:, None]
    if flags.get("mask_exact"):
        mask = np.arange(n)[None, :] < np.asarray(flags["mask_exact"])[:, None]
    if flags.get("edges"):
        edges = (rng.standard_normal((b, n, n, kwargs["edge_dim"])) * flags.get("edges_mul", 1.0)).astype(np.float32)
    if flags.get("adj") == "chain":
        i = np.arange(n)
        adj = np.abs(i[:, None] - i[None, :]) <= 1
    ref_node, ref_co = O.egnn_forward(cfg, params, feats, coors, edges, mask, adj)
    net = _module("layer", kwargs, params)
    node, co = net(_dev(feats), _dev(coors), _dev(edges), _dev(mask), _dev(adj))
    # 1e-4 absolute up to |out| = 256; beyond that (coordinates x 60: |x| ~ 600, one fp32 ulp is 6e-5 there) 4e-7 of the
    # output's scale, i.e. a few ulps
    np.testing.assert_allclose(node.cpu().numpy(), ref_node, atol=ATOL * max(1.0, float(np.abs(ref_node).max()) / 256.0), rtol=0)
    np.testing.assert_allclose(co.cpu().numpy(), ref_co, atol=ATOL * max(1.0, float(np.abs(ref_co).max()) / 256.0), rtol=0)


def test_network_c3_vs_oracle():
    """config 3: EGNN_Network(depth=3, dim=128, k=32), masked, reduced B."""
    kwargs = dict(depth=3, dim=128, num_nearest_neighbors=32)
    cfg = O.EGNNConfig(dim=128, num_nearest_neighbors=32, norm_feats=True)
    rng = np.random.default_rng(5)
    params = {}
    for layer in range(3):
        pl = O.random_params(cfg, seed=100 + layer, prefix=f"layers.{layer}.1.")
        # stacked xavier-scale layers blow activations up to |feats| ~ 700, where the fp32 oracle itself is
        # 9e-4 away from an fp64 run; these factors keep |feats| ~ 13 (oracle fp32-vs-fp64 noise 5e-6) while a
        # wrong geometry still moves the output by ~14, so the 1e-4 bar stays discriminating
        pl[f"layers.{layer}.1.coors_mlp.3.weight"] *= 0.1
        pl[f"layers.{layer}.1.node_mlp.3.weight"] *= 0.3
        pl[f"layers.{layer}.1.edge_mlp.3.weight"] *= 0.3
        params.update(pl)
    b, n = 2, 256
    feats = rng.standard_normal((b, n, 128)).astype(np.float32)
    coors = rng.standard_normal((b, n, 3)).astype(np.float32)
    mask = np.arange(n)[None, :] < np.array([[n], [n - 50]])
    ref_node, ref_co = O.egnn_network_forward(3, cfg, params, feats, coors, mask=mask)
    net = _module("network", kwargs, params)
    node, co = net(_dev(feats), _dev(coors), mask=_dev(mask))
    np.testing.assert_allclose(node.cpu().numpy(), ref_node, atol=ATOL, rtol=0)
    np.testing.assert_allclose(co.cpu().numpy(), ref_co, atol=ATOL, rtol=0)


# ------------------------------------------------------------------ the reference's own property tests
def _rot(a, b, c):
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    rz = lambda co, si: np.array([[co, -si, 0], [si, co, 0], [0, 0, 1]])
    ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    return (rz(ca, sa) @ ry @ rz(cc, sc)).astype(np.float32)


@pytest.mark.parametrize("kwargs,n,edge_dim", [
    (dict(dim=512, edge_dim=4), 16, 4),                                          # tests/test_equivariance.py:8-34
    (dict(dim=512, edge_dim=1, num_nearest_neighbors=8), 256, 1),                # :47-73
    (dict(dim=512, edge_dim=1, num_nearest_neighbors=8, norm_coors=True), 256, 1),   # :76-102
])
def test_equivariance(kwargs, n, edge_dim):
    """fp32 port of the reference's equivariance tests (default init, as upstream); the reference's own
    fp32 equivariance error on these shapes is <= 1e-6, tolerance 1e-5 here."""
    from egnn_pytorch_amd import EGNN
    torch.manual_seed(0)
    layer = EGNN(**kwargs).cuda().eval()
    rng = np.random.default_rng(3)
    R = _dev(_rot(*rng.random(3)))
    T = _dev(rng.standard_normal((1, 1, 3)).astype(np.float32))
    feats = _dev(rng.standard_normal((1, n, 512)).astype(np.float32))
    coors = _dev(rng.standard_normal((1, n, 3)).astype(np.float32))
    edges = _dev(rng.standard_normal((1, n, n, edge_dim)).astype(np.float32))
    mask = torch.ones(1, n, dtype=torch.bool, device="cuda")
    perm = feats.clone()
    perm[:, 0], perm[:, 1] = feats[:, 1], feats[:, 0]
    f1, c1 = layer(feats, coors @ R + T, edges, mask=mask)
    f2, c2 = layer(feats, coors, edges, mask=mask)
    f3, c3 = layer(perm, coors, edges, mask=mask)
    assert torch.allclose(f1, f2, atol=1e-5), "type 0 features are invariant"
    assert torch.allclose(c1, c2 @ R + T, atol=1e-5), "type 1 features are equivariant"
    assert not torch.allclose(f1, f3, atol=1e-6), "layer must be equivariant to permutations of node order"


# ------------------------------------------------------------------ full BASELINE sizes: size-independent properties
def test_higher_dimension():
    """tests/test_equivariance.py:36-45 (5-D coordinates run at all) -- plus what the reference does not check there:
    equivariance under a random orthogonal map + translation of R^5."""
    from egnn_pytorch_amd import EGNN
    torch.manual_seed(0)
    layer = EGNN(dim=512, edge_dim=4).cuda().eval()
    rng = np.random.default_rng(7)
    q, _ = np.linalg.qr(rng.standard_normal((5, 5)))
    R, T = _dev(q.astype(np.float32)), _dev(rng.standard_normal((1, 1, 5)).astype(np.float32))
    feats = _dev(rng.standard_normal((1, 16, 512)).astype(np.float32))
    coors = _dev(rng.standard_normal((1, 16, 5)).astype(np.float32))
    edges = _dev(rng.standard_normal((1, 16, 16, 4)).astype(np.float32))
    mask = torch.ones(1, 16, dtype=torch.bool, device="cuda")
    f1, c1 = layer(feats, coors @ R + T, edges, mask=mask)
    f2, c2 = layer(feats, coors, edges, mask=mask)
    assert c1.shape == (1, 16, 5)
    assert torch.allclose(f1, f2, atol=1e-5)
    assert torch.allclose(c1, c2 @ R + T, atol=1e-5)


def test_north_star_full_size_properties():
    """EGNN(dim=512, k=32), B=64, N=1024 (the metric's configuration), xavier-scale weights:
    (1) graphs are independent: running a sub-batch reproduces the same rows bit for bit;
    (2) repeat runs are bit-identical (deterministic reductions, no float atomics);
    (3) two graphs of the batch match the CPU oracle within 1e-4;
    (4) rotation + translation equivariance at full size."""
    kwargs = dict(dim=512, num_nearest_neighbors=32)
    cfg = O.EGNNConfig(**kwargs)
    params = O.random_params(cfg, seed=1)
    net = _module("layer", kwargs, params)
    g = torch.Generator().manual_seed(1234)
    b, n = 64, 1024
    feats = torch.randn(b, n, 512, generator=g)
    coors = torch.randn(b, n, 3, generator=g)
    lens = torch.randint(n // 2, n + 1, (b,), generator=g)
    mask = torch.arange(n)[None] < lens[:, None]
    fd, cd, md = feats.cuda(), coors.cuda(), mask.cuda()
    node, co = net(fd, cd, mask=md)
    node2, co2 = net(fd, cd, mask=md)
    assert torch.equal(node, node2) and torch.equal(co, co2)
    sub = [5, 63]
    ns, cs = net(fd[sub], cd[sub], mask=md[sub])
    assert torch.equal(ns, node[sub]) and torch.equal(cs, co[sub])
    for gi in sub:
        rn, rc = O.egnn_forward(cfg, params, feats[gi:gi + 1].numpy(), coors[gi:gi + 1].numpy(),
                                mask=mask[gi:gi + 1].numpy())
        np.testing.assert_allclose(node[gi:gi + 1].cpu().numpy(), rn, atol=ATOL, rtol=0)
        np.testing.assert_allclose(co[gi:gi + 1].cpu().numpy(), rc, atol=ATOL, rtol=0)
    R = _dev(_rot(0.3, 1.1, 2.0))
    T = torch.tensor([[[0.5, -1.0, 2.0]]], device="cuda")
    nr, cr = net(fd, cd @ R + T, mask=md)
    assert torch.allclose(nr, node, atol=2e-3)          # xavier-scale weights amplify fp32 noise in d
    assert torch.allclose(cr, co @ R + T, atol=2e-3)


FULL_SIZE = [
    # BASELINE.json configs at their full sizes (c5: the per-GPU shard of 64 graphs): name, kind, kwargs, B, N, flags
    ("c2_dense", "layer", dict(dim=512), 8, 256, dict(damp=True)),
    ("c3_network", "network", dict(depth=3, dim=128, num_nearest_neighbors=32), 64, 1024, dict(mask=True, damp=True)),
    ("c4_sparse", "layer", dict(dim=512, edge_dim=4, only_sparse_neighbors=True), 32, 2048,
     dict(mask=True, edges=True, adj="chain")),
    ("c5_shard", "network", dict(depth=6, dim=256, num_nearest_neighbors=32, norm_coors=True), 64, 1024,
     dict(mask=True, damp=True)),
]


@pytest.mark.parametrize("name,kind,kwargs,b,n,flags", FULL_SIZE, ids=[c[0] for c in FULL_SIZE])
def test_baseline_configs_full_size_properties(name, kind, kwargs, b, n, flags):
    """Size-independent properties at the BASELINE.json sizes (the oracle cannot run these in seconds):
    repeat runs bit-identical, a sub-batch reproduces its rows bit for bit, rotation + translation equivariance;
    plus ONE graph against the CPU oracle within 1e-4."""
    lk = {k: v for k, v in kwargs.items() if k != "depth"}
    if kind == "network":
        lk["norm_feats"] = True
    cfg = O.EGNNConfig(**lk)
    depth = kwargs.get("depth", 1)
    params = {}
    for layer in range(depth):
        prefix = f"layers.{layer}.1." if kind == "network" else ""
        pl = O.random_params(cfg, seed=40 + layer, prefix=prefix)
        if flags.get("damp"):              # keep stacked / many-neighbour outputs O(1..10) (see CONFIGS above)
            pl[prefix + "coors_mlp.3.weight"] *= np.float32(0.1)
            pl[prefix + "node_mlp.3.weight"] *= np.float32(0.3)
            pl[prefix + "edge_mlp.3.weight"] *= np.float32(0.3)
        params.update(pl)
    net = _module(kind, kwargs, params)
    g = torch.Generator().manual_seed(99)
    feats = torch.randn(b, n, kwargs["dim"], generator=g)
    coors = torch.randn(b, n, 3, generator=g)
    mask = edges = adj = None
    if flags.get("mask"):
        lens = torch.randint(n // 2, n + 1, (b,), generator=g)
        mask = torch.arange(n)[None] < lens[:, None]
    if flags.get("edges"):
        edges = torch.randn(b, n, n, kwargs["edge_dim"], generator=g)
    if flags.get("adj") == "chain":
        i = torch.arange(n)
        adj = (i[:, None] - i[None, :]).abs() <= 1
    dev = lambda t: None if t is None else t.cuda()

    def run(f, c, e, m):
        if kind == "network":
            return net(dev(f), dev(c), adj_mat=dev(adj), edges=dev(e), mask=dev(m))
        return net(dev(f), dev(c), dev(e), dev(m), dev(adj))

    node, co = run(feats, coors, edges, mask)
    node2, co2 = run(feats, coors, edges, mask)
    assert torch.equal(node, node2) and torch.equal(co, co2)
    assert torch.isfinite(node).all() and torch.isfinite(co).all()
    sub = [1, b - 1]
    take = lambda t: None if t is None else t[sub]
    ns, cs = run(feats[sub], coors[sub], take(edges), take(mask))
    assert torch.equal(ns, node[sub]) and torch.equal(cs, co[sub])
    gi = sub[0]
    one = lambda t: None if t is None else t[gi:gi + 1].numpy()
    adj_np = None if adj is None else adj.numpy()
    if kind == "network":
        rn, rc = O.egnn_network_forward(depth, cfg, params, one(feats), one(coors), adj_mat=adj_np, edges=one(edges),
                                        mask=one(mask))
    else:
        rn, rc = O.egnn_forward(cfg, params, one(feats), one(coors), one(edges), one(mask), adj_np)
    np.testing.assert_allclose(node[gi:gi + 1].cpu().numpy(), rn, atol=ATOL, rtol=0)
    np.testing.assert_allclose(co[gi:gi + 1].cpu().numpy(), rc, atol=ATOL, rtol=0)
    R = _rot(0.7, 0.2, 1.5)
    T = np.array([[[0.25, -0.5, 1.0]]], dtype=np.float32)
    nr, cr = run(feats, torch.from_numpy(coors.numpy() @ R + T), edges, mask)
    tol = 2e-3 * depth                                    # xavier-scale weights amplify fp32 noise in d, per layer
    co_rot = co @ _dev(R) + _dev(T)
    if kind == "layer":
        assert torch.allclose(nr, node, atol=tol) and torch.allclose(cr, co_rot, atol=tol)
    else:
        # stacked k-NN layers: rotating in fp32 perturbs squared distances by an ulp, and where the K-th and (K+1)-th
        # candidates of a node are that close its neighbour set flips (in the reference too); every node reachable
        # from such a node in the following layers then differs.  Equivariance must hold for all other nodes.
        bad = ((nr - node).abs().amax(-1) > tol) | ((cr - co_rot).abs().amax(-1) > tol)
        assert bad.float().mean().item() < 0.05, bad.float().mean().item()


def test_multi_round_stress_is_deterministic_and_correct():
    """Regression: dense multi-round launches (5 rounds per node group, every CU busy) once produced 1-2 % wrong
    coordinate weights, differently on every run, node features intact -- ds_bpermute_b32 (`__shfl_xor`) returning
    another value while LDS-DMA traffic of co-resident workgroups was in flight (see egnn_common.h: the kernels use
    DPP / explicit LDS exchanges only).  6 graphs x 600 nodes dense, 20 repeats: bit-identical, and on parity."""
    kwargs = dict(dim=32)
    cfg = O.EGNNConfig(**kwargs)
    params = O.random_params(cfg, seed=17)
    params["edge_mlp.3.weight"] = params["edge_mlp.3.weight"] * np.float32(0.1)
    params["coors_mlp.3.weight"] = params["coors_mlp.3.weight"] * np.float32(0.05)
    rng = np.random.default_rng(12345)
    b, n = 6, 600
    feats = rng.standard_normal((b, n, 32)).astype(np.float32)
    coors = rng.standard_normal((b, n, 3)).astype(np.float32)
    mask = np.arange(n)[None, :] < rng.integers(n // 2, n + 1, size=b)[:, None]
    rn, rc = O.egnn_forward(cfg, params, feats, coors, None, mask, None)
    net = _module("layer", kwargs, params)
    fd, cd, md = _dev(feats), _dev(coors), _dev(mask)
    first = None
    for _ in range(20):
        node, co = net(fd, cd, mask=md)
        if first is None:
            first = (node.clone(), co.clone())
        assert torch.equal(node, first[0]) and torch.equal(co, first[1])
    np.testing.assert_allclose(first[0].cpu().numpy(), rn, atol=ATOL, rtol=0)
    np.testing.assert_allclose(first[1].cpu().numpy(), rc, atol=ATOL, rtol=0)


def test_concurrent_launches_do_not_change_results():
    """Two launches of the layer running concurrently on different HIP streams (edge passes next to the other chunk's
    GEMMs, LDS-DMA traffic of foreign workgroups on every CU) must give bit for bit what the same chunks give one after
    the other.  (This is how the ds_bpermute_b32 problem of the stress test above reproduces within seconds.)"""
    from egnn_pytorch_amd import EGNN
    torch.manual_seed(0)
    layer = EGNN(dim=512, num_nearest_neighbors=32, soft_edges=True).cuda().eval()
    with torch.no_grad():
        for p_ in layer.parameters():                       # xavier-scale so that the gate / coordinate weights matter
            if p_.ndim == 2:
                p_.copy_(torch.randn_like(p_) * (2.0 / (p_.shape[0] + p_.shape[1])) ** 0.5)
    g = torch.Generator().manual_seed(1)
    b, n, nchunk = 32, 1024, 4
    feats, coors = torch.randn(b, n, 512, generator=g).cuda(), torch.randn(b, n, 3, generator=g).cuda()
    mask = torch.ones(b, n, dtype=torch.bool, device="cuda")
    cs = b // nchunk
    chunk = lambda c: (feats[c * cs:(c + 1) * cs], coors[c * cs:(c + 1) * cs], mask[c * cs:(c + 1) * cs])
    seq = [layer(f, x, mask=m) for f, x, m in map(chunk, range(nchunk))]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(4):
        main = torch.cuda.current_stream()
        for s_ in streams:
            s_.wait_stream(main)
        outs = []
        for c in range(nchunk):
            with torch.cuda.stream(streams[c % 2]):
                f, x, m = chunk(c)
                outs.append(layer(f, x, mask=m))
        for s_ in streams:
            main.wait_stream(s_)
        torch.cuda.synchronize()
        for o, r in zip(outs, seq):
            assert torch.equal(o[0], r[0]) and torch.equal(o[1], r[1])


def test_hip_graph_replay_matches_eager():
    """egnn_pytorch_amd.graphed: the launch sequence of a 3-layer network captured into a HIP graph replays bit for bit
    what the eager calls produce, also for new inputs of the same shape."""
    from egnn_pytorch_amd import EGNN_Network, graphed
    torch.manual_seed(0)
    net = EGNN_Network(depth=3, dim=64, num_nearest_neighbors=16, norm_coors=True).cuda().eval()
    g = torch.Generator().manual_seed(5)
    mk = lambda: (torch.randn(2, 200, 64, generator=g).cuda(), torch.randn(2, 200, 3, generator=g).cuda())
    mask = (torch.arange(200)[None] < torch.tensor([[200], [150]])).cuda()
    f0, c0 = mk()
    run = graphed(net, f0, c0, mask=mask)
    for _ in range(3):
        f, c = mk()
        want = net(f, c, mask=mask)
        got = run(f, c, mask=mask)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    with pytest.raises(ValueError):
        run(f[:1], c[:1], mask=mask[:1])


def test_empty_inputs_behave_like_the_reference():
    """B = 0 and (dense) N = 0 give empty outputs; N = 0 on the k-NN path raises topk's error (measured on the reference)."""
    from egnn_pytorch_amd import EGNN
    dense, knn = EGNN(dim=8).cuda().eval(), EGNN(dim=8, num_nearest_neighbors=2).cuda().eval()
    for layer in (dense, knn):
        f, c = layer(torch.randn(0, 5, 8).cuda(), torch.randn(0, 5, 3).cuda())
        assert f.shape == (0, 5, 8) and c.shape == (0, 5, 3)
    f, c = dense(torch.randn(2, 0, 8).cuda(), torch.randn(2, 0, 3).cuda())
    assert f.shape == (2, 0, 8) and c.shape == (2, 0, 3)
    with pytest.raises(RuntimeError, match="out of range"):
        knn(torch.randn(2, 0, 8).cuda(), torch.randn(2, 0, 3).cuda())


def test_cpu_input_raises():
    from egnn_pytorch_amd import EGNN
    layer = EGNN(dim=8)
    with pytest.raises(RuntimeError):
        layer(torch.randn(1, 4, 8), torch.randn(1, 4, 3))


# ---------------------------------------------------------------------------------------------- numerical range
def _range_layer(kw, seed=11):
    cfg = O.EGNNConfig(**kw)
    params = O.random_params(cfg, seed=seed)
    return cfg, params, _module("layer", kw, params)


def test_large_but_representable_inputs_match_at_scale():
    """Inputs well away from O(1) that the split-fp16 arithmetic still carries -- feats x 100, coordinates x 30 (dist^2 up to
    ~5e4), edge features x 1e3 -- against the oracle at 1e-4 RELATIVE TO THE OUTPUT'S SCALE, and without tripping the
    range status."""
    kw = dict(dim=32, num_nearest_neighbors=16, edge_dim=2)
    cfg, params, net = _range_layer(kw)
    rng = np.random.default_rng(zlib.crc32(b"large_inputs"))
    b, n = 2, 64
    feats = (rng.standard_normal((b, n, 32)) * 100.0).astype(np.float32)
    coors = (rng.standard_normal((b, n, 3)) * 30.0).astype(np.float32)
    edges = (rng.standard_normal((b, n, n, 2)) * 1e3).astype(np.float32)
    mask = np.arange(n)[None, :] < np.array([[n], [n - 9]])
    ref_node, ref_co = O.egnn_forward(cfg, params, feats, coors, edges, mask, None)
    node, co = net(_dev(feats), _dev(coors), _dev(edges), _dev(mask))          # (sync range check: would raise here)
    for got, ref in ((node, ref_node), (co, ref_co)):
        scale = float(np.abs(ref).max())
        assert np.isfinite(got.cpu().numpy()).all()
        np.testing.assert_allclose(got.cpu().numpy(), ref, atol=ATOL * max(1.0, scale), rtol=0)


@pytest.mark.parametrize("what,bit", [("feats", 1), ("coors", 4), ("edges", 4)])
def test_out_of_range_inputs_match_the_reference_through_the_wide_range_path(what, bit):
    """feats x 1e6 (|x| >= 65504 entering the fp16 split), coordinates x 1e5 (dist^2 ~ 1e10) and edge features x 1e9: the
    reference computes these in plain fp32 (egnn_pytorch.py:232-233, 287).  The split-fp16 kernels cannot carry them: they set a range
    status bit -- never a silently clamped number -- and the module re-runs the call on the plain-fp32 kernels (csrc/edge_exact.hip,
    egnn_linear_f32): the result matches the oracle at 1e-4 of the output's scale.  Where the word cannot be read for the call itself
    (deferred mode) the old contract holds: non-finite outputs, EGNNRangeError naming the cause at the next check."""
    from egnn_pytorch_amd import EGNNRangeError, _ops, exact_arithmetic
    kw = dict(dim=32, num_nearest_neighbors=8, edge_dim=2)
    cfg, params, net = _range_layer(kw)
    rng = np.random.default_rng(zlib.crc32(what.encode()))
    b, n = 2, 48
    feats = rng.standard_normal((b, n, 32)).astype(np.float32)
    coors = rng.standard_normal((b, n, 3)).astype(np.float32)
    edges = rng.standard_normal((b, n, n, 2)).astype(np.float32)
    if what == "feats":
        feats *= 1e6
    elif what == "coors":
        coors *= 1e5
    else:
        edges *= 1e9
    args = (_dev(feats), _dev(coors), _dev(edges))
    # sync mode (the default): the fast path trips the word, the call is answered by the wide-range path
    with _ops.phase_timer() as pt:
        node, co = net(*args)
    launched = set(pt.summary())
    assert "edge_fused" in launched and "edge_exact" in launched, launched          # the fast attempt, then the re-run
    ref_node, ref_co = O.egnn_forward(cfg, params, feats, coors, edges, None, None)
    ref64 = O.egnn_forward(cfg, {k: v.astype(np.float64) for k, v in params.items()}, feats.astype(np.float64), coors.astype(np.float64),
                           edges.astype(np.float64), None, None)
    for got, ref, r64 in ((node, ref_node, ref64[0]), (co, ref_co, ref64[1])):
        assert np.isfinite(ref).all()                            # (fp32 itself does not overflow on these inputs)
        scale = float(np.abs(ref).max())
        assert np.isfinite(got.cpu().numpy()).all()
        # against the fp32 oracle at 1e-4 of the output's scale, and no further from float64 than 1e-4 of it either
        np.testing.assert_allclose(got.cpu().numpy(), ref, atol=ATOL * max(1.0, scale), rtol=0)
        np.testing.assert_allclose(got.cpu().numpy().astype(np.float64), r64, atol=ATOL * max(1.0, scale), rtol=0)
    # deferred mode: the call returns (non-finite outputs), the next check raises
    old = _ops.RANGE_CHECK
    _ops.RANGE_CHECK = "deferred"
    try:
        node, co = net(*args)
        torch.cuda.synchronize()
        assert not (torch.isfinite(node).all() and torch.isfinite(co).all())
        with pytest.raises(EGNNRangeError) as err:
            _ops.check_range()
        assert _abi_bits(str(err.value)) & bit
        _ops.check_range()                                   # the word was cleared
    finally:
        _ops.RANGE_CHECK = old
    # under autograd (round 5): the call is re-run on the plain-fp32 kernels too and its backward is the recompute path in plain fp32 --
    # the inputs TRAIN: every gradient against float64 autograd of the restated layer over the same neighbour list, 1e-4 of its scale
    import copy
    from egnn_pytorch_amd import autograd as A
    net.train()                                              # (no dropout in this layer: train() only arms autograd-style use)
    with torch.enable_grad(), _ops.phase_timer() as pt:
        f, c, e = (t.clone().requires_grad_(True) for t in args)
        node_g, co_g = net(f, c, e)
        loss = (node_g / max(1.0, float(node_g.detach().abs().max()))).square().sum() + (co_g / max(1.0, float(co_g.detach().abs().max()))).square().sum()
        got = torch.autograd.grad(loss, [f, c, e] + list(net.parameters()), allow_unused=True)
    assert "edge_exact" in pt.summary()
    np.testing.assert_allclose(node_g.detach().cpu().numpy(), ref_node, atol=ATOL * max(1.0, float(np.abs(ref_node).max())), rtol=0)
    with torch.no_grad(), exact_arithmetic():
        idx, rank, radius = net._forward_hip_checked(args[0], args[1], args[2], None, None, None)[3:6]
    n64 = copy.deepcopy(net).double()
    f2, c2, e2 = (t.double().clone().requires_grad_(True) for t in args)
    with torch.enable_grad():
        n2, co2 = A.layer_given_neighbors(n64, f2, c2, e2, None, idx.long(), rank.double(), radius)
        loss2 = (n2 / max(1.0, float(n2.detach().abs().max()))).square().sum() + (co2 / max(1.0, float(co2.detach().abs().max()))).square().sum()
        want = torch.autograd.grad(loss2, [f2, c2, e2] + list(n64.parameters()), allow_unused=True)
    for a, r in zip(got, want):
        assert (a is None) == (r is None)
        if a is not None:
            assert torch.isfinite(a).all()
            assert float((a.double() - r).abs().max()) <= 2e-4 * max(float(r.abs().max()), 1e-30), (float((a.double() - r).abs().max()), float(r.abs().max()))
    net.eval()
    # and the module keeps working afterwards, on the fast path
    with _ops.phase_timer() as pt:
        small = net(_dev(feats * 0 + 1), _dev(rng.standard_normal((b, n, 3)).astype(np.float32)), _dev(edges * 0))
    assert torch.isfinite(small[0]).all() and "edge_exact" not in pt.summary()


@pytest.mark.parametrize("name", golden_names())
def test_golden_on_the_plain_fp32_kernels(name):
    """Every golden case of the reference through `exact_arithmetic()` -- exact-fp32 GEMMs, fp32 node_norm, csrc/edge_exact.hip: all
    layer options (fourier, edge features, dense / k-NN / adjacency, gates, CoorsNorm, clamps, mean pooling, coordinate dimensions,
    networks with their front-end and attention blocks)."""
    from egnn_pytorch_amd import exact_arithmetic, _ops
    meta, params, d = load_golden(name)
    net = _module(meta["kind"], meta["kwargs"], params)
    feats, coors = _dev(d["feats"]), _dev(d["coors"])
    edges, mask, adj = _dev(d.get("edges")), _dev(d.get("mask")), _dev(d.get("adj_mat"))
    with exact_arithmetic(), _ops.phase_timer() as pt:
        if meta["kind"] == "layer":
            node, co = net(feats, coors, edges, mask, adj)
        else:
            node, co = net(feats, coors, adj_mat=adj, edges=edges, mask=mask)
    launched = set(pt.summary())
    assert not launched & {"edge_fused", "node_proj", "node_mlp0", "node_mlp1"}, launched      # nothing of the split-fp16 path ran
    np.testing.assert_allclose(node.cpu().numpy(), d["node_out"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(co.cpu().numpy(), d["coors_out"], atol=ATOL, rtol=0)


def test_more_than_16_per_edge_scalars_run_on_the_plain_fp32_kernels():
    """fourier_features = 8, edge_dim = 5: 22 per-edge scalars -- more than the split-fp16 edge kernels carry (16): inference goes to
    csrc/edge_exact.hip (up to 64) instead of raising."""
    from egnn_pytorch_amd import _ops
    kw = dict(dim=24, fourier_features=8, edge_dim=5, num_nearest_neighbors=12, soft_edges=True, norm_coors=True)
    cfg, params, net = _range_layer(kw, seed=23)
    rng = np.random.default_rng(5)
    b, n = 2, 40
    feats = rng.standard_normal((b, n, 24)).astype(np.float32)
    coors = rng.standard_normal((b, n, 3)).astype(np.float32)
    edges = rng.standard_normal((b, n, n, 5)).astype(np.float32)
    mask = np.arange(n)[None, :] < np.array([[n], [n - 7]])
    with _ops.phase_timer() as pt:
        node, co = net(_dev(feats), _dev(coors), _dev(edges), _dev(mask))
    assert "edge_exact" in pt.summary() and "edge_fused" not in pt.summary()
    ref = O.egnn_forward(cfg, params, feats, coors, edges, mask, None)
    np.testing.assert_allclose(node.cpu().numpy(), ref[0], atol=ATOL, rtol=0)
    np.testing.assert_allclose(co.cpu().numpy(), ref[1], atol=ATOL, rtol=0)


def test_plain_fp32_kernels_agree_with_the_fast_path_at_the_north_star_width():
    """dim 512, k 32, ragged mask: the two arithmetic classes on the same inputs -- both within 1e-4 of the oracle, and of each other."""
    from egnn_pytorch_amd import exact_arithmetic
    kw = dict(dim=512, num_nearest_neighbors=32)
    cfg, params, net = _range_layer(kw, seed=17)
    rng = np.random.default_rng(99)
    b, n = 2, 160
    feats = rng.standard_normal((b, n, 512)).astype(np.float32)
    coors = rng.standard_normal((b, n, 3)).astype(np.float32)
    mask = np.arange(n)[None, :] < np.array([[n], [n - 31]])
    fast = net(_dev(feats), _dev(coors), mask=_dev(mask))
    with exact_arithmetic():
        exact = net(_dev(feats), _dev(coors), mask=_dev(mask))
    ref = O.egnn_forward(cfg, params, feats, coors, None, mask, None)
    for f, e, r in zip(fast, exact, ref):
        np.testing.assert_allclose(e.cpu().numpy(), r, atol=ATOL, rtol=0)
        np.testing.assert_allclose(e.cpu().numpy(), f.cpu().numpy(), atol=ATOL, rtol=0)


def _abi_bits(message):
    from egnn_pytorch_amd import _abi
    return sum(bit for bit, text in _abi.RANGE_BITS.items() if text in message)


def test_nan_padding_behind_the_mask_is_harmless():
    """Padded nodes may hold anything -- the reference zeroes masked pairs with masked_fill (egnn_pytorch.py:322) and never
    selects masked neighbours while enough valid ones exist.  NaN-padded and zero-padded batches must give bit-identical
    outputs on the valid nodes, and the range status must stay clear."""
    kw = dict(dim=32, num_nearest_neighbors=8)
    cfg, params, net = _range_layer(kw, seed=5)
    rng = np.random.default_rng(7)
    b, n = 2, 40
    lens = np.array([40, 23])
    mask = np.arange(n)[None, :] < lens[:, None]
    feats = rng.standard_normal((b, n, 32)).astype(np.float32)
    coors = rng.standard_normal((b, n, 3)).astype(np.float32)
    f0, c0 = feats.copy(), coors.copy()
    f0[~mask] = 0.0
    f1 = f0.copy()
    f1[~mask] = np.nan
    n0, co0 = net(_dev(f0), _dev(c0), mask=_dev(mask))
    n1, co1 = net(_dev(f1), _dev(c0), mask=_dev(mask))
    m = _dev(mask)
    assert torch.equal(n0[m], n1[m]) and torch.equal(co0[m], co1[m])
    assert torch.isfinite(n1[m]).all()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-4), (torch.bfloat16, 3e-2), (torch.float16, 4e-3)])
def test_other_float_dtypes_at_the_boundary(dtype, tol):
    """The reference is dtype-generic (its own tests run in float64, tests/test_equivariance.py:6).  The gfx950 path accepts
    float64 / bfloat16 / float16 modules and inputs and returns the callers' dtype; bfloat16 / float16 modules compute in its
    fp32-class arithmetic (float64 ones on the float64 kernels: tests/test_float64_property_suite.py has their 1e-10 bars):
    against the fp32 oracle within what the output dtype can hold."""
    from egnn_pytorch_amd import EGNN
    kw = dict(dim=32, num_nearest_neighbors=8, norm_feats=True)
    cfg = O.EGNNConfig(**kw)
    params = O.random_params(cfg, seed=21)
    rng = np.random.default_rng(3)
    feats = rng.standard_normal((2, 40, 32)).astype(np.float32)
    coors = rng.standard_normal((2, 40, 3)).astype(np.float32)
    mask = np.arange(40)[None, :] < np.array([[40], [31]])
    net = EGNN(**kw)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    net = net.to(dtype).cuda().eval()
    f, c = _dev(feats).to(dtype), _dev(coors).to(dtype)
    node, co = net(f, c, mask=_dev(mask))
    assert node.dtype == dtype and co.dtype == dtype
    # the oracle sees the inputs / weights as the module does: after rounding to `dtype`
    rp = {k: v.detach().float().cpu().numpy() for k, v in net.state_dict().items()}
    ref_node, ref_co = O.egnn_forward(cfg, rp, f.float().cpu().numpy(), c.float().cpu().numpy(), mask=mask)
    np.testing.assert_allclose(node.float().cpu().numpy(), ref_node, atol=tol * max(1.0, float(np.abs(ref_node).max())), rtol=0)
    np.testing.assert_allclose(co.float().cpu().numpy(), ref_co, atol=tol * max(1.0, float(np.abs(ref_co).max())), rtol=0)


def test_undamped_stacked_network_is_as_close_to_float64_as_the_fp32_reference():
    """VERDICT r1 (weak #2): an UNDAMPED 3-layer xavier-scale network.  Its activations reach the hundreds, where the
    fp32 reference itself sits 1e-4 ... 1e-3 away from a float64 evaluation, so an absolute 1e-4 against the fp32 oracle
    would measure summation order.  The meaningful bar: against the float64 oracle, the HIP path may be at most 4x as far
    as the fp32 oracle is (22-bit products against 24), and within 2e-5 of the output's scale.  Dense all-pairs graphs so
    that no neighbour selection can differ between the float64 and fp32 evaluations."""
    kwargs = dict(depth=3, dim=64, norm_coors=True)
    cfg = O.EGNNConfig(dim=64, norm_feats=True, norm_coors=True)
    rng = np.random.default_rng(zlib.crc32(b"undamped_network"))
    params = {}
    for layer in range(3):
        params.update(O.random_params(cfg, seed=300 + layer, prefix=f"layers.{layer}.1."))
    b, n = 2, 48
    feats = rng.standard_normal((b, n, 64)).astype(np.float32)
    coors = rng.standard_normal((b, n, 3)).astype(np.float32)
    mask = np.arange(n)[None, :] < np.array([[n], [n - 11]])
    ref32 = O.egnn_network_forward(3, cfg, params, feats, coors, mask=mask)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    ref64 = O.egnn_network_forward(3, cfg, p64, feats.astype(np.float64), coors.astype(np.float64), mask=mask)
    net = _module("network", kwargs, params)
    got = net(_dev(feats), _dev(coors), mask=_dev(mask))
    for g, r32, r64, what in zip(got, ref32, ref64, ("feats", "coors")):
        scale = float(np.abs(r64).max())
        e_ref = float(np.abs(r32.astype(np.float64) - r64).max())
        e_hip = float(np.abs(g.cpu().numpy().astype(np.float64) - r64).max())
        print(f"{what}: |out| <= {scale:.1f}, fp32 oracle vs float64 {e_ref:.2e}, HIP vs float64 {e_hip:.2e}")
        assert e_hip <= max(4.0 * e_ref, 1e-6 * scale), (what, scale, e_ref, e_hip)
        assert e_hip <= 2e-5 * max(scale, 1.0), (what, scale, e_hip)


@pytest.mark.parametrize("name,kwargs,b,n,chunk", [
    ("north_star", dict(dim=512, num_nearest_neighbors=32), 64, 1024, 16),
    ("c3_layer", dict(dim=128, num_nearest_neighbors=32, norm_feats=True), 64, 1024, 64),
])
def test_full_batch_against_the_reference_module_itself(name, kwargs, b, n, chunk):
    """VERDICT r1 (weak #2): the full-size configurations were compared on 1-2 graphs only (the numpy oracle needs seconds per
    graph).  Here ALL graphs of the metric's batch are compared with the REFERENCE MODULE ITSELF (oracle/_ref, byte-compiled
    from /root/reference) evaluated in fp32 on the same MI355X through PyTorch eager, `chunk` graphs at a time (it materialises
    E x (Din + 2H) activations): every row of node_out / coors_out, ragged masks, within 1e-4."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import build_ref
    if not build_ref.build():
        pytest.skip("oracle/_ref not built")
    ref = build_ref.import_reference()
    cfg = O.EGNNConfig(**kwargs)
    params = O.random_params(cfg, seed=7)
    net = _module("layer", kwargs, params)
    rnet = ref.EGNN(**kwargs)
    rnet.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    rnet = rnet.cuda().eval()
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    feats, coors = torch.randn(b, n, kwargs["dim"], generator=g).cuda(), torch.randn(b, n, 3, generator=g).cuda()
    lens = torch.randint(n // 2, n + 1, (b,), generator=g)
    mask = (torch.arange(n)[None] < lens[:, None]).cuda()
    node, co = net(feats, coors, mask=mask)
    worst_n = worst_c = 0.0
    for lo in range(0, b, chunk):
        rn, rc = rnet(feats[lo:lo + chunk], coors[lo:lo + chunk], mask=mask[lo:lo + chunk])
        worst_n = max(worst_n, float((node[lo:lo + chunk] - rn).abs().max()))
        worst_c = max(worst_c, float((co[lo:lo + chunk] - rc).abs().max()))
    print(f"{name}: {b} graphs x {n} nodes vs the reference module on the GPU: max|d feats| = {worst_n:.2e}, max|d coors| = {worst_c:.2e}")
    assert worst_n <= ATOL and worst_c <= ATOL, (worst_n, worst_c)


def test_network_edge_lookup_equals_the_materialised_tensor():
    """EGNN_Network's per-pair edge features (edge-token and adjacency-degree embeddings, egnn_pytorch.py:410-432): under no_grad
    they are looked up for the K selected pairs only (EdgeLookup / egnn_edge_features_gather_f32); under autograd the
    (B,N,N,edge_dim+adj_dim) tensor is materialised as upstream.  Same numbers, bit for bit."""
    from egnn_pytorch_amd import EGNN_Network
    torch.manual_seed(3)
    for kw in (dict(num_tokens=10, num_edge_tokens=5, edge_dim=4, num_adj_degrees=2, adj_dim=3, num_nearest_neighbors=8),
               dict(num_tokens=10, num_adj_degrees=3, adj_dim=2, only_sparse_neighbors=True),
               dict(num_tokens=10, num_edge_tokens=7, edge_dim=5)):                       # dense, edge tokens only
        net = EGNN_Network(depth=2, dim=32, **kw).cuda()
        with torch.no_grad():
            for p in net.parameters():
                if p.dim() > 1 and p.shape[0] > 1 and p.shape[1] > 1:
                    p.mul_(20.0)
        g = torch.Generator().manual_seed(1)
        b, n = 2, 40
        seq = torch.randint(0, 10, (b, n), generator=g).cuda()
        coors = torch.randn(b, n, 3, generator=g).cuda()
        i = torch.arange(n)
        adj = ((i[:, None] - i[None, :]).abs() <= 1).cuda() if "num_adj_degrees" in kw else None
        etok = torch.randint(0, kw["num_edge_tokens"], (b, n, n), generator=g).cuda() if "num_edge_tokens" in kw else None
        mask = (torch.arange(n)[None] < torch.tensor([[n], [n - 7]])).cuda()
        lazy = net(seq, coors, adj_mat=adj, edges=etok, mask=mask)                       # (autouse fixture: no_grad)
        with torch.enable_grad():
            full = net(seq, coors, adj_mat=adj, edges=etok, mask=mask)
        assert full[0].requires_grad
        assert torch.equal(lazy[0], full[0].detach()) and torch.equal(lazy[1], full[1].detach()), kw


@pytest.mark.parametrize("dim,heads,dim_head,tokens", [(32, 2, 16, 4), (64, 8, 64, 4), (48, 3, 80, 6)])
def test_global_attention_block_on_the_hip_kernels(dim, heads, dim_head, tokens):
    """GlobalLinearAttention (egnn_pytorch.py:115-144) on the device kernels -- projections on the split-f16 GEMM, LayerNorms
    in its operand packing, GELU in its epilogue, the two attention cores as HIP kernels -- against the oracle's restatement,
    ragged mask with one fully masked graph (uniform softmax, finite)."""
    from egnn_pytorch_amd.attention import GlobalLinearAttention
    torch.manual_seed(dim)
    blk = GlobalLinearAttention(dim=dim, heads=heads, dim_head=dim_head).cuda().eval()
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() > 1:
                p.mul_(2.0)
    g = torch.Generator().manual_seed(2)
    b, n = 3, 70
    x = torch.randn(b, n, dim, generator=g).cuda()
    q = torch.randn(b, tokens, dim, generator=g).cuda()
    mask = (torch.arange(n)[None] < torch.tensor([[n], [n - 20], [0]])).cuda()
    got_x, got_q = blk(x, q, mask=mask)
    params = {"blk." + k: v.detach().cpu().numpy() for k, v in blk.state_dict().items()}
    ref_x, ref_q = O.global_linear_attention(params, "blk.", x.cpu().numpy(), q.cpu().numpy(), heads, mask=mask.cpu().numpy())
    assert torch.isfinite(got_x).all() and torch.isfinite(got_q).all()
    np.testing.assert_allclose(got_x.cpu().numpy(), ref_x, atol=ATOL, rtol=0)
    np.testing.assert_allclose(got_q.cpu().numpy(), ref_q, atol=ATOL, rtol=0)
    # and the plain module (ATen) agrees too: the path autograd uses
    with torch.enable_grad():
        at_x, at_q = blk(x, q, mask=mask)
    np.testing.assert_allclose(got_x.cpu().numpy(), at_x.detach().cpu().numpy(), atol=ATOL, rtol=0)


def test_network_without_layers_returns_its_embeddings():
    """depth = 0: the reference runs an empty layer loop and returns the embedded features and the coordinates unchanged
    (egnn_pytorch.py:442-454); nothing of the first layer may be looked at."""
    from egnn_pytorch_amd import EGNN_Network
    torch.manual_seed(3)
    net = EGNN_Network(num_tokens=7, num_positions=12, dim=8, depth=0).cuda().eval()
    tokens = torch.randint(0, 7, (2, 9)).cuda()
    coors = torch.randn(2, 9, 3).cuda()
    with torch.no_grad():
        feats, out_coors = net(tokens, coors)
        want = net.token_emb(tokens) + net.pos_emb(torch.arange(9, device="cuda"))[None]
    assert torch.equal(feats, want) and torch.equal(out_coors, coors)


@pytest.mark.parametrize("pattern", ["ragged", "scattered", "one_graph_empty", "few_real", "blocks_of_four"])
@pytest.mark.parametrize("kwargs", [dict(dim=64, num_nearest_neighbors=32), dict(dim=32, num_nearest_neighbors=64, norm_coors=True, soft_edges=True,
                                                                               m_pool_method="mean", norm_feats=True)],
                         ids=["k32", "k64_flags"])
def test_padded_nodes_are_skipped_without_changing_anything(kwargs, pattern):
    """The wave-per-node edge kernel skips rounds whose edges are all masked out (padded nodes; whole workgroups when four consecutive
    positions of the Morton order are padding -- the order lists padded nodes last): the outputs of real AND padded nodes against the
    oracle (padded rows: node_mlp([LN(h) | 0]) + h and the input coordinates, bit for bit), for masks that are contiguous, scattered,
    empty for a whole graph, and nearly empty."""
    b, n, dim = 4, 160, kwargs["dim"]
    rng = np.random.default_rng(zlib.crc32((pattern + str(dim)).encode()))
    cfg = O.EGNNConfig(**kwargs)
    params = O.random_params(cfg, seed=23)
    feats = rng.standard_normal((b, n, dim)).astype(np.float32)
    coors = rng.standard_normal((b, n, 3)).astype(np.float32)
    k = kwargs["num_nearest_neighbors"]
    if pattern == "ragged":
        mask = np.arange(n)[None, :] < np.array([n, n - 37, n // 2 + 1, k + 3])[:, None]
    elif pattern == "scattered":
        mask = rng.random((b, n)) < 0.6
    elif pattern == "one_graph_empty":
        mask = rng.random((b, n)) < 0.8
        mask[2] = False
    elif pattern == "few_real":
        mask = np.zeros((b, n), dtype=bool)
        for g in range(b):
            mask[g, rng.choice(n, size=k + 1 + g, replace=False)] = True
    else:                                                                 # real nodes in aligned blocks of four positions
        mask = np.repeat(rng.random((b, n // 4)) < 0.5, 4, axis=1)
    ref_node, ref_co = O.egnn_forward(cfg, params, feats, coors, None, mask, None)
    net = _module("layer", kwargs, params)
    node, co = net(_dev(feats), _dev(coors), None, _dev(mask), None)
    node, co = node.cpu().numpy(), co.cpu().numpy()
    np.testing.assert_allclose(node, ref_node, atol=ATOL, rtol=0)
    np.testing.assert_allclose(co, ref_co, atol=ATOL, rtol=0)
    assert np.array_equal(co[~mask], coors[~mask])                        # padded nodes do not move: the same bits
    # ... and a padded node's features do not depend on anything but its own row (m_i = 0 exactly): move every other node
    coors2 = coors + np.where(mask[..., None], rng.standard_normal((b, n, 3)).astype(np.float32), 0).astype(np.float32)
    node2, _ = net(_dev(feats), _dev(coors2), None, _dev(mask), None)
    assert np.array_equal(node2.cpu().numpy()[~mask], node[~mask])


def test_inputs_on_another_device_are_rejected_not_read():
    """The kernels take raw device pointers: a mask / coors / adj_mat / edges tensor left on the host must raise (as torch's own
    device-mismatch error would upstream), not be dereferenced on the GPU."""
    from egnn_pytorch_amd import EGNN
    layer = EGNN(dim=16, num_nearest_neighbors=4).cuda().eval()
    feats, coors = torch.randn(2, 8, 16).cuda(), torch.randn(2, 8, 3).cuda()
    mask = torch.ones(2, 8, dtype=torch.bool)
    for kw in (dict(mask=mask), dict(adj_mat=torch.eye(8, dtype=torch.bool))):
        with pytest.raises(RuntimeError, match="same device"):
            layer(feats, coors, **kw)
    with pytest.raises(RuntimeError, match="same device"):
        layer(feats, coors.cpu())
    layer(feats, coors, mask=mask.cuda())
