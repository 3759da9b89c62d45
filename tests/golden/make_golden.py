"""Generate tests/golden/*.npz from the LIVE reference (lucidrains/egnn-pytorch at /root/reference).

Run in the dev container only (the reference does not travel to the GPU box):

    python tests/golden/make_golden.py

For every case: seeded inputs, the reference layer re-initialised with seeded xavier_normal_
weights (default init is std 1e-3, which makes feature parity vacuous -- SURVEY.md §4), the
reference forward in fp32 on CPU under no_grad/eval, and the (values, indices) the reference's
own `ranking.topk(...)` call returned at egnn_pytorch/egnn_pytorch.py:258 (captured by
wrapping torch.Tensor.topk for the duration of the call).  Everything needed to replay the
case without the reference is stored in the .npz: inputs, state_dict, outputs.
"""
import json
import os
import sys

import numpy as np
import torch

REF = os.environ.get("EGNN_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
from egnn_pytorch import EGNN, EGNN_Network  # noqa: E402  (the reference)

HERE = os.path.dirname(os.path.abspath(__file__))


def reinit(module, seed, coors_out_scale=1.0):
    """Seeded xavier-scale weights.  `coors_out_scale` damps the last coors_mlp Linear: stacked
    layers with O(1) coordinate weights blow the geometry up (|x| -> 1e2 after 3 layers) and turn
    the comparison chaotic; the multi-layer cases use 0.1 so coordinates move by O(0.1) per layer."""
    g = torch.Generator().manual_seed(seed)
    for name, p in module.named_parameters():
        with torch.no_grad():
            if p.ndim == 2:
                std = (2.0 / (p.shape[0] + p.shape[1])) ** 0.5
                if name.endswith("coors_mlp.3.weight"):
                    std *= coors_out_scale
                p.copy_(torch.randn(p.shape, generator=g) * std)
            elif name.endswith("node_norm.weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("node_norm.bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("coors_norm.scale"):
                p.fill_(0.5)
            # Linear biases keep PyTorch's default U(+-1/sqrt(fan_in)) but are re-drawn seeded:
            elif name.endswith("bias"):
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * p.abs().max().clamp(min=1e-3))


class TopkSpy:
    """Record the reference's own neighbour selection (egnn_pytorch/egnn_pytorch.py:258)."""

    def __init__(self):
        self.calls = []

    def __enter__(self):
        self._orig = torch.Tensor.topk
        spy = self

        def topk(t, *a, **kw):
            out = spy._orig(t, *a, **kw)
            spy.calls.append((out[0].clone(), out[1].clone()))
            return out

        torch.Tensor.topk = topk
        return self

    def __exit__(self, *exc):
        torch.Tensor.topk = self._orig


def ragged_mask(b, n, g):
    lens = torch.randint(n // 2, n + 1, (b,), generator=g)
    return torch.arange(n)[None, :] < lens[:, None]


def chain_adj(n):
    i = torch.arange(n)
    return (i[:, None] - i[None, :]).abs() <= 1          # README.md:89-90 (includes the diagonal)


def random_adj(n, g, max_deg=5, limit=7):
    """Symmetric random adjacency with the diagonal set and every off-diagonal degree <= limit
    (= K-1), so that the rank-0 tie group never straddles the top-K boundary and the selected
    neighbour SET is well defined."""
    a = torch.eye(n, dtype=torch.bool)
    deg = [0] * n
    for i in range(n):
        want = int(torch.randint(0, max_deg + 1, (1,), generator=g))
        for j in torch.randperm(n, generator=g)[:want].tolist():
            if j == i or a[i, j] or deg[i] >= limit or deg[j] >= limit:
                continue
            a[i, j] = a[j, i] = True
            deg[i] += 1
            deg[j] += 1
    return a


CASES = [
    # name, kind, ctor kwargs, B, N, flags
    ("c1_dense_dim32", "layer", dict(dim=32), 1, 16, dict()),
    ("dense_edges_mask", "layer", dict(dim=32, edge_dim=4), 2, 16, dict(edges=True, mask=True)),
    ("knn8_mask", "layer", dict(dim=64, num_nearest_neighbors=8), 2, 64, dict(mask=True)),
    ("knn8_nomask", "layer", dict(dim=64, num_nearest_neighbors=8), 2, 64, dict()),
    ("knn32_dim128_mask", "layer", dict(dim=128, num_nearest_neighbors=32), 2, 96, dict(mask=True)),
    ("sparse_chain_edges_mask", "layer", dict(dim=32, edge_dim=4, only_sparse_neighbors=True), 2, 32,
     dict(edges=True, mask=True, adj="chain")),
    ("sparse_chain_nomask", "layer", dict(dim=32, only_sparse_neighbors=True), 2, 32, dict(adj="chain")),
    ("knn8_adj_random_mask", "layer", dict(dim=32, edge_dim=2, num_nearest_neighbors=8), 2, 48,
     dict(edges=True, mask=True, adj="random")),
    ("knn8_adj_batched", "layer", dict(dim=32, num_nearest_neighbors=8), 2, 32,
     dict(mask=True, adj="random_batched")),
    ("all_flags", "layer", dict(dim=32, edge_dim=3, num_nearest_neighbors=12, norm_feats=True, norm_coors=True,
                                m_pool_method="mean", soft_edges=True, coor_weights_clamp_value=0.05,
                                valid_radius=3.0), 2, 40, dict(edges=True, mask=True)),
    ("mean_pool_nomask", "layer", dict(dim=32, num_nearest_neighbors=8, m_pool_method="mean"), 1, 24, dict()),
    ("fourier2_knn", "layer", dict(dim=32, fourier_features=2, num_nearest_neighbors=8), 2, 32, dict(mask=True)),
    ("fourier4_dense_edges", "layer", dict(dim=32, fourier_features=4, edge_dim=2), 1, 16, dict(edges=True)),
    ("no_coors_update", "layer", dict(dim=32, num_nearest_neighbors=8, update_coors=False), 1, 24, dict(mask=True)),
    ("no_feats_update", "layer", dict(dim=32, num_nearest_neighbors=8, update_feats=False), 1, 24, dict(mask=True)),
    ("m_dim8_dim20", "layer", dict(dim=20, m_dim=8, num_nearest_neighbors=5), 2, 20, dict(mask=True)),
    ("net3_dim32_k8_mask", "network", dict(depth=3, dim=32, num_nearest_neighbors=8), 2, 48, dict(mask=True)),
    ("net2_normcoors_clamp", "network", dict(depth=2, dim=32, num_nearest_neighbors=8, norm_coors=True,
                                             coor_weights_clamp_value=2.0), 2, 32, dict(mask=True)),
    # EGNN_Network front-end: token / position / edge-token embeddings, N-degree adjacency
    # expansion + adjacency embedding (README.md:95-120)
    ("net_tokens_pos", "network", dict(depth=2, dim=32, num_tokens=21, num_positions=64, num_nearest_neighbors=8,
                                       coor_weights_clamp_value=2.0), 2, 40, dict(mask=True, tokens=21)),
    ("net_adj_degrees_sparse", "network", dict(depth=2, dim=32, num_tokens=21, num_adj_degrees=3, adj_dim=8,
                                               only_sparse_neighbors=True), 2, 32, dict(mask=True, tokens=21, adj="chain_nodiag")),
    # (sparse-only neighbours: with an expanded adjacency the rank-0 tie group would straddle any fixed K)
    ("net_edge_tokens_adj2", "network", dict(depth=2, dim=32, num_edge_tokens=4, edge_dim=4, num_adj_degrees=2, adj_dim=4,
                                             only_sparse_neighbors=True), 2, 24,
     dict(mask=True, edge_tokens=4, adj="random")),
    # induced-set ("global linear") attention between the layers (egnn_pytorch.py:81-144, 376-388)
    ("net_global_attn", "network", dict(depth=3, dim=32, num_nearest_neighbors=8, global_linear_attn_every=2,
                                        global_linear_attn_heads=2, global_linear_attn_dim_head=8, num_global_tokens=4,
                                        coor_weights_clamp_value=2.0), 2, 40, dict(mask=True)),
    # coordinate dimension other than 3 (the reference works for any C; tests/test_equivariance.py uses 5)
    ("dense_coor_dim5", "layer", dict(dim=32), 1, 16, dict(coor_dim=5)),
    ("knn8_coor_dim5_normcoors_mask", "layer", dict(dim=32, num_nearest_neighbors=8, norm_coors=True), 2, 40,
     dict(mask=True, coor_dim=5)),
    ("knn8_coor_dim2_mask", "layer", dict(dim=32, num_nearest_neighbors=8), 2, 40, dict(mask=True, coor_dim=2)),
    # more than 8 coordinates (the fused kernels stop at 8: these run on the plain kernels; egnn_common.h::egnn_sqdist_any)
    ("knn8_coor_dim11_mask", "layer", dict(dim=32, num_nearest_neighbors=8), 2, 40, dict(mask=True, coor_dim=11)),
    ("knn8_coor_dim33_normcoors", "layer", dict(dim=24, num_nearest_neighbors=8, norm_coors=True, m_pool_method="mean"), 2, 36,
     dict(mask=True, coor_dim=33)),
]


def run_case(idx, name, kind, kwargs, b, n, flags):
    g = torch.Generator().manual_seed(1000 + idx)
    dim = kwargs["dim"]
    edge_dim = kwargs.get("edge_dim", 0)
    feats = torch.randn(b, n, dim, generator=g)
    if flags.get("tokens"):
        feats = torch.randint(0, flags["tokens"], (b, n), generator=g)
    coors = torch.randn(b, n, flags.get("coor_dim", 3), generator=g)
    edges = torch.randn(b, n, n, edge_dim, generator=g) if flags.get("edges") else None
    if flags.get("edge_tokens"):
        edges = torch.randint(0, flags["edge_tokens"], (b, n, n), generator=g)
    mask = ragged_mask(b, n, g) if flags.get("mask") else None
    adj = None
    if flags.get("adj") == "chain":
        adj = chain_adj(n)
    elif flags.get("adj") == "chain_nodiag":
        i = torch.arange(n)
        adj = (i[:, None] - i[None, :]).abs() == 1          # README.md:110 style: neighbours only, no diagonal
    elif flags.get("adj") == "random":
        adj = random_adj(n, g)
    elif flags.get("adj") == "random_batched":
        adj = torch.stack([random_adj(n, g) for _ in range(b)])

    torch.manual_seed(5000 + idx)       # default init feeds the bias magnitudes below: seed it so files regenerate bit for bit
    net = EGNN(**kwargs) if kind == "layer" else EGNN_Network(**kwargs)
    reinit(net, 7000 + idx, coors_out_scale=0.1 if kind == "network" else 1.0)
    net.eval()
    with torch.no_grad(), TopkSpy() as spy:
        if kind == "layer":
            out = net(feats, coors, edges, mask, adj)
        else:
            out = net(feats, coors, adj_mat=adj, edges=edges, mask=mask, return_coor_changes=True)
    node_out, coors_out = out[0], out[1]

    rec = {"feats": feats.numpy(), "coors": coors.numpy(),
           "node_out": node_out.numpy(), "coors_out": coors_out.numpy()}
    if edges is not None:
        rec["edges"] = edges.numpy()
    if mask is not None:
        rec["mask"] = mask.numpy()
    if adj is not None:
        rec["adj_mat"] = adj.numpy()
    if kind == "network":
        for i, c in enumerate(out[2]):
            rec[f"coor_change.{i}"] = c.numpy()
    for i, (v, ix) in enumerate(spy.calls):
        rec[f"topk_values.{i}"] = v.numpy()
        rec[f"topk_indices.{i}"] = ix.numpy().astype(np.int32)
    for k, v in net.state_dict().items():
        rec["param:" + k] = v.numpy()
    meta = {"name": name, "kind": kind, "kwargs": kwargs, "B": b, "N": n, "flags": flags,
            "torch": torch.__version__, "n_topk": len(spy.calls)}
    rec["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **rec)
    return path, os.path.getsize(path)


if __name__ == "__main__":
    torch.set_num_threads(4)
    total = 0
    only = set(sys.argv[1:])                    # optional: regenerate just the named cases
    for idx, case in enumerate(CASES):
        if only and case[0] not in only:
            continue
        path, size = run_case(idx, *case)
        total += size
        print(f"{os.path.basename(path):40s} {size/1024:8.1f} KiB")
    print(f"total {total/1024:.1f} KiB")
