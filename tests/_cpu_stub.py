"""TEST INFRASTRUCTURE: a CPU stand-in for the kernel layer of egnn_pytorch_amd, so that everything ABOVE the kernels -- EGNN.forward /
EGNN_Network.forward, the layer loop, sharding helpers, bench.py's rank logic -- can be driven end to end by world_size-2 `gloo`
processes on a machine without a GPU.  `install()` replaces EGNN._forward_with_hint (the method that launches the HIP kernels) with a
torch restatement: the neighbour selection of egnn_pytorch.py:232-260 as tensor ops + egnn_pytorch_amd.autograd.layer_given_neighbors.
Never imported by the package: the product has no CPU path (tests/test_host_logic.py::test_no_cpu_fallback)."""
import torch


def _select(layer, coors, mask, adj_mat):
    b, n, _ = coors.shape
    k = layer.num_nearest_neighbors
    radius = layer.valid_radius
    if not (k > 0 or layer.only_sparse_neighbors):
        return None, None, radius
    rel = coors[:, :, None, :] - coors[:, None, :, :]
    ranking = (rel ** 2).sum(dim=-1)
    if mask is not None:
        ranking = ranking.masked_fill(~(mask[:, :, None] & mask[:, None, :]), 1e5)
    if adj_mat is not None:
        adj = adj_mat if adj_mat.dim() == 3 else adj_mat[None].expand(b, n, n)
        if layer.only_sparse_neighbors:
            k = int(adj.float().sum(dim=-1).max().item())
            radius = 0.0
        eye = torch.eye(n, dtype=torch.bool, device=coors.device)[None]
        ranking = ranking.masked_fill(eye, -1.0).masked_fill(adj & ~eye, 0.0)
    rank, idx = ranking.topk(k, dim=-1, largest=False)
    return idx, rank, radius


def install():
    from egnn_pytorch_amd import _ops, autograd as A, layer as L

    def forward_stub(self, feats, coors, edges, mask, adj_mat, order_hint, want_u=False, drop_seed=None, presel=None, prefetch=None):
        idx, rank, radius = _select(self, coors, mask, adj_mat)
        with torch.no_grad():
            node, co = A.layer_given_neighbors(self, feats, coors, edges, mask, idx, rank, radius)
        return node, co, None, None if idx is None else idx.int(), rank, radius, None, None

    L.EGNN._forward_with_hint = forward_stub
    _ops.RANGE_CHECK = "off"                                # (the status word lives on a device)
