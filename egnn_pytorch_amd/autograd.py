"""Autograd support for the gfx950 EGNN layer (the reference is trained in practice,
denoise_sparse.py:70-78, and every op of egnn_pytorch.py:224-341 is differentiable).

Forward  = the HIP path (neighbour selection, fused edge pass, split-f16 GEMMs) -- nothing of size E x H is kept.
Backward = recompute-in-backward: the layer is re-evaluated for a few graphs at a time as a differentiable chain of ATen
           ops (library GEMMs, gathers along the neighbour list the HIP kernel selected) and differentiated by PyTorch's
           autograd engine; the chunking bounds the E x (Din + 2H) activations the reference materialises for the whole batch.
This is the functional backward (gradients of feats / coors / edges / every parameter agree with the reference's autograd,
tests/test_autograd.py); a fused HIP backward edge kernel (same tiling as the forward, deterministic scatter over the
transposed neighbour list) is the next step and would slot in behind the same autograd.Function.

`layer_given_neighbors` is a restatement of egnn_pytorch.py:262-341 that takes the neighbour list as an input (the selection
itself, :237-260, is not differentiable: topk indices and the `<= valid_radius` comparison carry no gradient upstream
either).  It is also what the CPU tests compare with the reference, independently of the GPU.
"""
from __future__ import annotations

import torch
from torch import nn

# activations of the recompute per edge: a few E x H tensors (pre-activation, activation, gradients); 16 GB of the 288 GB
_BYTES_PER_EDGE_FACTOR = 5.0
_CHUNK_BUDGET_BYTES = 16 << 30


def _fourier(dist, num_encodings):
    """[sin(d / 2^k) (k < F), cos(d / 2^k) (k < F), d]  (egnn_pytorch.py:34-41)."""
    scales = 2.0 ** torch.arange(num_encodings, device=dist.device, dtype=dist.dtype)
    x = dist / scales                                        # (..., 1) / (F) -> (..., F)
    return torch.cat((x.sin(), x.cos(), dist), dim=-1)


def layer_given_neighbors(layer, feats, coors, edges, mask, idx, rank, valid_radius, factorised=True):
    """EGNN.forward (egnn_pytorch.py:262-341) for given neighbours.
    idx (B,N,K) int64 / rank (B,N,K): the selection of :258 (None, None = dense all-pairs, K = N).
    Differentiable in feats, coors, edges and the parameters of `layer`.
    factorised: evaluate the first Linear of edge_mlp as (W_i h_i + b) + W_j h_j + W_s s_ij with the dim-wide products done
    once per node -- the same factorisation the HIP forward uses, 16x fewer flops than Linear(cat(...)) at
    the north-star shape, identical mathematics; False = the reference's literal cat + Linear."""
    b, n, dim = feats.shape
    dense = idx is None
    if dense:
        rel = coors[:, :, None, :] - coors[:, None, :, :]                          # (B,N,N,C)
        e_ij = edges
    else:
        k = idx.shape[-1]
        bi = torch.arange(b, device=feats.device)[:, None, None]
        rel = coors[:, :, None, :] - coors[bi, idx]                               # (B,N,K,C)
        e_ij = None if edges is None else torch.gather(edges, 2, idx[..., None].expand(b, n, k, edges.shape[-1]))
    dist = (rel ** 2).sum(dim=-1, keepdim=True)
    scal = _fourier(dist, layer.fourier_features) if layer.fourier_features > 0 else dist
    if e_ij is not None:
        scal = torch.cat((scal, e_ij), dim=-1)                                    # column order of edge_mlp.0.weight (:282-285)
    if factorised:
        lin = layer.edge_mlp[0]
        w_i, w_j, w_s = lin.weight[:, :dim], lin.weight[:, dim:2 * dim], lin.weight[:, 2 * dim:]
        p_i = feats @ w_i.t() + lin.bias                                          # (B,N,H)
        p_j = feats @ w_j.t()
        z = p_i[:, :, None, :] + (p_j[:, None, :, :] if dense else p_j[bi, idx]) + scal @ w_s.t()
        m_ij = z
        for mod in list(layer.edge_mlp)[1:]:                                      # dropout | Identity, SiLU, Linear, SiLU
            m_ij = mod(m_ij)
    else:
        feats_j = feats[:, None, :, :].expand(b, n, n, dim) if dense else feats[bi, idx]
        feats_i = feats[:, :, None, :].expand_as(feats_j)
        m_ij = layer.edge_mlp(torch.cat((feats_i, feats_j, scal), dim=-1))        # (:287)
    if layer.edge_gate is not None:
        m_ij = m_ij * layer.edge_gate(m_ij)                                       # (:289-290)

    pair_mask = None
    if mask is not None:                                                          # (:292-300) -- the radius / sparse-only cut
        if dense:                                                                 # exists only together with `mask`
            pair_mask = mask[:, :, None] & mask[:, None, :]
        else:
            pair_mask = mask[:, :, None] & mask[bi, idx] & (rank <= valid_radius)

    coors_out = coors
    if layer.coors_mlp is not None:
        w = layer.coors_mlp(m_ij).squeeze(-1)                                     # (:303-304)
        if layer.norm_coors:                                                      # CoorsNorm (:67-77)
            norm = rel.norm(dim=-1, keepdim=True)
            rel = rel / norm.clamp(min=layer.coors_norm.eps) * layer.coors_norm.scale
        if pair_mask is not None:
            w = w.masked_fill(~pair_mask, 0.0)
        if layer.coor_weights_clamp_value is not None:
            c = layer.coor_weights_clamp_value
            w = w.clamp(min=-c, max=c)
        coors_out = (w[..., None] * rel).sum(dim=2) + coors                      # (:315)

    node_out = feats
    if layer.node_mlp is not None:
        if pair_mask is not None:
            m_ij = m_ij.masked_fill(~pair_mask[..., None], 0.0)
        if layer.m_pool_method == "mean":
            if pair_mask is not None:                                             # masked mean with safe_div (:13-16, :326-328)
                cnt = pair_mask.sum(dim=-1, keepdim=True).to(m_ij.dtype)
                m_i = m_ij.sum(dim=2) / cnt.clamp(min=1e-8)
                m_i = m_i.masked_fill(cnt == 0, 0.0)
            else:
                m_i = m_ij.mean(dim=2)
        else:
            m_i = m_ij.sum(dim=2)
        node_out = layer.node_mlp(torch.cat((layer.node_norm(feats), m_i), dim=-1)) + feats      # (:335-337)
    return node_out, coors_out


def _chunk_graphs(layer, n, k, batch):
    din = 2 * layer.dim + 2 * layer.fourier_features + 1 + layer.edge_dim
    per_graph = n * k * (2 * din) * 4.0 * _BYTES_PER_EDGE_FACTOR          # E x H pre-activation, activation, their gradients
    return max(1, min(batch, int(_CHUNK_BUDGET_BYTES // max(per_graph, 1.0))))


class EGNNFunction(torch.autograd.Function):
    """forward: HIP kernels; backward: chunked recompute through autograd (module docstring)."""

    @staticmethod
    def forward(ctx, layer, order_hint, mask, adj_mat, feats, coors, edges, *params):
        with torch.no_grad():
            node_out, coors_out, order, idx, rank, valid_radius = layer._forward_hip_checked(feats, coors, edges, mask, adj_mat, order_hint)
        ctx.layer = layer
        ctx.valid_radius = valid_radius
        ctx.has_edges = edges is not None
        ctx.save_for_backward(feats, coors, edges if edges is not None else feats.new_empty(0),
                              mask if mask is not None else feats.new_empty(0),
                              idx if idx is not None else feats.new_empty(0), rank if rank is not None else feats.new_empty(0))
        ctx.flags = (mask is not None, idx is not None)
        ctx.order = order
        # outputs must not alias the inputs of a custom Function
        if node_out is feats:
            node_out = feats.clone()
        if coors_out is coors:
            coors_out = coors.clone()
        return node_out, coors_out

    @staticmethod
    def backward(ctx, g_node, g_coors):
        layer = ctx.layer
        feats, coors, edges, mask, idx, rank = ctx.saved_tensors
        has_mask, has_idx = ctx.flags
        edges = edges if ctx.has_edges else None
        mask = mask if has_mask else None
        idx = idx.long() if has_idx else None
        rank = rank if has_idx else None
        params = [p for p in layer.parameters()]
        b, n, _ = feats.shape
        k = idx.shape[-1] if idx is not None else n
        step = _chunk_graphs(layer, n, k, b)
        need = ctx.needs_input_grad                      # (layer, order_hint, mask, adj, feats, coors, edges, *params)
        g_feats = torch.zeros_like(feats) if need[4] else None
        g_coors_in = torch.zeros_like(coors) if need[5] else None
        g_edges = torch.zeros_like(edges) if (edges is not None and need[6]) else None
        g_params = [torch.zeros_like(p) if need[7 + i] else None for i, p in enumerate(params)]
        if g_node is None:
            g_node = torch.zeros_like(feats)
        if g_coors is None:
            g_coors = torch.zeros_like(coors)
        for lo in range(0, b, step):
            hi = min(b, lo + step)
            with torch.enable_grad():
                f = feats[lo:hi].detach().requires_grad_(need[4])
                c = coors[lo:hi].detach().requires_grad_(need[5])
                e = None if edges is None else edges[lo:hi].detach().requires_grad_(bool(need[6]))
                out_n, out_c = layer_given_neighbors(layer, f, c, e, None if mask is None else mask[lo:hi],
                                                     None if idx is None else idx[lo:hi], None if rank is None else rank[lo:hi],
                                                     ctx.valid_radius)
                wrt = [t for t in (f, c, e) if t is not None and t.requires_grad] + [p for p, g in zip(params, g_params) if g is not None]
                outs, gouts = [], []
                for o, g in ((out_n, g_node[lo:hi]), (out_c, g_coors[lo:hi])):
                    if o.requires_grad:
                        outs.append(o)
                        gouts.append(g)
                grads = torch.autograd.grad(outs, wrt, gouts, allow_unused=True) if outs and wrt else [None] * len(wrt)
            it = iter(grads)
            if f.requires_grad:
                g = next(it)
                if g is not None:
                    g_feats[lo:hi] = g
            if c.requires_grad:
                g = next(it)
                if g is not None:
                    g_coors_in[lo:hi] = g
            if e is not None and e.requires_grad:
                g = next(it)
                if g is not None:
                    g_edges[lo:hi] = g
            for i, gp in enumerate(g_params):
                if gp is not None:
                    g = next(it)
                    if g is not None:
                        gp.add_(g)
        return (None, None, None, None, g_feats, g_coors_in, g_edges, *g_params)


def wants_grad(layer, *tensors):
    """True when a forward under the current autograd mode has to record a graph."""
    if not torch.is_grad_enabled():
        return False
    if any(t is not None and torch.is_tensor(t) and t.is_floating_point() and t.requires_grad for t in tensors):
        return True
    return any(p.requires_grad for p in layer.parameters())
