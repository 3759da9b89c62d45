"""Autograd support for the gfx950 EGNN layer (the reference is trained in practice,
denoise_sparse.py:70-78, and every op of egnn_pytorch.py:224-341 is differentiable).

Forward  = the HIP path (neighbour selection, fused edge pass, split-f16 GEMMs -- or the plain fp32 / float64 kernels for the layers
           that run there).  Nothing of size E x H is kept: the saved tensors are the inputs, the neighbour list, the projection table
           (node-level) and u (E x m), the output of edge_mlp's second Linear, which the forward edge kernel writes on the side when a
           graph is being recorded.
Backward, three paths:
  `_backward_native` -- fp32 (half / bfloat16 modules through an fp32 shadow), every shape the fused forward kernels cover (m_dim <= 64,
           coordinate dimension 1 .. 8, up to 16 per-edge scalars, with or without training-mode dropout):
             * behind u: node_norm / node_mlp on the split-f16 GEMMs (`_node_mlp_backward`), the per-edge chain (second SiLU, gate, masks,
               coors_mlp, CoorsNorm, clamp, coordinate update, pooling) in closed form -- egnn_edge_tail_bwd_f32 (csrc/edge_tail.hip, matrix
               cores) for m_dim <= 16 and 3-D coordinates, egnn_edge_tail_exact_bwd_f32 (csrc/edge_exact_bwd.hip, one thread per edge) for
               wider heads / other coordinate dimensions; `tail_edge_backward` is their specification (autograd on E x m tensors,
               `layer_tail`, only on the CPU and behind EGNN_TAIL_GENERIC=0)  ->  gU, d/d (x_i - x_j) and those parameters' gradients;
             * the E x H work -- z = P_i + P_j + W_s s, a = SiLU(z), dz = (W2^T gU) SiLU'(z) and their contractions -- on
               egnn_edge_bwd_pass_f32 (csrc/edge_bwd.hip; `_edge_contract_fused`): one pass over the edges grouped by source node
               (d/d P_i, d/d W_s, d/d scalars) and one over the edges sorted by destination (d/d P_j, d/d W_2), everything recomputed
               and contracted in registers; heads wider than 16 channels: once per block of 16 channels (linear in gU);
             * the node-level products -- d/d feats, d/d edge_mlp.0, node_mlp -- on the forward's split-f16 GEMM (`_ops.grad_nn / grad_tn`).
  `_backward_exact` (round 5) -- the layers whose forward runs on the plain kernels (csrc/edge_exact.hip): float64 modules (the reference's
           own training recipe, denoise_sparse.py:11, 23-32), calls answered by the wide-range path (values beyond the split-fp16 range),
           more than 16 per-edge scalars / 8 coordinates / 64 message channels.  The per-edge tail through autograd on E x m tensors,
           the E x H work on egnn_edge_exact_bwd_f32 / _f64 + egnn_edge_exact_node_sums_* (csrc/edge_exact_bwd.hip), every contraction on
           the exact GEMMs (egnn_linear_f32 / egnn_linear_f64), in the arithmetic of the forward.
  `_backward_recompute` -- what is left: more per-edge scalars than the exact backward keeps in LDS (80 in fp32, 40 in float64), EGNN_NATIVE_BACKWARD=0 /
           EGNN_NATIVE_BACKWARD_EXACT=0, CPU tensors (the tests); also the native paths' reference in the tests.  The whole layer
           re-evaluated a few graphs at a time as a differentiable chain of ATen ops over the neighbour list the HIP kernel selected,
           factorised like the forward, and differentiated by autograd.
Gradients of feats / coors / edges / every parameter agree with the reference's autograd (tests/test_autograd.py).  Every sum
over edges has a fixed order (partial rows + egnn_rows_gather_sum_f32, per-workgroup partials summed by index: no float
atomics) -- the native backward is bit-reproducible.

`layer_given_neighbors` is a restatement of egnn_pytorch.py:262-341 that takes the neighbour list as an input (the selection
itself, :237-260, is not differentiable: topk indices and the `<= valid_radius` comparison carry no gradient upstream
either).  It is also what the CPU tests compare with the reference, independently of the GPU.
"""
from __future__ import annotations

import os

import torch
from torch import nn

# EGNN_NATIVE_BACKWARD=0: always the pure-ATen recompute (the reference implementation of the backward)
_NATIVE_MODE = os.environ.get("EGNN_NATIVE_BACKWARD", "1")
_NATIVE = _NATIVE_MODE != "0"
# which pass of the fused backward carries d/d W_2: "dest" (default), "both" = the by-source pass carries everything (tuning knob)
_FUSED_SPLIT = os.environ.get("EGNN_BWD_SPLIT", "dest")
_TAIL_KERNEL = os.environ.get("EGNN_BWD_TAIL_KERNEL", "1") != "0"      # 0: the per-edge chain behind u through autograd
_GRAD_GEMM = os.environ.get("EGNN_BWD_GRAD_GEMM", "1") != "0"          # 0: the node-level gradient products as fp32 library GEMMs
_TAIL_REDUCE = os.environ.get("EGNN_BWD_TAIL_REDUCE", "1") != "0"      # 0: the tail kernel writes its E x 64 factors out for library reductions
_NATIVE_EXACT = os.environ.get("EGNN_NATIVE_BACKWARD_EXACT", "1") != "0"   # 0: float64 / wide-range / wide-shape layers on the recompute path
_TAIL_GENERIC = os.environ.get("EGNN_TAIL_GENERIC", "1") != "0"        # 0: the per-edge chain of wide heads / other coordinate dimensions / the plain path through autograd
_EXACT_BWD_BYTES = int(float(os.environ.get("EGNN_EXACT_BWD_GB", "2")) * (1 << 30))   # budget of the a^T / dz^T tables per chunk of graphs
_KEEP_PROJ = os.environ.get("EGNN_BWD_KEEP_PROJ", "1") != "0"          # 0: the backward recomputes the P_i | P_j table (B N x 2 Hp fp32 less to keep)
_FUSED_MAX_GRAPHS = 0                 # tests: force the chunking over graphs that very large batches need (0 = by size only)

# activations of the recompute per edge: a few E x H tensors (pre-activation, activation, gradients); 16 GB of the 288 GB
_BYTES_PER_EDGE_FACTOR = 5.0
_CHUNK_BUDGET_BYTES = 16 << 30


def _fourier(dist, num_encodings):
    """[sin(d / 2^k) (k < F), cos(d / 2^k) (k < F), d]  (egnn_pytorch.py:34-41)."""
    scales = 2.0 ** torch.arange(num_encodings, device=dist.device, dtype=dist.dtype)
    x = dist / scales                                        # (..., 1) / (F) -> (..., F)
    return torch.cat((x.sin(), x.cos(), dist), dim=-1)


def _tn(a, b):
    """a^T @ b for tall a (R, ca), b (R, cb): the contraction runs over the R rows and the output is small, so a single library
    GEMM leaves most of the 256 CUs idle (2.2 ms for 65536 x 2080 x 512, 1.8 ms for 2M x 64 x 16).  Split-K by hand: slabs of rows
    as a batched GEMM, partial products summed in fixed order (1.2 / 0.2 ms)."""
    r, ca = a.shape
    cb = b.shape[1]
    if not a.is_cuda or r < (1 << 14):
        return a.t() @ b
    tiles = ((ca + 127) // 128) * ((cb + 127) // 128)
    s = 1
    while s < 256 and tiles * s < 512 and r % (2 * s) == 0 and r // (2 * s) >= 512:
        s *= 2
    if s == 1:
        return a.t() @ b
    return torch.bmm(a.view(s, r // s, ca).transpose(1, 2), b.view(s, r // s, cb)).sum(dim=0)


class _TallLinear(torch.autograd.Function):
    """F.linear for per-edge inputs (millions of rows, a few dozen features): the weight gradient is a `_tn` product."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        gx = (g2 @ w).view_as(x) if ctx.needs_input_grad[0] else None
        gw = _tn(g2, x.reshape(-1, x.shape[-1])) if ctx.needs_input_grad[1] else None
        gb = g2.sum(dim=0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb


def _per_edge(module, x):
    """module(x) for Sequentials of Linear / Dropout / SiLU / Sigmoid applied to very many rows -- the per-edge heads (edge_gate,
    coors_mlp) and the node-level node_mlp -- with the Linears' weight gradients computed by `_tn`."""
    if not x.is_cuda:
        return module(x)
    for sub in (module if isinstance(module, nn.Sequential) else [module]):
        if isinstance(sub, nn.Linear):
            x = _TallLinear.apply(x, sub.weight, sub.bias)
        else:
            x = sub(x)
    return x


def edge_scalars(layer, coors, edges, idx):
    """rel = x_i - x_j and the per-edge scalars [fourier(d), d, e_ij] in the column order of edge_mlp.0.weight (:282-285)."""
    b, n, _ = coors.shape
    if idx is None:
        rel = coors[:, :, None, :] - coors[:, None, :, :]                          # (B,N,N,C)
        e_ij = edges
    else:
        k = idx.shape[-1]
        bi = torch.arange(b, device=coors.device)[:, None, None]
        rel = coors[:, :, None, :] - coors[bi, idx]                               # (B,N,K,C)
        e_ij = None if edges is None else torch.gather(edges, 2, idx[..., None].expand(b, n, k, edges.shape[-1]))
    dist = (rel ** 2).sum(dim=-1, keepdim=True)
    scal = _fourier(dist, layer.fourier_features) if layer.fourier_features > 0 else dist
    if e_ij is not None:
        scal = torch.cat((scal, e_ij), dim=-1)
    return rel, scal


def _mlp_drop(module, x, drop, site, rows):
    """A Linear -> Dropout -> SiLU -> Linear [-> SiLU] Sequential with its dropout given by the kernels' hash mask
    (drop = (p, seed); rows = the mask row of every leading index of x) instead of the module's own nn.Dropout."""
    from . import _dropout
    mods = list(module)
    h = _TallLinear.apply(x, mods[0].weight, mods[0].bias) if x.is_cuda else mods[0](x)
    h = _dropout.apply(h, drop[1], site, rows, drop[0])
    for sub in mods[2:]:
        h = _TallLinear.apply(h, sub.weight, sub.bias) if (isinstance(sub, nn.Linear) and h.is_cuda) else sub(h)
    return h


def _edge_rows(b, n, k, device, graph_offset):
    """global edge ids (b N K + i K + k) of a chunk of graphs that starts at graph `graph_offset`: (b, n, k) int64"""
    return (torch.arange(b * n * k, device=device) + graph_offset * n * k).view(b, n, k)


def layer_tail(layer, feats, coors, u, rel, mask, idx, rank, valid_radius, drop=None, graph_offset=0):
    """Everything of EGNN.forward behind the second Linear of edge_mlp (:183-341): u (B,N,K,m_dim) = edge_mlp[3](...) ->
    second SiLU, gate, masks, coors_mlp / CoorsNorm / clamp / coordinate update, pooling, node_norm + node_mlp + residual.
    drop = (p, seed): training-mode dropout with the kernels' hash masks (egnn_pytorch_amd/_dropout.py)."""
    from . import _dropout
    b = feats.shape[0]
    m_ij = layer.edge_mlp[4](u)
    if layer.edge_gate is not None:
        m_ij = m_ij * _per_edge(layer.edge_gate, m_ij)                            # (:289-290)

    pair_mask = None
    if mask is not None:                                                          # (:292-300) -- the radius / sparse-only cut
        if idx is None:                                                           # exists only together with `mask`
            pair_mask = mask[:, :, None] & mask[:, None, :]
        else:
            bi = torch.arange(b, device=feats.device)[:, None, None]
            pair_mask = mask[:, :, None] & mask[bi, idx] & (rank <= valid_radius)

    coors_out = coors
    if layer.coors_mlp is not None:
        if drop is not None:
            w = _mlp_drop(layer.coors_mlp, m_ij, drop, _dropout.SITE_COORS,
                          _edge_rows(b, m_ij.shape[1], m_ij.shape[2], m_ij.device, graph_offset)).squeeze(-1)
        else:
            w = _per_edge(layer.coors_mlp, m_ij).squeeze(-1)                      # (:303-304)
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
        node_in = torch.cat((layer.node_norm(feats), m_i), dim=-1)
        if drop is not None:
            n_ = feats.shape[1]
            rows = (torch.arange(b * n_, device=feats.device) + graph_offset * n_).view(b, n_)
            node_out = _mlp_drop(layer.node_mlp, node_in, drop, _dropout.SITE_NODE, rows) + feats
        else:
            node_out = _per_edge(layer.node_mlp, node_in) + feats                 # (:335-337)
    return node_out, coors_out


def tail_edge_backward(layer, u, coors, idx, pair_mask, g_coors_out, g_msum):
    """The per-edge part of the backward of `layer_tail` in closed form (what csrc/edge_tail.hip evaluates one edge per lane;
    this torch version is its specification and the CPU tests' subject): given
        u (B,N,K,m) the output of edge_mlp's second Linear, g_coors_out (B,N,3) = d loss / d coors_out and
        g_msum (B,N,m) = d loss / d (sum over k of the pair-masked m_ij)  (sum pooling: d/d m_i; mean: that / count),
    returns d loss / d u (B,N,K,m), d loss / d rel (B,N,K,3; rel = x_i - x_j, without the distance path), and what the parameter
    gradients of coors_mlp / CoorsNorm are sums of: g_hid (E, 4m) = d/d (pre-activation of coors_mlp's SiLU), a3 (E, 4m) its
    activation, g_w (E,) = d/d (coors_mlp's output), g_scale (E,) the per-edge terms of d/d coors_norm.scale; with the edge gate
    (soft_edges, :289-290) also g_gate (E,) = d/d (the gate's pre-activation) and m0 = SiLU(u) next to the gated m.
    Covers: coordinate dimension 3; edge gate, masks, CoorsNorm, clamp as configured."""
    b, n, k, m = u.shape
    lin3, lin4 = layer.coors_mlp[0], layer.coors_mlp[3]
    w3, b3, w4, b4 = lin3.weight, lin3.bias, lin4.weight[0], lin4.bias[0]
    sg_u = torch.sigmoid(u)
    m0 = u * sg_u                                                               # SiLU(u)
    gate = layer.edge_gate is not None
    if gate:
        gw, gb = layer.edge_gate[0].weight[0], layer.edge_gate[0].bias[0]
        gt = torch.sigmoid(m0 @ gw + gb)[..., None]
        mm = m0 * gt                                                            # m_ij = SiLU(u) * sigmoid(gate(SiLU(u)))
    else:
        mm = m0
    hid = mm @ w3.t() + b3
    sg_h = torch.sigmoid(hid)
    a3 = hid * sg_h
    w = a3 @ w4 + b4                                                            # (B,N,K)
    if idx is None:
        rel = coors[:, :, None, :] - coors[:, None, :, :]
    else:
        bi = torch.arange(b, device=coors.device)[:, None, None]
        rel = coors[:, :, None, :] - coors[bi, idx]
    g_scale = None
    if layer.norm_coors:
        eps, scale = layer.coors_norm.eps, layer.coors_norm.scale
        rn = rel.norm(dim=-1, keepdim=True)
        den = rn.clamp(min=eps)
        relp = rel / den * scale
    else:
        relp = rel
    wm = w if pair_mask is None else w.masked_fill(~pair_mask, 0.0)
    c = layer.coor_weights_clamp_value
    wc = wm if c is None else wm.clamp(min=-c, max=c)
    g = g_coors_out[:, :, None, :]                                              # broadcast over k
    g_wc = (g * relp).sum(dim=-1)
    g_relp = wc[..., None] * g
    g_wm = g_wc if c is None else torch.where((wm >= -c) & (wm <= c), g_wc, torch.zeros_like(g_wc))
    g_w = g_wm if pair_mask is None else g_wm.masked_fill(~pair_mask, 0.0)
    g_a3 = g_w[..., None] * w4
    g_hid = g_a3 * (sg_h * (1 + hid * (1 - sg_h)))
    g_m = g_hid @ w3
    gm_pool = g_msum[:, :, None, :].expand(b, n, k, m)
    g_m = g_m + (gm_pool if pair_mask is None else gm_pool.masked_fill(~pair_mask[..., None], 0.0))
    g_gate = None
    if gate:
        g_s = (g_m * m0).sum(dim=-1, keepdim=True) * gt * (1 - gt)
        g_m = g_m * gt + g_s * gw
        g_gate = g_s.reshape(-1)
    g_u = g_m * (sg_u * (1 + u * (1 - sg_u)))
    if layer.norm_coors:
        dot = (g_relp * rel).sum(dim=-1, keepdim=True)
        g_scale = (dot / den).reshape(-1)
        g_rel = g_relp * (scale / den) - torch.where(rn >= eps, dot * scale / (den * den) * (rel / rn.clamp(min=1e-30)), torch.zeros_like(rel))
    else:
        g_rel = g_relp
    e = b * n * k
    return dict(g_u=g_u, g_rel=g_rel, g_hid=g_hid.reshape(e, -1), a3=a3.reshape(e, -1), g_w=g_w.reshape(e), g_scale=g_scale,
                m=mm.reshape(e, m), m0=m0.reshape(e, m), g_gate=g_gate)


def layer_given_neighbors(layer, feats, coors, edges, mask, idx, rank, valid_radius, factorised=True, drop=None, graph_offset=0):
    """EGNN.forward (egnn_pytorch.py:262-341) for given neighbours.
    idx (B,N,K) int64 / rank (B,N,K): the selection of :258 (None, None = dense all-pairs, K = N).
    Differentiable in feats, coors, edges and the parameters of `layer`.
    factorised: evaluate the first Linear of edge_mlp as (W_i h_i + b) + W_j h_j + W_s s_ij with the dim-wide products done
    once per node -- the same factorisation the HIP forward uses, 16x fewer flops than Linear(cat(...)) at
    the north-star shape, identical mathematics; False = the reference's literal cat + Linear."""
    b, n, dim = feats.shape
    dense = idx is None
    rel, scal = edge_scalars(layer, coors, edges, idx)
    lin = layer.edge_mlp[0]
    if factorised:
        w_i, w_j, w_s = lin.weight[:, :dim], lin.weight[:, dim:2 * dim], lin.weight[:, 2 * dim:]
        p_i = feats @ w_i.t() + lin.bias                                          # (B,N,H)
        p_j = feats @ w_j.t()
        bi = None if dense else torch.arange(b, device=feats.device)[:, None, None]
        z = p_i[:, :, None, :] + (p_j[:, None, :, :] if dense else p_j[bi, idx]) + scal @ w_s.t()
    else:
        bi = None if dense else torch.arange(b, device=feats.device)[:, None, None]
        feats_j = feats[:, None, :, :].expand(b, n, n, dim) if dense else feats[bi, idx]
        feats_i = feats[:, :, None, :].expand_as(feats_j)
        z = lin(torch.cat((feats_i, feats_j, scal), dim=-1))                      # (:287, first Linear)
    u = z
    if drop is not None:                                                          # (p, seed): the kernels' hash mask instead of nn.Dropout
        from . import _dropout
        u = _dropout.apply(u, drop[1], _dropout.SITE_EDGE, _edge_rows(b, u.shape[1], u.shape[2], u.device, graph_offset), drop[0])
        for mod in list(layer.edge_mlp)[2:4]:                                     # SiLU, Linear
            u = mod(u)
    else:
        for mod in list(layer.edge_mlp)[1:4]:                                     # dropout | Identity, SiLU, Linear
            u = mod(u)
    return layer_tail(layer, feats, coors, u, rel, mask, idx, rank, valid_radius, drop, graph_offset)


def _exact_active():
    from . import layer as _layer                         # (layer.py imports this module)
    return _layer.exact_active()


def _chunk_graphs(layer, n, k, batch):
    din = 2 * layer.dim + 2 * layer.fourier_features + 1 + layer.edge_dim
    per_graph = n * k * (2 * din) * 4.0 * _BYTES_PER_EDGE_FACTOR          # E x H pre-activation, activation, their gradients
    return max(1, min(batch, int(_CHUNK_BUDGET_BYTES // max(per_graph, 1.0))))


def _dropout_native_ok(layer):
    """Training-mode dropout on the native backward: the kernels that re-evaluate the forward's hash masks (egnn_edge_bwd_pass_f32 with
    drop_thr, the two tail kernels, egnn_silu_bwd_drop_f32) cover every shape of the fused forward kernels since round 5 -- heads up to 64
    channels, coordinate dimensions up to 8 (the generic tail kernel re-evaluates coors_mlp's mask), up to 16 per-edge scalars; the
    tuning switches that take pieces of the native backward away send the masked layer to the recompute path."""
    # (round 6: layers without coors_mlp / node_mlp -- update_coors=False / update_feats=False -- take the generic tail kernel, which
    # skips the absent module and its mask site; odd `dim`: egnn_silu_bwd_drop_f32 steps its (row, column) pair per element)
    return (_TAIL_KERNEL and _TAIL_REDUCE and _GRAD_GEMM and _FUSED_SPLIT == "dest" and _TAIL_GENERIC
            and 2 * layer.fourier_features + 1 + layer.edge_dim <= 16 and layer.m_dim <= 64
            and os.environ.get("EGNN_TAIL_SCALAR", "0") != "1" and os.environ.get("EGNN_BWD_DROP_NATIVE", "1") != "0")


class EGNNFunction(torch.autograd.Function):
    """forward: HIP kernels; backward: `_backward_native` / `_backward_exact` / `_backward_recompute` (module docstring)."""

    @staticmethod
    def forward(ctx, layer, order_hint, mask, adj_mat, feats, coors, edges, *params):
        # the native backward computes in fp32 like the forward: float64 / bfloat16 / float16 modules and inputs pass through
        # the same boundary conversion (gradients are returned in the callers' dtypes)
        # training-mode dropout: the forward kernels draw their masks from a hash of (seed, site, row, unit); the backward re-evaluates
        # the same masks -- inside its kernels (`_dropout_native_ok`) or on the recompute path (egnn_pytorch_amd/_dropout.py)
        drop = None
        if layer.dropout_active():
            from . import _dropout
            drop = (layer.dropout_p, _dropout.draw_seed())
        # (every shape the fused forward kernels cover -- m_dim <= 64, coordinate dimension 1 .. 8, up to 16 scalars -- is native: the E x H
        # work and the per-edge chain behind u, which has its closed-form kernels for all of them)
        native = (_NATIVE and layer.m_dim <= 64 and coors.shape[-1] <= 8 and (drop is None or _dropout_native_ok(layer))
                  and 2 * layer.fourier_features + 1 + layer.edge_dim <= 16
                  and not layer.float64_kernels()            # (a float64 module: float64 forward kernels, `_backward_exact` in float64)
                  and not _exact_active())                   # (the wide-range re-run: plain-fp32 forward kernels, `_backward_exact`)
        # the layers that run on the plain kernels (csrc/edge_exact.hip) -- float64 modules, the wide-range re-run, shapes beyond the fused
        # kernels' limits -- have their own native backward (round 5: `_backward_exact`, csrc/edge_exact_bwd.hip)
        s_in = 2 * layer.fourier_features + 1 + layer.edge_dim
        f64 = layer.float64_kernels()
        exact_path = f64 or _exact_active() or s_in > 16 or coors.shape[-1] > 8 or layer.m_dim > 64
        exact_native = bool(_NATIVE_EXACT and exact_path and s_in <= (40 if f64 else 80) and feats.is_cuda)
        with torch.no_grad():
            node_out, coors_out, order, idx, rank, valid_radius, u_pre, proj = layer._forward_hip_checked(
                feats, coors, edges, mask, adj_mat, order_hint, want_u=native or exact_native, drop_seed=None if drop is None else drop[1])
        ctx.drop = drop
        ctx.set_materialize_grads(False)
        ctx.exact_native = exact_native and u_pre is not None
        ctx.exact_dtype = torch.float64 if f64 else torch.float32
        use_nearest = layer.num_nearest_neighbors > 0 or layer.only_sparse_neighbors
        if use_nearest and idx is None:
            # neighbour path with K == 0 (only_sparse_neighbors and an empty adjacency): NO messages -- not the dense graph that
            # `idx is None` stands for everywhere else
            b, n = feats.shape[:2]
            idx = torch.empty(b, n, 0, dtype=torch.int32, device=feats.device)
            rank = torch.empty(b, n, 0, dtype=torch.float32, device=feats.device)
        ctx.layer = layer
        ctx.has_u = u_pre is not None                    # (E, 16 ceil(m_dim / 16)) fp32: E x m, not E x H
        ctx.valid_radius = valid_radius
        ctx.has_edges = edges is not None
        none = feats.new_empty(0)
        ctx.save_for_backward(feats, coors, edges if edges is not None else none, mask if mask is not None else none,
                              idx if idx is not None else none, rank if rank is not None else none,
                              u_pre if u_pre is not None else none,
                              proj[0] if (proj is not None and (_KEEP_PROJ or exact_native)) else none)
        # the projection table as the forward's edge pass read it: P_i as (fp16 hi, fp16 lo) words when pi_split -- the backward turns
        # them into fp32 in place, once (a second backward over a retained graph finds them decoded)
        ctx.proj_words = bool(proj is not None and proj[1])
        ctx.flags = (mask is not None, idx is not None)
        # the backward reads the layer's parameters (and their packed images) as they are THEN: an in-place update between
        # forward and backward must fail like it does for any saved tensor
        ctx.param_versions = tuple(p._version for p in params)
        ctx.order = order
        # outputs must not alias the inputs of a custom Function
        if node_out is feats:
            node_out = feats.clone()
        if coors_out is coors:
            coors_out = coors.clone()
        return node_out, coors_out

    @staticmethod
    def backward(ctx, g_node, g_coors):
        # (set_materialize_grads(False): an output the loss does not depend on arrives as None, not as zeros)
        ctx.dead_outputs = (g_node is None, g_coors is None)
        if g_node is None and g_coors is None:
            return (None,) * (7 + len(ctx.param_versions))
        for p, v in zip(ctx.layer.parameters(), ctx.param_versions):
            if p._version != v:
                raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace "
                                   "operation: a parameter of egnn_pytorch_amd.EGNN changed between forward and backward "
                                   f"(version {p._version}, expected {v})")
        if ctx.has_u:
            from . import _ops
            if getattr(ctx, "exact_native", False):
                return _backward_exact(ctx, g_node, g_coors)
            with _ops.backward_status():            # range bits of these kernels go to the backward's status word
                return _backward_native(ctx, g_node, g_coors)
        return _backward_recompute(ctx, g_node, g_coors)


def _pooled_messages(layer, u, m0, i64, r0, valid_radius):
    """(m_i, pair mask (B,N,K) bool or None, count or None) from u (B,N,K,m): egnn_pytorch.py:287-300, 319-333 on E x m tensors."""
    b = u.shape[0]
    k = u.shape[2]
    pm = None
    if m0 is not None:
        if i64 is None:
            pm = m0[:, :, None] & m0[:, None, :]
        else:
            bi = torch.arange(b, device=u.device)[:, None, None]
            pm = m0[:, :, None] & m0[bi, i64] & (r0 <= valid_radius)
    mm = torch.nn.functional.silu(u)
    if layer.edge_gate is not None:
        gl = layer.edge_gate[0]
        mm = mm * torch.sigmoid(mm @ gl.weight.detach().to(u.dtype)[0] + gl.bias.detach().to(u.dtype))[..., None]
    m_sum = (mm if pm is None else mm.masked_fill(~pm[..., None], 0.0)).sum(dim=2)
    cnt = None
    if layer.m_pool_method == "mean":
        if pm is not None:
            cnt = pm.sum(dim=-1, keepdim=True).to(m_sum.dtype)
            m_i = (m_sum / cnt.clamp(min=1e-8)).masked_fill(cnt == 0, 0.0)
        else:
            m_i = m_sum / k
    else:
        m_i = m_sum
    return m_i, pm, cnt


def _node_mlp_backward_exact(layer, f2d, m_i, g_out, grads, drop=None, row0=0):
    """node_norm + node_mlp + residual (egnn_pytorch.py:196-201, 335-337) differentiated on the exact GEMMs (egnn_linear_f32 / _f64) in
    f2d's dtype: returns (d loss / d feats of this part (rows, dim), d loss / d m_i (rows, m)) and accumulates the parameter gradients.
    node_norm itself (a LayerNorm on node-level rows) goes through autograd.  drop = (p, seed), row0: training-mode dropout between the
    first Linear and its SiLU -- the forward's hash mask of node rows row0 .. (the torch twin of the kernels' hash, node-level)."""
    from . import _ops
    dt = f2d.dtype
    dim = f2d.shape[1]
    n0, n3 = layer.node_mlp[0], layer.node_mlp[3]
    with torch.enable_grad():
        fl = f2d.detach().requires_grad_(True)
        ln = layer.node_norm(fl)
    w5, w6 = n0.weight.detach().to(dt).contiguous(), n3.weight.detach().to(dt).contiguous()       # (2 dim, dim + m), (dim, 2 dim)
    node_in = torch.cat((ln.detach(), m_i.to(dt)), dim=-1).contiguous()
    kin, hid = node_in.shape[1], w5.shape[0]
    rows = f2d.shape[0]
    z1 = _ops.linear_f32(node_in, w5, hid, kin, bias=n0.bias.detach().to(dt).contiguous(), name="bwd_exact_node_mlp")
    dk = None
    if drop is not None:                                                       # z_d = z keep / (1 - p); d z_d / d z = keep / (1 - p)
        from . import _dropout
        dk = _dropout.apply(torch.ones_like(z1), drop[1], _dropout.SITE_NODE, torch.arange(row0, row0 + rows, device=z1.device), drop[0])
        z1 = z1 * dk
    sg = torch.sigmoid(z1)
    a1 = z1 * sg
    g_out = g_out.contiguous()
    g_a1 = _ops.linear_f32(g_out, w6.t().contiguous(), hid, dim, name="bwd_exact_node_mlp")                    # g_out W6
    g_z1 = g_a1 * (sg * (1 + z1 * (1 - sg)))
    if dk is not None:
        g_z1 = g_z1 * dk
    got = g_out.t().contiguous()
    grads[id(n3.weight)] += _ops.linear_f32(got, a1.t().contiguous(), hid, rows, name="bwd_exact_node_mlp")  # g_out^T a1
    grads[id(n3.bias)] += g_out.sum(dim=0)
    gzt = g_z1.t().contiguous()
    grads[id(n0.weight)] += _ops.linear_f32(gzt, node_in.t().contiguous(), kin, rows, name="bwd_exact_node_mlp")
    grads[id(n0.bias)] += g_z1.sum(dim=0)
    g_in = _ops.linear_f32(g_z1, w5.t().contiguous(), kin, hid, name="bwd_exact_node_mlp")                      # g_z1 W5
    g_f = g_out.clone()                                                        # the residual
    if ln.requires_grad:
        norm_params = [p for p in layer.node_norm.parameters() if p.requires_grad]
        tg = torch.autograd.grad([ln], [fl] + norm_params, [g_in[:, :dim].contiguous()], allow_unused=True)
        if tg[0] is not None:
            g_f += tg[0]
        for p, g in zip(norm_params, tg[1:]):
            if g is not None:
                grads[id(p)] += g
    return g_f, g_in[:, dim:].contiguous()


def _tail_closed_form(layer, u2d, c0, i32, pm, g_coors_chunk, g_msum, grads, dl, b, n, k, drop=None, eid0=0):
    """The per-edge chain behind u on egnn_edge_tail_exact_bwd_* (csrc/edge_exact_bwd.hip; `tail_edge_backward` is the specification)
    in u2d's dtype, its parameter gradients contracted on the exact GEMMs (the kernel leaves their operands transposed: the edges are the
    K dimension).  Returns gU (E, m) and d loss / d coors of this part (B, N, C): x_i - x_j reaches the coordinates at the source (sum
    over a node's K edges) and, negated, at the neighbour (fixed-order sum over the CSR lists)."""
    from . import _ops
    dt = u2d.dtype
    e, m = u2d.shape
    cm = layer.coors_mlp
    gate = None if layer.edge_gate is None else (layer.edge_gate[0].weight, layer.edge_gate[0].bias)
    norm = layer.norm_coors and cm is not None
    pm8 = None if pm is None else pm.contiguous().view(torch.uint8)
    out = _ops.edge_tail_exact(u2d, c0, i32, pm8, g_coors_chunk.reshape(b * n, -1), g_msum,
                               None if cm is None else cm[0].weight, None if cm is None else cm[0].bias,
                               None if cm is None else cm[3].weight, None if cm is None else cm[3].bias,
                               layer.coors_norm.scale if norm else None, layer.coors_norm.eps if norm else 0.0,
                               layer.coor_weights_clamp_value, gate, b, n, k, drop, eid0)
    one = lambda v: v.view(1, e)                                                # noqa: E731
    if cm is not None:
        grads[id(cm[0].weight)] += _ops.linear_f32(out["ghid_t"], out["mm_t"], m, e, name="bwd_exact_tail")           # g_hid^T m
        grads[id(cm[0].bias)] += out["ghid_t"].sum(dim=1)
        grads[id(cm[3].weight)] += _ops.linear_f32(one(out["g_w"]), out["a3_t"], 4 * m, e, name="bwd_exact_tail")    # g_w^T a3
        grads[id(cm[3].bias)] += out["g_w"].sum().reshape(1)
        if norm:
            grads[id(layer.coors_norm.scale)] += out["g_scale"].sum().reshape(1)
    if gate is not None:
        grads[id(gate[0])] += _ops.linear_f32(one(out["g_gate"]), out["m0_t"], m, e, name="bwd_exact_tail")           # g_gate^T m0
        grads[id(gate[1])] += out["g_gate"].sum().reshape(1)
    src, _, dst, _ = _ops.edge_exact_node_sums(out["g_rel_t"], b * n, k, dl.order, dl.seg)
    return out["gU"], (src - dst).view(b, n, -1)


def _backward_exact(ctx, g_node, g_coors):
    """The backward of the layers that run on the plain kernels -- float64 modules (the reference's own training recipe,
    denoise_sparse.py:11, 23-32), calls answered by the wide-range path, shapes beyond the fused kernels' limits -- in the arithmetic of
    their forward (float64 / plain fp32), per chunk of graphs:
      1. behind u (saved by the forward kernel): the small per-edge tail and the node-level modules through autograd on E x m / node-level
         tensors (`layer_tail`)  ->  gU = d loss / d u, their parameters' gradients, the tail's share of d/d feats, d/d coors;
      2. the E x H work on egnn_edge_exact_bwd_* (csrc/edge_exact_bwd.hip): z, a recomputed, dz = (W2^T gU) SiLU'(z); a^T, dz^T (H, E) and
         d/d scalars; the per-node sums of dz over outgoing / incoming edges on egnn_edge_exact_node_sums_* (fixed order);
      3. every contraction a C = X W^T product of egnn_linear_f32 / _f64 (exact v_mfma_f32 / v_mfma_f64): d/d W2 = gU^T a and
         d/d W_s = dz^T s with the edges as the contraction, d/d feats = dP_i W_i + dP_j W_j, d/d W_i, W_j = dP^T feats;
      4. d/d scalars -> coordinates / edge features through the scalars' own small graph (E x S).
    Nothing of size E x H is touched by ATen; what stays there is E x m / E x S / node-level element-wise work, as in `_backward_native`."""
    from . import _abi, _ops
    dtype = ctx.exact_dtype
    esz = 8 if dtype == torch.float64 else 4
    orig_params = list(ctx.layer.parameters())
    layer = ctx.layer if dtype == torch.float64 else _f32_shadow(ctx.layer)
    feats, coors, edges, mask, idx32, rank = _unpack(ctx)
    in_dtypes = (feats.dtype, coors.dtype, None if edges is None else edges.dtype)
    feats, coors = feats.to(dtype), coors.to(dtype)
    edges = None if edges is None else edges.to(dtype)
    rank = None if rank is None else rank.to(dtype)
    params = list(layer.parameters())
    need = ctx.needs_input_grad
    b, n, dim = feats.shape
    dev = feats.device
    k = idx32.shape[-1] if idx32 is not None else n
    m, cdim = layer.m_dim, coors.shape[-1]
    s_in = 2 * layer.fourier_features + 1 + layer.edge_dim
    lin0, lin3 = layer.edge_mlp[0], layer.edge_mlp[3]
    h = lin0.weight.shape[0]
    head = {id(lin0.weight), id(lin0.bias), id(lin3.weight), id(lin3.bias)}
    tail_params = [p for p in params if id(p) not in head]
    grads = {id(p): torch.zeros_like(p, dtype=dtype) for p in params}
    g_feats, g_coors_in = torch.zeros_like(feats), torch.zeros_like(coors)
    want_ge = edges is not None and need[6]
    g_edges = torch.zeros_like(edges) if want_ge else None
    g_node = torch.zeros_like(feats) if g_node is None else g_node.to(dtype)
    g_coors = torch.zeros_like(coors) if g_coors is None else g_coors.to(dtype)
    u_all = ctx.saved_tensors[6].view(b, n, k, m)
    drop = getattr(ctx, "drop", None)                                     # (p, seed) of a training-mode forward: the same hash masks re-evaluated
    proj = ctx.saved_tensors[7]                                           # (B N, 2 hq): [P_i incl. bias | P_j], what the forward's edge pass read
    hq = proj.shape[1] // 2
    w1 = lin0.weight.detach().to(dtype).contiguous()                      # (H, Din): [W_i | W_j | scalar columns]
    w2 = lin3.weight.detach().to(dtype).contiguous()                      # (m, H)
    w_it, w_jt = w1[:, :dim].t().contiguous(), w1[:, dim:2 * dim].t().contiguous()       # (dim, H): B operands of d/d feats
    step = max(1, min(b, int(_EXACT_BWD_BYTES // max(1, 2 * n * k * h * esz))))
    for lo in range(0, b, step):
        hi_ = min(b, lo + step)
        bc = hi_ - lo
        ec, bn = bc * n * k, bc * n
        f0, c0 = feats[lo:hi_].contiguous(), coors[lo:hi_].contiguous()
        e0 = None if edges is None else edges[lo:hi_].contiguous()
        m0 = None if mask is None else mask[lo:hi_]
        i32 = None if idx32 is None else idx32[lo:hi_].contiguous()
        i64 = None if i32 is None else i32.long()
        r0 = None if rank is None else rank[lo:hi_]
        dl = _ops.dest_lists(i32, bc, n, k, dev)
        with torch.enable_grad():
            c = c0.detach().requires_grad_(True)
            e = None if e0 is None else e0.detach().requires_grad_(want_ge)
            rel, scal = edge_scalars(layer, c, e, i64)                            # (the scalars' own graph: step 4)
        if m <= 64 and _TAIL_GENERIC:
            # ---- 1. behind u, in closed form: the pooled messages (E x m element-wise), node_norm / node_mlp on the exact GEMMs, the
            # per-edge chain on egnn_edge_tail_exact_bwd_* with its parameter gradients contracted on the exact GEMMs
            with torch.no_grad():
                u4 = u_all[lo:hi_]
                m_i, pm, cnt = _pooled_messages(layer, u4, m0, i64, r0, ctx.valid_radius)
                g_msum = None
                if layer.node_mlp is not None:
                    g_f, g_mi = _node_mlp_backward_exact(layer, f0.view(bn, dim), m_i.reshape(bn, m), g_node[lo:hi_].reshape(bn, dim), grads,
                                                         drop, lo * n)
                    g_feats[lo:hi_] += g_f.view(bc, n, dim)
                    g_mi = g_mi.view(bc, n, m)
                    if layer.m_pool_method == "mean":
                        g_mi = (g_mi / cnt.clamp(min=1e-8)).masked_fill(cnt == 0, 0.0) if cnt is not None else g_mi / k
                    g_msum = g_mi.reshape(bn, m).contiguous()
                else:
                    g_feats[lo:hi_] += g_node[lo:hi_]                                # (update_feats=False: node_out is feats)
                g_coors_in[lo:hi_] += g_coors[lo:hi_]                                # the residual of the coordinate update
                g_u, g_c = _tail_closed_form(layer, u4.reshape(ec, m).contiguous(), c0, i32, pm, g_coors[lo:hi_], g_msum, grads, dl, bc, n, k,
                                             drop, lo * n * k)
                g_coors_in[lo:hi_] += g_c
        else:
            # ---- 1. the small tail, through autograd (E x m, node-level): heads wider than 64 channels
            with torch.enable_grad():
                f = f0.detach().requires_grad_(True)
                u = u_all[lo:hi_].detach().requires_grad_(True)
                out_n, out_c = layer_tail(layer, f, c, u, rel, m0, i64, r0, ctx.valid_radius, drop=drop, graph_offset=lo)
                outs, gouts = [], []
                for o, g in ((out_n, g_node[lo:hi_]), (out_c, g_coors[lo:hi_])):
                    if o.requires_grad:
                        outs.append(o)
                        gouts.append(g)
                live_tail = [p for p in tail_params if p.requires_grad]
                tg = torch.autograd.grad(outs, [u, f, c] + live_tail, gouts, allow_unused=True, retain_graph=True)
            g_u = (tg[0] if tg[0] is not None else torch.zeros_like(u)).reshape(ec, m).contiguous()
            if tg[1] is not None:
                g_feats[lo:hi_] += tg[1]
            if tg[2] is not None:
                g_coors_in[lo:hi_] += tg[2]
            for p, g in zip(live_tail, tg[3:]):
                if g is not None:
                    grads[id(p)] += g
        with torch.no_grad():
            # ---- 2. the E x H work
            a_t = _ops.empty(h, ec, dtype=dtype, device=dev)
            dz_t = _ops.empty(h, ec, dtype=dtype, device=dev)
            g_scal = _ops.empty(ec, s_in, dtype=dtype, device=dev)
            a = _abi.EdgeExactBwdArgs()
            a.B, a.N, a.K, a.m_dim, a.H = bc, n, k, m, h
            a.fourier, a.edge_dim, a.coor_dim, a.edges_by_k = layer.fourier_features, layer.edge_dim, cdim, 0
            pc = proj[lo * n:hi_ * n]
            a.Pi, a.Pj, a.ldp = pc.data_ptr(), pc.data_ptr() + esz * hq, 2 * hq
            a.Ws, a.ldws = w1.data_ptr() + esz * 2 * dim, w1.shape[1]
            a.W2, a.coors, a.edges, a.idx = w2.data_ptr(), c0.data_ptr(), _ops._ptr(e0), _ops._ptr(i32)
            a.gU, a.A_T, a.DZ_T, a.g_scal = g_u.data_ptr(), a_t.data_ptr(), dz_t.data_ptr(), g_scal.data_ptr()
            _ops.set_drop(a, drop, lo * n * k)
            _ops.edge_exact_bwd(a, dtype)
            gpi, gpi_t, gpj, gpj_t = _ops.edge_exact_node_sums(dz_t, bn, k, dl.order, dl.seg)
            # ---- 3. the contractions, on the exact GEMMs
            grads[id(lin3.weight)] += _ops.linear_f32(g_u.t().contiguous(), a_t, h, ec, name="bwd_exact_dw2")        # (m, H) = gU^T a
            grads[id(lin3.bias)] += g_u.sum(dim=0)
            scal_t = scal.detach().reshape(ec, s_in).t().contiguous()                                               # (S, E)
            gw1 = grads[id(lin0.weight)]
            gw1[:, 2 * dim:] += _ops.linear_f32(dz_t, scal_t, s_in, ec, name="bwd_exact_dws")                        # (H, S) = dz^T s
            del a_t, dz_t
            f2d = f0.view(bn, dim)
            t = _ops.linear_f32(gpi, w_it, dim, h, name="bwd_exact_dfeats")                                          # dP_i W_i
            t = _ops.linear_f32(gpj, w_jt, dim, h, residual=t, name="bwd_exact_dfeats")                              # + dP_j W_j
            g_feats[lo:hi_] += t.view(bc, n, dim)
            f_t = f2d.t().contiguous()                                                                              # (dim, B N)
            gw1[:, :dim] += _ops.linear_f32(gpi_t, f_t, dim, bn, name="bwd_exact_dw1")                               # dP_i^T feats
            gw1[:, dim:2 * dim] += _ops.linear_f32(gpj_t, f_t, dim, bn, name="bwd_exact_dw1")
            grads[id(lin0.bias)] += gpi.sum(dim=0)
            del gpi, gpi_t, gpj, gpj_t, f_t
        # ---- 4. d loss / d scalars -> coordinates (squared distance, fourier terms) and edge features
        sg = torch.autograd.grad([scal], [c] + ([e] if want_ge else []), [g_scal.view_as(scal)], allow_unused=True)
        if sg[0] is not None:
            g_coors_in[lo:hi_] += sg[0]
        if want_ge and sg[1] is not None:
            g_edges[lo:hi_] += sg[1]
    unused = _unused_params(layer, ctx)
    out_params = [grads[id(p)].to(op.dtype) if (need[7 + i] and id(p) not in unused) else None
                  for i, (p, op) in enumerate(zip(params, orig_params))]
    return (None, None, None, None, g_feats.to(in_dtypes[0]) if need[4] else None, g_coors_in.to(in_dtypes[1]) if need[5] else None,
            g_edges.to(in_dtypes[2]) if (want_ge and need[6]) else None) + tuple(out_params)


def _unused_params(layer, ctx=None):
    """ids of the parameters no output depends on -- node_norm without node_mlp (update_feats=False), CoorsNorm's scale without
    coors_mlp (update_coors=False; egnn_pytorch.py:302-306 applies it inside that branch): autograd leaves their .grad None in the
    reference, and so does this Function (an optimizer treats None and zeros differently: weight decay, state creation).  Likewise
    the parameters that reach the loss only through an output nobody used (ctx.dead_outputs: the last layer of a coordinate-denoising
    network -- denoise_sparse.py:70-72 -- never has its node_mlp / node_norm differentiated upstream)."""
    out = set()
    dead_node, dead_coors = getattr(ctx, "dead_outputs", (False, False)) if ctx is not None else (False, False)
    if layer.node_mlp is None or dead_node:
        out |= {id(p) for p in layer.node_norm.parameters()}
    if layer.coors_mlp is None or dead_coors:
        out |= {id(p) for p in layer.coors_norm.parameters()}
    if dead_node and layer.node_mlp is not None:
        out |= {id(p) for p in layer.node_mlp.parameters()}
    if dead_coors and layer.coors_mlp is not None:
        out |= {id(p) for p in layer.coors_mlp.parameters()}
    return out


def _unpack(ctx):
    feats, coors, edges, mask, idx, rank = ctx.saved_tensors[:6]
    has_mask, has_idx = ctx.flags
    return (feats, coors, edges if ctx.has_edges else None, mask if has_mask else None,
            idx if has_idx else None, rank if has_idx else None)


def _f32_shadow(layer):
    """The layer itself if its parameters are fp32, else an fp32 copy of it (cached per parameter version): the native backward
    differentiates in fp32 whatever dtype the module has, and hands the gradients back in the parameters' dtype."""
    from . import _weights
    if all(p.dtype == torch.float32 for p in layer.parameters()):
        return layer
    key = _weights.version_key(layer)
    cached = layer.__dict__.get("_shadow32")
    if cached is None or cached[0] != key:
        import copy
        packed, layer._packed = layer._packed, None          # (not the packed kernel weights, not an older shadow)
        layer.__dict__.pop("_shadow32", None)
        try:
            mod = copy.deepcopy(layer)
        finally:
            layer._packed = packed
        cached = (key, mod.float())
        layer.__dict__["_shadow32"] = cached
    return cached[1]


def _edge_tables(layer, w, f2d, pi_split):
    """fp32 P_i | P_j rows (incl. bias, in the forward's -log2(e) units) of the nodes in f2d: (rows, 2 Hp)."""
    from . import _ops
    hp = w["Hp"]
    feats_hl = _ops.split_f16(f2d)
    return _ops.linear_hl(feats_hl, w["Wcat_split"], 2 * hp, w["bcat"], name="bwd_node_proj", split_cols=hp if pi_split else 0)


def entry_list(eids, keys, n_keys):
    """The entry list of one egnn_edge_bwd_pass_f32 call (include/egnn_hip.h): edge ids `eids` (E,) ordered so that their keys
    `keys` (E,) -- the node each entry is grouped by -- are non-decreasing (None: E / n_keys consecutive entries per node).  Every
    node's entries are padded with -1 to whole
    16-entry tiles, the list to a multiple of 128.  Returns (ent int32 (L,), seg int64 (n_keys + 1,)): seg = the range of tiles
    (= partial rows) of each key."""
    dev = eids.device
    e = eids.numel()
    if keys is None:                                          # every key has the same number of consecutive entries (by source: K)
        per = e // n_keys
        t = (per + 15) // 16
        l = (n_keys * t * 16 + 127) // 128 * 128
        ent = torch.full((max(l, 128),), -1, dtype=torch.int32, device=dev)
        ent[:n_keys * t * 16].view(n_keys, t * 16)[:, :per] = eids.view(n_keys, per).to(torch.int32)
        return ent, torch.arange(n_keys + 1, device=dev) * t
    deg = torch.bincount(keys, minlength=n_keys)
    tiles = (deg + 15) // 16
    seg = torch.zeros(n_keys + 1, dtype=torch.int64, device=dev)
    torch.cumsum(tiles, 0, out=seg[1:])
    first = torch.zeros(n_keys + 1, dtype=torch.int64, device=dev)              # first entry of each key in the sorted list
    torch.cumsum(deg, 0, out=first[1:])
    pos = seg[:-1][keys] * 16 + (torch.arange(e, device=dev) - first[:-1][keys])
    l = (int(seg[-1]) * 16 + 127) // 128 * 128
    ent = torch.full((max(l, 128),), -1, dtype=torch.int32, device=dev)
    ent[pos] = eids.to(torch.int32)
    return ent, seg


def _edge_contract_fused(layer, w, f2d, c0, e0, sc2, i32, gu16, gu_scale, w_s, bc, n, k, pi_split, dest_lists=None, proj=None, drop=None, eid0=0):
    """egnn_edge_bwd_pass_f32 (csrc/edge_bwd.hip) twice -- entries grouped by source node, then by neighbour: z, SiLU(z) and dz
    are recomputed and contracted in registers, nothing of size E x H reaches memory.
    Returns d/d P_i (rows, Hp), d/d P_j (rows, Hp), d/d W_s (Hp, S), d/d scalars (E, S), d/d W_2 (16, Hp)."""
    from . import _ops
    dev = f2d.device
    ec = bc * n * k
    if proj is None:
        proj = _edge_tables(layer, w, f2d, False)
    ent, seg = entry_list(torch.arange(ec, device=dev), None, bc * n)
    # the contractions over all edges ride along: d/d W_s and d/d scalars with the first pass, d/d W_2 with the second (each keeps
    # its accumulators in registers; one pass carrying both drops from 3 to 2 workgroups per CU)
    s_first = _FUSED_SPLIT != "src"                          # ("src": d/d W_2 with the by-source pass, d/d W_s and d/d scalars with the other)
    want_w2_first = _FUSED_SPLIT == "src" or (_FUSED_SPLIT == "both" and w["S"] == 1)
    # by source every node has ceil(K / 16) tiles: two (16 < K <= 32) are summed inside the kernel, one IS the node's row -- no gather-sum
    pairs = 16 < k <= 32 and s_first and not want_w2_first
    o = _ops.edge_bwd_pass(w, proj, i32, gu16, gu_scale, sc2, ent, bc, n, k, by_dest=False, ws_nat=w_s if s_first else None,
                           want_w2=want_w2_first, row_pairs=pairs, drop=drop, eid0=eid0, want_amax=(pairs or k <= 16) and dev.type == "cuda")
    g_ws, g_scal, g_w2 = o.get("ws"), o.get("scal"), o.get("w2")
    if pairs or k <= 16:
        gz_i = o["rows"][:bc * n]
        if o.get("amax_bits") is not None:
            gz_i.amax_bits = _ops.HostRead(o["amax_bits"])   # (max |d/d P_i| as a by-product of the pass; read behind THIS pass)
    else:
        ident = torch.arange(o["rows"].shape[0], device=dev)
        gz_i = _ops.rows_gather_sum(o["rows"], ident, seg, bc * n)
    del o
    if dest_lists is None:
        dest_lists = _ops.dest_lists(i32, bc, n, k, dev)                 # (dense: destination = k)
    ent, seg = dest_lists.ent, dest_lists.tile_seg
    o = _ops.edge_bwd_pass(w, proj, i32, gu16, gu_scale, sc2, ent, bc, n, k, by_dest=True, ws_nat=None if s_first else w_s, want_w2=g_w2 is None,
                           drop=drop, eid0=eid0)
    if g_w2 is None:
        g_w2 = o["w2"]
    if not s_first:
        g_ws, g_scal = o["ws"], o["scal"]
    ident = torch.arange(o["rows"].shape[0], device=dev)
    if dev.type == "cuda":
        gz_j, bits = _ops.rows_gather_sum(o["rows"], ident, seg, bc * n, want_amax=True)
        gz_j.amax_bits = _ops.HostRead(bits)                  # (max |d/d P_j| as a by-product: the scale of its gradient GEMM operands)
    else:
        gz_j = _ops.rows_gather_sum(o["rows"], ident, seg, bc * n)
    return gz_i, gz_j, g_ws, g_scal, g_w2


def _node_mlp_backward(layer, w, f, m_i, g_out, grads_by_id, drop=None, row0=0, overlap=None):
    """Backward of the node update out = node_mlp(cat(node_norm(f), m_i)) + f (egnn_pytorch.py:335-337) on the split-f16 GEMMs:
    the hidden pre-activation recomputed by the forward's GEMM, then per Linear one NN product (d/d input) and one split-K TN product
    (d/d weight) that share one (plain, transposed) split of the incoming gradient; SiLU and its derivative in one pass
    (egnn_silu_bwd_f32).  node_norm (LayerNorm or Identity, element-wise per node) stays with autograd.  f: (bc, n, dim) leaf that
    requires grad; m_i (bc, n, m); g_out (bc, n, dim).  Adds the parameter gradients into grads_by_id; returns (d/d f, d/d m_i).
    drop = (p, seed), row0: training-mode dropout behind the first Linear -- the forward's hash mask of node rows row0 .. is
    re-evaluated inside the SiLU-backward pass.  overlap: a callable that queues launches which do not depend on this function's results;
    called while the SiLU pass -- whose max |.| words the next operand scales are chosen from -- is still running (_ops.HostRead)."""
    from . import _ops
    bc, n, dim = f.shape
    m = m_i.shape[-1]
    rows = bc * n
    lin5, lin6 = layer.node_mlp[0], layer.node_mlp[3]
    with torch.enable_grad():
        ln = layer.node_norm(f)
    with torch.no_grad():
        in32 = torch.cat((ln.detach(), m_i), dim=-1).view(rows, dim + m)
        g2d = g_out.reshape(rows, dim).contiguous()
        # (max |.| of the two operands that exist already: launched first, read behind the GEMM that is queued next)
        hr_g, hr_in = _ops.absmax_async(g2d), _ops.absmax_async(in32)
        z1 = _ops.linear_hl(_ops.split_f16(in32), w["W5_split"], 2 * dim, w["b5"], name="bwd_node_mlp")        # (rows, 2 dim) pre-activation
        go = _ops.GradOperand(g2d, amax=hr_g.floats()[0], colsum=lin6.bias.requires_grad)   # (column sums = d/d bias: by-products of the split)
        g_a1 = _ops.grad_nn(go, w["W6T_split"], 2 * dim, name="bwd_node_mlp")
        if z1.numel() % 4 == 0:
            a1, g_z1, bits = _ops.silu_bwd_(z1, g_a1, drop, row0)
            bits = _ops.HostRead(bits)
            if overlap is not None:
                overlap()
            amax_a1, amax_gz = bits.floats()                            # (by-products of the pass: no absmax launches for these two)
        else:
            sg = torch.sigmoid(z1)
            a1, g_z1 = z1 * sg, g_a1 * (sg * (1 + z1 * (1 - sg)))
            amax_a1 = amax_gz = None
        if lin6.weight.requires_grad:
            grads_by_id[id(lin6.weight)] += _ops.grad_tn(go, a1, name="bwd_node_mlp_w", x_operand=_ops.grad_tn_operand(a1, amax_a1))
        if lin6.bias.requires_grad:
            grads_by_id[id(lin6.bias)] += go.colsum
        del go, a1
        gz = _ops.GradOperand(g_z1, amax=amax_gz, colsum=lin5.bias.requires_grad)
        g_in = _ops.grad_nn(gz, w["W5T_split"], dim + m, name="bwd_node_mlp")
        if lin5.weight.requires_grad:
            grads_by_id[id(lin5.weight)] += _ops.grad_tn(gz, in32, name="bwd_node_mlp_w", x_operand=_ops.grad_tn_operand(in32, hr_in.floats()[0]))
        if lin5.bias.requires_grad:
            grads_by_id[id(lin5.bias)] += gz.colsum
        del gz, g_z1
        g_ln = g_in[:, :dim].reshape(bc, n, dim)
        g_mi = g_in[:, dim:].reshape(bc, n, m)
    if ln is f:                                                   # node_norm = Identity
        return g_ln + g_out, g_mi
    ln_params = [p for p in layer.node_norm.parameters() if p.requires_grad]
    tg = torch.autograd.grad([ln], [f] + ln_params, [g_ln], allow_unused=True)
    for p, g in zip(ln_params, tg[1:]):
        if g is not None:
            grads_by_id[id(p)] += g
    return (tg[0] if tg[0] is not None else torch.zeros_like(g_out)) + g_out, g_mi


def _backward_native(ctx, g_node, g_coors):
    """The backward on the HIP kernels (module docstring), per chunk of graphs:
       1. behind u = ctx.u_pre (edge_mlp's second Linear, written by the forward kernel): the pooled messages (egnn_edge_pool_f32), node_mlp
          on the split-f16 GEMMs (`_node_mlp_backward`; node_norm through autograd); the per-edge chain in closed form on
          egnn_edge_tail_bwd_f32, which also sums its parameter gradients (on the CPU, in the tests: through autograd, `layer_tail`)  ->  gU = d loss / d u, d loss / d (x_i - x_j), those modules' parameter gradients;
       2. the E x H work on egnn_edge_bwd_pass_f32 (`_edge_contract_fused`: by source and by destination, everything recomputed
          and contracted in registers)
          ->  d/d P_i, d/d P_j per node, d/d W_s, d/d scalars, d/d W_2;
       3. node-level products: d/d feats, d/d W_i, W_j, b_1 from the per-node sums and feats; d/d scalars -> coordinates (closed
          form when the distance is the only scalar, the scalars' own small graph otherwise) and edge features.
    Every sum over edges has a fixed order.  Chunks: the kernels' tables stay below 2 GB (signed 32-bit offsets)."""
    from . import _abi, _ops, _weights
    w = ctx.layer.packed_weights()
    orig_params = list(ctx.layer.parameters())
    layer = _f32_shadow(ctx.layer)                       # (the layer itself unless it is a float64 / half module)
    feats, coors, edges, mask, idx32, rank = _unpack(ctx)
    in_dtypes = (feats.dtype, coors.dtype, None if edges is None else edges.dtype)
    feats, coors = feats.float(), coors.float()
    edges = None if edges is None else edges.float()
    g_node = None if g_node is None else g_node.float()
    g_coors = None if g_coors is None else g_coors.float()
    params = list(layer.parameters())
    need = ctx.needs_input_grad                          # (layer, order_hint, mask, adj, feats, coors, edges, *params)
    b, n, dim = feats.shape
    k = idx32.shape[-1] if idx32 is not None else n
    m = layer.m_dim
    h, hp, s_in = w["H"], w["Hp"], w["S"]
    lin0, lin3 = layer.edge_mlp[0], layer.edge_mlp[3]
    head = {id(lin0.weight), id(lin0.bias), id(lin3.weight), id(lin3.bias)}
    tail_params = [p for p in params if id(p) not in head]
    # (one zero fill for all parameter gradients: views of a flat buffer)
    flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=feats.device)
    grads_by_id, off = {}, 0
    for p in params:
        grads_by_id[id(p)] = flat[off:off + p.numel()].view(p.shape)
        off += p.numel()
    g_feats = torch.zeros_like(feats)
    g_coors_in = torch.zeros_like(coors)
    want_ge = edges is not None and need[6]                 # (the (B,N,N,edge_dim) input: its gradient only if somebody asked for it)
    g_edges = torch.zeros_like(edges) if want_ge else None
    if g_node is None:
        g_node = torch.zeros_like(feats)
    if g_coors is None:
        g_coors = torch.zeros_like(coors)
    mp = 16 * _weights.m_blocks(m)                       # u rows: whole 16-channel blocks, pad channels 0
    u_all = ctx.saved_tensors[6].view(b, n, k, mp)
    drop = getattr(ctx, "drop", None)                    # (p, seed) of a training-mode forward (_dropout_native_ok), else None
    reduce = False
    # (egnn_edge_bwd_pass_f32: up to 16 per-edge scalars -- beyond five with the all-edge contractions split over the two passes)
    fused = (s_in <= 5 or (s_in <= 16 and _FUSED_SPLIT == "dest" and "WsTh" in w))
    if not fused:
        raise NotImplementedError("the native backward carries more than five per-edge scalars only with EGNN_BWD_SPLIT=dest (the default)")
    proj_all = None
    if len(ctx.saved_tensors) > 7 and ctx.saved_tensors[7].numel() and fused:
        proj_all = ctx.saved_tensors[7]                                   # (B N, 2 Hp): what the forward's edge pass read (P_i still as
                                                                          # (hi, lo) words when ctx.proj_words: decoded in `early` below)
    if True:
        # nothing of size E x H: graphs are only chunked to keep the P table below 4 GB (32-bit buffer offsets) and E below 2^31
        # (the kernel addresses both with signed 32-bit scalar offsets)
        step = max(1, min(b, int(((1 << 31) - 1) // (n * 2 * hp * 4)), int(((1 << 31) - 1) // (n * k)),
                          int(((1 << 31) - 1) // ((n * k // 16 + n + 16) * hp * 4))))         # (... and the partial rows)
        if _FUSED_MAX_GRAPHS > 0:
            step = min(step, _FUSED_MAX_GRAPHS)
    # the first Linear's blocks, zero padded to the kernel's hidden width Hp: everything below works on the contiguous
    # (E, Hp) buffers the kernel wrote (slicing [:, :H] first would copy 17 GB per use at the north-star shape)
    # (padded copies of parameters are kept with the packed weights: `w` is rebuilt whenever a parameter changes, so once per
    # optimizer step instead of a dozen small launches per backward)
    pads = w.get("bwd_pads")
    if pads is None or pads["device"] != feats.device:
        pads = w["bwd_pads"] = {"device": feats.device}
        w_s = torch.zeros(hp, lin0.weight.shape[1] - 2 * dim, dtype=torch.float32, device=feats.device)
        w_s[:h] = lin0.weight.detach()[:, 2 * dim:]
        pads["w_s"] = w_s
    w_s = pads["w_s"]
    w_i = w_j = None
    if not (feats.is_cuda and _GRAD_GEMM):               # (host tensors -- the CPU tests: plain matmuls instead of the split-f16 GEMM)
        w1p = torch.zeros(hp, lin0.weight.shape[1], dtype=torch.float32, device=feats.device)
        w1p[:h] = lin0.weight.detach()
        w_i, w_j = w1p[:, :dim].contiguous(), w1p[:, dim:2 * dim].contiguous()
    pi_split = k >= 6
    # the per-edge chain behind u in closed form on the device (egnn_edge_tail_bwd_f32) where it applies (m_dim <= 16, coors_mlp
    # hidden width <= 64); otherwise that part goes through autograd as well
    tail_kernel = (_TAIL_KERNEL and layer.coors_mlp is not None and layer.node_mlp is not None
                   and m <= 16 and layer.coors_mlp[0].weight.shape[0] <= 64 and coors.shape[-1] == 3)
    for lo in range(0, b, step):
        hi_ = min(b, lo + step)
        bc = hi_ - lo
        f0, c0 = feats[lo:hi_].contiguous(), coors[lo:hi_].contiguous()
        e0 = None if edges is None else edges[lo:hi_].contiguous()
        m0 = None if mask is None else mask[lo:hi_]
        i32 = None if idx32 is None else idx32[lo:hi_].contiguous()
        i64 = None if i32 is None else i32.long()
        r0 = None if rank is None else rank[lo:hi_]
        ec = bc * n * k
        dest_lists = None
        hr_f = _ops.absmax_async(f0.view(bc * n, dim)) if (f0.is_cuda and _GRAD_GEMM) else None      # (read in step 3)

        def early():
            """Launches that depend on the saved inputs only -- P_i decoded in place (once), the edges sorted by destination: queued where
            the node-level part waits for a scale, so that the device has work while the host is woken up (_ops.HostRead)."""
            nonlocal dest_lists
            if proj_all is not None and ctx.proj_words:
                _ops.unsplit_words_(proj_all, hp)
                ctx.proj_words = False
            if dest_lists is None and (i32 is not None or not tail_kernel):
                dest_lists = _ops.dest_lists(i32, bc, n, k, feats.device)                     # (shared by the tail and the E x H passes)
        if tail_kernel:
            # ---- 1. behind u: the node-level modules through autograd (node_norm, node_mlp, residual: from the pooled messages),
            # the per-edge chain (second SiLU, masks, coors_mlp, CoorsNorm, clamp, coordinate update, pooling) in closed form on
            # egnn_edge_tail_bwd_f32 -- `tail_edge_backward` is its specification
            with torch.no_grad():
                u16 = u_all[lo:hi_].contiguous()
                pm = None
                if m0 is not None:
                    if i64 is None:
                        pm = m0[:, :, None] & m0[:, None, :]
                    else:
                        bi = torch.arange(bc, device=feats.device)[:, None, None]
                        pm = m0[:, :, None] & m0[bi, i64] & (r0 <= ctx.valid_radius)
                reduce = f0.is_cuda and _TAIL_REDUCE       # (the kernels pool the messages and sum the parameter gradients' terms themselves)
                pm8 = None if pm is None else pm.contiguous().view(torch.uint8)
                gate, mm_pre, mm = None, None, None
                if layer.edge_gate is not None:                                              # soft_edges (:289-290)
                    if "gw16" not in pads:
                        gw16 = torch.zeros(16, dtype=torch.float32, device=feats.device)
                        gw16[:m] = layer.edge_gate[0].weight.detach()[0]
                        pads["gw16"] = gw16
                    gate = (pads["gw16"], layer.edge_gate[0].bias.detach().contiguous())
                if reduce:
                    m_sum = _ops.edge_pool(u16, gate, pm8, bc, n, k)
                else:
                    mm = torch.nn.functional.silu(u16)                                       # (bc, n, k, 16), columns >= m are 0
                    if gate is not None:
                        mm_pre = mm
                        mm = mm_pre * torch.sigmoid(mm_pre @ gate[0] + gate[1])[..., None]
                    m_sum = (mm if pm is None else mm.masked_fill(~pm[..., None], 0.0)).sum(dim=2)
                cnt = None
                if layer.m_pool_method == "mean":
                    if pm is not None:
                        cnt = pm.sum(dim=-1, keepdim=True).to(m_sum.dtype)
                        m_i = (m_sum / cnt.clamp(min=1e-8)).masked_fill(cnt == 0, 0.0)
                    else:
                        m_i = m_sum / k
                else:
                    m_i = m_sum
                del m_sum
            with torch.enable_grad():
                f = f0.detach().requires_grad_(True)
                c = c0.detach().requires_grad_(True)
                e = None if e0 is None else e0.detach().requires_grad_(want_ge)
                closed_dist = s_in == 1                       # the distance is the only per-edge scalar: its backward in closed form below
                if closed_dist:
                    if not reduce:                            # (reduce: x_i - x_j and the distance are by-products of the tail kernel)
                        with torch.no_grad():
                            rel, scal = edge_scalars(layer, c0, None, i64)
                else:
                    rel, scal = edge_scalars(layer, c, e, i64)                               # (only the scalars' graph is used below)
            if f0.is_cuda and _GRAD_GEMM:
                g_f, g_mi = _node_mlp_backward(layer, w, f, m_i[..., :m], g_node[lo:hi_], grads_by_id, drop, lo * n, overlap=early)
                g_feats[lo:hi_] += g_f
            else:
                with torch.enable_grad():
                    mi = m_i[..., :m].detach().requires_grad_(True)
                    out_n = _per_edge(layer.node_mlp, torch.cat((layer.node_norm(f), mi), dim=-1)) + f      # (split-K weight gradients)
                    # (only what requires grad may be differentiated: a frozen parameter in the list makes autograd.grad raise)
                    node_params = [p for p in list(layer.node_norm.parameters()) + list(layer.node_mlp.parameters()) if p.requires_grad]
                    tg = torch.autograd.grad([out_n], [f, mi] + node_params, [g_node[lo:hi_]], allow_unused=True)
                if tg[0] is not None:
                    g_feats[lo:hi_] += tg[0]
                for p, g in zip(node_params, tg[2:]):
                    if g is not None:
                        grads_by_id[id(p)] += g
                g_mi = tg[1]
            with torch.no_grad():
                g_msum = torch.zeros(bc, n, 16, dtype=torch.float32, device=feats.device)
                if g_mi is not None:
                    if layer.m_pool_method == "mean":
                        g_mi = (g_mi / cnt.clamp(min=1e-8)).masked_fill(cnt == 0, 0.0) if cnt is not None else g_mi / k
                    g_msum[..., :m] = g_mi
                lin_a, lin_b = layer.coors_mlp[0], layer.coors_mlp[3]
                hid3 = lin_a.weight.shape[0]
                if "w3p" not in pads:
                    w3p = torch.zeros(64, 16, dtype=torch.float32, device=feats.device)
                    w3p[:hid3, :m] = lin_a.weight.detach()
                    b3p = torch.zeros(64, dtype=torch.float32, device=feats.device)
                    b3p[:hid3] = lin_a.bias.detach()
                    w4p = torch.zeros(64, dtype=torch.float32, device=feats.device)
                    w4p[:hid3] = lin_b.weight.detach()[0]
                    pads["w3p"], pads["b3p"], pads["w4p"] = w3p, b3p, w4p
                w3p, b3p, w4p = pads["w3p"], pads["b3p"], pads["w4p"]
                norm = layer.norm_coors
                tail_args = (u16, c0, i32, pm8, g_coors[lo:hi_].contiguous(), g_msum, w3p, b3p, w4p, lin_b.bias.detach().contiguous(),
                             layer.coors_norm.scale.detach() if norm else None, layer.coors_norm.eps if norm else 0.0,
                             layer.coor_weights_clamp_value, bc, n, k)
                bias2 = gu_bits = None
                if reduce:
                    gu16, g_rel, sums, rel4, dist, gu_bits = _ops.edge_tail_bwd(*tail_args, gate=gate, reduce=True, want_rel=closed_dist,
                                                                               drop=drop, eid0=lo * n * k)
                    gu_bits = _ops.HostRead(gu_bits)                              # (read in step 2, behind the small launches below)
                    grads_by_id[id(lin_a.weight)] += sums[:1024].view(64, 16)[:hid3, :m]
                    grads_by_id[id(lin_a.bias)] += sums[1024:1024 + hid3]
                    grads_by_id[id(lin_b.weight)] += sums[1088:1088 + hid3][None, :]
                    grads_by_id[id(lin_b.bias)] += sums[1184:1185]
                    if norm:
                        grads_by_id[id(layer.coors_norm.scale)] += sums[1185:1186]
                    if gate is not None:
                        grads_by_id[id(layer.edge_gate[0].weight)] += sums[1168:1168 + m][None, :]
                        grads_by_id[id(layer.edge_gate[0].bias)] += sums[1186:1187]
                    bias2 = sums[1152:1152 + m]                                  # column sums of gU = d loss / d edge_mlp's last bias
                    if closed_dist:
                        rel, scal = rel4, dist.view(bc, n, k, 1)
                else:
                    tail_out = _ops.edge_tail_bwd(*tail_args, gate=gate)
                    gu16, g_rel, g_hid, a3, g_w, g_sc = tail_out[:6]
                    if gate is not None:
                        g_gate = tail_out[6]
                        grads_by_id[id(layer.edge_gate[0].weight)] += _tn(g_gate[:, None], mm_pre.view(ec, 16))[:, :m]
                        grads_by_id[id(layer.edge_gate[0].bias)] += g_gate.sum()[None]
                        del mm_pre, g_gate
                    grads_by_id[id(lin_a.weight)] += _tn(g_hid, mm.view(ec, 16))[:hid3, :m]
                    grads_by_id[id(lin_a.bias)] += g_hid.sum(dim=0)[:hid3]
                    grads_by_id[id(lin_b.weight)] += _tn(g_w[:, None], a3)[:, :hid3]
                    grads_by_id[id(lin_b.bias)] += g_w.sum()[None]
                    if norm:
                        grads_by_id[id(layer.coors_norm.scale)] += g_sc.sum()[None]
                    del g_hid, a3, mm
                g_coors_in[lo:hi_] += g_coors[lo:hi_]                              # (the residual; g_rel reaches the coordinates below)
                early()
        elif f0.is_cuda and _TAIL_GENERIC and _GRAD_GEMM and m <= 64:
            # ---- 1. behind u for the heads of 17 .. 64 channels and the other coordinate dimensions (round 5): the pooled messages on
            # E x m tensors, node_mlp on the split-f16 GEMMs, the per-edge chain in closed form on egnn_edge_tail_exact_bwd_f32 (one
            # thread per edge; csrc/edge_exact_bwd.hip) with its parameter gradients on the exact-fp32 GEMM
            closed_dist, g_rel, bias2, gu_bits = False, None, None, None
            with torch.enable_grad():
                c = c0.detach().requires_grad_(True)
                e = None if e0 is None else e0.detach().requires_grad_(want_ge)
                rel, scal = edge_scalars(layer, c, e, i64)                        # (only the scalars' graph is used below)
            with torch.no_grad():
                u4 = u_all[lo:hi_, :, :, :m]
                m_i, pm, cnt = _pooled_messages(layer, u4, m0, i64, r0, ctx.valid_radius)
                g_msum = None
                if layer.node_mlp is not None:
                    f = f0.detach().requires_grad_(True)
                    g_f, g_mi = _node_mlp_backward(layer, w, f, m_i, g_node[lo:hi_], grads_by_id, drop, lo * n, overlap=early)
                    g_feats[lo:hi_] += g_f
                    if layer.m_pool_method == "mean":
                        g_mi = (g_mi / cnt.clamp(min=1e-8)).masked_fill(cnt == 0, 0.0) if cnt is not None else g_mi / k
                    g_msum = g_mi.reshape(bc * n, m).contiguous()
                else:
                    g_feats[lo:hi_] += g_node[lo:hi_]
                g_coors_in[lo:hi_] += g_coors[lo:hi_]
                early()
                g_u, g_c = _tail_closed_form(layer, u4.reshape(ec, m).contiguous(), c0, i32, pm, g_coors[lo:hi_], g_msum, grads_by_id,
                                             dest_lists, bc, n, k, drop, lo * n * k)
                g_coors_in[lo:hi_] += g_c
                gu16 = torch.zeros(ec, mp, dtype=torch.float32, device=feats.device)
                gu16[:, :m] = g_u
        else:
            closed_dist, g_rel, bias2, gu_bits = False, None, None, None
            # ---- 1. the small tail, through autograd
            with torch.enable_grad():
                f = f0.detach().requires_grad_(True)
                c = c0.detach().requires_grad_(True)
                e = None if e0 is None else e0.detach().requires_grad_(want_ge)
                u = u_all[lo:hi_, :, :, :m].detach().requires_grad_(True)
                rel, scal = edge_scalars(layer, c, e, i64)
                out_n, out_c = layer_tail(layer, f, c, u, rel, m0, i64, r0, ctx.valid_radius)
                outs, gouts = [], []
                for o, g in ((out_n, g_node[lo:hi_]), (out_c, g_coors[lo:hi_])):
                    if o.requires_grad:
                        outs.append(o)
                        gouts.append(g)
                live_tail = [p for p in tail_params if p.requires_grad]
                wrt = [u, f, c] + live_tail
                tg = torch.autograd.grad(outs, wrt, gouts, allow_unused=True, retain_graph=True)
            g_u = tg[0] if tg[0] is not None else torch.zeros_like(u)
            if tg[1] is not None:
                g_feats[lo:hi_] += tg[1]
            if tg[2] is not None:
                g_coors_in[lo:hi_] += tg[2]
            for p, g in zip(live_tail, tg[3:]):
                if g is not None:
                    grads_by_id[id(p)] += g
            gu16 = torch.zeros(ec, mp, dtype=torch.float32, device=feats.device)
            gu16[:, :m] = g_u.reshape(ec, m)
        # ---- 2. the E x H work: d/d P_i, d/d P_j (per node), d/d W_s, d/d scalars, d/d W_2
        early()
        amax = _ops.bits_to_floats(gu_bits)[0] if gu_bits is not None else _ops.absmax(gu16)
        gu_scale = _weights.pow2_scale(amax) if amax > 0 else 1.0
        with torch.no_grad():
            f2d = f0.view(bc * n, dim)
            sc2 = scal.detach().reshape(ec, s_in).contiguous()
            contract = _edge_contract_fused
            proj = None if proj_all is None else proj_all[lo * n:hi_ * n]
            if drop is not None and mp == 16:
                assert fused                                                           # (_dropout_native_ok)
                gz_i, gz_j, g_ws, g_scal, g_w2 = contract(layer, w, f2d, c0, e0, sc2, i32, gu16, gu_scale, w_s, bc, n, k, pi_split, dest_lists, proj,
                                                          drop, lo * n * k)
            elif mp == 16:
                gz_i, gz_j, g_ws, g_scal, g_w2 = contract(layer, w, f2d, c0, e0, sc2, i32, gu16, gu_scale, w_s, bc, n, k, pi_split, dest_lists, proj)
            else:
                # m_dim > 16: dz = (W2^T gU) SiLU'(z) and every contraction of it are linear in gU, so the passes run once per block of
                # 16 message channels (its columns of gU, its rows of W2) and the results are added; d/d W_2 is per block anyway
                if dest_lists is None and i32 is not None:
                    dest_lists = _ops.dest_lists(i32, bc, n, k, feats.device)
                parts = []
                # (only blocks that hold real channels: m_dim 33 .. 48 has mp = 64 -- the forward's NB = 4 -- but its fourth block of gU
                # and W2 is all padding, a by-source + by-destination pass that would add exact zeros)
                nblk = (m + 15) // 16
                try:
                    for blk in range(nblk):
                        w["W2Th"], w["w2_block"] = w["W2Th_blocks"][blk], blk   # (in place: what the passes cache in `w` -- bwd_pads -- survives;
                                                                                #  w2_block: read by the CPU tests' kernel emulation)
                        parts.append(contract(layer, w, f2d, c0, e0, sc2, i32, gu16[:, 16 * blk:16 * blk + 16].contiguous(), gu_scale, w_s, bc, n,
                                              k, pi_split, dest_lists, proj, drop, lo * n * k))       # (the mask of z: the same for every block)
                finally:
                    w["W2Th"] = w["W2Th_blocks"][0]
                    w.pop("w2_block", None)
                gz_i, gz_j, g_ws, g_scal = (sum(p[q] for p in parts[1:]) + parts[0][q] for q in range(4))
                g_w2 = torch.cat([p[4] for p in parts] + [torch.zeros_like(parts[0][4])] * (mp // 16 - nblk), dim=0)
                del parts
            # ---- 3. node-level products: d/d feats = dP_i W_i + dP_j W_j, d/d W_i = dP_i^T feats, d/d W_j = dP_j^T feats.  On the
            # device: the forward's split-f16 matrix-core GEMM (operands pre-scaled by powers of two, the weight gradients split-K
            # over the B N nodes with the parts summed in fixed order) -- the fp32 library GEMMs they replace ran at 60 - 130 TFLOP/s
            gw1 = grads_by_id[id(lin0.weight)]
            if f2d.is_cuda and _GRAD_GEMM:
                # (each matrix: one absmax, one read for its plain and transposed images; feats^T split once for both weight gradients)
                # (launch order: everything of d/d P_i first -- its max |.| word was written by the by-source pass, long finished -- so that
                # the device is busy with it while the host waits for the word of d/d P_j, which the last kernel queued above writes)
                ib, jb = getattr(gz_i, "amax_bits", None), getattr(gz_j, "amax_bits", None)
                op_i = _ops.GradOperand(gz_i, amax=None if ib is None else _ops.bits_to_floats(ib)[0], colsum=True)
                gb1 = op_i.colsum
                t = _ops.grad_nn(op_i, w["WiT_split"], dim, name="bwd_dfeats")
                f_op = _ops.grad_tn_operand(f2d, None if hr_f is None else hr_f.floats()[0])
                gw1[:, :dim] += _ops.grad_tn(op_i, f2d, name="bwd_dw1", x_operand=f_op)[:h]
                del op_i
                op_j = _ops.GradOperand(gz_j, amax=None if jb is None else _ops.bits_to_floats(jb)[0])
                t = _ops.grad_nn(op_j, w["WjT_split"], dim, residual=t, name="bwd_dfeats")
                g_feats[lo:hi_] += t.view(bc, n, dim)
                gw1[:, dim:2 * dim] += _ops.grad_tn(op_j, f2d, name="bwd_dw1", x_operand=f_op)[:h]
                del f_op, op_j
            else:
                g_feats[lo:hi_] += (gz_i @ w_i + gz_j @ w_j).view(bc, n, dim)
                gw1[:, :dim] += _tn(gz_i, f2d)[:h]
                gw1[:, dim:2 * dim] += _tn(gz_j, f2d)[:h]
                gb1 = gz_i.sum(dim=0)
            gw1[:, 2 * dim:] += g_ws[:h]
            grads_by_id[id(lin0.bias)] += gb1[:h]
            g_scal = g_scal.view_as(scal)
            grads_by_id[id(lin3.weight)] += g_w2[:m, :h]
            grads_by_id[id(lin3.bias)] += bias2 if bias2 is not None else gu16[:, :m].sum(dim=0)
            del gz_i, gz_j
        # d loss / d scalars -> coordinates (through d = |x_i - x_j|^2 and the fourier terms) and edge features
        if closed_dist:
            with torch.no_grad():                                   # d = |rel|^2:  d loss / d rel += 2 g_d rel
                if rel.shape[-1] == 4:                              # (the tail kernel's (E, 4) rows, 4th column 0)
                    g_rel.addcmul_(g_scal.reshape(ec, 1), rel, value=2.0)
                else:
                    g_rel[:, :3] += (2.0 * g_scal.reshape(ec, 1)) * rel.reshape(ec, 3)
        else:
            sg = torch.autograd.grad([scal], [c] + ([e] if want_ge else []), [g_scal], allow_unused=True)
            if sg[0] is not None:
                g_coors_in[lo:hi_] += sg[0]
            if want_ge and sg[1] is not None:
                g_edges[lo:hi_] += sg[1]
        if g_rel is not None:
            # rel = x_i - x_j reaches the coordinates at the source (sum over a node's edges) and, negated, at the neighbour
            # (fixed-order gather over the edges sorted by destination)
            with torch.no_grad():
                g_coors_in[lo:hi_] += g_rel.view(bc, n, k, 4).sum(dim=2)[..., :3]
                if i64 is None:
                    g_coors_in[lo:hi_] -= g_rel.view(bc, n, n, 4).sum(dim=1)[..., :3]
                else:
                    g_coors_in[lo:hi_] -= _ops.rows_gather_sum(g_rel, dest_lists.order, dest_lists.seg, bc * n).view(bc, n, 4)[..., :3]
    unused = _unused_params(layer, ctx)
    out_params = [grads_by_id[id(p)].to(op.dtype) if (need[7 + i] and id(p) not in unused) else None
                  for i, (p, op) in enumerate(zip(params, orig_params))]
    return (None, None, None, None, g_feats.to(in_dtypes[0]) if need[4] else None, g_coors_in.to(in_dtypes[1]) if need[5] else None,
            g_edges.to(in_dtypes[2]) if want_ge else None, *out_params)


def _backward_recompute(ctx, g_node, g_coors):
    """The pure-ATen backward: chunked recompute of the whole layer through autograd (module docstring).  Used where neither native
    path applies -- more per-edge scalars than `_backward_exact` carries, CPU tensors,
    the EGNN_NATIVE_BACKWARD* switches -- and as the native paths' reference in the tests."""
    layer = ctx.layer
    feats, coors, edges, mask, idx, rank = _unpack(ctx)
    idx = None if idx is None else idx.long()
    params = [p for p in layer.parameters()]
    # inputs whose dtype differs from the parameters' (the forward converts at the boundary): differentiate in the parameters'
    # dtype, return every gradient in the dtype of the tensor it belongs to
    in_dtypes = (feats.dtype, coors.dtype, None if edges is None else edges.dtype)
    pd = params[0].dtype if params else feats.dtype
    feats, coors = feats.to(pd), coors.to(pd)
    edges = None if edges is None else edges.to(pd)
    rank = None if rank is None else rank.to(pd)
    g_node = None if g_node is None else g_node.to(pd)
    g_coors = None if g_coors is None else g_coors.to(pd)
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
                                                 ctx.valid_radius, drop=getattr(ctx, "drop", None), graph_offset=lo)
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
    cast = lambda g, dt: None if g is None else g.to(dt)                                        # noqa: E731
    unused = _unused_params(layer, ctx)
    g_params = [None if id(p) in unused else g for p, g in zip(params, g_params)]
    return (None, None, None, None, cast(g_feats, in_dtypes[0]), cast(g_coors_in, in_dtypes[1]), cast(g_edges, in_dtypes[2]), *g_params)


def wants_grad(layer, *tensors):
    """True when a forward under the current autograd mode has to record a graph."""
    if not torch.is_grad_enabled():
        return False
    if any(torch.is_tensor(t) and t.is_floating_point() and t.requires_grad for t in tensors):
        return True
    return any(p.requires_grad for p in layer.parameters())
