"""CPU oracle for the EGNN.forward hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a plain-numpy restatement of the algorithm in the reference
(lucidrains/egnn-pytorch v0.2.8).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the *checker*: the shipped
path (`egnn_pytorch_amd`) never imports anything under `oracle/` and raises when its HIP
library is missing.

Parity pin: the reference publishes no golden vectors.  This oracle is
pinned instead against outputs of the reference itself, generated in the dev container by
`tests/golden/make_golden.py` (imports `/root/reference`) and committed as
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks the oracle against every one of
them, and `tests/test_oracle_vs_reference.py` re-runs the live reference when it is
present.

Every function cites the reference lines it restates (paths relative to /root/reference).
The arithmetic is deliberately the reference's *unfactorised* op order (materialised
edge_input, one Linear over the concatenation) so that the oracle is an independent check
of the factorised HIP path.
"""
from __future__ import annotations

import numpy as np

RANK_MASKED = 1e5      # egnn_pytorch/egnn_pytorch.py:242
RANK_SELF = -1.0       # egnn_pytorch/egnn_pytorch.py:255
RANK_ADJ = 0.0         # egnn_pytorch/egnn_pytorch.py:256


# --------------------------------------------------------------------------- helpers

def silu(x):
    """nn.SiLU  (egnn_pytorch/egnn_pytorch.py:56-60): x * sigmoid(x)."""
    return x / (1.0 + np.exp(-x, dtype=x.dtype))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x, dtype=x.dtype))


def linear(x, w, b=None):
    """nn.Linear: x @ w.T + b."""
    y = x @ w.T
    if b is not None:
        y = y + b
    return y


def layer_norm(x, weight, bias, eps=1e-5):
    """nn.LayerNorm(dim) as used at egnn_pytorch/egnn_pytorch.py:191,335 (biased variance)."""
    mu = x.mean(axis=-1, keepdims=True, dtype=x.dtype)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=x.dtype)
    return xc / np.sqrt(var + x.dtype.type(eps)) * weight + bias


def safe_div(num, den, eps=1e-8):
    """egnn_pytorch/egnn_pytorch.py:13-16."""
    res = num / np.maximum(den, num.dtype.type(eps))
    res = np.where(den == 0, num.dtype.type(0), res)
    return res


def fourier_encode_dist(x, num_encodings):
    """egnn_pytorch/egnn_pytorch.py:34-41.  x: (...,) -> (..., 2F+1) = [sin.., cos.., x]."""
    x = x[..., None]
    scales = (2.0 ** np.arange(num_encodings)).astype(x.dtype)
    xs = x / scales
    return np.concatenate([np.sin(xs), np.cos(xs), x], axis=-1)


def pairwise(coors):
    """egnn_pytorch/egnn_pytorch.py:232-233.

    rel_coors[b,i,j,:] = coors[b,i] - coors[b,j];  rel_dist = sum_c rel^2.
    For C == 3 the reference's CPU result is bit-identical to ((dx*dx + dy*dy) + dz*dz) with
    separately rounded multiplies and adds; that order is pinned here.
    Other C: `inner_sum` below -- the summation tree of ATen's inner-dimension sum, which the ranking has to follow bit for bit.
    """
    rel = coors[:, :, None, :] - coors[:, None, :, :]
    sq = rel * rel
    return rel, inner_sum(sq)


def inner_sum(x):
    """`x.sum(dim=-1)` of a contiguous tensor as the reference's CPU path computes it (ATen, aten/src/ATen/native/cpu/SumKernel.cpp:
    vectorized_inner_sum / scalar_inner_sum over row_sum with four interleaved partial sums), restated and checked bit for bit against
    torch 2.10 for C = 1 .. 513 in float32 and 1 .. 200 in float64 (tests/test_oracle_vs_reference.py).  V = lanes of the vector type:
    8 for float32, 4 for float64.
      C <  V: partial sums p[j] = x[j] + x[4 + j] + ..., the tail x[4 (C // 4):] added to p[0], then ((p0 + p1) + p2) + p3;
      C >= V: whole vectors v_i = x[V i : V i + V] -- four interleaved partial vectors over groups of four, the remaining vectors added
              to the first, the four folded into it; then a scalar that starts at 0, takes the tail x[V (C // V):] first and the V lanes
              after.
    (Rows longer than 512 / 256 elements cascade in blocks and are not covered.)  For C <= 8 in float32 this is: left to right for
    C in {1, 2, 3, 4, 8}; s0, s4, ..., s_{C-1}, s1, s2, s3 for C in {5, 6, 7}."""
    c = x.shape[-1]
    v = 4 if x.dtype == np.float64 else 8
    zero = np.zeros(x.shape[:-1], dtype=x.dtype)
    if c < v:
        p = [zero.copy() for _ in range(4)]
        for i in range(c // 4):
            for j in range(4):
                p[j] = p[j] + x[..., 4 * i + j]
        for k in range(4 * (c // 4), c):
            p[0] = p[0] + x[..., k]
        return ((p[0] + p[1]) + p[2]) + p[3]
    nv = c // v
    assert nv // 4 <= 16, "rows this long cascade in ATen: not restated"
    p = [np.zeros(x.shape[:-1] + (v,), dtype=x.dtype) for _ in range(4)]
    for i in range(nv // 4):
        for j in range(4):
            p[j] = p[j] + x[..., v * (4 * i + j):v * (4 * i + j) + v]
    for i in range(4 * (nv // 4), nv):
        p[0] = p[0] + x[..., v * i:v * i + v]
    lanes = ((p[0] + p[1]) + p[2]) + p[3]
    acc = zero.copy()
    for k in range(v * nv, c):
        acc = acc + x[..., k]
    for k in range(v):
        acc = acc + lanes[..., k]
    return acc


def sum_order(c):
    """Order in which the reference's fp32 `(rel_coors ** 2).sum(-1)` adds its C <= 8 terms (`inner_sum` is the general statement)."""
    assert c <= 8
    if c in (5, 6, 7):
        return [0] + list(range(4, c)) + [1, 2, 3]
    return list(range(c))


def build_ranking(rel_dist, mask, adj_mat):
    """egnn_pytorch/egnn_pytorch.py:237-256.  Returns (ranking (B,N,N), adj_mat without diagonal)."""
    b, n, _ = rel_dist.shape
    ranking = rel_dist.copy()
    if mask is not None:
        rank_mask = mask[:, :, None] & mask[:, None, :]
        ranking[~rank_mask] = rel_dist.dtype.type(RANK_MASKED)
    adj = None
    if adj_mat is not None:
        adj = np.broadcast_to(adj_mat, (b, n, n)).copy() if adj_mat.ndim == 2 else adj_mat.copy()
        eye = np.eye(n, dtype=bool)[None]
        adj = adj & ~eye
        ranking[np.broadcast_to(eye, ranking.shape)] = rel_dist.dtype.type(RANK_SELF)
        ranking[adj] = rel_dist.dtype.type(RANK_ADJ)
    return ranking, adj


def topk_smallest(ranking, k):
    """Tensor.topk(k, dim=-1, largest=False) at egnn_pytorch/egnn_pytorch.py:258.

    Values ascending.  Tie order: the reference's (ATen) order is implementation-defined;
    the oracle and the HIP kernel both use ascending index."""
    if k > ranking.shape[-1]:
        raise RuntimeError("selected index k out of range")       # what torch.topk raises
    idx = np.argsort(ranking, axis=-1, kind="stable")[..., :k]
    val = np.take_along_axis(ranking, idx, axis=-1)
    return val, idx


def sparse_num_nearest(adj_mat):
    """egnn_pytorch/egnn_pytorch.py:249: int(adj_mat.float().sum(-1).max()) -- computed BEFORE the
    diagonal is cleared, so a set diagonal is counted."""
    if adj_mat.size == 0:
        return 0
    return int(adj_mat.astype(np.float32).sum(axis=-1).max())


# --------------------------------------------------------------------------- the layer

class EGNNConfig:
    """Constructor arguments of EGNN (egnn_pytorch/egnn_pytorch.py:149-168)."""

    def __init__(self, dim, edge_dim=0, m_dim=16, fourier_features=0, num_nearest_neighbors=0,
                 dropout=0.0, init_eps=1e-3, norm_feats=False, norm_coors=False,
                 norm_coors_scale_init=1e-2, update_feats=True, update_coors=True,
                 only_sparse_neighbors=False, valid_radius=float("inf"), m_pool_method="sum",
                 soft_edges=False, coor_weights_clamp_value=None):
        assert m_pool_method in {"sum", "mean"}
        assert update_feats or update_coors
        self.dim = dim
        self.edge_dim = edge_dim
        self.m_dim = m_dim
        self.fourier_features = fourier_features
        self.num_nearest_neighbors = num_nearest_neighbors
        self.norm_feats = norm_feats
        self.norm_coors = norm_coors
        self.update_feats = update_feats
        self.update_coors = update_coors
        self.only_sparse_neighbors = only_sparse_neighbors
        self.valid_radius = valid_radius
        self.m_pool_method = m_pool_method
        self.soft_edges = soft_edges
        self.coor_weights_clamp_value = coor_weights_clamp_value


def egnn_forward(cfg, params, feats, coors, edges=None, mask=None, adj_mat=None, prefix="",
                 return_neighbors=False):
    """EGNN.forward (egnn_pytorch/egnn_pytorch.py:224-341), eval mode (dropout = identity).

    `params`: mapping of reference state_dict key -> ndarray (keys as listed in SURVEY.md §8b).
    Returns (node_out, coors_out) and, if return_neighbors, also (nbhd_ranking, nbhd_indices)
    (None on the dense path)."""
    p = lambda k: params[prefix + k]
    dt = feats.dtype
    b, n, d = feats.shape
    num_nearest = cfg.num_nearest_neighbors
    valid_radius = cfg.valid_radius
    use_nearest = num_nearest > 0 or cfg.only_sparse_neighbors                       # :230

    rel_coors, rel_dist = pairwise(coors)                                           # :232-233
    nbhd_ranking = nbhd_indices = None

    if use_nearest:
        ranking, _ = build_ranking(rel_dist, mask, adj_mat)                         # :237-256
        if adj_mat is not None and cfg.only_sparse_neighbors:
            adj_full = np.broadcast_to(adj_mat, (b, n, n)) if adj_mat.ndim == 2 else adj_mat
            num_nearest = sparse_num_nearest(adj_full)                              # :249
            valid_radius = 0                                                        # :250
        nbhd_ranking, nbhd_indices = topk_smallest(ranking, num_nearest)            # :258
        nbhd_mask = nbhd_ranking <= valid_radius                                    # :260
        bi = np.arange(b)[:, None, None]
        ii = np.arange(n)[None, :, None]
        rel_coors = rel_coors[bi, ii, nbhd_indices]                                 # :262
        rel_dist = rel_dist[bi, ii, nbhd_indices]                                   # :263
        if edges is not None:
            edges = edges[bi, ii, nbhd_indices]                                     # :266
        feats_j = feats[bi, nbhd_indices]                                           # :275
        k = num_nearest
    else:
        feats_j = np.broadcast_to(feats[:, None, :, :], (b, n, n, d))               # :277
        k = n

    if cfg.fourier_features > 0:
        dist_feat = fourier_encode_dist(rel_dist, cfg.fourier_features)             # :270-272
    else:
        dist_feat = rel_dist[..., None]

    feats_i = np.broadcast_to(feats[:, :, None, :], (b, n, k, d))                   # :279-280
    edge_input = np.concatenate([feats_i, feats_j, dist_feat], axis=-1)             # :282
    if edges is not None:
        edge_input = np.concatenate([edge_input, edges], axis=-1)                   # :285

    h = silu(linear(edge_input, p("edge_mlp.0.weight"), p("edge_mlp.0.bias")))      # :287 (:178-184)
    m_ij = silu(linear(h, p("edge_mlp.3.weight"), p("edge_mlp.3.bias")))
    del h, edge_input

    if cfg.soft_edges:                                                              # :289-290
        gate = sigmoid(linear(m_ij, p("edge_gate.0.weight"), p("edge_gate.0.bias")))
        m_ij = m_ij * gate

    emask = None
    if mask is not None:                                                            # :292-300
        mask_i = mask[:, :, None]
        if use_nearest:
            mask_j = mask[np.arange(b)[:, None, None], nbhd_indices]
            emask = (mask_i & mask_j) & nbhd_mask
        else:
            emask = mask_i & mask[:, None, :]

    if cfg.update_coors:                                                            # :302-317
        cw = silu(linear(m_ij, p("coors_mlp.0.weight"), p("coors_mlp.0.bias")))
        cw = linear(cw, p("coors_mlp.3.weight"), p("coors_mlp.3.bias"))[..., 0]
        rc = rel_coors
        if cfg.norm_coors:                                                          # CoorsNorm :67-77
            nrm = np.sqrt((rc * rc).sum(axis=-1, keepdims=True, dtype=dt))
            rc = rc / np.maximum(nrm, dt.type(1e-8)) * p("coors_norm.scale")
        if emask is not None:
            cw = np.where(emask, cw, dt.type(0))                                    # :308-309
        if cfg.coor_weights_clamp_value is not None:                                # :311-313
            cv = dt.type(cfg.coor_weights_clamp_value)
            cw = np.clip(cw, -cv, cv)
        coors_out = np.einsum("bij,bijc->bic", cw, rc).astype(dt) + coors           # :315
    else:
        coors_out = coors

    if cfg.update_feats:                                                            # :319-339
        if emask is not None:
            m_ij = np.where(emask[..., None], m_ij, dt.type(0))                     # :320-322
        if cfg.m_pool_method == "mean":
            if emask is not None:
                mask_sum = emask[..., None].sum(axis=-2).astype(dt)                 # :326-327
                m_i = safe_div(m_ij.sum(axis=-2, dtype=dt), mask_sum)
            else:
                m_i = m_ij.mean(axis=-2, dtype=dt)                                  # :330
        else:
            m_i = m_ij.sum(axis=-2, dtype=dt)                                       # :333
        normed = feats
        if cfg.norm_feats:
            normed = layer_norm(feats, p("node_norm.weight"), p("node_norm.bias"))  # :335
        node_in = np.concatenate([normed, m_i], axis=-1)                            # :336
        hid = silu(linear(node_in, p("node_mlp.0.weight"), p("node_mlp.0.bias")))
        node_out = linear(hid, p("node_mlp.3.weight"), p("node_mlp.3.bias")) + feats  # :337
    else:
        node_out = feats

    node_out = node_out.astype(dt, copy=False)
    coors_out = coors_out.astype(dt, copy=False)
    if return_neighbors:
        return node_out, coors_out, nbhd_ranking, nbhd_indices
    return node_out, coors_out


def adjacency_degrees(adj_mat, num_adj_degrees):
    """N-degree adjacency expansion (egnn_pytorch/egnn_pytorch.py:414-427).  adj_mat (B,N,N) bool.
    Returns (adj_indices int64 (B,N,N), expanded adj_mat bool): degree d >= 2 labels the entries where
    (adj @ adj > 0) differs from adj (the reference's `(next.float() - adj.float()).bool()` is an XOR), and the
    adjacency handed to the layers is the expanded one."""
    adj = adj_mat.copy()
    adj_indices = adj.astype(np.int64)
    for ind in range(num_adj_degrees - 1):
        degree = ind + 2
        a = adj.astype(np.float32)
        nxt = (a @ a) > 0
        mask = nxt != adj
        adj_indices[mask] = degree
        adj = nxt
    return adj_indices, adj


def network_frontend(params, feats, coors, adj_mat=None, edges=None, num_adj_degrees=None):
    """EGNN_Network.forward up to the layer loop (egnn_pytorch/egnn_pytorch.py:401-432): token / position / edge-token
    embeddings, adjacency-degree expansion and its embedding concatenated onto the edge features.
    Embedding tables are taken from `params` when present (token_emb / pos_emb / edge_emb / adj_emb .weight)."""
    b = feats.shape[0]
    if "token_emb.weight" in params:
        feats = params["token_emb.weight"][feats]                                    # :401-402
    if "pos_emb.weight" in params:
        n = feats.shape[1]
        feats = feats + params["pos_emb.weight"][np.arange(n)][None]                 # :404-408
    if edges is not None and "edge_emb.weight" in params:
        edges = params["edge_emb.weight"][edges]                                     # :410-411
    if num_adj_degrees is not None:
        assert adj_mat is not None
        n = adj_mat.shape[-1]
        adj = np.broadcast_to(adj_mat, (b, n, n)).copy() if adj_mat.ndim == 2 else adj_mat.copy()
        adj_indices, adj_mat = adjacency_degrees(adj, num_adj_degrees)               # :414-427
        if "adj_emb.weight" in params:
            adj_emb = params["adj_emb.weight"][adj_indices]                          # :429-431
            edges = np.concatenate([edges, adj_emb], axis=-1) if edges is not None else adj_emb
    return feats, coors, adj_mat, edges


def gelu(x):
    """nn.GELU() default (exact erf form), egnn_pytorch.py:129."""
    import math
    erf = np.vectorize(math.erf, otypes=[x.dtype])
    return (0.5 * x * (1.0 + erf(x / np.sqrt(2.0).astype(x.dtype)))).astype(x.dtype)


def attention(params, prefix, x, context, heads, mask=None):
    """Attention.forward (egnn_pytorch/egnn_pytorch.py:93-113): multi-head softmax attention of x over context;
    `mask` (B, n_context) removes context positions (filled with -finfo.max before the softmax, :104-107)."""
    q = linear(x, params[prefix + "to_q.weight"])                                   # :96
    kv = linear(context, params[prefix + "to_kv.weight"])                            # :97
    inner = q.shape[-1]
    k, v = kv[..., :inner], kv[..., inner:]
    dh = inner // heads
    split = lambda t: t.reshape(t.shape[0], t.shape[1], heads, dh).transpose(0, 2, 1, 3)      # b h n d, :99
    q, k, v = split(q), split(k), split(v)
    dots = np.einsum("bhid,bhjd->bhij", q, k) * np.asarray(dh ** -0.5, dtype=q.dtype)           # :100, scale :86
    if mask is not None:
        dots = np.where(mask[:, None, None, :], dots, -np.finfo(dots.dtype).max)                 # :102-105
    dots = dots - dots.max(-1, keepdims=True)
    attn = np.exp(dots)
    attn = attn / attn.sum(-1, keepdims=True)                                        # :107
    out = np.einsum("bhij,bhjd->bhid", attn, v).astype(q.dtype)                      # :108
    out = out.transpose(0, 2, 1, 3).reshape(x.shape[0], x.shape[1], inner)           # :110
    return linear(out, params[prefix + "to_out.weight"], params[prefix + "to_out.bias"])         # :111


def global_linear_attention(params, prefix, x, queries, heads, mask=None):
    """GlobalLinearAttention.forward (egnn_pytorch/egnn_pytorch.py:133-144): the global tokens attend over the
    (masked) sequence, the sequence attends over the induced tokens; residuals; LayerNorm-Linear-GELU-Linear."""
    res_x, res_q = x, queries
    x = layer_norm(x, params[prefix + "norm_seq.weight"], params[prefix + "norm_seq.bias"])             # :135
    queries = layer_norm(queries, params[prefix + "norm_queries.weight"], params[prefix + "norm_queries.bias"])
    induced = attention(params, prefix + "attn1.", queries, x, heads, mask=mask)      # :137
    out = attention(params, prefix + "attn2.", x, induced, heads)                      # :138
    x = out + res_x                                                                    # :140
    queries = induced + res_q                                                          # :141
    h = layer_norm(x, params[prefix + "ff.0.weight"], params[prefix + "ff.0.bias"])
    h = gelu(linear(h, params[prefix + "ff.1.weight"], params[prefix + "ff.1.bias"]))
    x = linear(h, params[prefix + "ff.3.weight"], params[prefix + "ff.3.bias"]) + x   # :143
    return x, queries


def egnn_network_forward(depth, cfg, params, feats, coors, adj_mat=None, edges=None, mask=None,
                         return_coor_changes=False, num_adj_degrees=None, global_linear_attn_every=0,
                         global_linear_attn_heads=8):
    """EGNN_Network.forward (egnn_pytorch/egnn_pytorch.py:390-454): the front-end (network_frontend), the optional
    global attention blocks (:376-388, :434-446) and the layer loop (:442-454).  `cfg` is the per-layer EGNN
    configuration (edge_dim already includes adj_dim) and must have norm_feats=True (forced at :387).
    State-dict prefixes: layers.{l}.0. (attention, layers with l % every == 0), layers.{l}.1. (EGNN)."""
    assert cfg.norm_feats
    feats, coors, adj_mat, edges = network_frontend(params, feats, coors, adj_mat, edges, num_adj_degrees)
    feats = feats.astype(coors.dtype, copy=False)
    if edges is not None:
        edges = edges.astype(coors.dtype, copy=False)
    global_tokens = None
    if global_linear_attn_every > 0:
        global_tokens = np.broadcast_to(params["global_tokens"][None], (feats.shape[0],) + params["global_tokens"].shape)
    coor_changes = [coors]
    for layer in range(depth):
        if global_linear_attn_every > 0 and layer % global_linear_attn_every == 0:                    # :382
            feats, global_tokens = global_linear_attention(params, f"layers.{layer}.0.", feats, global_tokens,
                                                           global_linear_attn_heads, mask=mask)        # :445-446
        feats, coors = egnn_forward(cfg, params, feats, coors, edges=edges, mask=mask,
                                    adj_mat=adj_mat, prefix=f"layers.{layer}.1.")
        coor_changes.append(coors)
    if return_coor_changes:
        return feats, coors, coor_changes
    return feats, coors


# --------------------------------------------------------------------------- parameter helpers

def param_shapes(cfg):
    """Shapes of the reference state_dict, in registration order."""
    din = 2 * cfg.fourier_features + 2 * cfg.dim + cfg.edge_dim + 1                # :175
    m = cfg.m_dim
    shapes = {
        "edge_mlp.0.weight": (2 * din, din), "edge_mlp.0.bias": (2 * din,),
        "edge_mlp.3.weight": (m, 2 * din), "edge_mlp.3.bias": (m,),
    }
    if cfg.soft_edges:
        shapes.update({"edge_gate.0.weight": (1, m), "edge_gate.0.bias": (1,)})
    if cfg.norm_feats:
        shapes.update({"node_norm.weight": (cfg.dim,), "node_norm.bias": (cfg.dim,)})
    if cfg.norm_coors:
        shapes.update({"coors_norm.scale": (1,)})
    if cfg.update_feats:
        shapes.update({"node_mlp.0.weight": (2 * cfg.dim, cfg.dim + m), "node_mlp.0.bias": (2 * cfg.dim,),
                       "node_mlp.3.weight": (cfg.dim, 2 * cfg.dim), "node_mlp.3.bias": (cfg.dim,)})
    if cfg.update_coors:
        shapes.update({"coors_mlp.0.weight": (4 * m, m), "coors_mlp.0.bias": (4 * m,),
                       "coors_mlp.3.weight": (1, 4 * m), "coors_mlp.3.bias": (1,)})
    return shapes


def random_params(cfg, seed, prefix="", dtype=np.float32, scale="xavier"):
    """Seeded parameters at a scale that makes parity discriminating (the
    reference's default std-1e-3 init makes feature parity vacuous)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in param_shapes(cfg).items():
        if name.endswith("weight") and len(shp) == 2:
            std = np.sqrt(2.0 / (shp[0] + shp[1])) if scale == "xavier" else 1e-3
            out[prefix + name] = (rng.standard_normal(shp) * std).astype(dtype)
        elif name == "node_norm.weight":
            out[prefix + name] = (1.0 + 0.1 * rng.standard_normal(shp)).astype(dtype)
        elif name == "node_norm.bias":
            out[prefix + name] = (0.1 * rng.standard_normal(shp)).astype(dtype)
        elif name == "coors_norm.scale":
            out[prefix + name] = np.full(shp, 1e-2 if scale != "xavier" else 0.5, dtype=dtype)
        else:   # Linear biases: U(+-1/sqrt(fan_in))
            wshape = param_shapes(cfg)[name.replace("bias", "weight")]
            bound = 1.0 / np.sqrt(wshape[1])
            out[prefix + name] = rng.uniform(-bound, bound, shp).astype(dtype)
    return out
