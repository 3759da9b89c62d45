"""Induced-set ("global linear") attention of EGNN_Network (egnn_pytorch/egnn_pytorch.py:81-144).

Outside the per-edge hot path: a handful of global tokens attend over the node features
and the nodes attend back over the induced tokens -- O(N * num_global_tokens) work per graph, done with stock
device ops (hipBLASLt GEMMs + fused softmax).  Parameter names and shapes are the reference's, so its
`state_dict` loads unchanged (`layers.{l}.0.*`, `global_tokens`)."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn


class Attention(nn.Module):
    """Multi-head softmax attention of `x` over `context` (egnn_pytorch.py:83-113).  `mask` (B, n_context) removes
    context positions."""

    def __init__(self, dim, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim)

    def forward(self, x, context, mask=None):
        b, n, _ = x.shape
        h = self.heads
        q = self.to_q(x).view(b, n, h, -1).transpose(1, 2)                           # (b, h, n, d)
        k, v = self.to_kv(context).view(b, context.shape[1], 2, h, -1).permute(2, 0, 3, 1, 4)
        attn_mask = None
        if mask is not None:
            # the reference fills masked logits with -finfo.max (:102-105): a row whose context is entirely masked (a fully
            # padded graph) then softmaxes to a uniform distribution with finite outputs; a boolean attn_mask would give NaN
            attn_mask = torch.zeros(mask.shape, dtype=q.dtype, device=q.device).masked_fill_(~mask, -torch.finfo(q.dtype).max)
            attn_mask = attn_mask[:, None, None, :]
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, scale=self.scale)
        return self.to_out(out.transpose(1, 2).reshape(b, n, -1))


class GlobalLinearAttention(nn.Module):
    """norm -> tokens attend over the (masked) sequence -> sequence attends over the induced tokens -> residuals ->
    LayerNorm-Linear-GELU-Linear feed-forward with residual (egnn_pytorch.py:115-144)."""

    def __init__(self, *, dim, heads=8, dim_head=64):
        super().__init__()
        self.norm_seq = nn.LayerNorm(dim)
        self.norm_queries = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads, dim_head)
        self.attn2 = Attention(dim, heads, dim_head)
        self.ff = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, dim * 4), nn.GELU(), nn.Linear(dim * 4, dim))

    def forward(self, x, queries, mask=None):
        seq, tok = self.norm_seq(x), self.norm_queries(queries)
        induced = self.attn1(tok, seq, mask=mask)
        x = self.attn2(seq, induced) + x
        return self.ff(x) + x, induced + queries
