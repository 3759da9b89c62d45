"""Induced-set ("global linear") attention of EGNN_Network (egnn_pytorch/egnn_pytorch.py:81-144).

A handful of global tokens attend over the node features and the nodes attend back over the induced tokens.  Parameter
names and shapes are the reference's, so its `state_dict` loads unchanged (`layers.{l}.0.*`, `global_tokens`).

Inference on the MI355X (`GlobalLinearAttention._forward_hip`): every per-node projection -- attn1.to_kv, attn2.to_q,
attn2.to_out (+ residual), the feed-forward (Linear, exact GELU in the epilogue, Linear + residual) -- runs on the split-f16
GEMM (egnn_linear_hl_f32) with the LayerNorms fused into its operand packing (egnn_node_prep_hl), and the two attention cores
are HIP kernels (csrc/global_attn.hip: egnn_induced_attn_f32, egnn_token_attn_f32).  What stays in ATen is token-sized: the
LayerNorm and the three small Linears over the (B, T, dim) global tokens.  Under autograd (and on the CPU, where only the
tests run it) the block is the plain differentiable module below."""
from __future__ import annotations

import torch
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
        dots = (q @ k.transpose(-1, -2)) * self.scale                                 # (b, h, n, n_context)
        if mask is not None:
            # as the reference (:102-105): masked logits become -finfo.max, so a context that is entirely masked (a fully
            # padded graph) softmaxes to a uniform distribution with finite outputs.  Spelled out instead of
            # F.scaled_dot_product_attention: its fused GPU kernels return something else for fully masked rows.
            dots = dots.masked_fill(~mask[:, None, None, :], -torch.finfo(dots.dtype).max)
        out = dots.softmax(dim=-1) @ v
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
        from . import layer as _layer
        if x.is_cuda and not torch.is_grad_enabled() and x.dtype == torch.float32 and queries.dtype == torch.float32 \
                and queries.shape[1] <= 8 and self.attn1.to_q.weight.shape[0] // self.attn1.heads <= 256 \
                and not _layer.exact_active():                    # (plain-fp32 mode: the differentiable module below, in fp32)
            return self._forward_hip(x, queries, mask)
        seq, tok = self.norm_seq(x), self.norm_queries(queries)
        induced = self.attn1(tok, seq, mask=mask)
        x = self.attn2(seq, induced) + x
        return self.ff(x) + x, induced + queries

    # ------------------------------------------------------------------ gfx950 inference path
    def _packed(self):
        from . import _weights
        key = _weights.version_key(self)
        if getattr(self, "_pk", None) is None or self._pk_key != key:
            f = lambda w: _weights.split_f16(w.detach().float())
            v = lambda t: t.detach().float().contiguous()
            self._pk = dict(kv1=f(self.attn1.to_kv.weight), q2=f(self.attn2.to_q.weight), o2=f(self.attn2.to_out.weight),
                            ff1=f(self.ff[1].weight), ff2=f(self.ff[3].weight), bo2=v(self.attn2.to_out.bias),
                            bf1=v(self.ff[1].bias), bf2=v(self.ff[3].bias), g_seq=v(self.norm_seq.weight),
                            b_seq=v(self.norm_seq.bias), g_ff=v(self.ff[0].weight), b_ff=v(self.ff[0].bias))
            self._pk_key = key
        return self._pk

    def _forward_hip(self, x, queries, mask):
        from . import _ops
        b, n, dim = x.shape
        a1, a2 = self.attn1, self.attn2
        heads = a1.heads
        inner = a1.to_q.weight.shape[0]
        dh = inner // heads
        w = self._packed()
        x2d = x.contiguous().view(b * n, dim)
        seq_hl = _ops.node_prep_hl(x2d, None, w["g_seq"], w["b_seq"], self.norm_seq.eps, 0)            # LayerNorm(x), packed (hi, lo)
        kv = _ops.linear_hl(seq_hl, w["kv1"], 2 * inner, name="attn_kv")                               # attn1.to_kv
        tok = self.norm_queries(queries)                                                               # (B, T, dim): token-sized ATen
        induced = _ops.induced_attn(a1.to_q(tok), kv, mask, b, n, heads, dh, a1.scale)
        induced = a1.to_out(induced)                                                                   # (B, T, dim)
        q2 = _ops.linear_hl(seq_hl, w["q2"], inner, name="attn_q")                                     # attn2.to_q
        att = _ops.token_attn(q2, a2.to_kv(induced), b, n, heads, dh, a2.scale)
        x1 = _ops.linear_hl(_ops.split_f16(att), w["o2"], dim, w["bo2"], residual=x2d, name="attn_out")   # to_out + residual
        h_hl = _ops.node_prep_hl(x1, None, w["g_ff"], w["b_ff"], self.ff[0].eps, 0)
        hid = _ops.linear_hl(h_hl, w["ff1"], 4 * dim, w["bf1"], act=2, out_f32=False, out_hl=True, name="attn_ff0")
        x2 = _ops.linear_hl(hid, w["ff2"], dim, w["bf2"], residual=x1, name="attn_ff1")
        return x2.view(b, n, dim), induced + queries
