"""Re-layout of the reference's parameters for the gfx950 kernels (pure tensor plumbing).

The module keeps the reference's parameters / state_dict untouched; the kernels read
derived, padded copies built here and cached per parameter version:

  edge_mlp.0.weight (H, Din), columns [h_i | h_j | fourier sin, cos | d | e]  (egnn_pytorch.py:282-285)
      -> Wcat (2*Hp, dim): rows [0,H) = W_i, rows [Hp, Hp+H) = W_j   (node-level projection weights)
      -> bcat (2*Hp):      [0,H) = edge_mlp.0.bias                   (folded into P_i)
      -> Ws   (Sp, Hp):    the per-edge scalar columns, transposed
  edge_mlp.3.weight (m, H) -> W2f (Hp/16, 64, 4): MFMA-fragment order, lane = 16*g + channel,
                                                   element t = W2[channel, 16*step + 4*g + t]
  coors_mlp.* / edge_gate.* -> zero padded to 16 channels / 64 hidden units
Zero padding is exact: padded hidden units see x = 0 -> SiLU(0) = 0 and meet zero W2 columns.
"""
from __future__ import annotations

import torch

SP_SUPPORTED = (1, 2, 3, 5, 8, 16)     # template instantiations of the edge kernel
M_PAD = 16                              # channels of one 16x16 MFMA tile
C_PAD = 64                              # coors_mlp hidden units (4 * 16)


def padded_hidden(h: int) -> int:
    return (h + 31) // 32 * 32


def padded_scalars(s: int) -> int:
    for sp in SP_SUPPORTED:
        if sp >= s:
            return sp
    raise NotImplementedError(
        f"{s} per-edge scalar inputs (2*fourier_features + 1 + edge_dim) exceed the {SP_SUPPORTED[-1]} "
        f"the gfx950 edge kernel is built for")


def pack(layer) -> dict:
    """Build the kernel-side weight set of one EGNN layer.  All outputs are fp32, contiguous, on the
    parameters' device."""
    w1 = layer.edge_mlp[0].weight.detach().float()
    b1 = layer.edge_mlp[0].bias.detach().float()
    w2 = layer.edge_mlp[3].weight.detach().float()
    b2 = layer.edge_mlp[3].bias.detach().float()
    dev = w1.device
    dim, m = layer.dim, layer.m_dim
    h, din = w1.shape
    s = din - 2 * dim
    if m > M_PAD:
        raise NotImplementedError(f"m_dim={m} > {M_PAD} is not supported by the gfx950 edge kernel")
    hp = padded_hidden(h)
    sp = padded_scalars(s)
    z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)

    wcat = z(2 * hp, dim)
    wcat[:h] = w1[:, :dim]
    wcat[hp:hp + h] = w1[:, dim:2 * dim]
    bcat = z(2 * hp)
    bcat[:h] = b1
    ws = z(sp, hp)
    ws[:s, :h] = w1[:, 2 * dim:].t()

    w2p = z(M_PAD, hp)
    w2p[:m, :h] = w2
    # (channel, step, g, t) -> (step, g, channel, t) -> (step, lane = 16 g + channel, t)
    w2f = w2p.view(M_PAD, hp // 16, 4, 4).permute(1, 2, 0, 3).contiguous().view(hp // 16, 64, 4)
    b2p = z(M_PAD)
    b2p[:m] = b2

    out = dict(H=h, Hp=hp, S=s, Sp=sp, Wcat=wcat, bcat=bcat, Ws=ws, W2f=w2f, b2=b2p)

    if layer.edge_gate is not None:
        gw = z(M_PAD)
        gw[:m] = layer.edge_gate[0].weight.detach().float()[0]
        out["gate_w"] = gw
        out["gate_b"] = layer.edge_gate[0].bias.detach().float().contiguous()
    if layer.coors_mlp is not None:
        w3 = layer.coors_mlp[0].weight.detach().float()          # (4m, m)
        w3p = z(C_PAD, M_PAD)
        w3p[:4 * m, :m] = w3
        b3p = z(C_PAD)
        b3p[:4 * m] = layer.coors_mlp[0].bias.detach().float()
        w4p = z(C_PAD)
        w4p[:4 * m] = layer.coors_mlp[3].weight.detach().float()[0]
        out.update(W3=w3p, b3=b3p, W4=w4p, b4=layer.coors_mlp[3].bias.detach().float().contiguous())
    if layer.norm_coors:
        out["coors_scale"] = layer.coors_norm.scale.detach().float().contiguous()
    if layer.node_mlp is not None:
        out.update(W5=layer.node_mlp[0].weight.detach().float().contiguous(),
                   b5=layer.node_mlp[0].bias.detach().float().contiguous(),
                   W6=layer.node_mlp[3].weight.detach().float().contiguous(),
                   b6=layer.node_mlp[3].bias.detach().float().contiguous())
        if layer.norm_feats:
            out.update(gamma=layer.node_norm.weight.detach().float().contiguous(),
                       beta=layer.node_norm.bias.detach().float().contiguous(),
                       ln_eps=float(layer.node_norm.eps))
    return out


def version_key(layer):
    """Cache key: identity + in-place version of every parameter (optimizer steps / load_state_dict bump it)."""
    return tuple((p.data_ptr(), p._version, str(p.device)) for p in layer.parameters())
