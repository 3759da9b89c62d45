"""Re-layout of the reference's parameters for the gfx950 kernels (pure tensor plumbing).

The module keeps the reference's parameters / state_dict untouched; the kernels read
derived, padded copies built here and cached per parameter version:

  edge_mlp.0.weight (H, Din), columns [h_i | h_j | fourier sin, cos | d | e]  (egnn_pytorch.py:282-285)
      -> Wcat (2*Hp, dim): rows [0,H) = W_i, rows [Hp, Hp+H) = W_j   (node-level projection weights)
      -> bcat (2*Hp):      [0,H) = edge_mlp.0.bias                   (folded into P_i)
      -> Ws   (Sp, Hp):    the per-edge scalar columns, transposed
     all three multiplied by -log2(e): the edge kernel evaluates SiLU(x) = -ln2 * y / (1 + 2^y), y = -log2(e) x,
     with v_exp_f32 (= 2^y) and no extra multiply.
  edge_mlp.3.weight (m, H) -> W2h (Hp/32, 2, 64, 8) fp16: (-ln2 * w2_scale * W2) split into hi + lo halves
     (hi = fp16(w), lo = fp16(w - hi): 22 significant bits), in v_mfma_f32_16x16x32_f16 fragment order:
     [step][hi|lo][lane = 16*g + channel][t] = W2[channel, 32*step + 8*g + t].  w2_scale is the power of two that
     brings max|W2| into [1, 2) so hi and lo stay in fp16's normal range; the kernel multiplies by 1/w2_scale.
  coors_mlp.* / edge_gate.* -> zero padded to 16 channels / 64 hidden units
Zero padding is exact: padded hidden units see y = 0 -> 0 / (1 + 1) = 0 and meet zero W2 columns.
"""
from __future__ import annotations

import math

import torch

NEG_LOG2E = -1.4426950408889634
NEG_LN2 = -0.6931471805599453

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


def pack_tiles(x: torch.Tensor) -> torch.Tensor:
    """Row-major (R, Kp) -> the packed tile-major layout of include/egnn_hip.h (R % 32 == 0, Kp % 32 == 0):
    [R/32][Kp/16][32][2][8] with the 16-byte chunk index XOR-swizzled by ((row >> 3) & 1).  Returns a flat tensor."""
    r, kp = x.shape
    assert r % 32 == 0 and kp % 32 == 0
    t = x.reshape(r // 32, 32, kp // 16, 2, 8).permute(0, 2, 1, 3, 4).contiguous()      # (rb, kt, r, ck, e)
    sw = (torch.arange(32, device=x.device) >> 3) & 1
    swapped = t.flip(3)
    out = torch.where(sw.view(1, 1, 32, 1, 1).bool(), swapped, t)
    return out.reshape(-1)


def unpack_tiles(flat: torch.Tensor, rows: int, kp: int) -> torch.Tensor:
    """Inverse of pack_tiles (tests / debugging): flat packed buffer -> row-major (ceil(rows/32)*32, kp)."""
    rp = (rows + 31) // 32 * 32
    t = flat[: rp * kp].reshape(rp // 32, kp // 16, 32, 2, 8)
    sw = (torch.arange(32, device=flat.device) >> 3) & 1
    t = torch.where(sw.view(1, 1, 32, 1, 1).bool(), t.flip(3), t)
    return t.permute(0, 2, 1, 3, 4).reshape(rp, kp)


def split_f16(w: torch.Tensor):
    """(N, K) fp32 weight -> (W_hi, W_lo, inv_scale, w_rows) for egnn_linear_hl_f32: packed fp16 images of scale * W,
    scale = the power of two that brings max|W| into [1, 2) (hi and lo then sit in fp16's normal range),
    hi = fp16(w), lo = fp16(w - hi), zero padded to (ceil(N/256)*256, ceil(K/32)*32) so that both the 128- and the
    256-wide output tiles can read whole tiles."""
    n, k = w.shape
    npad, kpad = (n + 255) // 256 * 256, (k + 31) // 32 * 32
    amax = float(w.abs().max()) if w.numel() else 0.0
    scale = 2.0 ** (-math.floor(math.log2(amax))) if amax > 0 and math.isfinite(amax) else 1.0
    ws = torch.zeros(npad, kpad, dtype=torch.float32, device=w.device)
    ws[:n, :k] = w.float() * scale
    hi = ws.half()
    lo = (ws - hi.float()).half()
    return pack_tiles(hi), pack_tiles(lo), 1.0 / scale, npad


def split_f16_rowmajor(w: torch.Tensor):
    """Row-major variant for the reference kernel egnn_linear_split_f32 (A split on the fly)."""
    n, k = w.shape
    npad, kpad = (n + 127) // 128 * 128, (k + 31) // 32 * 32
    amax = float(w.abs().max()) if w.numel() else 0.0
    scale = 2.0 ** (-math.floor(math.log2(amax))) if amax > 0 and math.isfinite(amax) else 1.0
    ws = torch.zeros(npad, kpad, dtype=torch.float32, device=w.device)
    ws[:n, :k] = w.float() * scale
    hi = ws.half()
    lo = (ws - hi.float()).half()
    return hi.contiguous(), lo.contiguous(), 1.0 / scale


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
    wcat[:h] = w1[:, :dim] * NEG_LOG2E
    wcat[hp:hp + h] = w1[:, dim:2 * dim] * NEG_LOG2E
    bcat = z(2 * hp)
    bcat[:h] = b1 * NEG_LOG2E
    ws = z(sp, hp)
    ws[:s, :h] = w1[:, 2 * dim:].t() * NEG_LOG2E

    w2p = z(M_PAD, hp)
    w2p[:m, :h] = w2 * NEG_LN2
    amax = float(w2p.abs().max())
    w2_scale = 2.0 ** (-math.floor(math.log2(amax))) if amax > 0 and math.isfinite(amax) else 1.0
    w2s = w2p * w2_scale                                   # max |.| in [1, 2)
    w2_hi = w2s.half()
    w2_lo = (w2s - w2_hi.float()).half()
    # (channel, step, g, t) -> (step, g, channel, t) -> (step, lane = 16 g + channel, t)
    frag = lambda t: t.view(M_PAD, hp // 32, 4, 8).permute(1, 2, 0, 3).contiguous().view(hp // 32, 64, 8)
    w2h = torch.stack([frag(w2_hi), frag(w2_lo)], dim=1).contiguous()      # (Hp/32, 2, 64, 8) fp16
    b2p = z(M_PAD)
    b2p[:m] = b2

    out = dict(H=h, Hp=hp, S=s, Sp=sp, Wcat=wcat, Wcat_split=split_f16(wcat), bcat=bcat, Ws=ws, W2h=w2h,
               w2_inv_scale=1.0 / w2_scale, b2=b2p)

    if layer.edge_gate is not None:
        gw = z(M_PAD)
        gw[:m] = layer.edge_gate[0].weight.detach().float()[0]
        out["gate_w"] = gw
        out["gate_b"] = layer.edge_gate[0].bias.detach().float().contiguous()
    if layer.coors_mlp is not None:
        w3 = layer.coors_mlp[0].weight.detach().float()          # (4m, m)
        w3p = z(C_PAD, M_PAD)
        w3p[:4 * m, :m] = w3
        a3 = float(w3p.abs().max())
        w3_scale = 2.0 ** (-math.floor(math.log2(a3))) if a3 > 0 and math.isfinite(a3) else 1.0
        w3s = w3p * w3_scale
        w3_hi = w3s.half()
        w3h = torch.stack([w3_hi, (w3s - w3_hi.float()).half()]).contiguous()        # (2, 64, 16) fp16: hi | lo
        b3p = z(C_PAD)
        b3p[:4 * m] = layer.coors_mlp[0].bias.detach().float()
        w4p = z(C_PAD)
        w4p[:4 * m] = layer.coors_mlp[3].weight.detach().float()[0]
        out.update(W3=w3p, W3h=w3h, w3_inv_scale=1.0 / w3_scale, b3=b3p, W4=w4p,
                   b4=layer.coors_mlp[3].bias.detach().float().contiguous())
    if layer.norm_coors:
        out["coors_scale"] = layer.coors_norm.scale.detach().float().contiguous()
    if layer.node_mlp is not None:
        out.update(W5=layer.node_mlp[0].weight.detach().float().contiguous(),
                   b5=layer.node_mlp[0].bias.detach().float().contiguous(),
                   W6=layer.node_mlp[3].weight.detach().float().contiguous(),
                   b6=layer.node_mlp[3].bias.detach().float().contiguous())
        out["W5_split"] = split_f16(out["W5"])
        out["W6_split"] = split_f16(out["W6"])
        if layer.norm_feats:
            out.update(gamma=layer.node_norm.weight.detach().float().contiguous(),
                       beta=layer.node_norm.bias.detach().float().contiguous(),
                       ln_eps=float(layer.node_norm.eps))
    return out


def version_key(layer):
    """Cache key: identity + in-place version of every parameter (optimizer steps / load_state_dict bump it)."""
    return tuple((p.data_ptr(), p._version, str(p.device)) for p in layer.parameters())
