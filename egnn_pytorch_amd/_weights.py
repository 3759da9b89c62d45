"""Re-layout of the reference's parameters for the gfx950 kernels (pure tensor plumbing).

The module keeps the reference's parameters / state_dict untouched; the kernels read
derived, padded copies built here and cached per parameter version:

  edge_mlp.0.weight (H, Din), columns [h_i | h_j | fourier sin, cos | d | e]  (egnn_pytorch.py:282-285)
      -> Wcat (2*Hp, dim): rows [0,H) = W_i, rows [Hp, Hp+H) = W_j   (node-level projection weights)
      -> bcat (2*Hp):      [0,H) = edge_mlp.0.bias                   (folded into P_i)
      -> Wst  (Hp, 4 NM, 2) fp16: the per-edge scalar columns as A fragments of the first-layer MFMA (see
                           `scalar_table`)
     all three multiplied by -log2(e): the edge kernel evaluates SiLU(x) = -ln2 * y / (1 + 2^y), y = -log2(e) x,
     with v_exp_f32 (= 2^y) and no extra multiply.
  edge_mlp.3.weight (m, H) -> W2h (Hp/32, NB, 2, 64, 8) fp16: (-ln2 * w2_scale * W2) split into hi + lo halves
     (hi = fp16(w), lo = fp16(w - hi): 22 significant bits), in v_mfma_f32_16x16x32_f16 fragment order, NB = m_blocks(m)
     blocks of 16 channels:
     [step][nb][hi|lo][lane = 16*g + c][t] = W2[16*nb + c, 32*step + hidden_slot(g, t)], hidden_slot(g, t) =
     4*g + t (t < 4), 16 + 4*g + (t - 4) (t >= 4) -- the order in which the first-layer MFMA leaves the hidden
     units in a lane.  w2_scale is the power of two that brings max|W2| into [1, 2) so hi and lo stay in fp16's
     normal range; the kernel multiplies by 1/w2_scale.
  coors_mlp.* / edge_gate.* -> zero padded to 16 NB channels / 64 NB hidden units
Zero padding is exact: padded hidden units see y = 0 -> 0 / (1 + 1) = 0 and meet zero W2 columns.
"""
from __future__ import annotations

import math
import os

import torch

NEG_LOG2E = -1.4426950408889634
NEG_LN2 = -0.6931471805599453

S_MAX = 16                              # per-edge scalar inputs the edge kernel is instantiated for
SCALAR_SHIFT = 1024.0                   # 2^10: the coarse part of a per-edge scalar is carried as fp16(s / 2^10)
M_MAX = 64                              # largest m_dim the edge kernel is instantiated for


def m_blocks(m: int) -> int:
    """16-channel accumulator tiles per edge tile (the kernel's NB): 1 for m_dim <= 16, 2 up to 32, 4 up to 64."""
    if m > M_MAX:
        raise NotImplementedError(f"m_dim={m} > {M_MAX} is not supported by the gfx950 edge kernel")
    return 1 if m <= 16 else (2 if m <= 32 else 4)


def padded_hidden(h: int) -> int:
    return (h + 31) // 32 * 32


def check_scalars(s: int) -> int:
    if s > S_MAX:
        raise NotImplementedError(
            f"{s} per-edge scalar inputs (2*fourier_features + 1 + edge_dim) exceed the {S_MAX} "
            f"the gfx950 edge kernel is built for")
    return s


def pow2_scale(amax: float) -> float:
    """The power of two that brings amax into [1, 2) (1 for 0 / non-finite)."""
    return 2.0 ** (-math.floor(math.log2(amax))) if amax > 0 and math.isfinite(amax) else 1.0


def edge_mfmas(s: int) -> int:
    """Mirror of egnn_edge_mfmas (include/egnn_hip.h): first-layer MFMAs the edge kernel chains for s scalars."""
    return 1 if s <= 1 else (3 if s <= 4 else (4 if s <= 5 else (6 if s <= 8 else 12)))


def scalar_table(ws: torch.Tensor, amax=None):
    """(S, Hp) fp32 scalar weights -> (Wst, ws_inv_scale).  Wst is (Hp, 4 * edge_mfmas(S), 2) fp16 (stored through `swizzle_terms`): per hidden unit h
    the (fp16, fp16) words that sit in K-slots 4g+2, 4g+3 of lane group g of first-layer MFMA m
    (v_mfma_f32_16x16x16_f16; K-slots 4g, 4g+1 belong to the (hi, lo) pair of P_i), term index 4 m + g = 3 s + kind:
        kind 0: (hi, lo) of c * W * 2^10      x  B = (s1, s1),      s1 = fp16(s' / 2^10)
        kind 1: (hi, lo) of c * W             x  B = (r_hi, r_hi),  r = s' - 2^10 s1
        kind 2: (hi, 0)  of c * W             x  B = (r_lo, 0)
    with s' = s / c the per-edge scalar and c = ws_scale the power of two that brings max|W| into [1, 2).  The five
    products add up to W * s with ~2^-22 relative error for |s'| < 6e7 (dist^2: |rel| < ~7000 length units)."""
    s_, hp = ws.shape
    c = pow2_scale((float(ws.abs().max()) if ws.numel() else 0.0) if amax is None else amax)       # (amax: max |ws| if the caller has it)
    w = (ws * c).t().contiguous()                                    # (Hp, S)
    wa = w * SCALAR_SHIFT
    hi, ahi = w.half(), wa.half()
    lo, alo = (w - hi.float()).half(), (wa - ahi.float()).half()
    zero = torch.zeros_like(hi)
    terms = torch.stack([torch.stack([ahi, alo], -1), torch.stack([hi, lo], -1), torch.stack([hi, zero], -1)], dim=2)
    tab = torch.zeros(hp, 4 * edge_mfmas(s_), 2, dtype=torch.float16, device=ws.device)
    tab[:, :3 * s_] = terms.reshape(hp, 3 * s_, 2)                   # (Hp, S, 3, 2) -> term index 3 s + kind
    return swizzle_terms(tab).contiguous(), 1.0 / c


def swizzle_terms(tab: torch.Tensor) -> torch.Tensor:
    """The table as the kernels read it: units 8 .. 15 of every 16-block keep the term pairs (0, 1) and (2, 3) of each four-term group
    swapped -- lane (unit e, group g) reads position g ^ 2 there, so that units e and e + 8 (32 dwords apart) do not meet in one LDS bank
    (round 5: SQ_LDS_BANK_CONFLICT of the edge pass 16.9 M -> 0).  An involution: applying it twice gives the plain table back."""
    hp, nt, _ = tab.shape
    t = tab.reshape(hp, nt // 4, 2, 2, 2)
    upper = ((torch.arange(hp, device=tab.device) & 8) != 0).view(hp, 1, 1, 1, 1)
    return torch.where(upper, t.flip(2), t).reshape(hp, nt, 2)


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


def node_mlp_fused_supported(dim: int, m: int) -> bool:
    """Mirror of egnn_node_mlp_fused_halves(dim, m_dim) > 0 (csrc/node_mlp_fused.hip: the widths the one-launch node_mlp is built for)."""
    return m == 16 and dim in (32, 64, 128, 256)


def node_mlp_fused_image(w5_split, w6_split, dim: int, m: int) -> torch.Tensor:
    """Tensor-op twin of egnn_node_mlp_fused_pack_f16 (the specification the kernel is tested against, and what a CPU module packs):
    per block hb of 32 hidden units  [W5 fragments (ht, ks, hi|lo, lane, 8)] [W6 fragments (dt, hi|lo, lane, 8)]  of
    v_mfma_f32_16x16x32_f16 A operands -- lane = 16 kq + r holds row r of its 16-row tile and K-slots 8 kq .. 8 kq + 7; W5's rows are
    hidden units 32 hb + 16 ht + r with k = 32 ks + slot; W6's rows are output features 16 dt + r and slot s carries hidden unit
    32 hb + pi(s), pi(8 q + t) = 4 q + t (t < 4), 16 + 4 q + (t - 4) -- the order in which the first product's accumulators hold them."""
    assert node_mlp_fused_supported(dim, m)
    ndt = nhb = dim // 16
    kp1 = (dim + m + 31) // 32 * 32
    k1s = kp1 // 32
    pi = torch.tensor([4 * (s_ >> 3) + (s_ & 7) if (s_ & 7) < 4 else 16 + 4 * (s_ >> 3) + (s_ & 7) - 4 for s_ in range(32)],
                      device=w5_split[0].device)
    f5, f6 = [], []
    for part in (0, 1):
        w5 = unpack_tiles(w5_split[part], w5_split[3], kp1)[:2 * dim]                    # (2 dim, Kp1) fp16, scaled
        w6 = unpack_tiles(w6_split[part], w6_split[3], 2 * dim)[:dim]                    # (dim, 2 dim)
        f5.append(w5.reshape(nhb, 2, 16, k1s, 4, 8).permute(0, 1, 3, 4, 2, 5).reshape(nhb, 2, k1s, 64, 8))      # (hb, ht, ks, kq r, e)
        g = w6.reshape(ndt, 16, nhb, 32)[..., pi].reshape(ndt, 16, nhb, 4, 8)            # (dt, r, hb, kq, e)
        f6.append(g.permute(2, 0, 3, 1, 4).reshape(nhb, ndt, 64, 8))
    f5 = torch.stack(f5, dim=3).reshape(nhb, -1)                                          # (hb, ht, ks, part, lane, e)
    f6 = torch.stack(f6, dim=2).reshape(nhb, -1)                                          # (hb, dt, part, lane, e)
    return torch.cat((f5, f6), dim=1).reshape(-1).contiguous()


def _f32_mul(a: float, b: float) -> float:
    """a * b rounded to fp32 (what max |w * b| is for fp32 tensors when a = max |w|: rounding is monotone)."""
    import numpy as np
    return float(np.float32(a) * np.float32(abs(b)))


def split_f16_device(pieces, n_rows: int, k: int, amax: float, factor: float = 1.0):
    """split_f16 of the (n_rows, k) matrix made of `pieces` = [(X, first image row, transposed)], X a 2-D fp32 view with unit column
    stride holding the rows from `first image row` on (or, transposed, X^T does), everything else zero, times `factor`: the same
    (W_hi, W_lo, inv_scale, w_rows) bit for bit -- one memset and one egnn_split_scaled_f16 launch per image / piece instead of ~22
    small tensor ops and a host read.  amax = max |X| over the pieces (the caller reads all of a layer's maxima in ONE transfer)."""
    from . import _abi, _ops
    dev = pieces[0][0].device
    npad, kpad = (n_rows + 255) // 256 * 256, (k + 31) // 32 * 32
    scale = pow2_scale(_f32_mul(amax, factor))
    hi = torch.zeros(npad * kpad, dtype=torch.float16, device=dev)
    lo = torch.zeros(npad * kpad, dtype=torch.float16, device=dev)
    lib = _abi.load()
    for x, row0, transposed in pieces:
        assert x.dtype == torch.float32 and x.stride(1) == 1 and row0 % 32 == 0
        off = row0 * kpad * 2                                        # bytes: whole 32-row blocks are contiguous in the packed layout
        rc = lib.egnn_split_scaled_f16(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], float(scale * factor), int(transposed),
                                       hi.data_ptr() + off, lo.data_ptr() + off, kpad, None, _ops._stream())
        _abi.check(rc, "egnn_split_scaled_f16")
    return hi, lo, 1.0 / scale, npad


def pack(layer) -> dict:
    """Build the kernel-side weight set of one EGNN layer.  All outputs are fp32, contiguous, on the
    parameters' device.  On the GPU the seven GEMM weight images come from split_f16_device and every maximum the scales are
    chosen from is read in one transfer (an optimizer step invalidates the cache: this runs once per layer and training step)."""
    dev0 = layer.edge_mlp[0].weight.device
    if dev0.type == "cuda":
        with torch.cuda.device(dev0):                             # (the packing kernels launch on the parameters' device)
            return _pack(layer)
    return _pack(layer)


def _pack(layer) -> dict:
    w1 = layer.edge_mlp[0].weight.detach().float()
    b1 = layer.edge_mlp[0].bias.detach().float()
    w2 = layer.edge_mlp[3].weight.detach().float()
    b2 = layer.edge_mlp[3].bias.detach().float()
    dev = w1.device
    dim, m = layer.dim, layer.m_dim
    h, din = w1.shape
    s = din - 2 * dim
    nb = m_blocks(m)
    M_PAD, C_PAD = 16 * nb, 64 * nb
    hp = padded_hidden(h)
    check_scalars(s)
    z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
    on_dev = w1.is_cuda
    am = None
    if on_dev:
        # every maximum the power-of-two scales below are chosen from, in ONE host read
        zero = torch.zeros((), dtype=torch.float32, device=dev)
        mx = lambda t: t.abs().max() if t.numel() else zero
        tens = [mx(w1[:, :dim]), mx(w1[:, dim:2 * dim]), mx(w1[:, 2 * dim:]), mx(w2)]
        tens.append(mx(layer.coors_mlp[0].weight.detach().float()) if layer.coors_mlp is not None else zero)
        if layer.node_mlp is not None:
            tens += [mx(layer.node_mlp[0].weight.detach().float()), mx(layer.node_mlp[3].weight.detach().float())]
        else:
            tens += [zero, zero]
        am = dict(zip(("wi", "wj", "ws", "w2", "w3", "w5", "w6"), torch.stack(tens).tolist()))

    wcat = z(2 * hp, dim)
    wcat[:h] = w1[:, :dim] * NEG_LOG2E
    wcat[hp:hp + h] = w1[:, dim:2 * dim] * NEG_LOG2E
    bcat = z(2 * hp)
    bcat[:h] = b1 * NEG_LOG2E
    ws = z(s, hp)
    ws[:, :h] = w1[:, 2 * dim:].t() * NEG_LOG2E
    wst, ws_inv_scale = scalar_table(ws, None if am is None else _f32_mul(am["ws"], NEG_LOG2E))

    w2p = z(M_PAD, hp)
    w2p[:m, :h] = w2 * NEG_LN2
    amax = float(w2p.abs().max()) if am is None else _f32_mul(am["w2"], NEG_LN2)
    w2_scale = 2.0 ** (-math.floor(math.log2(amax))) if amax > 0 and math.isfinite(amax) else 1.0
    w2s = w2p * w2_scale                                   # max |.| in [1, 2)
    w2_hi = w2s.half()
    w2_lo = (w2s - w2_hi.float()).half()
    # channel = 16 nb + c, h = 32 step + 16 hb + 4 g + r:
    # (nb, c, step, hb, g, r) -> (step, nb, g, c, hb, r) -> (step, nb, lane = 16 g + c, t = 4 hb + r)
    frag = lambda t: t.view(nb, 16, hp // 32, 2, 4, 4).permute(2, 0, 4, 1, 3, 5).contiguous().view(hp // 32, nb, 64, 8)
    w2h = torch.stack([frag(w2_hi), frag(w2_lo)], dim=2).contiguous()      # (Hp/32, NB, 2, 64, 8) fp16
    b2p = z(M_PAD)
    b2p[:m] = b2

    w_i, w_j = w1[:, :dim], w1[:, dim:2 * dim]
    if on_dev:
        wcat_split = split_f16_device([(w_i, 0, False), (w_j, hp, False)], 2 * hp, dim, max(am["wi"], am["wj"]), NEG_LOG2E)
    else:
        wcat_split = split_f16(wcat)
    out = dict(H=h, Hp=hp, S=s, Wcat=wcat, Wcat_split=wcat_split, bcat=bcat, Ws=ws, Wst=wst,
               ws_inv_scale=ws_inv_scale, W2h=w2h, w2_inv_scale=1.0 / w2_scale, b2=b2p)
    # backward, d/d feats = dP_i W_i + dP_j W_j (natural units): the W operands of that GEMM are the transposes, (dim, Hp)
    if on_dev:
        out["WiT_split"] = split_f16_device([(w_i, 0, True)], dim, hp, am["wi"])
        out["WjT_split"] = split_f16_device([(w_j, 0, True)], dim, hp, am["wj"])
    else:
        wit, wjt = z(dim, hp), z(dim, hp)
        wit[:, :h] = w_i.t()
        wjt[:, :h] = w_j.t()
        out["WiT_split"] = split_f16(wit)
        out["WjT_split"] = split_f16(wjt)
    if True:
        # backward (egnn_edge_bwd_pass_f32): W2^T in natural units as A fragments of v_mfma_f32_16x16x16_f16,
        # [step][hb][hi|lo][lane = 16 g + r][u] = W2[4 g + u][32 step + 16 hb + r] -- one image per block of 16 message channels
        # (m_dim > 16: the pass is linear in gU, so it runs once per block and the results are added; "W2Th" = block 0)
        w2t = z(hp, 16 * nb)
        w2t[:h, :m] = w2.t()
        t_scale = pow2_scale((float(w2t.abs().max()) if w2t.numel() else 0.0) if am is None else am["w2"])
        fragt = lambda t: t.view(hp // 32, 2, 16, 4, 4).permute(0, 1, 3, 2, 4).contiguous().view(hp // 32, 2, 64, 4)
        blocks = []
        for blk in range(nb):
            w2ts = w2t[:, 16 * blk:16 * blk + 16].contiguous() * t_scale
            t_hi = w2ts.half()
            t_lo = (w2ts - t_hi.float()).half()
            blocks.append(torch.stack([fragt(t_hi), fragt(t_lo)], dim=2).contiguous())      # (Hp/32, 2, 2, 64, 4) fp16
        out["W2Th"] = blocks[0]
        out["W2Th_blocks"] = blocks
        out["w2t_scale"] = t_scale
        if s > 5:
            # backward with more than five per-edge scalars (egnn_edge_bwd_pass_f32, DSM): W_s^T in natural units as A fragments,
            # [step][hb][hi|lo][lane = 16 g + s][u] = W_s[32 step + 16 hb + 4 g + u][s]
            wsn = z(hp, 16)
            wsn[:h, :s] = w1[:, 2 * dim:]
            s_scale = pow2_scale(float(wsn.abs().max()) if am is None else am["ws"])
            wss = wsn * s_scale
            s_hi = wss.half()
            s_lo = (wss - s_hi.float()).half()
            # (A operand: lane (g, m = s) holds hidden units 4 g .. 4 g + 3 of the 16-block -- W2Th above is the B-operand form)
            frags = lambda t: t.view(hp // 32, 2, 4, 4, 16).permute(0, 1, 2, 4, 3).contiguous().view(hp // 32, 2, 64, 4)
            out["WsTh"] = torch.stack([frags(s_hi), frags(s_lo)], dim=2).contiguous()
            out["wst_scale"] = s_scale

    if layer.edge_gate is not None:
        gw = z(M_PAD)
        gw[:m] = layer.edge_gate[0].weight.detach().float()[0]
        out["gate_w"] = gw
        out["gate_b"] = layer.edge_gate[0].bias.detach().float().contiguous()
    if layer.coors_mlp is not None:
        w3 = layer.coors_mlp[0].weight.detach().float()          # (4m, m)
        w3p = z(C_PAD, M_PAD)
        w3p[:4 * m, :m] = w3
        a3 = float(w3p.abs().max()) if am is None else am["w3"]
        w3_scale = 2.0 ** (-math.floor(math.log2(a3))) if a3 > 0 and math.isfinite(a3) else 1.0
        w3s = w3p * w3_scale
        w3_hi = w3s.half()
        w3h = torch.stack([w3_hi, (w3s - w3_hi.float()).half()]).contiguous()        # (2, 64 NB, 16 NB) fp16: hi | lo
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
        w5, w6 = out["W5"], out["W6"]
        # (backward: d/d (node_mlp input) = g W5, d/d (hidden) = g W6 -- the W operands are the transposes)
        if on_dev:
            out["W5_split"] = split_f16_device([(w5, 0, False)], w5.shape[0], w5.shape[1], am["w5"])
            out["W6_split"] = split_f16_device([(w6, 0, False)], w6.shape[0], w6.shape[1], am["w6"])
            out["W5T_split"] = split_f16_device([(w5, 0, True)], w5.shape[1], w5.shape[0], am["w5"])
            out["W6T_split"] = split_f16_device([(w6, 0, True)], w6.shape[1], w6.shape[0], am["w6"])
        else:
            out["W5_split"] = split_f16(w5)
            out["W6_split"] = split_f16(w6)
            out["W5T_split"] = split_f16(w5.t().contiguous())
            out["W6T_split"] = split_f16(w6.t().contiguous())
        if node_mlp_fused_supported(dim, m):                       # narrow layers: node_mlp in one launch (csrc/node_mlp_fused.hip)
            if on_dev:
                from . import _ops
                out["nmf_img"] = _ops.node_mlp_fused_image(out["W5_split"], out["W6_split"], dim, m)
            else:
                out["nmf_img"] = node_mlp_fused_image(out["W5_split"], out["W6_split"], dim, m)
        if layer.norm_feats:
            out.update(gamma=layer.node_norm.weight.detach().float().contiguous(),
                       beta=layer.node_norm.bias.detach().float().contiguous(),
                       ln_eps=float(layer.node_norm.eps))
    return out


def version_key(layer):
    """Cache key: identity + in-place version of every parameter (optimizer steps / load_state_dict bump it).  The parameter list itself
    is cached on the layer (walking nn.Module.parameters() costs 50 us per forward); it is rebuilt when the registered Parameter objects
    change (re-assignment; `_apply` with overwrite_module_params_on_conversion)."""
    cached = layer.__dict__.get("_param_cache")
    mods = layer.__dict__.get("_param_cache_mods")
    tree = layer.__dict__.get("_param_cache_tree")
    # the cache is valid while (a) every registered child of every module of the tree is still the object it was (a replaced submodule --
    # `layer.node_norm = nn.LayerNorm(..)`, `edge_mlp[0] = new_linear`, an adapter swap -- keeps the OLD module's _parameters unchanged, so
    # (b) alone would go on serving the stale packed tables) and no child was added or removed, and (b) every Parameter is the
    # registered one
    if (cached is None
            or any(len(c._modules) != cnt for c, cnt, _ in tree)
            or any(c._modules.get(n) is not ch for c, _, kids in tree for n, ch in kids)
            or any(len(m._parameters) != cnt for m, cnt in layer.__dict__["_param_cache_counts"])
            or any(m._parameters.get(n) is not p for (m, n), p in zip(mods, cached))):
        mods, cached, tree, counts = [], [], [], []
        for m in layer.modules():
            tree.append((m, len(m._modules), tuple(m._modules.items())))
            counts.append((m, len(m._parameters)))
            for n, p in m._parameters.items():
                if p is not None:
                    mods.append((m, n))
                    cached.append(p)
        layer.__dict__["_param_cache"], layer.__dict__["_param_cache_mods"] = cached, mods
        layer.__dict__["_param_cache_tree"], layer.__dict__["_param_cache_counts"] = tree, counts
    return tuple((p.data_ptr(), p._version, p.device) for p in cached)
