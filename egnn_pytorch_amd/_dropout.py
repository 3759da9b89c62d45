"""Training-mode dropout of the gfx950 layer: the torch twin of the kernels' counter-based mask (csrc/egnn_common.h).

The reference has ONE `nn.Dropout` shared by its three MLPs, each time behind the first Linear (egnn_pytorch.py:176, 178-184,
196-208).  The fused edge pass never materialises the E x H pre-activation of edge_mlp, so it cannot read a mask tensor either:
element (row, col) of a site is kept iff a hash of (seed, site, row, col) is >= p * 2^32.  Everything that has to agree with the
kernels -- the backward's recompute, the tests -- evaluates the same function here, with int64 tensors standing in for uint32
registers (products wrap modulo 2^64, their low 32 bits are what the kernels compute).

Like the reference's dropout it is a Bernoulli(1 - p) mask with 1 / (1 - p) rescaling, drawn anew for every forward call (the seed
comes from torch's default CPU generator, so `torch.manual_seed` makes a run reproducible); it is NOT the same random stream as
`nn.Dropout`'s Philox generator -- no two dropout implementations share one."""
from __future__ import annotations

import torch

SITE_EDGE, SITE_COORS, SITE_NODE = 0, 1, 2
_M32 = 0xFFFFFFFF
_A, _S, _C = 0x9E3779B1, 0x27D4EB2F, 0x85EBCA77
_M1, _M2 = 0x2C1B3C6D, 0x297A2D39


def threshold(p: float) -> int:
    """drop_thr of the C ABI: keep <=> hash >= round(p * 2^32) (clamped into [1, 2^32 - 1] for p in (0, 1))."""
    return max(1, min(_M32, int(round(p * 4294967296.0))))


def inv_keep(p: float) -> float:
    """1 / (1 - p), the scale of a kept unit.  p = 1 (nn.Dropout(1.0) is legal upstream and outputs zeros): every unit is dropped
    -- threshold() keeps one with probability 2^-32 -- and the scale of that unit stays finite (1) instead of dividing by zero."""
    return 1.0 / (1.0 - p) if p < 1.0 else 1.0


def draw_seed() -> int:
    """A fresh 31-bit seed from torch's default CPU generator (no device synchronisation)."""
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


def hash32(seed: int, site: int, rows: torch.Tensor, cols: torch.Tensor) -> torch.Tensor:
    """egnn_drop_hash(egnn_drop_base(seed, site, row), col) for every (row, col): rows (R,) and cols (C,) integer tensors ->
    (R, C) int64 holding the uint32 hash."""
    r = rows.to(torch.int64)[:, None]
    c = cols.to(torch.int64)[None, :]
    x = (r * _A + (seed + site * _S) + c * _C) & _M32
    x = x ^ (x >> 15)
    x = (x * _M1) & _M32
    x = x ^ (x >> 12)
    x = (x * _M2) & _M32
    x = x ^ (x >> 15)
    return x


def keep_mask(seed: int, site: int, rows: torch.Tensor, cols: torch.Tensor, p: float) -> torch.Tensor:
    return hash32(seed, site, rows, cols) >= threshold(p)


def apply(z: torch.Tensor, seed: int, site: int, rows: torch.Tensor, p: float) -> torch.Tensor:
    """dropout of z (..., C) whose leading dimensions flatten to `rows` (R,): z * keep / (1 - p)."""
    c = z.shape[-1]
    keep = keep_mask(seed, site, rows.reshape(-1), torch.arange(c, device=z.device), p).view(z.shape)
    return torch.where(keep, z * inv_keep(p), torch.zeros_like(z))
