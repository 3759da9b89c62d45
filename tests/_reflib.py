"""ctypes binding of tests/libegnn_hip_ref.so (include/egnn_hip_ref.h): the TEST-ONLY reference kernel -- the GEMM with A split
on the fly -- plus test-side wrappers of the product library's plain-fp32 kernels (exact fp32 GEMM, fp32 node_norm + concat:
the wide-range path, which doubles as the A/B reference of the split-f16 kernels).  Nothing under egnn_pytorch_amd/ imports this."""
import ctypes
import math
import os
from ctypes import c_float, c_int, c_int64, c_void_p

import torch

from egnn_pytorch_amd import _abi
from egnn_pytorch_amd._ops import _ptr, _stream, _timed, empty

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libegnn_hip_ref.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        _abi.load()                                        # the HIP runtime torch initialised, then ours
        lib = ctypes.CDLL(_PATH)
        lib.egnn_linear_split_f32.restype = c_int
        lib.egnn_linear_split_f32.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p,
                                              c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]
        _lib = lib
    return _lib


def split_f16_rowmajor(w):
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


def linear(a, w, bias=None, residual=None, act=0, name="linear"):
    """act(a @ w.T + bias) (+ residual) -- egnn_linear_f32.  a: (M,K) fp32 contiguous; w: (N,K)."""
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k and a.is_contiguous() and w.is_contiguous()
    c = empty(m, n, dtype=torch.float32, device=a.device)
    ldr = 0
    if residual is not None:
        assert residual.shape == (m, n) and residual.is_contiguous()
        ldr = n
    with _timed(name):
        rc = _abi.load().egnn_linear_f32(_ptr(a), k, _ptr(w), k, _ptr(bias), _ptr(residual), ldr,
                                         _ptr(c), n, m, n, k, act, _stream())
    _abi.check(rc, "egnn_linear_f32")
    return c


def linear_split(a, wsplit, n, bias=None, residual=None, act=0, name="linear"):
    """act(a @ W.T + bias) (+ residual) on the matrix cores -- egnn_linear_split_f32.
    `wsplit` = (W_hi, W_lo, inv_scale) from split_f16_rowmajor; n = true number of output columns."""
    whi, wlo, inv = wsplit
    m, k = a.shape
    assert a.is_contiguous() and whi.shape == wlo.shape and whi.shape[0] >= n and whi.shape[1] >= k
    c = empty(m, n, dtype=torch.float32, device=a.device)
    ldr = 0
    if residual is not None:
        assert residual.shape == (m, n) and residual.is_contiguous()
        ldr = n
    with _timed(name):
        rc = load().egnn_linear_split_f32(_ptr(a), k, _ptr(whi), _ptr(wlo), whi.shape[1], float(inv), _ptr(bias),
                                               _ptr(residual), ldr, _ptr(c), n, m, n, k, act, _stream())
    _abi.check(rc, "egnn_linear_split_f32")
    return c


def node_prep(feats2d, m_i, gamma, beta, eps, m_dim):
    rows, dim = feats2d.shape
    out = empty(rows, dim + m_dim, dtype=torch.float32, device=feats2d.device)
    with _timed("node_prep"):
        rc = _abi.load().egnn_node_prep_f32(_ptr(feats2d), _ptr(m_i), _ptr(gamma), _ptr(beta), float(eps),
                                            _ptr(out), rows, dim, m_dim, _stream())
    _abi.check(rc, "egnn_node_prep_f32")
    return out


