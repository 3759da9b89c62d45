"""Thin wrappers: torch tensors (device memory + stream plumbing) -> C-ABI calls of libegnn_hip.so.

Nothing here computes: each function validates, allocates outputs with torch (caching allocator) and
enqueues one HIP kernel on torch's current stream.
"""
from __future__ import annotations

import contextlib
import contextvars
import threading
import time
import os
from ctypes import byref

import torch

from . import _abi


class PhaseTimer:
    """Optional per-kernel timing with events on the launch stream (used by bench.py / profiling).

    with phase_timer() as t: layer(...)   ->  t.summary() = {kernel: [ms, ...]}
    """

    def __init__(self):
        self.records = []

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1 in self.records:
            out.setdefault(name, []).append(e0.elapsed_time(e1))
        return out


_timer = None


@contextlib.contextmanager
def phase_timer():
    global _timer
    prev, _timer = _timer, PhaseTimer()
    try:
        yield _timer
    finally:
        _timer = prev


@contextlib.contextmanager
def _timed(name):
    if _timer is None:
        yield
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    yield
    e1.record()
    _timer.records.append((name, e0, e1))


_POISON = os.environ.get("EGNN_POISON_ALLOC", "0") == "1"


def empty(*shape, dtype, device):
    """torch.empty for kernel outputs / workspaces.  EGNN_POISON_ALLOC=1 (test aid) fills them with NaN / 0x7f bytes
    first: an element a kernel forgets to write can then not hide behind stale-but-correct data that the caching
    allocator hands back from the previous call."""
    t = torch.empty(*shape, dtype=dtype, device=device)
    if _POISON:
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        else:
            t.view(torch.uint8).fill_(0x7F)
    return t


_RAW_STREAM = os.environ.get("EGNN_RAW_STREAM", "1") != "0"


def _stream():
    """The raw hipStream_t of torch's current stream on the current device.  Through the C binding when it exists: the public
    torch.cuda.current_stream() builds a Stream object through three layers of device-index helpers (8 us a call, thirteen calls per
    forward: a third of the host time of a forward, tools/host_overhead_probe.py)."""
    if _RAW_STREAM:
        try:
            return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
        except AttributeError:
            pass
    return torch.cuda.current_stream().cuda_stream


_side = {}


def side_stream(device):
    """One extra HIP stream per device for work that is independent of the main launch sequence (layer.py)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    st = _side.get(key)
    if st is None:
        st = _side[key] = torch.cuda.Stream(device=torch.device("cuda", key))
    return st


_side_handles = {}


def side_handles(device):
    """(raw hipStream_t of the side stream, raw hipEvent_t fork, raw hipEvent_t join) for the C whole-layer entry (egnn_forward_opts);
    one pair of events per device and thread, kept alive here (an event may be re-recorded once the call that used it is enqueued)."""
    import threading
    key = (device.index, threading.get_ident())
    h = _side_handles.get(key)
    if h is None:
        side = side_stream(device)
        evs = (torch.cuda.Event(), torch.cuda.Event())
        for ev in evs:
            ev.record(side)                                   # (torch creates the hipEvent_t at the first record)
        h = _side_handles[key] = (side.cuda_stream, evs[0].cuda_event, evs[1].cuda_event, side, evs)
    return h[:3]


_fork_events = {}


def fork_event(device):
    """An event recorded NOW on the current stream of `device` (one cached event object per device and thread: a later record
    replaces the earlier one, and whoever waits for it does so right behind the record -- layer.py::_forward_hip_impl)."""
    import threading
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), threading.get_ident())
    ev = _fork_events.get(key)
    if ev is None:
        ev = _fork_events[key] = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(torch.device("cuda", key[0])))
    return ev


def _ptr(t):
    return None if t is None else t.data_ptr()


def _u8(t):
    """bool tensor -> same storage viewed as bytes."""
    if t is None:
        return None
    if t.dtype != torch.bool:
        raise TypeError(f"mask / adj_mat must be torch.bool (got {t.dtype})")
    return t.contiguous().view(torch.uint8)


# ---------------------------------------------------------------------------------------------- numerical-range status
# One int32 status word per device (include/egnn_hip.h: EGNN_RANGE_*).  Kernels OR bits into it when a finite value leaves
# what the split-fp16 arithmetic carries (the value turns into inf / NaN).  When it is read back is the caller's choice:
#   EGNN_RANGE_CHECK=sync      (default) every forward ends with one 4-byte device->host read and raises EGNNRangeError
#   EGNN_RANGE_CHECK=deferred  no synchronisation: the word is copied to pinned memory after each forward and examined at
#                              the start of the next one (or by egnn_pytorch_amd.check_range()); bench.py uses this
#   EGNN_RANGE_CHECK=off       never read (outputs are still non-finite when it happens)
RANGE_CHECK = os.environ.get("EGNN_RANGE_CHECK", "sync")
_SPIN = os.environ.get("EGNN_RANGE_SPIN", "1") != "0"
_SPIN_SECONDS = float(os.environ.get("EGNN_RANGE_SPIN_SECONDS", "2.0"))   # spinning on the pinned word, then the copy + synchronise fallback
_status = {}


class _Status:
    """Two words per device: [0] is OR-ed by the kernels of a forward, [1] by the kernels of a backward (`backward_status()`), so that
    bits a backward leaves behind are reported as what they are instead of being attributed to -- and re-run as -- the next forward."""

    def __init__(self, device):
        self.dev = torch.zeros(2, dtype=torch.int32, device=device)
        self.host = torch.zeros(2, dtype=torch.int32).pin_memory()
        self.event = None
        # sync mode: the words + a sequence number written by a one-thread kernel into pinned memory the host spins on (egnn_status_publish)
        self.pub = torch.zeros(4, dtype=torch.int32).pin_memory()
        self.pub_np = self.pub.numpy()
        self.seq = 0
        self.lock = threading.Lock()


_status_slot = contextvars.ContextVar("egnn_status_slot", default=0)


@contextlib.contextmanager
def backward_status():
    """Kernels launched inside (the native backward) report range problems in the backward's status word."""
    tok = _status_slot.set(1)
    try:
        yield
    finally:
        _status_slot.reset(tok)


def _status_ptr(device):
    return status_word(device).dev.data_ptr() + 4 * _status_slot.get()


def status_word(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    st = _status.get(key)
    if st is None:
        st = _status[key] = _Status(torch.device("cuda", key))
    return st


def _raise_range(bits, when, origin="call"):
    what = "; ".join(msg for bit, msg in _abi.RANGE_BITS.items() if bits & bit)
    err = _abi.EGNNRangeError(
        f"egnn_pytorch_amd ({when}): a value left the range of the split-fp16 arithmetic of the gfx950 path: {what}. "
        f"The outputs of that call are non-finite. (The reference computes in plain fp32 and has no such limit; rescale the "
        f"inputs / weights, or see DESIGN.md section 2.)")
    err.origin, err.bits = origin, int(bits)
    raise err


def range_check_after_forward(device, mode=None):
    """Called by the modules when a forward has been enqueued."""
    mode = mode or RANGE_CHECK
    if mode == "off" or torch.cuda.is_current_stream_capturing():      # (graph capture: graphed() checks after each replay)
        return
    st = status_word(device)
    if mode == "sync":
        # the one host synchronisation of the forward: 8 bytes into pinned memory + a stream synchronisation (a .tolist() / .item() goes
        # through a staged pageable copy: slower per forward; a one-thread kernel writing the words into pinned memory for the host to spin
        # on was measured too: no faster -- what the synchronisation costs is the host time from forward() entry to its first launch)
        fwd = bwd = None
        if _SPIN:
            # publish + spin under the device's lock: the sequence number and the pinned words are per device, and two threads running
            # forwards on one device would overwrite each other's (one of them then spun out its whole limit).  The spin is bounded by
            # TIME (a count of Python iterations is seconds on one host and minutes on another); past it: the copy + synchronise below.
            with st.lock:
                st.seq = (st.seq % 0x7ffffff0) + 1
                with torch.cuda.device(st.dev.device):
                    rc = _abi.load().egnn_status_publish(st.dev.data_ptr(), st.pub.data_ptr(), 2, st.seq, _stream())
                _abi.check(rc, "egnn_status_publish")
                view, seq, spins = st.pub_np, st.seq, 0
                deadline = None
                while view[2] != seq:
                    spins += 1
                    if (spins & 0xFFF) == 0:
                        now = time.perf_counter()
                        if deadline is None:
                            deadline = now + _SPIN_SECONDS
                        elif now > deadline:
                            break
                if view[2] == seq:
                    fwd, bwd = int(view[0]), int(view[1])
        if fwd is None:                                 # (EGNN_RANGE_SPIN=0, or the pinned word never changed: copy + synchronise)
            st.host.copy_(st.dev, non_blocking=True)
            torch.cuda.current_stream(st.dev.device).synchronize()
            fwd, bwd = int(st.host[0]), int(st.host[1])
        if fwd or bwd:
            st.dev.zero_()
            st.host.zero_()
        if bwd and not fwd:                             # left by an earlier backward: not this call's, never a reason to re-run it
            _raise_range(bwd, "an earlier backward", origin="backward")
        if fwd:
            _raise_range(fwd | bwd, "this call")
        return
    # deferred: examined by the next call / check_range().  Attribution is relaxed in this mode: the copy runs on the side stream and may
    # overlap the next forward's kernels OR-ing into the word (and check_range()'s zero_() on the launch stream), so a bit set by step
    # n + 1 can be reported with step n, or be reported twice -- the OR is sticky, detection itself is never lost.  The copy is a blit kernel: on the side stream, behind this forward's
    # last kernel, it does not sit between this step and the next one on the launch stream (two launch gaps + 4 us per step)
    cur = torch.cuda.current_stream(st.dev.device)
    side = side_stream(st.dev.device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        st.host.copy_(st.dev, non_blocking=True)
        st.event = torch.cuda.Event()
        st.event.record(side)


def check_range(device=None, wait=True):
    """Examine the status word of earlier (deferred-mode) forwards; raises EGNNRangeError if a value left the range."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return                                          # event queries are illegal inside a stream capture (graphed())
    for key, st in list(_status.items()):
        if device is not None and torch.device(device).index not in (None, key):
            continue
        if st.event is None:
            continue
        if wait:
            st.event.synchronize()
        elif not st.event.query():
            continue
        bits = int(st.host[0]) | int(st.host[1])
        st.event = None
        if bits:
            st.dev.zero_()
            st.host.zero_()
            _raise_range(bits, "an earlier call", origin="earlier")


def knn_select(coors, mask, adj_mat, k, out=None):
    """(idx int32 (B,N,K), rank (B,N,K) in the coordinates' dtype) -- egnn_knn_select_f32; float64 coordinates (a float64 module):
    egnn_knn_select_f64.  out = (idx, rank): caller-allocated outputs (the side-stream fork allocates them on the launch stream)."""
    b, n, cdim = coors.shape
    f64 = coors.dtype == torch.float64
    if out is not None:
        idx, rank = out
    else:
        idx = empty(b, n, k, dtype=torch.int32, device=coors.device)
        rank = empty(b, n, k, dtype=coors.dtype, device=coors.device)
    m8 = _u8(mask)
    a8 = _u8(adj_mat)
    stride = 0
    if a8 is not None:
        if a8.dim() == 3:
            if a8.shape != (b, n, n):
                raise ValueError(f"adj_mat shape {tuple(a8.shape)} != {(b, n, n)}")
            stride = n * n
        elif a8.shape != (n, n):
            raise ValueError(f"adj_mat shape {tuple(a8.shape)} != {(n, n)}")
    with _timed("knn_select"):
        fn = _abi.load().egnn_knn_select_f64 if f64 else _abi.load().egnn_knn_select_f32
        rc = fn(_ptr(coors), _ptr(m8), _ptr(a8), stride, b, n, k, cdim, _ptr(idx), _ptr(rank), _stream())
    _abi.check(rc, "egnn_knn_select_f64" if f64 else "egnn_knn_select_f32")
    return idx, rank


def spatial_order(coors, out=None, mask8=None):
    """(B,N) int32 Morton permutation -- egnn_spatial_order_masked_f32 (scheduling aid for the edge pass; the padded nodes of mask8,
    (B,N) bytes, behind the real ones)."""
    b, n, _ = coors.shape
    order = out if out is not None else empty(b, n, dtype=torch.int32, device=coors.device)
    with _timed("spatial_order"):
        rc = _abi.load().egnn_spatial_order_masked_f32(_ptr(coors), _ptr(mask8), b, n, _ptr(order), _stream())
    _abi.check(rc, "egnn_spatial_order_masked_f32")
    return order


def slot_prep(coors, mask8, idx, rank, order, valid_radius, out=None):
    """(B*N*K, 4) int32 per-slot records {j | pair_ok << 31, x_i - x_j} in the edge pass's consumption order --
    egnn_slot_prep_f32 (flattens the setup's index chain: include/egnn_hip.h).  idx None: the dense all-pairs layer (K = N, j = k)."""
    if idx is None:
        b, n, k = coors.shape[0], coors.shape[1], coors.shape[1]
    else:
        b, n, k = idx.shape
    slots = out if out is not None else empty(b * n * k, 4, dtype=torch.int32, device=coors.device)
    with _timed("slot_prep"):
        rc = _abi.load().egnn_slot_prep_f32(_ptr(coors), _ptr(mask8), _ptr(idx), _ptr(rank), _ptr(order),
                                            float(min(valid_radius, 3.0e38)), b, n, k, _ptr(slots), _stream())
    _abi.check(rc, "egnn_slot_prep_f32")
    return slots


def adj_expand(adj_mat, b, num_adj_degrees):
    """N-degree adjacency expansion (egnn_pytorch.py:414-427) -- egnn_adj_expand_u8.
    Returns (expanded adjacency (B,N,N) bool, adj_indices (B,N,N) uint8)."""
    a8 = _u8(adj_mat)
    n = a8.shape[-1]
    stride = n * n if a8.dim() == 3 else 0
    if a8.dim() == 3 and a8.shape[0] != b:
        raise ValueError(f"adj_mat batch {a8.shape[0]} != {b}")
    dev = a8.device
    adj_out = empty(b, n, n, dtype=torch.uint8, device=dev)
    deg = empty(b, n, n, dtype=torch.uint8, device=dev)
    ws = empty(_abi.load().egnn_adj_expand_workspace_bytes(b, n), dtype=torch.uint8, device=dev)
    with _timed("adj_expand"):
        rc = _abi.load().egnn_adj_expand_u8(_ptr(a8), stride, b, n, num_adj_degrees, _ptr(adj_out), _ptr(deg), _ptr(ws),
                                            _stream())
    _abi.check(rc, "egnn_adj_expand_u8")
    return adj_out.view(torch.bool), deg


def adj_max_degree(adj_mat):
    """int(adj_mat.float().sum(-1).max()) -- one device->host read, like the reference's .item() (:249)."""
    a8 = _u8(adj_mat)
    n = a8.shape[-1]
    rows = a8.numel() // n
    out = torch.empty(1, dtype=torch.int32, device=a8.device)
    rc = _abi.load().egnn_adj_max_degree_u8(_ptr(a8), rows, n, _ptr(out), _stream())
    _abi.check(rc, "egnn_adj_max_degree_u8")
    return int(out.item())


def _kpad(k):
    return (k + 31) // 32 * 32


def _packed_empty(rows, kp, device, zero=False):
    n = (rows + 31) // 32 * 32 * kp
    if zero:
        return torch.zeros(n, dtype=torch.float16, device=device)
    return empty(n, dtype=torch.float16, device=device)


class PackedHL:
    """A (rows, kp) matrix as fp16 (hi, lo) images in the packed tile-major layout (include/egnn_hip.h)."""
    __slots__ = ("hi", "lo", "rows", "kp")

    def __init__(self, hi, lo, rows, kp):
        self.hi, self.lo, self.rows, self.kp = hi, lo, rows, kp

    def dense(self):
        """Row-major fp32 reconstruction hi + lo (tests / debugging)."""
        from ._weights import unpack_tiles
        return (unpack_tiles(self.hi, self.rows, self.kp).float() + unpack_tiles(self.lo, self.rows, self.kp).float())[: self.rows]


def split_f16(x2d):
    """fp32 (rows, cols) -> packed fp16 (hi, lo) pair for egnn_linear_hl_f32 -- egnn_split_f16."""
    rows, cols = x2d.shape
    kp = _kpad(cols)
    hi = _packed_empty(rows, kp, x2d.device)
    lo = _packed_empty(rows, kp, x2d.device)
    with _timed("split_f16"):
        rc = _abi.load().egnn_split_f16(_ptr(x2d), cols, rows, cols, _ptr(hi), _ptr(lo), kp, _status_ptr(x2d.device),
                                        _stream())
    _abi.check(rc, "egnn_split_f16")
    return PackedHL(hi, lo, rows, kp)


def linear_hl(a: "PackedHL", wsplit, n, bias=None, residual=None, act=0, out_f32=True, out_hl=False, name="linear",
              split_cols=0, drop=None, row_mask=None):
    """act(A @ W.T + bias) (+ residual) with pre-split packed fp16 (hi, lo) operands -- egnn_linear_hl_f32.
    Returns fp32 C, or a PackedHL (padded to 32 columns for the next GEMM) when out_hl, or both.
    split_cols: columns [0, split_cols) of C hold (fp16 hi, fp16 lo) words instead of fp32 values.
    row_mask: (M,) bytes -- M-tiles without a set row are skipped, their rows of C stay unwritten (egnn_linear_hl_lda_rows_f32: the
    projection table of a padded batch, inference only)."""
    whi, wlo, inv, w_rows = wsplit
    m, kp = a.rows, a.kp
    kp_a = 0
    if whi.numel() != w_rows * kp:
        # the A image is wider than the contraction: the product runs over its first kp columns (egnn_linear_hl_lda_f32 -- the
        # projection reads feats out of the [feats | m_i] image of node_mlp's input)
        kp_a, kp = kp, whi.numel() // w_rows
        assert kp < kp_a and kp % 32 == 0 and drop is None
    assert whi.numel() == w_rows * kp and w_rows >= n
    dev = a.hi.device
    c = empty(m, n, dtype=torch.float32, device=dev) if out_f32 else None
    out = None
    kp_out = 0
    if out_hl:
        kp_out = _kpad(n)
        zero = kp_out != n                                   # pad columns must read as zero in the next GEMM
        out = PackedHL(_packed_empty(m, kp_out, dev, zero), _packed_empty(m, kp_out, dev, zero), m, kp_out)
    ldr = 0
    if residual is not None:
        assert residual.shape == (m, n) and residual.is_contiguous()
        ldr = n
    with _timed(name):
        if row_mask is not None:
            assert drop is None and row_mask.numel() == m and row_mask.is_contiguous()
            rc = _abi.load().egnn_linear_hl_lda_rows_f32(_ptr(a.hi), _ptr(a.lo), kp_a, _ptr(whi), _ptr(wlo), float(inv), _ptr(bias),
                                                         _ptr(residual), ldr, _ptr(c), n, _ptr(out.hi) if out else None,
                                                         _ptr(out.lo) if out else None, kp_out, m, n, kp, w_rows, act, int(split_cols),
                                                         _ptr(row_mask), _status_ptr(dev), _stream())
        elif kp_a:
            rc = _abi.load().egnn_linear_hl_lda_f32(_ptr(a.hi), _ptr(a.lo), kp_a, _ptr(whi), _ptr(wlo), float(inv), _ptr(bias),
                                                    _ptr(residual), ldr, _ptr(c), n, _ptr(out.hi) if out else None,
                                                    _ptr(out.lo) if out else None, kp_out, m, n, kp, w_rows, act, int(split_cols),
                                                    _status_ptr(dev), _stream())
        elif drop is None:
            rc = _abi.load().egnn_linear_hl_f32(_ptr(a.hi), _ptr(a.lo), _ptr(whi), _ptr(wlo), float(inv), _ptr(bias),
                                                _ptr(residual), ldr, _ptr(c), n, _ptr(out.hi) if out else None,
                                                _ptr(out.lo) if out else None, kp_out, m, n, kp, w_rows, act, int(split_cols),
                                                _status_ptr(dev), _stream())
        else:                                               # (p, seed): nn.Dropout between the Linear and its activation
            from . import _dropout
            p_drop, seed = drop
            rc = _abi.load().egnn_linear_hl_drop_f32(_ptr(a.hi), _ptr(a.lo), _ptr(whi), _ptr(wlo), float(inv), _ptr(bias),
                                                     _ptr(residual), ldr, _ptr(c), n, _ptr(out.hi) if out else None,
                                                     _ptr(out.lo) if out else None, kp_out, m, n, kp, w_rows, act, int(split_cols),
                                                     _dropout.threshold(p_drop), int(seed), _dropout.inv_keep(p_drop),
                                                     _status_ptr(dev), _stream())
    _abi.check(rc, "egnn_linear_hl_f32")
    if out_f32 and out_hl:
        return c, out
    return out if out_hl else c


def node_mlp_fused_supported(dim, m_dim):
    """egnn_node_mlp_fused_f32 is built for this layer width (m_dim = 16, dim in {32, 64, 128, 256})."""
    return _abi.load().egnn_node_mlp_fused_halves(int(dim), int(m_dim)) > 0


def node_mlp_fused_image(w5_split, w6_split, dim, m_dim):
    """The fused kernel's weight image (fp16 tensor) from the two packed (hi, lo) weight images of egnn_linear_hl_f32."""
    lib = _abi.load()
    n = lib.egnn_node_mlp_fused_halves(int(dim), int(m_dim))
    assert n > 0
    img = empty(n, dtype=torch.float16, device=w5_split[0].device)
    rc = lib.egnn_node_mlp_fused_pack_f16(_ptr(w5_split[0]), _ptr(w5_split[1]), _ptr(w6_split[0]), _ptr(w6_split[1]), int(dim), int(m_dim),
                                          _ptr(img), _stream())
    _abi.check(rc, "egnn_node_mlp_fused_pack_f16")
    return img


def node_mlp_fused(node_in: "PackedHL", image, w5_inv, b5, w6_inv, b6, residual, dim, m_dim):
    """W6 SiLU(W5 [LayerNorm(h) | m_i] + b5) + b6 + residual in one launch (egnn_node_mlp_fused_f32): (rows, dim) fp32."""
    m = node_in.rows
    assert residual.shape == (m, dim) and residual.is_contiguous() and node_in.kp == _kpad(dim + m_dim)
    out = empty(m, dim, dtype=torch.float32, device=residual.device)
    with _timed("node_mlp"):
        rc = _abi.load().egnn_node_mlp_fused_f32(_ptr(node_in.hi), _ptr(node_in.lo), _ptr(image), float(w5_inv), _ptr(b5), float(w6_inv),
                                                 _ptr(b6), _ptr(residual), _ptr(out), m, int(dim), int(m_dim),
                                                 _status_ptr(residual.device), _stream())
    _abi.check(rc, "egnn_node_mlp_fused_f32")
    return out


# ---- the wide-range path: the same layer in plain fp32 (include/egnn_hip.h, "The wide-range path")
def linear_f32(a, w, n, k, bias=None, residual=None, act=0, out=None, name="linear_f32"):
    """act(a @ w[:n, :k].T + bias) (+ residual) in exact fp32 (v_mfma_f32_32x32x2_f32) -- egnn_linear_f32.  a (M, >= k) and w
    (>= n rows, row stride w.stride(0)) may be column slices of wider tensors (unit column stride); `out` likewise (M, >= n)."""
    m = a.shape[0]
    assert a.stride(1) == 1 and w.stride(1) == 1 and a.dtype == w.dtype and a.dtype in (torch.float32, torch.float64)
    f64 = a.dtype == torch.float64                   # (a float64 module: egnn_linear_f64, v_mfma_f64_16x16x4_f64)
    if out is None:
        out = empty(m, n, dtype=a.dtype, device=a.device)
    assert out.stride(1) == 1 and out.shape[0] == m
    ld = lambda t, width: t.stride(0) if t.shape[0] > 1 else max(t.stride(0), width)      # noqa: E731  (a one-row matrix may carry any stride)
    ldr = 0
    if residual is not None:
        assert residual.stride(1) == 1 and residual.shape[0] == m
        ldr = ld(residual, n)
    with _timed(name):
        fn = _abi.load().egnn_linear_f64 if f64 else _abi.load().egnn_linear_f32
        rc = fn(_ptr(a), ld(a, k), _ptr(w), ld(w, k), _ptr(bias), _ptr(residual), ldr, _ptr(out), ld(out, n), m, n, k, act, _stream())
    _abi.check(rc, "egnn_linear_f64" if f64 else "egnn_linear_f32")
    return out


def node_prep_f32(feats2d, m_i, gamma, beta, eps, m_dim):
    """[LayerNorm(feats) | m_i] as a plain (rows, dim + m_dim) matrix in feats2d's dtype -- egnn_node_prep_f32 / egnn_node_prep_f64."""
    rows, dim = feats2d.shape
    out = empty(rows, dim + m_dim, dtype=feats2d.dtype, device=feats2d.device)
    with _timed("node_prep_f32"):
        fn = _abi.load().egnn_node_prep_f64 if feats2d.dtype == torch.float64 else _abi.load().egnn_node_prep_f32
        rc = fn(_ptr(feats2d), _ptr(m_i), _ptr(gamma), _ptr(beta), float(eps), _ptr(out), rows, dim, m_dim, _stream())
    _abi.check(rc, "egnn_node_prep_f32")
    return out


def edge_exact(args: "_abi.EdgeExactArgs", device, dtype=torch.float32):
    """egnn_edge_exact_f32 / egnn_edge_exact_f64 (dtype: what the data pointers of args hold); allocates the per-edge workspace the
    entry asks for."""
    lib = _abi.load()
    nbytes = lib.egnn_edge_exact_workspace_bytes(args.B, args.N, args.K, args.m_dim, args.coor_dim)
    ws = empty(max(1, nbytes // 4), dtype=dtype, device=device)
    args.edge_ws = ws.data_ptr()
    with _timed("edge_exact"):
        rc = (lib.egnn_edge_exact_f64 if dtype == torch.float64 else lib.egnn_edge_exact_f32)(byref(args), _stream())
    _abi.check(rc, "egnn_edge_exact_f64" if dtype == torch.float64 else "egnn_edge_exact_f32")
    return ws


def set_drop(args, drop, eid0=0):
    """drop = (p, seed) -> the drop_* fields of an argument block of the plain kernels (None: no dropout)"""
    if drop is not None:
        from . import _dropout
        args.drop_thr, args.drop_seed, args.drop_inv_keep, args.drop_eid0 = (_dropout.threshold(drop[0]), int(drop[1]),
                                                                             _dropout.inv_keep(drop[0]), int(eid0))


def drop_silu_(z, drop, row0=0):
    """z (rows, cols) <- SiLU(dropout(z)) in place (egnn_drop_silu_f32 / _f64): nn.Dropout between node_mlp's first Linear and its SiLU on
    the plain kernels, the mask = the kernels' hash of node rows row0 .. (drop = (p, seed); None: plain SiLU)."""
    from . import _dropout
    assert z.dim() == 2 and z.stride(1) == 1 and z.dtype in (torch.float32, torch.float64)
    lib = _abi.load()
    thr, seed, inv = (0, 0, 1.0) if drop is None else (_dropout.threshold(drop[0]), int(drop[1]), _dropout.inv_keep(drop[0]))
    with _timed("drop_silu"):
        rc = (lib.egnn_drop_silu_f64 if z.dtype == torch.float64 else lib.egnn_drop_silu_f32)(_ptr(z), z.stride(0) if z.shape[0] > 1 else z.shape[1],
                                                                                            z.shape[0], z.shape[1], thr, seed, inv, int(row0), _stream())
    _abi.check(rc, "egnn_drop_silu")
    return z


def edge_exact_bwd(args: "_abi.EdgeExactBwdArgs", dtype):
    """egnn_edge_exact_bwd_f32 / _f64: a^T, dz^T (H, E) and d/d scalars (E, S) of a chunk of graphs (include/egnn_hip.h)."""
    lib = _abi.load()
    f64 = dtype == torch.float64
    with _timed("edge_exact_bwd"):
        rc = (lib.egnn_edge_exact_bwd_f64 if f64 else lib.egnn_edge_exact_bwd_f32)(byref(args), _stream())
    _abi.check(rc, "egnn_edge_exact_bwd_f64" if f64 else "egnn_edge_exact_bwd_f32")


def edge_exact_node_sums(dz_t, nodes, k, order, seg):
    """egnn_edge_exact_node_sums_*: (d/d P_i, its transpose, d/d P_j, its transpose) -- (nodes, H), (H, nodes) twice -- from dz^T (H, E):
    the K edges leaving a node and the edges arriving at it (CSR lists of egnn_dest_lists_i32), summed in a fixed order."""
    h, e = dz_t.shape
    f64 = dz_t.dtype == torch.float64
    gpi, gpj = (empty(nodes, h, dtype=dz_t.dtype, device=dz_t.device) for _ in range(2))
    gpi_t, gpj_t = (empty(h, nodes, dtype=dz_t.dtype, device=dz_t.device) for _ in range(2))
    lib = _abi.load()
    with _timed("edge_exact_node_sums"):
        rc = (lib.egnn_edge_exact_node_sums_f64 if f64 else lib.egnn_edge_exact_node_sums_f32)(
            _ptr(dz_t), e, h, nodes, k, _ptr(order), _ptr(seg), _ptr(gpi), _ptr(gpi_t), _ptr(gpj), _ptr(gpj_t), _stream())
    _abi.check(rc, "egnn_edge_exact_node_sums")
    return gpi, gpi_t, gpj, gpj_t


def edge_tail_exact(u, coors, idx32, pair_mask, g_coors_out, g_msum, w3, b3, w4, b4, scale, eps, clamp, gate, b, n, k, drop=None, eid0=0):
    """egnn_edge_tail_exact_bwd_f32 / _f64: the per-edge chain behind u in closed form for any head width <= 64 / coordinate dimension,
    in u's dtype.  u (E, m); coors (B, N, C); g_coors_out (B N, C); g_msum (B N, m) or None; w3 .. b4 = coors_mlp's tensors or None
    (update_coors=False); scale = coors_norm.scale or None; gate = (weight (m), bias (1)) or None.  Returns a dict: gU (E, m), g_rel_t
    (C, E) -- transposed: its per-node sums come from egnn_edge_exact_node_sums_* -- and the transposed operands of the parameter
    gradients ghid_t, a3_t (4m, E), mm_t, m0_t (m, E), g_w, g_scale, g_gate (E).  drop = (p, seed), eid0: training-mode dropout in
    coors_mlp -- the forward's hash mask of edge rows eid0 .. re-evaluated."""
    e, m = u.shape
    dt, dev = u.dtype, u.device
    cdim = coors.shape[-1]
    a = _abi.EdgeTailExactArgs()
    a.B, a.N, a.K, a.m_dim, a.coor_dim, a.norm_coors = b, n, k, m, cdim, int(scale is not None)
    a.eps, a.clamp = float(eps), -1.0 if clamp is None else float(clamp)
    out = dict(gU=empty(e, m, dtype=dt, device=dev), g_rel_t=empty(cdim, e, dtype=dt, device=dev), mm_t=empty(m, e, dtype=dt, device=dev))
    keep = [t.contiguous() for t in (u, coors, g_coors_out)]
    a.u, a.coors, a.g_coors_out = (t.data_ptr() for t in keep)
    a.idx, a.pair_mask = _ptr(idx32), _ptr(pair_mask)
    if g_msum is not None:
        keep.append(g_msum.contiguous())
        a.g_msum = keep[-1].data_ptr()
    if w3 is not None:
        keep += [t.detach().to(dt).contiguous() for t in (w3, b3, w4.reshape(-1), b4.reshape(-1))]
        a.W3, a.b3, a.W4, a.b4 = (t.data_ptr() for t in keep[-4:])
        out.update(ghid_t=empty(4 * m, e, dtype=dt, device=dev), a3_t=empty(4 * m, e, dtype=dt, device=dev), g_w=empty(e, dtype=dt, device=dev))
        a.ghid_t, a.a3_t, a.g_w = out["ghid_t"].data_ptr(), out["a3_t"].data_ptr(), out["g_w"].data_ptr()
        if scale is not None:
            keep.append(scale.detach().to(dt).contiguous())
            a.scale = keep[-1].data_ptr()
            out["g_scale"] = empty(e, dtype=dt, device=dev)
            a.g_scale = out["g_scale"].data_ptr()
    else:
        a.norm_coors = 0
    if gate is not None:
        keep += [gate[0].detach().to(dt).reshape(-1).contiguous(), gate[1].detach().to(dt).reshape(-1).contiguous()]
        a.gate_w, a.gate_b = keep[-2].data_ptr(), keep[-1].data_ptr()
        out.update(m0_t=empty(m, e, dtype=dt, device=dev), g_gate=empty(e, dtype=dt, device=dev))
        a.m0_t, a.g_gate = out["m0_t"].data_ptr(), out["g_gate"].data_ptr()
    a.gU, a.g_rel, a.mm_t = out["gU"].data_ptr(), out["g_rel_t"].data_ptr(), out["mm_t"].data_ptr()
    if drop is not None:
        from . import _dropout
        a.drop_thr, a.drop_seed, a.drop_inv_keep, a.drop_eid0 = _dropout.threshold(drop[0]), int(drop[1]), _dropout.inv_keep(drop[0]), int(eid0)
    lib = _abi.load()
    f64 = dt == torch.float64
    with _timed("edge_tail_exact"):
        rc = (lib.egnn_edge_tail_exact_bwd_f64 if f64 else lib.egnn_edge_tail_exact_bwd_f32)(byref(a), _stream())
    _abi.check(rc, "egnn_edge_tail_exact_bwd")
    return out


def node_prep_hl(feats2d, m_i, gamma, beta, eps, m_dim, with_raw=False):
    """[LayerNorm(feats) | m_i] as a packed fp16 (hi, lo) pair -- egnn_node_prep_hl.  m_i None: those columns are
    zero (the edge pass writes them in place).  with_raw: also return feats itself as a packed pair (the projection's
    A operand), produced in the same pass."""
    rows, dim = feats2d.shape
    kp = _kpad(dim + m_dim)
    hi = _packed_empty(rows, kp, feats2d.device)
    lo = _packed_empty(rows, kp, feats2d.device)
    raw = None
    if with_raw:
        rkp = _kpad(dim)
        raw = PackedHL(_packed_empty(rows, rkp, feats2d.device), _packed_empty(rows, rkp, feats2d.device), rows, rkp)
    with _timed("node_prep"):
        rc = _abi.load().egnn_node_prep_hl(_ptr(feats2d), _ptr(m_i), _ptr(gamma), _ptr(beta), float(eps), _ptr(hi), _ptr(lo),
                                           kp, _ptr(raw.hi) if raw else None, _ptr(raw.lo) if raw else None,
                                           raw.kp if raw else 0, rows, dim, m_dim, _status_ptr(feats2d.device),
                                           _stream())
    _abi.check(rc, "egnn_node_prep_hl")
    out = PackedHL(hi, lo, rows, kp)
    return (out, raw) if with_raw else out


def edge_fused(args: _abi.EdgeArgs, device):
    args.status = _status_ptr(device)
    with _timed("edge_fused"):
        rc = _abi.load().egnn_edge_fused_f32(byref(args), _stream())
    _abi.check(rc, "egnn_edge_fused_f32")


def edge_features_gather(lookup, idx, b, n, k):
    """(B,N,K,edge_dim) features of the selected pairs from EGNN_Network's look-up tables -- egnn_edge_features_gather_f32."""
    dev = (lookup.deg if lookup.deg is not None else (lookup.tok if lookup.tok is not None else lookup.edges)).device
    out = empty(b, n, k, lookup.width, dtype=torch.float32, device=dev)
    with _timed("edge_features"):
        rc = _abi.load().egnn_edge_features_gather_f32(_ptr(lookup.edges), _ptr(lookup.tok), _ptr(lookup.tok_emb), lookup.d1,
                                                       _ptr(lookup.deg), _ptr(lookup.deg_emb), lookup.d2, _ptr(idx), b, n, k,
                                                       _ptr(out), _stream())
    _abi.check(rc, "egnn_edge_features_gather_f32")
    return out


def induced_attn(q, kv, mask, b, n, heads, dim_head, scale):
    """attn1 core of the induced-set attention block: T tokens attend over the nodes -- egnn_induced_attn_f32."""
    t = q.shape[1]
    out = empty(b, t, heads * dim_head, dtype=torch.float32, device=kv.device)
    with _timed("induced_attn"):
        rc = _abi.load().egnn_induced_attn_f32(_ptr(q.contiguous()), _ptr(kv), kv.stride(0), _ptr(_u8(mask)), b, n, t, heads, dim_head,
                                               float(scale), _ptr(out), _stream())
    _abi.check(rc, "egnn_induced_attn_f32")
    return out


def token_attn(q, kv_tok, b, n, heads, dim_head, scale):
    """attn2 core: every node attends over the T induced tokens -- egnn_token_attn_f32."""
    t = kv_tok.shape[1]
    out = empty(b * n, heads * dim_head, dtype=torch.float32, device=q.device)
    with _timed("token_attn"):
        rc = _abi.load().egnn_token_attn_f32(_ptr(q), q.stride(0), _ptr(kv_tok.contiguous()), b, n, t, heads, dim_head, float(scale),
                                             _ptr(out), out.stride(0), _stream())
    _abi.check(rc, "egnn_token_attn_f32")
    return out


_HOST_READ_EVENTS = os.environ.get("EGNN_HOST_READ_EVENTS", "1") != "0"     # 0: every small read-back drains the stream (`.tolist()`)


class HostRead:
    """A few words of device memory on their way to the host WITHOUT draining the stream: created right behind the kernel that writes
    them -- a copy to pinned memory and an event behind that copy -- and waited for (`tensor()` / `floats()` / `ints()`) only where the
    values are needed.  The backward chooses its power-of-two scales from max |x| words that its kernels leave behind; read with
    `.tolist()` each of them waited for EVERYTHING queued so far and left the device idle until the host had caught up (eight times per
    training step: profiles/r05_experiments/train_step_timeline.txt).  With the wait pinned to the producing kernel the host goes on
    queueing independent work in between and wakes up while the device is still busy."""

    def __init__(self, t):
        self.dtype = t.dtype
        if t.is_cuda and _HOST_READ_EVENTS:
            self.buf = torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)
            with torch.cuda.device(t.device):                # (the copy and the event on the tensor's device's current stream)
                self.buf.copy_(t.reshape(-1), non_blocking=True)
                self.ev = torch.cuda.Event()
                self.ev.record(torch.cuda.current_stream(t.device))
        else:
            self.buf, self.ev = t.reshape(-1), None

    def tensor(self):
        if self.ev is not None:
            self.ev.synchronize()
            self.ev = None
        return self.buf

    def floats(self):
        """max-|x| bit patterns (egnn_absmax_f32's contract) as Python floats"""
        t = self.tensor()
        return (t.view(torch.float32) if t.dtype == torch.int32 else t).tolist()

    def ints(self):
        return self.tensor().tolist()


def bits_to_floats(bits):
    """The floats behind a tensor of max-|x| bit patterns (egnn_absmax_f32's contract), in ONE host read (or behind a HostRead)."""
    if isinstance(bits, HostRead):
        return bits.floats()
    return bits.view(torch.float32).tolist()


def rows_gather_sum(rows, order, seg_ptr, n_out, want_amax=False):
    """out[r] = sum of rows[order[p]] over p in [seg_ptr[r], seg_ptr[r+1]), fixed order -- egnn_rows_gather_sum_f32.
    want_amax: returns (out, bits) with bits = a 1-element int32 tensor holding the bit pattern of max |out| (a by-product)."""
    cols = rows.shape[1]
    out = empty(n_out, cols, dtype=torch.float32, device=rows.device)
    bits = torch.empty(1, dtype=torch.int32, device=rows.device) if want_amax else None
    with _timed("rows_gather_sum"):
        rc = _abi.load().egnn_rows_gather_sum_f32(_ptr(rows), rows.stride(0), _ptr(order), _ptr(seg_ptr), n_out, cols, _ptr(out),
                                                  cols, _ptr(bits), _stream())
    _abi.check(rc, "egnn_rows_gather_sum_f32")
    return (out, bits) if want_amax else out


def split_scaled(x2d, scale, transposed=False, w_image=False):
    """packed fp16 (hi, lo) images of scale * x2d (or of its transpose) -- egnn_split_scaled_f16.  w_image: allocate whole 128-row
    tiles (zero filled), as the W operand of egnn_linear_hl_f32 wants."""
    rows, cols = x2d.shape
    img_rows, k_extent = (cols, rows) if transposed else (rows, cols)
    kp = _kpad(k_extent)
    alloc_rows = (img_rows + 127) // 128 * 128 if w_image else img_rows
    hi = _packed_empty(alloc_rows, kp, x2d.device, zero=w_image)
    lo = _packed_empty(alloc_rows, kp, x2d.device, zero=w_image)
    with _timed("split_scaled"):
        rc = _abi.load().egnn_split_scaled_f16(_ptr(x2d), x2d.stride(0), rows, cols, float(scale), int(transposed), _ptr(hi), _ptr(lo), kp,
                                               _status_ptr(x2d.device), _stream())
    _abi.check(rc, "egnn_split_scaled_f16")
    return PackedHL(hi, lo, img_rows, kp), alloc_rows


def split_scaled_both(x2d, scale, colsum=False):
    """(plain image, transposed image) of scale * x2d from one read -- egnn_split_scaled_both_f16.  colsum: also the column sums of
    x2d (cols,), a by-product of the same read: per 64-row block inside the kernel, the blocks added up in fixed order (`sum_rows`)."""
    rows, cols = x2d.shape
    kp, kpt = _kpad(cols), _kpad(rows)
    hi, lo = _packed_empty(rows, kp, x2d.device), _packed_empty(rows, kp, x2d.device)
    hit, lot = _packed_empty(cols, kpt, x2d.device), _packed_empty(cols, kpt, x2d.device)
    lib = _abi.load()
    parts, ld = None, (cols + 3) // 4 * 4
    if colsum:
        parts = empty(int(lib.egnn_split_scaled_colsum_rows(rows, kpt)), ld, dtype=torch.float32, device=x2d.device)
    with _timed("split_scaled"):
        rc = lib.egnn_split_scaled_both_f16(_ptr(x2d), x2d.stride(0), rows, cols, float(scale), _ptr(hi), _ptr(lo), kp,
                                            _ptr(hit), _ptr(lot), kpt, _status_ptr(x2d.device), _ptr(parts), ld, _stream())
    _abi.check(rc, "egnn_split_scaled_both_f16")
    if colsum:
        return PackedHL(hi, lo, rows, kp), PackedHL(hit, lot, cols, kpt), sum_rows(parts, 1.0 / float(scale))[:cols]
    return PackedHL(hi, lo, rows, kp), PackedHL(hit, lot, cols, kpt)


def silu_bwd_(z, g, drop=None, row0=0):
    """z <- SiLU(z), g <- g * SiLU'(z), in place, one pass (egnn_silu_bwd_f32).  Returns (z, g, bits): bits = the bit patterns of
    max |SiLU(z)| and max |g SiLU'(z)| (2-element int32 tensor; bits_to_floats).  drop = (p, seed): training-mode dropout between
    the Linear that produced z (rows, cols) and the SiLU -- the forward's hash mask of rows row0 .. (egnn_silu_bwd_drop_f32)."""
    bits = torch.empty(2, dtype=torch.int32, device=z.device)
    with _timed("silu_bwd"):
        if drop is None:
            rc = _abi.load().egnn_silu_bwd_f32(_ptr(z), _ptr(g), _ptr(z), _ptr(g), z.numel(), _ptr(bits), _stream())
        else:
            from . import _dropout
            rc = _abi.load().egnn_silu_bwd_drop_f32(_ptr(z), _ptr(g), _ptr(z), _ptr(g), z.numel(), _ptr(bits), _dropout.threshold(drop[0]),
                                                    int(drop[1]), _dropout.inv_keep(drop[0]), int(row0), z.shape[-1], _stream())
    _abi.check(rc, "egnn_silu_bwd_f32")
    return z, g, bits


class GradOperand:
    """A gradient matrix g (R, M) prepared once for its two products: .plain = image of s g (the A operand of g @ W), .t = image of
    (s g)^T (the A operand of g^T @ x), .scale = s (a power of two, grad_scale); .zero: g is all zero / not finite."""

    def __init__(self, g2d, amax=None, colsum=False):
        self.shape = g2d.shape
        self.scale = grad_scale(absmax(g2d) if amax is None else amax)
        self.zero = self.scale is None
        self.plain = self.t = self.colsum = None
        if not self.zero and self.scale is not NONFINITE:
            if colsum:                       # (the column sums -- a bias gradient -- ride along with the split's read of g)
                self.plain, self.t, self.colsum = split_scaled_both(g2d, self.scale, colsum=True)
            else:
                self.plain, self.t = split_scaled_both(g2d, self.scale)
        elif colsum:
            self.colsum = g2d.sum(dim=0)     # (all zero, or not finite: the sum says so)


def absmax(x):
    """max |x| as a Python float (one host read) -- egnn_absmax_f32: one pass over x where `x.abs().max()` makes two and a copy.
    What the power-of-two scales of the gradient GEMMs and of the edge backward are chosen from.  (Host tensors -- the CPU tests of
    the backward's host logic -- go through torch.)"""
    if not x.is_cuda:
        return float(x.abs().max())
    return absmax_async(x).floats()[0]


def absmax_async(x):
    """`absmax` whose host read can wait: the launch now, a HostRead of the bit pattern (`.floats()[0]` = max |x|)."""
    if not x.is_cuda:
        return HostRead(x.abs().max().reshape(1).float())
    x = x if x.is_contiguous() else x.contiguous()
    out = torch.empty(1, dtype=torch.int32, device=x.device)
    with _timed("absmax"):
        rc = _abi.load().egnn_absmax_f32(_ptr(x), x.numel(), _ptr(out), _stream())
    _abi.check(rc, "egnn_absmax_f32")
    return HostRead(out)


def unsplit_words_(table, cols):
    """egnn_unsplit_words_f32: columns [0, cols) of the fp32 table hold (fp16 hi, fp16 lo) words -> hi + lo, in place."""
    with _timed("unsplit_words"):
        rc = _abi.load().egnn_unsplit_words_f32(_ptr(table), table.stride(0), table.shape[0], cols, _stream())
    _abi.check(rc, "egnn_unsplit_words_f32")
    return table


NONFINITE = "nonfinite"          # grad_scale of an operand that holds a NaN / inf: its products are NaN, like any fp32 product would be


def grad_scale(amax):
    """The power of two that brings max |x| into [2^12, 2^13) (fp16 hi halves well inside the range, lo halves off the subnormals);
    None for an all-zero operand (the product is zero), NONFINITE for one that holds a NaN or an inf (the product is NaN)."""
    from . import _weights
    if amax != amax or amax == float("inf"):
        return NONFINITE
    if not (amax > 0.0):
        return None
    # (clamped: for max |x| below ~2^-88 the scale -- a C float in the ABI -- and the inverse scales built from it would leave fp32's
    # normal range; such an operand is split with 2^100 instead, its values then sit in fp16's subnormals or flush to zero, which is
    # what the fp32 product they replace would have given at that magnitude: ~0)
    return min(_weights.pow2_scale(amax) * 4096.0, 2.0 ** 100)


def grad_nn(g2d, wsplit_t, n, residual=None, name="grad_nn", amax=None):
    """g2d (R, K) fp32 @ W (K, n), W given as the packed split image of W^T (n rows, K): on the split-f16 GEMM of the forward,
    g2d pre-scaled by a power of two (gradients are small: their fp16 lo halves must stay off the subnormals).  amax: max |g2d| if
    the caller has it already; g2d may be a GradOperand (both of its products share one split)."""
    if isinstance(g2d, GradOperand):
        op = g2d
        scale, a, rows, dev = op.scale, op.plain, op.shape[0], (residual.device if residual is not None else wsplit_t[0].device)
    else:
        scale, rows, dev = grad_scale(absmax(g2d) if amax is None else amax), g2d.shape[0], g2d.device
        a = None if (scale is None or scale is NONFINITE) else split_scaled(g2d, scale)[0]      # max |scale * g| in [2^12, 2^13)
    if scale is NONFINITE:
        return torch.full((rows, n), float("nan"), dtype=torch.float32, device=dev)
    if scale is None:
        out = torch.zeros(rows, n, dtype=torch.float32, device=dev)
        return out if residual is None else out + residual
    whi, wlo, inv, w_rows = wsplit_t
    return linear_hl(a, (whi, wlo, inv / scale, w_rows), n, None, residual=residual, name=name)


def grad_tn_operand(x2d, amax=None):
    """The right-hand operand of grad_tn, prepared once for several products with the same x2d: (packed image of (s x2d)^T, rows, s)
    or None when x2d is all zero / not finite.  amax: max |x2d| if the caller has it."""
    sx = grad_scale(absmax(x2d) if amax is None else amax)
    if sx is None or sx is NONFINITE:
        return sx
    w, w_rows = split_scaled(x2d, sx, transposed=True, w_image=True)  # (n rows, K = r)
    return w, w_rows, sx


def grad_tn(g2d, x2d, k_splits=None, name="grad_tn", amax=None, x_operand=None):
    """g2d (R, M)^T @ x2d (R, N) -> (M, N): the weight-gradient contraction over R = B N nodes, split-K on the split-f16 GEMM.
    amax: max |g2d| if known; x_operand: grad_tn_operand(x2d) if several products share x2d."""
    op = g2d if isinstance(g2d, GradOperand) else None
    r, m = (op.shape if op is not None else g2d.shape)
    n = x2d.shape[1]
    sg = op.scale if op is not None else grad_scale(absmax(g2d) if amax is None else amax)
    if x_operand is None:
        x_operand = grad_tn_operand(x2d)
    if sg is NONFINITE or x_operand is NONFINITE:
        return torch.full((m, n), float("nan"), dtype=torch.float32, device=x2d.device)
    if sg is None or x_operand is None:
        return torch.zeros(m, n, dtype=torch.float32, device=x2d.device)
    w, w_rows, sx = x_operand
    a = op.t if op is not None else split_scaled(g2d, sg, transposed=True)[0]                    # (m rows, K = r)
    nkt = a.kp // 16
    if k_splits is None:
        tiles = ((m + 127) // 128) * ((n + 127) // 128)
        k_splits = 1
        while tiles * k_splits < 512 and nkt // (2 * k_splits) >= 64:
            k_splits *= 2
    mp = m
    parts = empty(k_splits, mp, n, dtype=torch.float32, device=x2d.device)
    with _timed(name):
        rc = _abi.load().egnn_linear_hl_splitk_f32(_ptr(a.hi), _ptr(a.lo), _ptr(w.hi), _ptr(w.lo), 1.0, _ptr(parts), n, m, n, a.kp, w_rows,
                                                   k_splits, _stream())
    _abi.check(rc, "egnn_linear_hl_splitk_f32")
    out = empty(m, n, dtype=torch.float32, device=x2d.device)
    count = m * n
    if count % 4 != 0:
        return parts.sum(dim=0) / (sg * sx)
    rc = _abi.load().egnn_sum_parts_f32(_ptr(parts), k_splits, count, 1.0 / (sg * sx), _ptr(out), _stream())
    _abi.check(rc, "egnn_sum_parts_f32")
    return out


class DestLists:
    """The edges sorted stably by destination (egnn_dest_lists_i32): `ent` / `tile_seg` = the padded entry list of
    egnn_edge_bwd_pass_f32 (by_dest = 1), `order` / `seg` = the CSR form egnn_rows_gather_sum_f32 reads."""

    def __init__(self, ent, tile_seg, order, seg, tiles=None):
        self._ent, self.tile_seg, self.order, self.seg, self._tiles = ent, tile_seg, order, seg, tiles

    @property
    def ent(self):
        """the entry list cut to its length -- the one host read (the number of tiles), taken when the list is first used"""
        if self._tiles is not None:
            tiles = int(self._tiles.ints()[0])
            self._ent, self._tiles = self._ent[:max(128, (tiles * 16 + 127) // 128 * 128)], None
        return self._ent


def dest_lists(idx32, b, n, k, device):
    """idx32 (B,N,K) int32 or None (dense: destination = k).  One host read: the number of tiles (sizes the entry list)."""
    lib = _abi.load()
    cap = lib.egnn_dest_lists_capacity(b, n, k)
    ent = torch.empty(cap, dtype=torch.int32, device=device)
    tile_seg = torch.empty(b * n + 1, dtype=torch.int64, device=device)
    order = torch.empty(b * n * k, dtype=torch.int64, device=device)
    seg = torch.empty(b * n + 1, dtype=torch.int64, device=device)
    scratch = torch.empty(b, dtype=torch.int64, device=device)
    with _timed("dest_lists"):
        rc = lib.egnn_dest_lists_i32(_ptr(idx32), b, n, k, _ptr(ent), cap, _ptr(tile_seg), _ptr(order), _ptr(seg), _ptr(scratch), _stream())
    _abi.check(rc, "egnn_dest_lists_i32")
    return DestLists(ent, tile_seg, order, seg, HostRead(tile_seg[-1:]))


def sum_rows(part, scale=1.0):
    """scale * column sums of a (rows, count) fp32 array with many rows, in a fixed order, on egnn_sum_parts_f32 (which walks its parts
    one after the other per output element: made for a handful of split-K parts): first the rows p G + g over p for each of G groups
    (the array read as (rows / G) parts of G * count elements), then the G group sums."""
    rows, count = part.shape
    lib = _abi.load()
    g = 1
    while g < 256 and rows % (2 * g) == 0 and rows // (2 * g) >= 1:
        g *= 2
    out = empty(count, dtype=torch.float32, device=part.device)
    with _timed("sum_rows"):
        if g > 1 and rows // g > 1:
            mid = empty(g, count, dtype=torch.float32, device=part.device)
            _abi.check(lib.egnn_sum_parts_f32(_ptr(part), rows // g, g * count, 1.0, _ptr(mid), _stream()), "egnn_sum_parts_f32")
            part, rows = mid, g
        _abi.check(lib.egnn_sum_parts_f32(_ptr(part), rows, count, float(scale), _ptr(out), _stream()), "egnn_sum_parts_f32")
    return out


def edge_pool(u16, gate, pair_mask, b, n, k):
    """egnn_edge_pool_f32: (B, N, 16) sum over k of pair_mask * SiLU(u) * gate -- the pooled messages the backward's node-level part
    starts from.  gate = (gate_w (16) zero padded, gate_b (1)) or None; pair_mask (E) uint8 or None."""
    out = empty(b, n, 16, dtype=torch.float32, device=u16.device)
    with _timed("edge_pool"):
        rc = _abi.load().egnn_edge_pool_f32(_ptr(u16), _ptr(gate[0]) if gate is not None else None, _ptr(gate[1]) if gate is not None else None,
                                            _ptr(pair_mask), b, n, k, _ptr(out), _stream())
    _abi.check(rc, "egnn_edge_pool_f32")
    return out


def edge_tail_bwd(u16, coors, idx32, pair_mask, g_coors_out, g_msum16, w3p, b3p, w4p, b4, scale, eps, clamp, b, n, k, gate=None,
                  reduce=False, want_rel=False, drop=None, eid0=0):
    """egnn_edge_tail_bwd_f32 (include/egnn_hip.h): the per-edge closed-form backward behind edge_mlp's second Linear.
    Returns (gU (E, 16), g_rel (E, 4), g_hid (E, 64), a3 (E, 64), g_w (E,), g_scale (E,) or None); with gate = (gate_w (16) zero
    padded, gate_b (1)) -- soft_edges -- a seventh element g_gate (E,) = d loss / d (gate pre-activation).
    reduce: the kernel sums the parameter gradients' per-edge terms itself -- returns (gU, g_rel, sums (1192,), rel (E, 4) or None,
    dist (E,) or None, bits of max |gU| (1-element int32 tensor)): sums = [d/d W3 (64 x 16) | d/d b3 (64) | d/d W4 (64) | column sums of gU (16) | d/d gate_w (16) | d/d b4 |
    d/d CoorsNorm.scale | d/d gate_b | 0...]; want_rel: also x_i - x_j and |x_i - x_j|^2 per edge."""
    dev = u16.device
    e = b * n * k
    f32 = dict(dtype=torch.float32, device=dev)
    gu = empty(e, 16, **f32)
    g_rel = empty(e, 4, **f32)
    if reduce:
        lib = _abi.load()
        pf = lib.egnn_edge_tail_part_floats()
        n_waves = (e + 255) // 256 * 4
        part = empty(n_waves, pf, **f32)
        a = _abi.EdgeTailArgs()
        a.B, a.N, a.K, a.norm_coors = b, n, k, int(scale is not None)
        a.clamp = -1.0 if clamp is None else float(clamp)
        a.eps = float(eps)
        a.u, a.coors, a.idx, a.pair_mask = u16.data_ptr(), coors.data_ptr(), _ptr(idx32), _ptr(pair_mask)
        a.g_coors_out, a.g_msum = g_coors_out.data_ptr(), g_msum16.data_ptr()
        a.W3, a.b3, a.W4, a.b4, a.scale = w3p.data_ptr(), b3p.data_ptr(), w4p.data_ptr(), b4.data_ptr(), _ptr(scale)
        a.gU, a.g_rel, a.part = gu.data_ptr(), g_rel.data_ptr(), part.data_ptr()
        gu_bits = torch.empty(1, dtype=torch.int32, device=dev)
        a.amax_gu = gu_bits.data_ptr()
        if drop is not None:                                  # training-mode dropout in coors_mlp: the forward's hash mask (p, seed), edges numbered from eid0
            from . import _dropout
            a.drop_thr, a.drop_seed, a.drop_inv_keep, a.drop_eid0 = _dropout.threshold(drop[0]), int(drop[1]), _dropout.inv_keep(drop[0]), int(eid0)
        if gate is not None:
            a.gate_w, a.gate_b = gate[0].data_ptr(), gate[1].data_ptr()
        rel = dist = None
        if want_rel:
            rel, dist = empty(e, 4, **f32), empty(e, **f32)
            a.rel_out, a.dist_out = rel.data_ptr(), dist.data_ptr()
        with _timed("edge_tail_bwd"):
            rc = lib.egnn_edge_tail_bwd_f32(byref(a), _stream())
        _abi.check(rc, "egnn_edge_tail_bwd_f32")
        return gu, g_rel, sum_rows(part), rel, dist, gu_bits
    g_hid = empty(e, 64, **f32)
    a3 = empty(e, 64, **f32)
    g_w = empty(e, **f32)
    g_scale = empty(e, **f32) if scale is not None else None
    a = _abi.EdgeTailArgs()
    a.B, a.N, a.K, a.norm_coors = b, n, k, int(scale is not None)
    a.clamp = -1.0 if clamp is None else float(clamp)
    a.eps = float(eps)
    a.u, a.coors, a.idx, a.pair_mask = u16.data_ptr(), coors.data_ptr(), _ptr(idx32), _ptr(pair_mask)
    a.g_coors_out, a.g_msum = g_coors_out.data_ptr(), g_msum16.data_ptr()
    a.W3, a.b3, a.W4, a.b4, a.scale = w3p.data_ptr(), b3p.data_ptr(), w4p.data_ptr(), b4.data_ptr(), _ptr(scale)
    a.gU, a.g_rel, a.g_hid, a.a3, a.g_w, a.g_scale = gu.data_ptr(), g_rel.data_ptr(), g_hid.data_ptr(), a3.data_ptr(), g_w.data_ptr(), _ptr(g_scale)
    g_gate = None
    if gate is not None:
        g_gate = empty(e, **f32)
        a.gate_w, a.gate_b, a.g_gate = gate[0].data_ptr(), gate[1].data_ptr(), g_gate.data_ptr()
    with _timed("edge_tail_bwd"):
        rc = _abi.load().egnn_edge_tail_bwd_f32(byref(a), _stream())
    _abi.check(rc, "egnn_edge_tail_bwd_f32")
    if gate is not None:
        return gu, g_rel, g_hid, a3, g_w, g_scale, g_gate
    return gu, g_rel, g_hid, a3, g_w, g_scale


# rounds (of 128 entries) one workgroup of egnn_edge_bwd_pass_f32 streams through its column chunk: short enough that the
# workgroups of one graph and one chunk (they share the gathered rows) are many and run side by side on an XCD
ROUNDS_PER_SLAB = int(os.environ.get("EGNN_BWD_ROUNDS_PER_SLAB", "8"))


def edge_bwd_pass(w, proj, idx32, gu16, gu_scale, scal, ent, b, n, k, by_dest, ws_nat=None, want_w2=False, n_slabs=None, row_pairs=False,
                  drop=None, eid0=0, want_amax=False):
    """One pass of egnn_edge_bwd_pass_f32 (include/egnn_hip.h) over the entry list ent (autograd.entry_list).  proj = (B*N, 2 Hp)
    fp32 P_i | P_j rows.  Returns a dict: rows (L / 16, Hp) partial rows, one per tile; with want_w2: w2 = d/d W_2 (16, Hp); with ws_nat (the
    natural-units scalar weights (Hp, S)): ws = d/d W_s (Hp, S) and scal = d/d scalars (E, S) -- the partial arrays of the
    kernel already summed (fixed order).  row_pairs (by source, 16 < K <= 32, with ws_nat): rows = one row per node (L / 32, Hp)."""
    lib = _abi.load()
    hp, s_in = w["Hp"], w["S"]
    dev = proj.device
    e = b * n * k
    l = ent.numel()
    if n_slabs is None:
        n_slabs = max(1, (l // 128 + ROUNDS_PER_SLAB - 1) // ROUNDS_PER_SLAB)
    n_rows = l // 32 if row_pairs else l // 16
    rows = empty(n_rows + 1, hp, dtype=torch.float32, device=dev)
    a = _abi.EdgeBwdArgs()
    a.B, a.N, a.K, a.Hp, a.S, a.by_dest, a.n_slabs = b, n, k, hp, s_in, int(by_dest), n_slabs
    a.wst_terms = w["Wst"].shape[1]
    a.L, a.E = l, e
    a.ent = ent.data_ptr()
    a.Pi, a.Pj, a.ldp = proj.data_ptr(), proj.data_ptr() + 4 * hp, proj.stride(0)
    a.Wst, a.ws_inv_scale = w["Wst"].data_ptr(), w["ws_inv_scale"]
    a.idx = _ptr(idx32)
    a.W2Th, a.gU, a.gu_scale = w["W2Th"].data_ptr(), gu16.data_ptr(), gu_scale
    a.inv_scale = 1.0 / (gu_scale * w["w2t_scale"])
    a.scal = scal.data_ptr()
    a.part_rows, a.ld_rows = rows.data_ptr(), hp
    a.row_pairs = int(row_pairs)
    amax_bits = None
    if want_amax and (s_in == 1 or ws_nat is None):           # (by-product of the fp32 tile sums: max |rows|)
        amax_bits = torch.empty(1, dtype=torch.int32, device=dev)
        a.rows_amax = amax_bits.data_ptr()
    if drop is not None:                                      # training-mode dropout in edge_mlp: the forward's hash mask (p, seed)
        from . import _dropout
        a.drop_thr, a.drop_seed, a.drop_inv_keep, a.drop_eid0 = _dropout.threshold(drop[0]), int(drop[1]), _dropout.inv_keep(drop[0]), int(eid0)
    if want_w2:
        dw2 = empty(n_slabs, 16, hp, dtype=torch.float32, device=dev)
        a.dW2_part = dw2.data_ptr()
    if ws_nat is not None:
        ch = lib.egnn_edge_bwd_chunk_steps()
        n_chunks = (hp // 32 + ch - 1) // ch
        ws_nat = ws_nat.contiguous()
        dws = empty(n_slabs * 16 if s_in > 1 else n_slabs, s_in, hp, dtype=torch.float32, device=dev)
        ds = empty(n_chunks, e, s_in, dtype=torch.float32, device=dev)
        a.Ws, a.dWs_part, a.ds_part = ws_nat.data_ptr(), dws.data_ptr(), ds.data_ptr()
        if s_in > 5:                                          # d/d s on the matrix cores: W_s^T fragments (_weights.pack)
            a.WsTh, a.wst_inv_scale = w["WsTh"].data_ptr(), 1.0 / w["wst_scale"]
        if s_in > 1:
            # d/d W_s on the matrix cores: the scalars enter as split-fp16 fragments, each column brought into [1, 2) at its maximum
            amax = scal.abs().amax(dim=0)
            col_scale = torch.where(amax > 0, torch.exp2(-torch.floor(torch.log2(amax.clamp_min(1e-30)))), torch.ones_like(amax)).contiguous()
            a.scal_scale = col_scale.data_ptr()
    wb = lib.egnn_edge_bwd_work_bytes(l, s_in, int(want_w2), int(ws_nat is not None))
    work = empty(max(wb, 16), dtype=torch.uint8, device=dev)
    a.work, a.work_bytes = work.data_ptr(), wb
    with _timed("edge_bwd_by_dest" if by_dest else "edge_bwd_by_src"):
        rc = lib.egnn_edge_bwd_pass_f32(byref(a), _stream())
    _abi.check(rc, "egnn_edge_bwd_pass_f32")
    out = {"rows": rows[:n_rows], "amax_bits": amax_bits}
    if want_w2:
        out["w2"] = dw2.sum(dim=0)          # (340 MB of per-slab partials at the north-star shape: 0.19 ms; `sum_rows` measured 0.21)
    if ws_nat is not None:
        if s_in > 1:                                          # one partial per wave (every 4th row of the array), times col_scale
            out["ws"] = (dws.view(n_slabs * 4, 4, s_in, hp)[:, 0].sum(dim=0) / col_scale[:, None]).t().contiguous()
        else:
            out["ws"] = dws.sum(dim=0).t().contiguous()
        out["scal"] = ds.sum(dim=0)
    return out


# ---------------------------------------------------------------------------------------------- whole-layer C interface
def pack_weights_c(layer):
    """egnn_pack_weights_host on the module's parameters: (desc, info, blob uint8 CPU tensor).  The Python module itself
    re-lays its weights on the device with _weights.pack (same layout, bit-identical: tests/test_host_logic.py); this is the
    path a binding without torch takes, and what `forward_c` uses."""
    import ctypes
    lib = _abi.load()
    desc = _abi.layer_desc(layer)
    sd = {k.replace(".", "_"): v.detach().float().cpu().contiguous() for k, v in layer.state_dict().items()}
    params = _abi.LayerParams()
    for f in _abi.PARAM_FIELDS:
        setattr(params, f, sd[f].data_ptr() if f in sd else None)
    nbytes = lib.egnn_packed_weights_bytes(byref(desc))
    if nbytes == 0:
        raise _abi.EGNNHipError("egnn_packed_weights_bytes: descriptor outside what the gfx950 kernels are built for")
    blob = torch.zeros(nbytes, dtype=torch.uint8)
    info = _abi.PackedInfo()
    _abi.check(lib.egnn_pack_weights_host(byref(desc), byref(params), blob.data_ptr(), byref(info)), "egnn_pack_weights_host")
    assert info.bytes == nbytes
    return desc, info, blob


def pack_weights_blob(layer, w, device):
    """The blob of the C whole-layer entry assembled ON THE DEVICE from the module's own re-laid-out tensors `w` (_weights.pack: the same
    formats, bit for bit -- tests/test_host_logic.py::test_c_weight_packer_matches_python holds the piece-by-piece mapping used here):
    (desc, info, blob uint8 on `device`).  ~0.3 ms of small copies where egnn_pack_weights_host needs 60 ms at dim 512.
    Raises EGNNHipError for a descriptor outside what the C entry is built for."""
    lib = _abi.load()
    desc = _abi.layer_desc(layer)
    info = _abi.PackedInfo()
    _abi.check(lib.egnn_packed_layout(byref(desc), byref(info)), "egnn_packed_layout")
    if (info.H, info.Hp, info.S) != (w["H"], w["Hp"], w["S"]) or 4 * info.NM != w["Wst"].shape[1]:
        raise _abi.EGNNHipError("egnn_packed_layout disagrees with _weights.pack")
    blob = torch.zeros(info.bytes, dtype=torch.uint8, device=device)

    def put(off, t):
        flat = t.to(device).contiguous().reshape(-1).view(torch.uint8)
        blob[off:off + flat.numel()].copy_(flat)

    def put_split(key, o_hi, o_lo, rows_c):
        hi, lo, inv, rows = w[key]
        if rows != rows_c:
            raise _abi.EGNNHipError(f"{key}: {rows} packed rows, the C layout expects {rows_c}")
        put(o_hi, hi)
        put(o_lo, lo)
        return float(inv)

    info.wcat_inv_scale = put_split("Wcat_split", info.wcat_hi, info.wcat_lo, info.wcat_rows)
    put(info.bcat, w["bcat"])
    put(info.wst, w["Wst"])
    put(info.w2h, w["W2h"])
    put(info.b2, w["b2"])
    info.ws_inv_scale, info.w2_inv_scale = float(w["ws_inv_scale"]), float(w["w2_inv_scale"])
    if "gate_w" in w:
        put(info.gate_w, w["gate_w"])
        put(info.gate_b, w["gate_b"])
    if "W3h" in w:
        put(info.w3h, w["W3h"])
        put(info.b3, w["b3"])
        put(info.w4, w["W4"])
        put(info.b4, w["b4"])
        info.w3_inv_scale = float(w["w3_inv_scale"])
    if "coors_scale" in w:
        put(info.coors_scale, w["coors_scale"])
    if "W5_split" in w:
        info.w5_inv_scale = put_split("W5_split", info.w5_hi, info.w5_lo, info.w5_rows)
        info.w6_inv_scale = put_split("W6_split", info.w6_hi, info.w6_lo, info.w6_rows)
        put(info.b5, w["b5"])
        put(info.b6, w["b6"])
    if "gamma" in w:
        put(info.gamma, w["gamma"])
        put(info.beta, w["beta"])
    return desc, info, blob


def forward_c(layer, feats, coors, edges=None, mask=None, adj_mat=None, packed=None):
    """One EGNN.forward through the single-call C entry egnn_layer_forward_f32 (what a non-Python binding uses); returns
    (node_out, coors_out).  `packed` = (desc, info, blob_on_device) to reuse a previous pack."""
    lib = _abi.load()
    if packed is None:
        desc, info, blob = pack_weights_c(layer)
        packed = (desc, info, blob.to(feats.device))
    desc, info, blob_dev = packed
    b, n, _ = feats.shape
    feats, coors = feats.contiguous(), coors.contiguous()
    if edges is not None:
        edges = edges.contiguous().float()
    m8, a8 = _u8(mask), _u8(adj_mat)
    stride = n * n if (a8 is not None and a8.dim() == 3) else 0
    k = layer.num_nearest_neighbors
    if adj_mat is not None and layer.only_sparse_neighbors:
        k = adj_max_degree(adj_mat)
    if not (layer.num_nearest_neighbors > 0 or layer.only_sparse_neighbors):
        k = n
    nbytes = lib.egnn_workspace_bytes(byref(desc), b, n, min(k, n))
    ws = empty(max(nbytes, 1), dtype=torch.uint8, device=feats.device)
    node_out, coors_out = torch.empty_like(feats), torch.empty_like(coors)
    rc = lib.egnn_layer_forward_f32(byref(desc), byref(info), _ptr(blob_dev), _ptr(feats), _ptr(coors), _ptr(edges), _ptr(m8),
                                    _ptr(a8), stride, b, n, k, coors.shape[-1], _ptr(node_out), _ptr(coors_out), _ptr(ws),
                                    nbytes, _status_ptr(feats.device), _stream())
    _abi.check(rc, "egnn_layer_forward_f32")
    return node_out, coors_out
