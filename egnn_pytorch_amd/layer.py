"""Drop-in `EGNN` / `EGNN_Network` modules whose forward runs on the gfx950 HIP kernels.

Boundary mirrored: constructor keywords, `forward` signatures (note the different
positional order of the two entry points), parameter names / shapes of the reference's
`state_dict`, error behaviour (asserts at construction, `topk` out-of-range when K > N).
Reference: egnn_pytorch/egnn_pytorch.py:148-341 (EGNN), :343-454 (EGNN_Network).

The forward pass runs ONLY on a CUDA(HIP) device, in fp32 (or, for a module converted with .double(), in float64 on the float64
kernels).  There is no CPU or PyTorch-eager fallback for it: anything the kernels do not cover raises.  Under autograd (grad mode on
and an input or a parameter requires grad) the same HIP forward is wrapped in an autograd.Function whose backward runs on the HIP
kernels (the E x H work of every shape the fused forward covers) or recomputes the layer a few graphs at a time
(egnn_pytorch_amd/autograd.py).
"""
from __future__ import annotations

import contextlib
import contextvars
import os
import warnings

import torch
from torch import nn

from . import _abi, _dropout, _ops, _weights, autograd as _autograd
from .attention import GlobalLinearAttention

_SPATIAL_ORDER = os.environ.get("EGNN_SPATIAL_ORDER", "1") != "0"     # scheduling knob only; results do not depend on it
_SLOT_PREP = os.environ.get("EGNN_SLOT_PREP", "1") != "0"             # per-slot records for the edge pass's setup (same results)
_SIDE_STREAM = os.environ.get("EGNN_SIDE_STREAM", "1") != "0"         # neighbour selection beside the projection GEMM
# EGNN_Network: the next layer's neighbour selection started right behind this layer's edge pass (it needs the new coordinates only).  Off:
# measured at c3 / c5 full size it is worth -1.8 % / +0.5 % -- the selection already runs beside the next layer's projection, and two
# kernels that share the CUs share their throughput (profiles/r04_experiments/network_selection_look_ahead.txt)
_PREFETCH = os.environ.get("EGNN_PREFETCH", "0") != "0"
# egnn_edge_args.algo: 0 = the library chooses (persistent wave-per-node kernel where it applies), 1 = the general edge kernel always
# (A/B measurements; tests/test_gpu_kernels.py checks the two against each other)
_EDGE_ALGO = int(os.environ.get("EGNN_EDGE_ALGO", "0"))
# Arithmetic.  "fast" (default): split-fp16 products on the matrix cores (fp32-class accuracy, values up to fp16's 65504 at the cast
# sites); a call that leaves that range is re-run automatically in plain fp32 -- the reference's arithmetic class, several times
# slower -- when the range status word is read synchronously (EGNN_RANGE_CHECK=sync, the default) and no graph is being recorded;
# otherwise it raises EGNNRangeError.  "exact": always the plain-fp32 kernels (inference only).
_PRECISION = os.environ.get("EGNN_PRECISION", "fast")
_DENSE_PW = os.environ.get("EGNN_DENSE_PW", "1") != "0"            # dense layers with N % 32 == 0 on the wave-per-node edge kernel
_SHARED_FEATS_IMAGE = os.environ.get("EGNN_SHARED_FEATS_IMAGE", "1") != "0"   # 0: a second packed image of feats for the projection
_NODE_MLP_FUSED = os.environ.get("EGNN_NODE_MLP_FUSED", "1") != "0"   # node_mlp of narrow layers in one launch (csrc/node_mlp_fused.hip)
_ENTRY_FORK = os.environ.get("EGNN_ENTRY_FORK", "1") != "0"        # ... which then waits for an event recorded at the layer's entry, not for them
_LATE_SELECT = os.environ.get("EGNN_LATE_SELECT", "1") != "0"      # node-level launches before the neighbour selection (see _forward_hip)
# Inference forwards as ONE call of the C whole-layer entry (egnn_layer_forward_opts_f32: the same kernels in the same order, enqueued from
# C) instead of a dozen Python-side launches: ~200 us of host time per forward become a few tens.  Only with every scheduling switch
# above at its default -- the C entry implements the default policy -- and never while per-kernel timing is on.
_C_FORWARD = os.environ.get("EGNN_C_FORWARD", "1") != "0"
_PROJ_ROW_MASK = os.environ.get("EGNN_PROJ_ROW_MASK", "1") != "0"     # projection GEMM skips M-tiles of padded nodes (inference)


def _default_policy():
    """every scheduling switch at its default (read at call time: tests flip them on the module)"""
    return (_SPATIAL_ORDER and _SLOT_PREP and _EDGE_ALGO == 0 and _SHARED_FEATS_IMAGE and _NODE_MLP_FUSED and _ENTRY_FORK and _LATE_SELECT
            and not _PREFETCH and _PROJ_ROW_MASK)
_exact_now = contextvars.ContextVar("egnn_exact_now", default=False)       # per thread / context: concurrent forwards do not see each other's
_warned_rerun = False


@contextlib.contextmanager
def exact_arithmetic():
    """Run the enclosed forwards on the plain-fp32 (wide-range) kernels: include/egnn_hip.h, "The wide-range path"."""
    tok = _exact_now.set(True)
    try:
        yield
    finally:
        _exact_now.reset(tok)


def exact_active():
    return _exact_now.get() or _PRECISION == "exact"


def _warn_rerun(err):
    """Once per process: the automatic plain-fp32 re-run is several times slower than the fast path and costs a host synchronisation."""
    global _warned_rerun
    if not _warned_rerun:
        _warned_rerun = True
        warnings.warn("egnn_pytorch_amd: a forward left the range of the split-fp16 fast path and is being re-run on the plain-fp32 "
                      f"kernels (several times slower; this warning is issued once). Cause: {err}", RuntimeWarning, stacklevel=3)


def _rerun_exact_ok(module, *tensors):
    """A forward that tripped the range status word may be re-run in plain fp32: the word was read for THIS call (sync mode) and no
    stream capture is going on.  Under autograd too (round 5): the re-run's forward is the plain-fp32 kernels and its backward the
    recompute path -- plain fp32 ATen arithmetic over the neighbour list those kernels selected, no fp16 cast site anywhere -- so that
    reference-legal inputs such as feats x 1e6 TRAIN instead of raising (slower: it is the wide-range path).  With training-mode
    dropout as well: the plain kernels evaluate the same hash masks (the re-run draws a fresh seed, like any other forward)."""
    if _ops.RANGE_CHECK != "sync" or exact_active():
        return False
    return not torch.cuda.is_current_stream_capturing()
# The kernels compute in fp32-class arithmetic (split-f16 products, fp32 accumulation: DESIGN.md §2).  Other floating dtypes
# -- the reference is dtype-generic and its own tests run in float64 -- are accepted at the boundary: inputs are converted to
# fp32, outputs back to the callers' dtype.  For bf16 / fp16 that is at least the reference's precision; for float64 it is
# NOT (products carry ~22 significant bits): the reference's 1e-6 fp64 equivariance bars hold only for small weights.
_FLOAT_DTYPES = (torch.float32, torch.float64, torch.float16, torch.bfloat16)


class EdgeLookup:
    """Per-pair edge features given as look-up tables instead of a materialised (B,N,N,edge_dim) tensor: what EGNN_Network's
    front-end produces (egnn_pytorch.py:410-432) -- [edge_emb(edge tokens) or float edges | adj_emb(adjacency-degree labels)].
    The edge kernel reads the tables for the K selected pairs of each node (include/egnn_hip.h: egnn_edge_args.edge_tok ...);
    inference only (under autograd the network materialises the tensor so that the embeddings receive gradients)."""

    def __init__(self, edges=None, tok=None, tok_emb=None, deg=None, deg_emb=None):
        self.edges = None if edges is None else edges.contiguous().float()
        self.tok = None if tok is None else tok.contiguous().long()
        self.tok_emb = None if tok_emb is None else tok_emb.detach().contiguous().float()
        self.deg = None if deg is None else deg.contiguous()
        self.deg_emb = None if deg_emb is None else deg_emb.detach().contiguous().float()
        self.d1 = self.tok_emb.shape[1] if self.tok is not None else (self.edges.shape[-1] if self.edges is not None else 0)
        self.d2 = self.deg_emb.shape[1] if self.deg is not None else 0

    @property
    def width(self):
        return self.d1 + self.d2


def _mlp(d_in, d_hidden, d_out, dropout, final_act):
    """Linear -> dropout|Identity -> SiLU -> Linear [-> SiLU]; indices 0 and 3 hold the Linears, which is
    what gives the reference's state_dict keys (`edge_mlp.0.*`, `edge_mlp.3.*`, ...)."""
    mods = [nn.Linear(d_in, d_hidden), dropout, nn.SiLU(), nn.Linear(d_hidden, d_out)]
    if final_act:
        mods.append(nn.SiLU())
    return nn.Sequential(*mods)


class CoorsNorm(nn.Module):
    """Holder of the learned `scale` of the reference's CoorsNorm (egnn_pytorch.py:67-77); the
    normalisation itself (x / max(|x|, eps) * scale) happens inside the fused edge kernel."""

    def __init__(self, eps=1e-8, scale_init=1.0):
        super().__init__()
        self.eps = eps
        self.scale = nn.Parameter(torch.full((1,), float(scale_init)))


class EGNN(nn.Module):
    def __init__(self, dim, edge_dim=0, m_dim=16, fourier_features=0, num_nearest_neighbors=0,
                 dropout=0.0, init_eps=1e-3, norm_feats=False, norm_coors=False,
                 norm_coors_scale_init=1e-2, update_feats=True, update_coors=True,
                 only_sparse_neighbors=False, valid_radius=float("inf"), m_pool_method="sum",
                 soft_edges=False, coor_weights_clamp_value=None):
        super().__init__()
        assert m_pool_method in {"sum", "mean"}, "pool method must be either sum or mean"
        assert update_feats or update_coors, "you must update either features, coordinates, or both"

        self.dim = dim
        self.edge_dim = edge_dim
        self.m_dim = m_dim
        self.fourier_features = fourier_features
        self.num_nearest_neighbors = num_nearest_neighbors
        self.only_sparse_neighbors = only_sparse_neighbors
        self.valid_radius = valid_radius
        self.m_pool_method = m_pool_method
        self.coor_weights_clamp_value = coor_weights_clamp_value
        self.norm_feats = norm_feats
        self.norm_coors = norm_coors
        self.dropout_p = dropout
        self.init_eps = init_eps

        edge_input_dim = 2 * fourier_features + 2 * dim + edge_dim + 1
        drop = nn.Dropout(dropout) if dropout > 0 else nn.Identity()     # one shared instance, as upstream

        self.edge_mlp = _mlp(edge_input_dim, 2 * edge_input_dim, m_dim, drop, final_act=True)
        self.edge_gate = nn.Sequential(nn.Linear(m_dim, 1), nn.Sigmoid()) if soft_edges else None
        self.node_norm = nn.LayerNorm(dim) if norm_feats else nn.Identity()
        self.coors_norm = CoorsNorm(scale_init=norm_coors_scale_init) if norm_coors else nn.Identity()
        self.node_mlp = _mlp(dim + m_dim, 2 * dim, dim, drop, final_act=False) if update_feats else None
        self.coors_mlp = _mlp(m_dim, 4 * m_dim, 1, drop, final_act=False) if update_coors else None

        for mod in self.modules():                                       # upstream init_: N(0, init_eps)
            if type(mod) is nn.Linear:
                nn.init.normal_(mod.weight, std=init_eps)

        self._packed = None
        self._packed_key = None

    def __getstate__(self):
        """copy.deepcopy / pickle / torch.save(module): without the kernel-side caches (re-laid-out weights on the device, the C entry's
        blob, the cached parameter list) -- they are rebuilt at the copy's first forward."""
        st = self.__dict__.copy()
        st["_packed"] = st["_packed_key"] = None
        for k in ("_c_packed", "_param_cache", "_param_cache_mods", "_param_cache_tree", "_param_cache_counts"):
            st.pop(k, None)
        return st

    # ------------------------------------------------------------------ kernel-side weights
    def packed_weights(self):
        key = _weights.version_key(self)
        if self._packed is None or key != self._packed_key:
            self._packed = _weights.pack(self)
            self._packed_key = key
        return self._packed

    # ------------------------------------------------------------------ forward
    def _check_inputs(self, feats, coors, edges, mask, adj_mat):
        if not feats.is_cuda:
            raise RuntimeError("egnn_pytorch_amd.EGNN runs only on an MI355X (cuda/HIP) device; "
                               "there is no CPU fallback (move the module and inputs with .cuda())")
        if feats.dtype not in _FLOAT_DTYPES or coors.dtype not in _FLOAT_DTYPES:
            raise NotImplementedError(f"the gfx950 path takes floating-point feats / coors (got {feats.dtype}/{coors.dtype})")
        if feats.dim() != 3 or coors.dim() != 3 or feats.shape[:2] != coors.shape[:2]:
            raise ValueError(f"feats {tuple(feats.shape)} / coors {tuple(coors.shape)}: expected (B,N,dim) and (B,N,C)")
        if not 1 <= coors.shape[-1] <= 64:
            raise NotImplementedError("the gfx950 path supports coordinate dimensions 1..64 (beyond 8 on the plain kernels)")
        if feats.shape[-1] != self.dim:
            raise ValueError(f"feats last dim {feats.shape[-1]} != dim {self.dim}")
        if (edges is not None) != (self.edge_dim > 0):
            raise ValueError("`edges` must be passed if and only if edge_dim > 0")
        b, n = feats.shape[:2]
        if isinstance(edges, EdgeLookup):
            if edges.width != self.edge_dim:
                raise ValueError(f"edge look-up tables give {edges.width} features per pair, edge_dim is {self.edge_dim}")
        elif edges is not None and tuple(edges.shape) != (b, n, n, self.edge_dim):
            raise ValueError(f"edges shape {tuple(edges.shape)} != {(b, n, n, self.edge_dim)}")
        if mask is not None and tuple(mask.shape) != (b, n):
            raise ValueError(f"mask shape {tuple(mask.shape)} != {(b, n)}")
        # the kernels take raw device pointers: a tensor on another device (or on the host) would be read as garbage, not rejected
        for name, t in (("coors", coors), ("mask", mask), ("adj_mat", adj_mat), ("edges", None if isinstance(edges, EdgeLookup) else edges)):
            if t is not None and t.device != feats.device:
                raise RuntimeError(f"Expected all tensors to be on the same device: feats is on {feats.device}, {name} on {t.device}")

    def forward(self, feats, coors, edges=None, mask=None, adj_mat=None):
        out = None
        try:
            out = self._call(feats, coors, edges, mask, adj_mat, None)[:2]
            _ops.range_check_after_forward(feats.device)            # EGNN_RANGE_CHECK: sync (default) | deferred | off
        except _abi.EGNNRangeError as err:
            if err.origin == "backward" or not _rerun_exact_ok(self, feats, coors, edges):
                raise
            _warn_rerun(err)
            out = None                                              # (under autograd the first attempt holds u and the projection table)
            with exact_arithmetic():                                # reference-legal inputs beyond the fp16 cast sites: plain fp32
                out = self._call(feats, coors, edges, mask, adj_mat, None)[:2]
        return out

    def _call(self, feats, coors, edges, mask, adj_mat, order_hint, presel=None, prefetch=None):
        """(node_out, coors_out, order): inference under no_grad, or -- when a graph has to be recorded -- through
        autograd.EGNNFunction (HIP forward, recompute-in-backward).  presel / prefetch: EGNN_Network's look-ahead of the neighbour
        selection (`_select_neighbors`), inference only."""
        if _ops.RANGE_CHECK == "deferred" and feats.is_cuda:
            _ops.check_range(feats.device, wait=False)              # an earlier call's status, if it has arrived
        if _autograd.wants_grad(self, feats, coors, edges):
            node_out, coors_out = _autograd.EGNNFunction.apply(self, order_hint, mask, adj_mat, feats, coors, edges,
                                                               *self.parameters())
            order = None                                            # (scheduling hint only; recomputed by the next layer)
        else:
            with torch.no_grad():
                node_out, coors_out, order = self._forward_with_hint(feats, coors, edges, mask, adj_mat, order_hint, presel=presel,
                                                                     prefetch=prefetch)[:3]
        return node_out, coors_out, order

    def _forward_hip_checked(self, feats, coors, edges, mask, adj_mat, order_hint, want_u=False, drop_seed=None):
        """(node_out, coors_out, order, idx, rank, valid_radius, u, proj) -- what autograd.EGNNFunction.forward needs; u = the
        (B*N*K, 16) pre-activation of edge_mlp's second SiLU when `want_u` (the native backward differentiates from it), proj =
        ((B*N, 2 Hp) projection table P_i | P_j, pi_split) the edge pass read (kept for the backward instead of a second GEMM)."""
        return self._forward_with_hint(feats, coors, edges, mask, adj_mat, order_hint, want_u=want_u, drop_seed=drop_seed, selection=True)

    def dropout_active(self):
        """training mode with dropout > 0: every forward draws a fresh mask seed (egnn_pytorch_amd/_dropout.py)"""
        return self.training and self.dropout_p > 0

    def _forward_with_hint(self, feats, coors, edges, mask, adj_mat, order_hint, want_u=False, drop_seed=None, presel=None, prefetch=None,
                           selection=False):
        """forward + the scheduling permutation it used (EGNN_Network hands layer 0's on to the next layers).  selection: the caller
        reads the neighbour list (idx, rank) from the returned tuple -- the one-call C forward keeps it in its workspace."""
        self._check_inputs(feats, coors, edges, mask, adj_mat)
        _abi.load()
        f_dtype, c_dtype = feats.dtype, coors.dtype
        if self.float64_kernels():
            # a float64 module computes in float64, as the reference does (its own tests run in float64, tests/test_equivariance.py:6):
            # the plain kernels instantiated for double (include/egnn_hip.h, "The float64 path")
            if isinstance(edges, EdgeLookup):
                raise NotImplementedError("float64 modules take the materialised (B,N,N,edge_dim) edge features")
            with torch.cuda.device(feats.device):
                if self.dropout_active() and drop_seed is None:
                    drop_seed = _dropout.draw_seed()
                out = self._forward_exact(feats.double(), coors.double(), None if edges is None else edges.double(), mask, adj_mat,
                                          dtype=torch.float64, want_u=want_u,
                                          drop=(self.dropout_p, drop_seed) if self.dropout_active() else None)
            if f_dtype != torch.float64 or c_dtype != torch.float64:
                out = (out[0].to(f_dtype), out[1].to(c_dtype)) + tuple(out[2:])
            return out
        if f_dtype == torch.float64 or c_dtype == torch.float64:
            _warn_float64_once()
        if (_C_FORWARD and _default_policy() and not want_u and not selection and drop_seed is None and presel is None and prefetch is None and f_dtype == torch.float32
                and c_dtype == torch.float32 and _ops._timer is None and not self.dropout_active() and not exact_active()):
            out = self._forward_c(feats, coors, edges, mask, adj_mat, order_hint)
            if out is not None:
                return out
        with torch.cuda.device(feats.device):
            if self.dropout_active() and drop_seed is None:
                drop_seed = _dropout.draw_seed()
            out = self._forward_hip(feats.float(), coors.float(),
                                    edges if (edges is None or isinstance(edges, EdgeLookup)) else edges.float(), mask, adj_mat, order_hint,
                                    want_u=want_u, drop=(self.dropout_p, drop_seed) if self.dropout_active() else None,
                                    presel=presel, prefetch=prefetch)
        if f_dtype != torch.float32 or c_dtype != torch.float32:
            out = (out[0].to(f_dtype), out[1].to(c_dtype)) + tuple(out[2:])
        return out

    def compute_dtype(self):
        """float64 for a module converted with .double() (every kernel of its forward then is a float64 kernel), else float32 (half /
        bfloat16 modules are converted at the boundary: at least their own precision)."""
        p = next(self.parameters(), None)
        return torch.float64 if (p is not None and p.dtype == torch.float64) else torch.float32

    def float64_kernels(self):
        """A float64 module runs on the float64 kernels (training-mode dropout included: the plain kernels evaluate the same hash masks)."""
        return self.compute_dtype() == torch.float64

    def _select_outputs(self, coors, adj_mat, order_hint, k):
        """The four outputs of the k-NN selection (idx, rank, order or None, slots or None), allocated on the LAUNCH stream."""
        b, n = coors.shape[:2]
        dev = coors.device
        want_order = adj_mat is None and 64 <= n <= 4096 and _SPATIAL_ORDER and coors.shape[-1] == 3
        have_hint = order_hint is not None and tuple(order_hint.shape) == (b, n)
        return (_ops.empty(b, n, k, dtype=torch.int32, device=dev), _ops.empty(b, n, k, dtype=torch.float32, device=dev),
                _ops.empty(b, n, dtype=torch.int32, device=dev) if (want_order and not have_hint) else None,
                _ops.empty(b * n * k, 4, dtype=torch.int32, device=dev) if (_SLOT_PREP and coors.shape[-1] == 3) else None)

    def _select_neighbors(self, coors, mask, adj_mat, order_hint, fork=None):
        """(idx, rank, order, slots, K, valid_radius) of egnn_pytorch.py:230-260 for fp32 coordinates on the device: the K nearest
        neighbours (None, None, None, None, N on the dense path), the Morton order the edge pass schedules by, the per-slot records.
        Launched on the side stream (EGNN_SIDE_STREAM=0: the current one); whoever consumes the result joins that stream first.
        fork = (outputs of _select_outputs, event): the side stream waits for that EVENT of the launch stream -- recorded at the layer's
        entry, before its node-level launches -- instead of for the stream's tail, so the selection runs beside node_prep and the
        projection although it is enqueued behind them (the outputs were allocated at the same point: a block the allocator recycles
        from a tensor freed AFTER the event could still be in use by a launch the side stream does not wait for)."""
        b, n = coors.shape[:2]
        num_nearest = self.num_nearest_neighbors
        valid_radius = self.valid_radius
        use_nearest = num_nearest > 0 or self.only_sparse_neighbors
        idx = rank = order = slots = None
        if not use_nearest:
            # dense all-pairs.  With N % 32 == 0 and the standard layer's shape the wave-per-node edge kernel applies (csrc/edge_pw.hip):
            # it reads per-slot records, here with j = k.  Only when the batch fills the chip with one wave per node (measured, round 5:
            # B N = 16 384 ... 32 768 nodes -2 ... -6.5 %; BASELINE.json's c2 -- 2048 nodes, each wave walking 8 rounds -- +25 %, so c2
            # stays on the general kernel, which splits a node's slots over workgroups).  The records are 16 B N^2 bytes: bounded.
            if (_SLOT_PREP and _DENSE_PW and b * n >= 8192 and n % 32 == 0 and 32 <= n <= 4096 and coors.shape[-1] == 3 and self.m_dim <= 16
                    and self.edge_dim == 0 and self.fourier_features == 0 and not self.dropout_active() and b * n * n * 16 <= (1 << 28)):
                slots = _ops.slot_prep(coors, _ops._u8(mask), None, None, None, valid_radius)
            return None, None, None, slots, n, valid_radius
        if adj_mat is not None and self.only_sparse_neighbors:
            num_nearest = _ops.adj_max_degree(adj_mat)                # host sync, as upstream (:249)
            valid_radius = 0.0
        k = num_nearest
        if k > n:
            raise RuntimeError("selected index k out of range")      # torch.topk's error upstream
        if k > 0:
            # Neighbour selection (VALU / scalar bound, no MFMA) and the Morton order (one workgroup per graph) depend on
            # the coordinates only; the projection GEMM that follows (MFMA bound) depends on the features only.  Forked onto
            # a side stream they share the CUs instead of queueing (EGNN_SIDE_STREAM=0: one stream); joined before the
            # edge pass.
            # k-NN path: neighbours are spatial -> workgroups that own Morton-adjacent nodes share gathered rows in L1.
            # Scheduling only (results do not depend on it), so a stack of layers reuses the first layer's order:
            # coordinates move by small steps per layer and the locality survives.
            mask8 = _ops._u8(mask)
            want_order = adj_mat is None and 64 <= n <= 4096 and _SPATIAL_ORDER and coors.shape[-1] == 3
            have_hint = order_hint is not None and tuple(order_hint.shape) == (b, n)

            # The four outputs are allocated HERE, on the launch stream, and written by the side stream: the launch stream waits for the
            # side stream before it reads them, so when they are freed every use is ordered before the launch stream's later work and the
            # allocator may reuse the blocks at once.  (Allocated inside the fork they belonged to the side stream's pool and needed
            # record_stream(): four calls per forward and frees deferred to events.)
            idx_o, rank_o, order_o, slots_o = fork[0] if fork is not None else self._select_outputs(coors, adj_mat, order_hint, k)

            def select():
                # (the Morton order launched AHEAD of the selection was measured in round 6: +- 0.3 %, not kept)
                idx_, rank_ = _ops.knn_select(coors, mask, adj_mat, k, out=(idx_o, rank_o))
                order_ = (order_hint if have_hint else _ops.spatial_order(coors, out=order_o, mask8=mask8)) if want_order else None
                # the edge pass's setup as one coalesced record per slot instead of a chain of dependent loads
                slots_ = _ops.slot_prep(coors, mask8, idx_, rank_, order_, valid_radius, out=slots_o) if slots_o is not None else None
                return idx_, rank_, order_, slots_

            # (not while per-kernel timing is on: events on two streams would charge one kernel's wait to another)
            use_side = _SIDE_STREAM and _ops._timer is None
            side = _ops.side_stream(coors.device) if use_side else None
            if side is not None:
                if fork is not None and fork[1] is not None:
                    side.wait_event(fork[1])
                else:
                    side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    idx, rank, order, slots = select()
            else:
                idx, rank, order, slots = select()
        return idx, rank, order, slots, k, valid_radius

    def _c_state(self, device):
        """(desc, info, blob on `device`, fused node_mlp image or None) of the C whole-layer entry for the current parameters; None when the
        C entry does not cover this layer's shape."""
        w = self.packed_weights()                                    # (re-laid out on the device once per parameter version)
        key = (self._packed_key, device)
        st = self.__dict__.get("_c_packed")
        if st is None or st[0] != key:
            try:
                desc, info, blob_dev = _ops.pack_weights_blob(self, w, device)
            except _abi.EGNNHipError:
                st = (key, None)
            else:
                img = w.get("nmf_img")
                st = (key, (desc, info, blob_dev, img.to(device) if img is not None else None))
            self.__dict__["_c_packed"] = st
        return st[1]

    def _forward_c(self, feats, coors, edges, mask, adj_mat, order_hint):
        """The fp32 inference forward as one call of egnn_layer_forward_opts_f32 (include/egnn_hip.h; csrc/layer_api.hip mirrors
        `_forward_hip_impl` launch for launch, so the outputs are the same bits).  Returns `_forward_hip`'s tuple, or None when the call is
        outside what the C entry covers (edge look-up tables, wide shapes, the wave-per-node kernel on dense batches, empty inputs, another
        current device): the Python launch sequence below then runs it."""
        b, n, dim = feats.shape
        cdim = coors.shape[-1]
        use_nearest = self.num_nearest_neighbors > 0 or self.only_sparse_neighbors
        dev = feats.device
        if (isinstance(edges, EdgeLookup) or b == 0 or n == 0 or cdim > 8 or self.m_dim > 64
                or 2 * self.fourier_features + 1 + self.edge_dim > 16 or dev.index != torch.cuda.current_device()
                or (adj_mat is not None and (not adj_mat.is_cuda or adj_mat.dtype != torch.bool))):
            return None
        if not use_nearest and (b * n >= 8192 and n % 32 == 0 and 32 <= n <= 4096 and cdim == 3 and self.m_dim <= 16 and self.edge_dim == 0
                                and self.fourier_features == 0 and _DENSE_PW and b * n * n * 16 <= (1 << 28)):
            return None                                             # (dense batches that fill the chip: the wave-per-node kernel)
        st = self._c_state(dev)
        if st is None:
            return None
        desc, info, blob_dev, img = st
        valid_radius = self.valid_radius
        if use_nearest:
            k = self.num_nearest_neighbors
            if adj_mat is not None and self.only_sparse_neighbors:
                k = _ops.adj_max_degree(adj_mat)                    # host sync, as upstream (:249)
                valid_radius = 0.0
            if k > n:
                raise RuntimeError("selected index k out of range")  # torch.topk's error upstream
            if k == 0:
                return None
        else:
            k = n
        feats, coors = feats.contiguous(), coors.contiguous()
        if edges is not None:
            edges = edges.contiguous().float()
        m8, a8 = _ops._u8(mask), _ops._u8(adj_mat)
        stride = n * n if (a8 is not None and a8.dim() == 3) else 0
        lib = _abi.load()
        nbytes = lib.egnn_workspace_bytes(desc, b, n, k)
        # every buffer is allocated HERE, on the launch stream (the side stream writes the selection's part of the workspace between the
        # two events of this call: `_select_neighbors` has the argument)
        ws = _ops.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        node_out = _ops.empty(b, n, dim, dtype=torch.float32, device=dev) if self.node_mlp is not None else feats
        coors_out = _ops.empty(b, n, cdim, dtype=torch.float32, device=dev) if self.coors_mlp is not None else coors
        opts = _abi.ForwardOpts()
        order = None
        if use_nearest and adj_mat is None and 64 <= n <= 4096 and cdim == 3:
            if order_hint is not None and tuple(order_hint.shape) == (b, n) and order_hint.dtype == torch.int32 and order_hint.is_contiguous():
                order, opts.order_is_hint = order_hint, 1
            else:
                order = _ops.empty(b, n, dtype=torch.int32, device=dev)
            opts.order = order.data_ptr()
        if use_nearest and _SIDE_STREAM:
            opts.side_stream, opts.ev_fork, opts.ev_join = _ops.side_handles(dev)
        if img is not None:
            opts.nmf_img = img.data_ptr()
        rc = lib.egnn_layer_forward_opts_f32(desc, info, blob_dev.data_ptr(), feats.data_ptr(), coors.data_ptr(), _ops._ptr(edges), _ops._ptr(m8),
                                             _ops._ptr(a8), stride, b, n, k, cdim, node_out.data_ptr(), coors_out.data_ptr(), ws.data_ptr(),
                                             nbytes, _ops._status_ptr(dev), _ops._stream(), opts)
        if rc != 0 and opts.side_stream:
            # the selection may still be writing into the workspace on the side stream: join it before the buffers are freed
            torch.cuda.current_stream().wait_stream(_ops.side_stream(dev))
        _abi.check(rc, "egnn_layer_forward_opts_f32")
        return node_out, coors_out, order, None, None, valid_radius, None, None

    def _forward_hip(self, feats, coors, edges, mask, adj_mat, order_hint=None, want_u=False, drop=None, presel=None, prefetch=None):
        try:
            return self._forward_hip_impl(feats, coors, edges, mask, adj_mat, order_hint, want_u, drop, presel, prefetch)
        except BaseException:
            # the neighbour selection may still be writing its outputs on the side stream: join it before they are freed (they belong
            # to the launch stream's pool: `_select_neighbors`)
            if _SIDE_STREAM and feats.is_cuda:
                torch.cuda.current_stream().wait_stream(_ops.side_stream(feats.device))
            raise

    def _forward_hip_impl(self, feats, coors, edges, mask, adj_mat, order_hint, want_u, drop, presel, prefetch):
        # more per-edge scalars than the split-fp16 edge kernels carry (2 fourier + 1 + edge_dim > 16, up to 160): the plain-fp32 kernels
        # ... and more than 8 coordinates (the fused kernels keep x_i - x_j in registers up to 8)
        # ... and heads wider than 64 message channels (the fused kernels hold up to four 16-channel accumulator tiles per edge tile)
        wide_shape = 2 * self.fourier_features + 1 + self.edge_dim > 16 or coors.shape[-1] > 8 or self.m_dim > 64
        if exact_active() or wide_shape:
            return self._forward_exact(feats, coors, edges, mask, adj_mat, want_u=want_u, drop=drop)
        b, n, dim = feats.shape
        w = self.packed_weights()
        feats = feats.contiguous()
        coors = coors.contiguous()
        feats2d = feats.view(b * n, dim)
        lookup = edges if isinstance(edges, EdgeLookup) else None
        if lookup is not None:
            edges = lookup.edges
        elif edges is not None:
            edges = edges.contiguous().float()
        mask8 = _ops._u8(mask)

        # ---- neighbour selection (egnn_pytorch.py:230-260)
        use_nearest = self.num_nearest_neighbors > 0 or self.only_sparse_neighbors
        if b == 0 or (n == 0 and not use_nearest):
            # empty batch / empty dense graphs: the reference returns empty outputs (N = 0 on the k-NN path: topk's error)
            return torch.empty_like(feats), torch.empty_like(coors), None, None, None, self.valid_radius, None, None
        # (EGNN_Network hands over what the previous layer started on the side stream right behind its edge pass)
        # When K is known without looking at the data (the k-NN path; not only_sparse_neighbors, whose K is the adjacency's maximum degree),
        # the node-level launches go FIRST and the selection -- on the side stream, needed by the edge pass only -- is enqueued behind
        # them: the first kernel of the forward starts ~30 us earlier, which is what a synchronous range check exposes per call.
        sel = presel
        late_select = (_LATE_SELECT and sel is None and use_nearest and 0 < self.num_nearest_neighbors <= n
                       and not (adj_mat is not None and self.only_sparse_neighbors))
        if sel is None and not late_select:
            sel = self._select_neighbors(coors, mask, adj_mat, order_hint)
        fork = None
        if late_select and _SIDE_STREAM and _ENTRY_FORK and _ops._timer is None:
            # the selection is enqueued behind the node-level launches but depends on nothing they write: fork point = here
            fork = (self._select_outputs(coors, adj_mat, order_hint, self.num_nearest_neighbors), _ops.fork_event(feats.device))
        idx = rank = order = slots = None
        k, valid_radius = self.num_nearest_neighbors, self.valid_radius
        if sel is not None:
            idx, rank, order, slots, k, valid_radius = sel

        node_out, coors_out = feats, coors
        node_in = u_pre = proj_kept = None
        if k > 0:
            # ---- node-level projections P = feats [W_i ; W_j]^T + [b1 ; 0]
            # (K >= 6: the edge pass feeds P_i to its first-layer MFMA as (fp16 hi, fp16 lo) words)
            hp = w["Hp"]
            pi_split = k >= 6
            if self.node_mlp is not None:
                # one pass over feats: its (hi, lo) split for the projection AND [LayerNorm(feats) | 0] for node_mlp
                # (egnn_pytorch.py:335-336); the edge pass drops m_i into the zero columns
                # (node_norm = Identity, the reference's default: the two are the same values -- the projection reads the first
                # `dim` columns of the [feats | 0] image, egnn_linear_hl_lda_f32; 134 MB less to write at the north-star shape)
                shared = _SHARED_FEATS_IMAGE and w.get("gamma") is None
                if shared:
                    node_in = feats_hl = _ops.node_prep_hl(feats2d, None, None, None, 1e-5, self.m_dim)
                else:
                    node_in, feats_hl = _ops.node_prep_hl(feats2d, None, w.get("gamma"), w.get("beta"), w.get("ln_eps", 1e-5),
                                                          self.m_dim, with_raw=True)
            else:
                feats_hl = _ops.split_f16(feats2d)
            # (a padded batch: the rows of padded nodes are read by masked-out edges only -- whole tiles of them are not computed.
            # Not under autograd: the backward differentiates through every edge)
            proj = _ops.linear_hl(feats_hl, w["Wcat_split"], 2 * hp, w["bcat"], name="node_proj",
                                  split_cols=hp if pi_split else 0,
                                  row_mask=mask8.view(-1) if (mask8 is not None and not want_u and drop is None and _PROJ_ROW_MASK
                                                              and use_nearest and pi_split and _SLOT_PREP and _EDGE_ALGO == 0
                                                              and _abi.load().egnn_edge_pw_covers(b, n, k, w["S"], self.fourier_features,
                                                                                                  self.edge_dim, self.m_dim, coors.shape[-1],
                                                                                                  2 * hp) == 1) else None)
            del feats_hl
            if sel is None:
                sel = self._select_neighbors(coors, mask, adj_mat, order_hint, fork=fork)
            idx, rank, order, slots, k_sel, valid_radius = sel
            assert k_sel == k
            side_join = use_nearest and k > 0 and _SIDE_STREAM and _ops._timer is None
            a = _abi.EdgeArgs()
            a.B, a.N, a.K, a.dim, a.m_dim = b, n, k, dim, self.m_dim
            a.H, a.Hp = w["H"], hp
            a.fourier, a.edge_dim, a.S, a.pi_split = self.fourier_features, self.edge_dim, w["S"], int(pi_split)
            a.Pi = proj.data_ptr()
            a.Pj = proj.data_ptr() + 4 * hp
            a.ldp = 2 * hp
            a.Wst, a.W2h, a.b2 = w["Wst"].data_ptr(), w["W2h"].data_ptr(), w["b2"].data_ptr()
            a.ws_inv_scale, a.wst_terms = w["ws_inv_scale"], w["Wst"].shape[1]
            a.w2_inv_scale = w["w2_inv_scale"]
            if self.edge_gate is not None:
                a.gate_w, a.gate_b = w["gate_w"].data_ptr(), w["gate_b"].data_ptr()
            if self.coors_mlp is not None:
                a.W3h, a.b3, a.W4, a.b4 = (w[x].data_ptr() for x in ("W3h", "b3", "W4", "b4"))
                a.w3_inv_scale = w["w3_inv_scale"]
                coors_out = _ops.empty(*coors.shape, dtype=coors.dtype, device=coors.device)
                a.coors_out = coors_out.data_ptr()
            if self.norm_coors:
                a.coors_scale = w["coors_scale"].data_ptr()
            a.coors, a.coor_dim = coors.data_ptr(), coors.shape[-1]
            if lookup is not None:
                # the features of the K selected pairs only, read by the kernel in neighbour-list order
                if side_join:
                    torch.cuda.current_stream().wait_stream(_ops.side_stream(feats.device))
                edges = _ops.edge_features_gather(lookup, idx, b, n, k)
                a.edges_by_k = 1
            a.edges = _ops._ptr(edges)
            a.mask = _ops._ptr(mask8)
            if side_join:
                torch.cuda.current_stream().wait_stream(_ops.side_stream(feats.device))
            a.idx, a.rank = _ops._ptr(idx), _ops._ptr(rank)
            a.order, a.slots = _ops._ptr(order), _ops._ptr(slots)
            a.valid_radius = float(min(valid_radius, 3.0e38))
            cv = self.coor_weights_clamp_value
            a.clamp = -1.0 if cv is None else float(cv)
            a.pool_mean = int(self.m_pool_method == "mean")
            if self.node_mlp is not None:
                a.node_hi, a.node_lo, a.node_kp = node_in.hi.data_ptr(), node_in.lo.data_ptr(), node_in.kp
            if drop is not None:                                  # training-mode dropout: the mask is a hash, see _dropout.py
                a.drop_thr, a.drop_seed, a.drop_inv_keep = _dropout.threshold(drop[0]), int(drop[1]), _dropout.inv_keep(drop[0])
            if want_u:                                            # (rows of whole 16-channel accumulator tiles, pad channels 0)
                u_pre = _ops.empty(b * n * k, 16 * _weights.m_blocks(self.m_dim), dtype=torch.float32, device=feats.device)
                a.U_out = u_pre.data_ptr()
            a.algo = _EDGE_ALGO
            _ops.edge_fused(a, feats.device)
            if prefetch is not None:
                # the next layer's neighbour selection needs this layer's coordinates and nothing else: started here, on the side
                # stream, it runs beside this layer's node_mlp and the next layer's projection instead of in front of them
                prefetch(coors_out, order)
            if u_pre is not None:
                proj_kept = (proj, pi_split)
            del proj
        elif self.node_mlp is not None:                                   # K == 0: no messages, m_i = 0
            node_in = _ops.node_prep_hl(feats2d, None, w.get("gamma"), w.get("beta"), w.get("ln_eps", 1e-5), self.m_dim)

        # ---- node update (egnn_pytorch.py:335-337)
        if self.node_mlp is not None:
            if _NODE_MLP_FUSED and drop is None and "nmf_img" in w:
                # narrow layers (dim <= 256): both Linears in one launch, the hidden activation stays in registers
                node_out = _ops.node_mlp_fused(node_in, w["nmf_img"], w["W5_split"][2], w["b5"], w["W6_split"][2], w["b6"], feats2d,
                                               dim, self.m_dim).view(b, n, dim)
            else:
                hid = _ops.linear_hl(node_in, w["W5_split"], 2 * dim, w["b5"], act=1, out_f32=False, out_hl=True,
                                     name="node_mlp0", drop=drop)
                node_out = _ops.linear_hl(hid, w["W6_split"], dim, w["b6"], residual=feats2d, name="node_mlp1").view(b, n, dim)
        return node_out, coors_out, order, idx, rank, valid_radius, u_pre, proj_kept


    def _forward_exact(self, feats, coors, edges, mask, adj_mat, dtype=torch.float32, want_u=False, drop=None):
        """The layer on the plain-fp32 kernels (include/egnn_hip.h, "The wide-range path"): exact-fp32 GEMMs, fp32 node_norm, the edge
        pass as fp32 VALU arithmetic on the module's own weight tensors.  Same neighbour selection, same return tuple as _forward_hip.
        dtype = torch.float64: the same kernels instantiated for double ("The float64 path"; feats / coors / edges are float64).
        drop = (p, seed): training-mode dropout -- the fused kernels' hash masks at the three sites (egnn_edge_exact_args.drop_*,
        egnn_drop_silu_*)."""
        esz = 8 if dtype == torch.float64 else 4
        b, n, dim = feats.shape
        feats = feats.contiguous()
        coors = coors.contiguous()
        feats2d = feats.view(b * n, dim)
        lookup = edges if isinstance(edges, EdgeLookup) else None
        if lookup is not None:
            edges = lookup.edges
        elif edges is not None:
            edges = edges.contiguous().to(dtype)
        mask8 = _ops._u8(mask)
        num_nearest, valid_radius = self.num_nearest_neighbors, self.valid_radius
        use_nearest = num_nearest > 0 or self.only_sparse_neighbors
        idx = rank = None
        if b == 0 or (n == 0 and not use_nearest):
            return torch.empty_like(feats), torch.empty_like(coors), None, None, None, valid_radius, None, None
        if use_nearest:
            if adj_mat is not None and self.only_sparse_neighbors:
                num_nearest = _ops.adj_max_degree(adj_mat)
                valid_radius = 0.0
            k = num_nearest
            if k > n:
                raise RuntimeError("selected index k out of range")
            if k > 0:
                idx, rank = _ops.knn_select(coors, mask, adj_mat, k)
        else:
            k = n
        f32 = lambda t: t.detach().to(dtype).contiguous()                # noqa: E731  (the module's tensors in the compute dtype)
        node_out, coors_out, m_i = feats, coors, None
        u_pre = proj_keep = None                                         # want_u (forward under autograd): u (E, m_dim) and the projection table
        if k > 0:
            lin0, lin3 = self.edge_mlp[0], self.edge_mlp[3]
            w1 = f32(lin0.weight)                                        # (H, Din): [W_i | W_j | scalar columns]
            h, din = w1.shape
            hq = (h + 3) // 4 * 4
            proj = _ops.empty(b * n, 2 * hq, dtype=dtype, device=feats.device)
            _ops.linear_f32(feats2d, w1, h, dim, bias=f32(lin0.bias), out=proj[:, :h], name="node_proj_f32")
            _ops.linear_f32(feats2d, w1[:, dim:], h, dim, out=proj[:, hq:hq + h], name="node_proj_f32")
            a = _abi.EdgeExactArgs()
            a.B, a.N, a.K, a.m_dim, a.H = b, n, k, self.m_dim, h
            a.fourier, a.edge_dim, a.coor_dim = self.fourier_features, self.edge_dim, coors.shape[-1]
            a.pool_mean = int(self.m_pool_method == "mean")
            a.Pi, a.Pj, a.ldp = proj.data_ptr(), proj.data_ptr() + esz * hq, 2 * hq
            a.Ws, a.ldws = w1.data_ptr() + esz * 2 * dim, din
            keep = [w1, f32(lin3.weight), f32(lin3.bias)]
            a.W2, a.b2 = keep[1].data_ptr(), keep[2].data_ptr()
            if self.edge_gate is not None:
                keep += [f32(self.edge_gate[0].weight).view(-1), f32(self.edge_gate[0].bias)]
                a.gate_w, a.gate_b = keep[-2].data_ptr(), keep[-1].data_ptr()
            if self.coors_mlp is not None:
                c0, c3 = self.coors_mlp[0], self.coors_mlp[3]
                keep += [f32(c0.weight), f32(c0.bias), f32(c3.weight).view(-1), f32(c3.bias)]
                a.W3, a.b3, a.W4, a.b4 = (t.data_ptr() for t in keep[-4:])
                coors_out = _ops.empty(*coors.shape, dtype=dtype, device=coors.device)
                a.coors_out = coors_out.data_ptr()
            if self.norm_coors:
                keep.append(f32(self.coors_norm.scale))
                a.coors_scale = keep[-1].data_ptr()
            a.coors = coors.data_ptr()
            if lookup is not None:
                edges = _ops.edge_features_gather(lookup, idx, b, n, k)
                a.edges_by_k = 1
            a.edges, a.mask = _ops._ptr(edges), _ops._ptr(mask8)
            a.idx, a.rank = _ops._ptr(idx), _ops._ptr(rank)
            a.valid_radius = float(valid_radius) if dtype == torch.float64 else float(min(valid_radius, 3.0e38))
            cv = self.coor_weights_clamp_value
            a.clamp = -1.0 if cv is None else float(cv)
            if self.node_mlp is not None:
                m_i = _ops.empty(b * n, self.m_dim, dtype=dtype, device=feats.device)
                a.m_i = m_i.data_ptr()
            if want_u:
                u_pre = _ops.empty(b * n * k, self.m_dim, dtype=dtype, device=feats.device)
                a.U_out = u_pre.data_ptr()
                proj_keep = (proj, False)
            _ops.set_drop(a, drop)
            _ops.edge_exact(a, feats.device, dtype)
            del proj, keep
        elif self.node_mlp is not None:
            m_i = torch.zeros(b * n, self.m_dim, dtype=dtype, device=feats.device)        # K == 0: no messages
        if self.node_mlp is not None:
            ln = self.node_norm if isinstance(self.node_norm, nn.LayerNorm) else None
            node_in = _ops.node_prep_f32(feats2d, m_i, f32(ln.weight) if ln is not None else None,
                                         f32(ln.bias) if ln is not None else None, ln.eps if ln is not None else 1e-5, self.m_dim)
            n0, n3 = self.node_mlp[0], self.node_mlp[3]
            if drop is None:
                hid = _ops.linear_f32(node_in, f32(n0.weight), 2 * dim, dim + self.m_dim, bias=f32(n0.bias), act=1, name="node_mlp0_f32")
            else:                                                        # Linear -> Dropout -> SiLU (egnn_pytorch.py:196-201)
                hid = _ops.drop_silu_(_ops.linear_f32(node_in, f32(n0.weight), 2 * dim, dim + self.m_dim, bias=f32(n0.bias), name="node_mlp0_f32"), drop)
            node_out = _ops.linear_f32(hid, f32(n3.weight), dim, 2 * dim, bias=f32(n3.bias), residual=feats2d,
                                       name="node_mlp1_f32").view(b, n, dim)
        return node_out, coors_out, None, idx, rank, valid_radius, u_pre, proj_keep


_FP64_WARNED = False


def _warn_float64_once():
    """float64 callers get float64 tensors back, computed with fp32-class arithmetic (split-fp16 products, ~22 significant
    bits, fp32 accumulation): say so once per process instead of silently."""
    global _FP64_WARNED
    if _FP64_WARNED:
        return
    _FP64_WARNED = True
    import warnings
    warnings.warn("egnn_pytorch_amd: float64 inputs / modules are converted at the boundary and computed with fp32-class "
                  "arithmetic on the gfx950 kernels (about 22 significant bits; results are returned as float64). "
                  "The reference computes float64 in float64.", RuntimeWarning, stacklevel=4)


class EGNN_Network(nn.Module):
    """Stack of EGNN layers (egnn_pytorch.py:343-454).  The layer loop -- the part BASELINE.json's
    configs exercise -- runs on the HIP kernels; the token / position / edge / adjacency-degree
    front-end is ordinary tensor plumbing on the device, and so is the optional induced-set attention between
    layers (`global_linear_attn_every > 0`, egnn_pytorch_amd/attention.py: outside the per-edge hot path)."""

    def __init__(self, *, depth, dim, num_tokens=None, num_edge_tokens=None, num_positions=None,
                 edge_dim=0, num_adj_degrees=None, adj_dim=0, global_linear_attn_every=0,
                 global_linear_attn_heads=8, global_linear_attn_dim_head=64, num_global_tokens=4,
                 **kwargs):
        super().__init__()
        assert not (num_adj_degrees is not None and num_adj_degrees < 1), \
            "make sure adjacent degrees is greater than 1"
        self.num_positions = num_positions
        self.token_emb = nn.Embedding(num_tokens, dim) if num_tokens is not None else None
        self.pos_emb = nn.Embedding(num_positions, dim) if num_positions is not None else None
        self.edge_emb = nn.Embedding(num_edge_tokens, edge_dim) if num_edge_tokens is not None else None
        self.has_edges = edge_dim > 0
        self.num_adj_degrees = num_adj_degrees
        self.adj_emb = nn.Embedding(num_adj_degrees + 1, adj_dim) \
            if (num_adj_degrees is not None and adj_dim > 0) else None
        edge_dim = edge_dim if self.has_edges else 0
        adj_dim = adj_dim if num_adj_degrees is not None else 0
        has_global_attn = global_linear_attn_every > 0
        self.global_tokens = nn.Parameter(torch.randn(num_global_tokens, dim)) if has_global_attn else None

        self.layers = nn.ModuleList()
        for ind in range(depth):
            # index 0 of each pair is the optional attention block: keys `layers.{l}.0.*` / `layers.{l}.1.*` (:381-388)
            is_global = has_global_attn and ind % global_linear_attn_every == 0
            self.layers.append(nn.ModuleList([
                GlobalLinearAttention(dim=dim, heads=global_linear_attn_heads,
                                      dim_head=global_linear_attn_dim_head) if is_global else None,
                EGNN(dim=dim, edge_dim=edge_dim + adj_dim, norm_feats=True, **kwargs)]))

    def forward(self, feats, coors, adj_mat=None, edges=None, mask=None, return_coor_changes=False):
        if _ops.RANGE_CHECK == "deferred" and coors.is_cuda:
            _ops.check_range(coors.device, wait=False)
        # under autograd the embeddings / attention blocks are ordinary differentiable modules and every EGNN layer records
        # its own autograd.Function; otherwise nothing is recorded
        grad = torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or
                                            any(torch.is_tensor(t) and t.is_floating_point() and t.requires_grad
                                                for t in (feats, coors, edges)))
        try:
            with torch.enable_grad() if grad else torch.no_grad():
                out = self._forward(feats, coors, adj_mat, edges, mask, return_coor_changes)
            _ops.range_check_after_forward(coors.device)            # once per network forward, not per layer
        except _abi.EGNNRangeError as err:
            if err.origin == "backward" or not _rerun_exact_ok(self, feats, coors, edges):
                raise
            _warn_rerun(err)
            with exact_arithmetic(), (torch.enable_grad() if grad else torch.no_grad()):    # the whole stack again, in plain fp32
                out = self._forward(feats, coors, adj_mat, edges, mask, return_coor_changes)
        return out

    def _forward(self, feats, coors, adj_mat, edges, mask, return_coor_changes):
        b = feats.shape[0]
        if self.token_emb is not None:
            feats = self.token_emb(feats)
        if self.pos_emb is not None:
            n = feats.shape[1]
            assert n <= self.num_positions, \
                f"given sequence length {n} must be less than the number of positions {self.num_positions} set at init"
            feats = feats + self.pos_emb(torch.arange(n, device=feats.device))[None]
        # Edge features for the layers.  Inference: look-up tables (EdgeLookup) -- the (B,N,N,edge_dim+adj_dim) tensor of :410-432
        # is never materialised, the edge kernel reads the embedding rows of the K selected pairs of each node.  Under autograd:
        # the tensor, so that the embeddings receive gradients.
        lazy = not torch.is_grad_enabled() and not any(l.float64_kernels() for _, l in self.layers)       # (depth = 0: an empty loop, as upstream)
        tok = tok_emb = None
        if edges is not None and self.edge_emb is not None:
            if lazy:
                tok, tok_emb, edges = edges, self.edge_emb.weight, None
            else:
                edges = self.edge_emb(edges)

        if self.num_adj_degrees is not None:
            assert adj_mat is not None, "adjacency matrix must be passed in (keyword argument adj_mat)"
            if not adj_mat.is_cuda:
                raise RuntimeError("egnn_pytorch_amd.EGNN_Network runs only on an MI355X (cuda/HIP) device")
            # N-degree expansion (egnn_pytorch.py:414-427) as bit-set algebra on the device instead of float matmuls
            adj_mat, adj_indices = _ops.adj_expand(adj_mat, b, self.num_adj_degrees)
            if self.adj_emb is not None:
                if lazy:
                    edges = EdgeLookup(edges=edges, tok=tok, tok_emb=tok_emb, deg=adj_indices, deg_emb=self.adj_emb.weight)
                    tok = None
                else:
                    adj_emb = self.adj_emb(adj_indices.long())
                    edges = torch.cat((edges, adj_emb), dim=-1) if edges is not None else adj_emb
        if tok is not None:                                         # edge tokens without adjacency degrees
            edges = EdgeLookup(tok=tok, tok_emb=tok_emb)

        global_tokens = None
        if self.global_tokens is not None:
            global_tokens = self.global_tokens[None].expand(b, -1, -1)
        coor_changes = [coors]
        order = None
        # Inference look-ahead (EGNN_PREFETCH=1; off by default, see _PREFETCH): layer l + 1's neighbour selection depends on layer l's
        # coordinates only, which its edge pass writes -- before node_mlp.  Started right behind that edge pass on the side stream, it
        # runs beside node_mlp (and the next layer's attention block / projection) as well.
        look_ahead = (_PREFETCH and not torch.is_grad_enabled() and coors.is_cuda and coors.dtype == torch.float32
                      and _SIDE_STREAM and _ops._timer is None and not exact_active())
        pending = [None]
        layers = list(self.layers)
        for li, (global_attn, egnn) in enumerate(layers):
            if global_attn is not None:
                feats, global_tokens = global_attn(feats, global_tokens, mask=mask)           # :445-446
            presel, pending[0] = pending[0], None
            prefetch = None
            if look_ahead and li + 1 < len(layers):
                nxt = layers[li + 1][1]
                if (nxt.num_nearest_neighbors > 0 or nxt.only_sparse_neighbors) and not nxt.float64_kernels():
                    def prefetch(coors_out, order_used, nxt=nxt):
                        pending[0] = nxt._select_neighbors(coors_out, mask, adj_mat, order_used)
            feats, coors, order = egnn._call(feats, coors, edges, mask, adj_mat, order, presel=presel, prefetch=prefetch)
            coor_changes.append(coors)
        if return_coor_changes:
            return feats, coors, coor_changes
        return feats, coors
