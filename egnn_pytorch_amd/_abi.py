"""ctypes binding of libegnn_hip.so (C ABI declared in include/egnn_hip.h).

The library is the product: there is NO fallback.  If it is missing or does not export the
symbols of the header this module raises, and every forward call raises with it.

`import torch` happens before `ctypes.CDLL` on purpose: torch bundles its own libamdhip64.so.7
(same SONAME as /opt/rocm's); loading torch first makes our library bind to the HIP runtime torch
already initialised, so `torch.cuda.current_stream().cuda_stream` and `tensor.data_ptr()` are valid
inside the library (two HIP runtimes in one process would make stream handles unusable).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_void_p

import torch  # noqa: F401  (must precede CDLL, see module docstring)

ABI_VERSION = 37
_LIB_NAME = "libegnn_hip.so"
_PKG_DIR = os.path.dirname(os.path.abspath(__file__))

# every symbol include/egnn_hip.h declares
SYMBOLS = (
    "egnn_abi_version", "egnn_error_string", "egnn_padded_hidden", "egnn_knn_select_f32",
    "egnn_adj_max_degree_u8",
    "egnn_edge_fused_f32", "egnn_spatial_order_f32", "egnn_linear_hl_f32", "egnn_split_f16", "egnn_node_prep_hl",
    "egnn_packed_halves", "egnn_adj_expand_u8", "egnn_adj_expand_workspace_bytes", "egnn_edge_mfmas",
    "egnn_packed_weights_bytes", "egnn_packed_layout", "egnn_pack_weights_host", "egnn_workspace_bytes", "egnn_layer_forward_f32", "egnn_layer_forward_opts_f32",
    "egnn_edge_bwd_pass_f32", "egnn_edge_bwd_chunk_steps", "egnn_edge_bwd_work_bytes", "egnn_edge_tail_bwd_f32", "egnn_edge_tail_part_floats", "egnn_edge_pool_f32", "egnn_rows_gather_sum_f32", "egnn_edge_features_gather_f32",
    "egnn_induced_attn_f32", "egnn_token_attn_f32", "egnn_slot_prep_f32", "egnn_spatial_order_masked_f32", "egnn_struct_bytes",
    "egnn_dest_lists_capacity", "egnn_dest_lists_i32", "egnn_split_scaled_f16", "egnn_linear_hl_splitk_f32", "egnn_sum_parts_f32", "egnn_absmax_f32", "egnn_unsplit_words_f32", "egnn_split_scaled_both_f16", "egnn_split_scaled_colsum_rows", "egnn_drop_silu_f32", "egnn_drop_silu_f64", "egnn_silu_bwd_f32", "egnn_silu_bwd_drop_f32",
    "egnn_linear_hl_drop_f32", "egnn_linear_hl_lda_f32", "egnn_linear_hl_lda_rows_f32", "egnn_edge_pw_covers",
    "egnn_linear_f32", "egnn_node_prep_f32", "egnn_edge_exact_f32", "egnn_edge_exact_workspace_bytes",
    "egnn_knn_select_f64", "egnn_linear_f64", "egnn_node_prep_f64", "egnn_edge_exact_f64",
    "egnn_edge_exact_bwd_f32", "egnn_edge_exact_bwd_f64", "egnn_edge_exact_node_sums_f32", "egnn_edge_exact_node_sums_f64",
    "egnn_edge_tail_exact_bwd_f32", "egnn_edge_tail_exact_bwd_f64", "egnn_status_publish",
    "egnn_node_mlp_fused_halves", "egnn_node_mlp_fused_pack_f16", "egnn_node_mlp_fused_f32",
)


class EdgeArgs(Structure):
    """Mirror of `struct egnn_edge_args` (include/egnn_hip.h) -- field order and types must match."""
    _fields_ = [
        ("B", c_int32), ("N", c_int32), ("K", c_int32), ("dim", c_int32), ("m_dim", c_int32),
        ("H", c_int32), ("Hp", c_int32), ("fourier", c_int32), ("edge_dim", c_int32),
        ("S", c_int32), ("pi_split", c_int32),
        ("Pi", c_void_p), ("Pj", c_void_p), ("ldp", c_int64),
        ("Wst", c_void_p), ("wst_terms", c_int32), ("ws_inv_scale", c_float), ("W2h", c_void_p), ("w2_inv_scale", c_float), ("b2", c_void_p),
        ("gate_w", c_void_p), ("gate_b", c_void_p),
        ("W3h", c_void_p), ("w3_inv_scale", c_float), ("b3", c_void_p), ("W4", c_void_p), ("b4", c_void_p),
        ("coors_scale", c_void_p),
        ("coors", c_void_p), ("coor_dim", c_int32), ("edges", c_void_p), ("mask", c_void_p), ("idx", c_void_p), ("rank", c_void_p),
        ("order", c_void_p),
        ("valid_radius", c_float), ("clamp", c_float), ("pool_mean", c_int32),
        ("m_i", c_void_p), ("coors_out", c_void_p),
        ("node_hi", c_void_p), ("node_lo", c_void_p), ("node_kp", c_int32),
        ("status", c_void_p),
        ("U_out", c_void_p),
        ("edges_by_k", c_int32),
        ("slots", c_void_p),
        ("drop_thr", ctypes.c_uint32), ("drop_seed", ctypes.c_uint32), ("drop_inv_keep", c_float),
        ("algo", c_int32),
    ]


class EdgeExactArgs(Structure):
    """Mirror of `struct egnn_edge_exact_args` (include/egnn_hip.h): the edge pass in plain fp32 (wide-range path) / float64."""
    _fields_ = [
        ("B", c_int32), ("N", c_int32), ("K", c_int32), ("m_dim", c_int32), ("H", c_int32), ("fourier", c_int32),
        ("edge_dim", c_int32), ("coor_dim", c_int32), ("pool_mean", c_int32), ("edges_by_k", c_int32),
        ("Pi", c_void_p), ("Pj", c_void_p), ("ldp", c_int64), ("Ws", c_void_p), ("ldws", c_int64),
        ("W2", c_void_p), ("b2", c_void_p), ("gate_w", c_void_p), ("gate_b", c_void_p),
        ("W3", c_void_p), ("b3", c_void_p), ("W4", c_void_p), ("b4", c_void_p), ("coors_scale", c_void_p),
        ("coors", c_void_p), ("edges", c_void_p), ("mask", c_void_p), ("idx", c_void_p), ("rank", c_void_p),
        ("valid_radius", c_double), ("clamp", c_double),
        ("m_i", c_void_p), ("coors_out", c_void_p), ("edge_ws", c_void_p), ("U_out", c_void_p),
        ("drop_thr", c_uint32), ("drop_seed", c_uint32), ("drop_inv_keep", c_float), ("drop_eid0", c_int64),
    ]


class EdgeExactBwdArgs(Structure):
    """Mirror of `struct egnn_edge_exact_bwd_args` (include/egnn_hip.h): the E x H work of the plain-fp32 / float64 backward."""
    _fields_ = [
        ("B", c_int32), ("N", c_int32), ("K", c_int32), ("m_dim", c_int32), ("H", c_int32), ("fourier", c_int32),
        ("edge_dim", c_int32), ("coor_dim", c_int32), ("edges_by_k", c_int32), ("reserved", c_int32),
        ("Pi", c_void_p), ("Pj", c_void_p), ("ldp", c_int64), ("Ws", c_void_p), ("ldws", c_int64),
        ("W2", c_void_p), ("coors", c_void_p), ("edges", c_void_p), ("idx", c_void_p), ("gU", c_void_p),
        ("A_T", c_void_p), ("DZ_T", c_void_p), ("g_scal", c_void_p),
        ("drop_thr", c_uint32), ("drop_seed", c_uint32), ("drop_inv_keep", c_float), ("drop_eid0", c_int64),
    ]


class EdgeTailExactArgs(Structure):
    """Mirror of `struct egnn_edge_tail_exact_args` (include/egnn_hip.h): the per-edge chain behind u in closed form, any head / C."""
    _fields_ = [
        ("B", c_int32), ("N", c_int32), ("K", c_int32), ("m_dim", c_int32), ("coor_dim", c_int32), ("norm_coors", c_int32),
        ("eps", c_double), ("clamp", c_double),
        ("u", c_void_p), ("coors", c_void_p), ("idx", c_void_p), ("pair_mask", c_void_p), ("g_coors_out", c_void_p), ("g_msum", c_void_p),
        ("W3", c_void_p), ("b3", c_void_p), ("W4", c_void_p), ("b4", c_void_p), ("scale", c_void_p), ("gate_w", c_void_p), ("gate_b", c_void_p),
        ("gU", c_void_p), ("g_rel", c_void_p), ("ghid_t", c_void_p), ("a3_t", c_void_p), ("mm_t", c_void_p), ("m0_t", c_void_p),
        ("g_w", c_void_p), ("g_scale", c_void_p), ("g_gate", c_void_p),
        ("drop_thr", c_uint32), ("drop_seed", c_uint32), ("drop_inv_keep", c_float), ("drop_eid0", c_int64),
    ]


class EdgeBwdArgs(Structure):
    """Mirror of `struct egnn_edge_bwd_args` (include/egnn_hip.h) -- field order and types must match."""
    _fields_ = [
        ("B", c_int32), ("N", c_int32), ("K", c_int32), ("Hp", c_int32), ("S", c_int32), ("by_dest", c_int32),
        ("n_slabs", c_int32), ("wst_terms", c_int32),
        ("L", c_int64), ("E", c_int64),
        ("ent", c_void_p), ("Pi", c_void_p), ("Pj", c_void_p), ("ldp", c_int64),
        ("Wst", c_void_p), ("ws_inv_scale", c_float), ("idx", c_void_p), ("W2Th", c_void_p), ("gU", c_void_p),
        ("gu_scale", c_float), ("inv_scale", c_float), ("scal", c_void_p), ("Ws", c_void_p), ("scal_scale", c_void_p),
        ("part_rows", c_void_p), ("ld_rows", c_int64),
        ("dW2_part", c_void_p), ("dWs_part", c_void_p), ("ds_part", c_void_p),
        ("WsTh", c_void_p), ("wst_inv_scale", c_float),
        ("drop_thr", c_uint32), ("drop_seed", c_uint32), ("drop_inv_keep", c_float), ("drop_eid0", c_int64),
        ("row_pairs", c_int32), ("rows_amax", c_void_p), ("work", c_void_p), ("work_bytes", c_int64),
    ]


class EdgeTailArgs(Structure):
    """Mirror of `struct egnn_edge_tail_args` (include/egnn_hip.h)."""
    _fields_ = [
        ("B", c_int32), ("N", c_int32), ("K", c_int32), ("norm_coors", c_int32), ("clamp", c_float), ("eps", c_float),
        ("u", c_void_p), ("coors", c_void_p), ("idx", c_void_p), ("pair_mask", c_void_p), ("g_coors_out", c_void_p),
        ("g_msum", c_void_p), ("W3", c_void_p), ("b3", c_void_p), ("W4", c_void_p), ("b4", c_void_p), ("scale", c_void_p),
        ("gU", c_void_p), ("g_rel", c_void_p), ("g_hid", c_void_p), ("a3", c_void_p), ("g_w", c_void_p), ("g_scale", c_void_p),
        ("gate_w", c_void_p), ("gate_b", c_void_p), ("g_gate", c_void_p),
        ("part", c_void_p), ("rel_out", c_void_p), ("dist_out", c_void_p), ("amax_gu", c_void_p),
        ("drop_thr", c_uint32), ("drop_seed", c_uint32), ("drop_inv_keep", c_float), ("drop_eid0", c_int64),
    ]


class LayerDesc(Structure):
    """Mirror of `struct egnn_layer_desc` (include/egnn_hip.h)."""
    _fields_ = [("dim", c_int32), ("edge_dim", c_int32), ("m_dim", c_int32), ("fourier_features", c_int32),
                ("num_nearest_neighbors", c_int32), ("norm_feats", c_int32), ("norm_coors", c_int32), ("update_feats", c_int32),
                ("update_coors", c_int32), ("only_sparse_neighbors", c_int32), ("soft_edges", c_int32), ("pool_mean", c_int32),
                ("valid_radius", c_float), ("coor_weights_clamp_value", c_float), ("ln_eps", c_float)]


PARAM_FIELDS = ("edge_mlp_0_weight", "edge_mlp_0_bias", "edge_mlp_3_weight", "edge_mlp_3_bias", "edge_gate_0_weight",
                "edge_gate_0_bias", "node_norm_weight", "node_norm_bias", "coors_norm_scale", "node_mlp_0_weight",
                "node_mlp_0_bias", "node_mlp_3_weight", "node_mlp_3_bias", "coors_mlp_0_weight", "coors_mlp_0_bias",
                "coors_mlp_3_weight", "coors_mlp_3_bias")


class LayerParams(Structure):
    """Mirror of `struct egnn_layer_params`: host pointers to the reference's state_dict tensors (field = key with '.' -> '_')."""
    _fields_ = [(f, c_void_p) for f in PARAM_FIELDS]


INFO_OFFSETS = ("wcat_hi", "wcat_lo", "bcat", "wst", "w2h", "b2", "gate_w", "gate_b", "w3h", "b3", "w4", "b4", "coors_scale",
                "w5_hi", "w5_lo", "b5", "w6_hi", "w6_lo", "b6", "gamma", "beta")


class PackedInfo(Structure):
    """Mirror of `struct egnn_packed_info`."""
    _fields_ = [("H", c_int32), ("Hp", c_int32), ("S", c_int32), ("NM", c_int32),
                ("wcat_rows", c_int32), ("w5_rows", c_int32), ("w6_rows", c_int32),
                ("wcat_inv_scale", c_float), ("ws_inv_scale", c_float), ("w2_inv_scale", c_float), ("w3_inv_scale", c_float),
                ("w5_inv_scale", c_float), ("w6_inv_scale", c_float)] + \
               [(f, ctypes.c_uint64) for f in INFO_OFFSETS] + [("bytes", ctypes.c_uint64)]


class ForwardOpts(Structure):
    """Mirror of `struct egnn_forward_opts`."""
    _fields_ = [("side_stream", c_void_p), ("ev_fork", c_void_p), ("ev_join", c_void_p), ("order", c_void_p), ("nmf_img", c_void_p),
                ("order_is_hint", c_int32), ("reserved", c_int32)]


def layer_desc(layer) -> "LayerDesc":
    """egnn_layer_desc of an egnn_pytorch_amd.EGNN (or reference EGNN-like) module."""
    import math
    d = LayerDesc()
    d.dim, d.edge_dim, d.m_dim, d.fourier_features = layer.dim, layer.edge_dim, layer.m_dim, layer.fourier_features
    d.num_nearest_neighbors = layer.num_nearest_neighbors
    d.norm_feats, d.norm_coors = int(layer.norm_feats), int(layer.norm_coors)
    d.update_feats, d.update_coors = int(layer.node_mlp is not None), int(layer.coors_mlp is not None)
    d.only_sparse_neighbors, d.soft_edges = int(layer.only_sparse_neighbors), int(layer.edge_gate is not None)
    d.pool_mean = int(layer.m_pool_method == "mean")
    d.valid_radius = 3.0e38 if math.isinf(layer.valid_radius) else float(layer.valid_radius)
    cv = layer.coor_weights_clamp_value
    d.coor_weights_clamp_value = -1.0 if cv is None else float(cv)
    d.ln_eps = float(layer.node_norm.eps) if layer.norm_feats else 1e-5
    return d


class EGNNHipError(RuntimeError):
    pass


class EGNNRangeError(EGNNHipError):
    """A finite value left the range the split-fp16 arithmetic of the gfx950 path can carry (include/egnn_hip.h:
    EGNN_RANGE_*).  The outputs of that call are non-finite; the reference (plain fp32) has no such limit.
    origin: "call" (this forward), "backward" (bits an earlier backward left: the forward that reports them is valid and is not
    re-run) or "earlier" (deferred mode: some earlier call); bits: the status bits."""
    origin = "call"
    bits = 0


RANGE_BITS = {
    1: "a GEMM input (feats / [LayerNorm(feats) | m_i] / node_mlp hidden activation) with |x| >= 65504",
    2: "a node projection P_i = -log2(e) (W_i h_i + b) with |P| >= 65504",
    4: "a per-edge scalar (squared distance, fourier term or edge feature) with |s| >= 6e7 * max|its weight column| scale",
    8: "an edge_mlp hidden activation beyond fp16 range (the edge message came out non-finite)",
    16: "an edge message / pooled message with |m| >= 65504",
}


_lib = None


def lib_path() -> str:
    return os.environ.get("EGNN_HIP_LIB", os.path.join(_PKG_DIR, _LIB_NAME))


def load():
    """Load (once) and return the ctypes handle.  Raises EGNNHipError if the library is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise EGNNHipError(
            f"{path} not found: the HIP extension is not built. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or egnn_pytorch_amd/csrc/build.sh). "
            f"egnn_pytorch_amd has no CPU / PyTorch fallback.")
    try:
        lib = ctypes.CDLL(path)
    except OSError as exc:  # pragma: no cover
        raise EGNNHipError(f"cannot load {path}: {exc}") from exc
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise EGNNHipError(f"{path} does not export {missing}; rebuild it")

    lib.egnn_abi_version.restype = c_int
    lib.egnn_abi_version.argtypes = []
    lib.egnn_error_string.restype = c_char_p
    lib.egnn_error_string.argtypes = [c_int]
    lib.egnn_padded_hidden.restype = c_int
    lib.egnn_padded_hidden.argtypes = [c_int]
    lib.egnn_knn_select_f32.restype = c_int
    lib.egnn_knn_select_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p]
    lib.egnn_adj_max_degree_u8.restype = c_int
    lib.egnn_adj_max_degree_u8.argtypes = [c_void_p, c_int64, c_int, c_void_p, c_void_p]
    lib.egnn_adj_expand_workspace_bytes.restype = c_size_t
    lib.egnn_adj_expand_workspace_bytes.argtypes = [c_int, c_int]
    lib.egnn_adj_expand_u8.restype = c_int
    lib.egnn_adj_expand_u8.argtypes = [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.egnn_packed_halves.restype = c_int64
    lib.egnn_packed_halves.argtypes = [c_int64, c_int]
    lib.egnn_linear_hl_f32.restype = c_int
    lib.egnn_linear_hl_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p,
                                       c_int64, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.egnn_linear_hl_lda_f32.restype = c_int
    lib.egnn_linear_hl_lda_f32.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p,
                                           c_int64, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.egnn_edge_pw_covers.restype = c_int
    lib.egnn_edge_pw_covers.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64]
    lib.egnn_linear_hl_lda_rows_f32.restype = c_int
    lib.egnn_linear_hl_lda_rows_f32.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p,
                                                c_int64, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                                c_void_p]
    lib.egnn_linear_hl_drop_f32.restype = c_int
    lib.egnn_linear_hl_drop_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p,
                                            c_int64, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int,
                                            ctypes.c_uint32, ctypes.c_uint32, c_float, c_void_p, c_void_p]
    lib.egnn_node_mlp_fused_halves.restype = c_int64
    lib.egnn_node_mlp_fused_halves.argtypes = [c_int, c_int]
    lib.egnn_node_mlp_fused_pack_f16.restype = c_int
    lib.egnn_node_mlp_fused_pack_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.egnn_node_mlp_fused_f32.restype = c_int
    lib.egnn_node_mlp_fused_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64,
                                            c_int, c_int, c_void_p, c_void_p]
    lib.egnn_split_f16.restype = c_int
    lib.egnn_split_f16.argtypes = [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]
    lib.egnn_node_prep_hl.restype = c_int
    lib.egnn_node_prep_hl.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int,
                                      c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p]
    lib.egnn_spatial_order_f32.restype = c_int
    lib.egnn_spatial_order_f32.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.egnn_spatial_order_masked_f32.restype = c_int
    lib.egnn_spatial_order_masked_f32.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.egnn_slot_prep_f32.restype = c_int
    lib.egnn_slot_prep_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.egnn_dest_lists_capacity.restype = c_size_t
    lib.egnn_dest_lists_capacity.argtypes = [c_int, c_int, c_int]
    lib.egnn_dest_lists_i32.restype = c_int
    lib.egnn_dest_lists_i32.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.egnn_split_scaled_f16.restype = c_int
    lib.egnn_split_scaled_f16.argtypes = [c_void_p, c_int64, c_int64, c_int, c_float, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]
    lib.egnn_linear_hl_splitk_f32.restype = c_int
    lib.egnn_linear_hl_splitk_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                              c_int, c_void_p]
    lib.egnn_sum_parts_f32.restype = c_int
    lib.egnn_sum_parts_f32.argtypes = [c_void_p, c_int, c_int64, c_float, c_void_p, c_void_p]
    lib.egnn_absmax_f32.restype = c_int
    lib.egnn_absmax_f32.argtypes = [c_void_p, c_int64, c_void_p, c_void_p]
    lib.egnn_split_scaled_both_f16.restype = c_int
    lib.egnn_split_scaled_both_f16.argtypes = [c_void_p, c_int64, c_int64, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                               c_void_p, c_void_p, c_int64, c_void_p]
    for fn in (lib.egnn_drop_silu_f32, lib.egnn_drop_silu_f64):
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_int64, c_int64, c_int, c_uint32, c_uint32, c_float, c_int64, c_void_p]
    lib.egnn_split_scaled_colsum_rows.restype = c_int64
    lib.egnn_split_scaled_colsum_rows.argtypes = [c_int64, c_int]
    lib.egnn_silu_bwd_f32.restype = c_int
    lib.egnn_silu_bwd_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]
    lib.egnn_silu_bwd_drop_f32.restype = c_int
    lib.egnn_silu_bwd_drop_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_uint32, c_uint32, c_float, c_int64, c_int, c_void_p]
    lib.egnn_unsplit_words_f32.restype = c_int
    lib.egnn_unsplit_words_f32.argtypes = [c_void_p, c_int64, c_int64, c_int, c_void_p]
    lib.egnn_edge_mfmas.restype = c_int
    lib.egnn_edge_mfmas.argtypes = [c_int]
    lib.egnn_edge_fused_f32.restype = c_int
    lib.egnn_edge_fused_f32.argtypes = [POINTER(EdgeArgs), c_void_p]
    lib.egnn_edge_bwd_pass_f32.restype = c_int
    lib.egnn_edge_bwd_pass_f32.argtypes = [POINTER(EdgeBwdArgs), c_void_p]
    lib.egnn_edge_tail_bwd_f32.restype = c_int
    lib.egnn_edge_tail_bwd_f32.argtypes = [POINTER(EdgeTailArgs), c_void_p]
    lib.egnn_edge_tail_part_floats.restype = c_int
    lib.egnn_edge_tail_part_floats.argtypes = []
    lib.egnn_edge_pool_f32.restype = c_int
    lib.egnn_edge_pool_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.egnn_edge_bwd_chunk_steps.restype = c_int
    lib.egnn_edge_bwd_chunk_steps.argtypes = []
    lib.egnn_edge_bwd_work_bytes.restype = c_size_t
    lib.egnn_edge_bwd_work_bytes.argtypes = [c_int64, c_int, c_int, c_int]
    lib.egnn_induced_attn_f32.restype = c_int
    lib.egnn_induced_attn_f32.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                          c_void_p]
    lib.egnn_token_attn_f32.restype = c_int
    lib.egnn_token_attn_f32.argtypes = [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int64,
                                        c_void_p]
    lib.egnn_edge_features_gather_f32.restype = c_int
    lib.egnn_edge_features_gather_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                                  c_int, c_int, c_int, c_void_p, c_void_p]
    lib.egnn_rows_gather_sum_f32.restype = c_int
    lib.egnn_rows_gather_sum_f32.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p]

    lib.egnn_packed_weights_bytes.restype = c_size_t
    lib.egnn_packed_weights_bytes.argtypes = [POINTER(LayerDesc)]
    lib.egnn_packed_layout.restype = c_int
    lib.egnn_packed_layout.argtypes = [POINTER(LayerDesc), POINTER(PackedInfo)]
    lib.egnn_pack_weights_host.restype = c_int
    lib.egnn_pack_weights_host.argtypes = [POINTER(LayerDesc), POINTER(LayerParams), c_void_p, POINTER(PackedInfo)]
    lib.egnn_workspace_bytes.restype = c_size_t
    lib.egnn_workspace_bytes.argtypes = [POINTER(LayerDesc), c_int, c_int, c_int]
    lib.egnn_layer_forward_f32.restype = c_int
    lib.egnn_layer_forward_f32.argtypes = [POINTER(LayerDesc), POINTER(PackedInfo), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                           c_void_p, c_void_p]
    lib.egnn_layer_forward_opts_f32.restype = c_int
    lib.egnn_layer_forward_opts_f32.argtypes = lib.egnn_layer_forward_f32.argtypes + [POINTER(ForwardOpts)]

    lib.egnn_linear_f32.restype = c_int
    lib.egnn_linear_f32.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                    c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]
    lib.egnn_node_prep_f32.restype = c_int
    lib.egnn_node_prep_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int, c_int, c_void_p]
    lib.egnn_edge_exact_f32.restype = c_int
    lib.egnn_edge_exact_f32.argtypes = [POINTER(EdgeExactArgs), c_void_p]
    lib.egnn_edge_exact_f64.restype = c_int
    lib.egnn_edge_exact_f64.argtypes = [POINTER(EdgeExactArgs), c_void_p]
    lib.egnn_knn_select_f64.restype = c_int
    lib.egnn_knn_select_f64.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.egnn_linear_f64.restype = c_int
    lib.egnn_linear_f64.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                    c_int, c_void_p]
    lib.egnn_node_prep_f64.restype = c_int
    lib.egnn_node_prep_f64.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_int64, c_int, c_int, c_void_p]
    for fn in (lib.egnn_edge_exact_bwd_f32, lib.egnn_edge_exact_bwd_f64):
        fn.restype = c_int
        fn.argtypes = [POINTER(EdgeExactBwdArgs), c_void_p]
    for fn in (lib.egnn_edge_tail_exact_bwd_f32, lib.egnn_edge_tail_exact_bwd_f64):
        fn.restype = c_int
        fn.argtypes = [POINTER(EdgeTailExactArgs), c_void_p]
    for fn in (lib.egnn_edge_exact_node_sums_f32, lib.egnn_edge_exact_node_sums_f64):
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_int64, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.egnn_status_publish.restype = c_int
    lib.egnn_status_publish.argtypes = [c_void_p, c_void_p, c_int, c_int32, c_void_p]
    lib.egnn_edge_exact_workspace_bytes.restype = c_size_t
    lib.egnn_edge_exact_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int, c_int]
    if lib.egnn_abi_version() != ABI_VERSION:
        raise EGNNHipError(f"{path}: ABI version {lib.egnn_abi_version()} != {ABI_VERSION}; rebuild it")
    lib.egnn_struct_bytes.restype = c_int64
    lib.egnn_struct_bytes.argtypes = [c_int]
    for which, mirror in enumerate((EdgeArgs, EdgeBwdArgs, EdgeTailArgs, LayerDesc, PackedInfo, EdgeExactArgs, EdgeExactBwdArgs, EdgeTailExactArgs,
                                    ForwardOpts)):
        if lib.egnn_struct_bytes(which) != ctypes.sizeof(mirror):
            raise EGNNHipError(f"{path}: sizeof({mirror.__name__}) = {ctypes.sizeof(mirror)} here, {lib.egnn_struct_bytes(which)} in the "
                               f"library: the ctypes mirror in _abi.py and include/egnn_hip.h disagree")
    _lib = lib
    return lib


def check(code: int, what: str):
    """Map a C-ABI return code to an exception (0 = ok, <0 = EGNN_E_*, >0 = hipError_t)."""
    if code == 0:
        return
    msg = load().egnn_error_string(code).decode()
    if code == -5:   # EGNN_E_K_GT_N -- same text torch.topk raises in the reference (egnn_pytorch.py:258)
        raise RuntimeError(f"{what}: {msg}")
    raise EGNNHipError(f"{what} failed with code {code}: {msg}")
