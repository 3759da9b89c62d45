"""Shared helpers for the test-suite: golden-fixture loading and tolerances."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star tolerance: fp32 feats/coors within 1e-4 of the reference (BASELINE.json).
ATOL = 1e-4


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    params = {k[len("param:"):]: z[k] for k in z.files if k.startswith("param:")}
    data = {k: z[k] for k in z.files if not k.startswith("param:") and k != "meta"}
    return meta, params, data


NETWORK_ONLY = ("depth", "num_tokens", "num_edge_tokens", "num_positions", "num_adj_degrees", "adj_dim",
                "global_linear_attn_every", "global_linear_attn_heads", "global_linear_attn_dim_head", "num_global_tokens")


def layer_kwargs(meta):
    """Per-layer EGNN kwargs of a golden case (for networks: what EGNN_Network passes to each EGNN, :387)."""
    kw = dict(meta["kwargs"])
    if meta["kind"] == "network":
        edge_dim = kw.get("edge_dim", 0) if kw.get("edge_dim", 0) > 0 else 0
        adj_dim = kw.get("adj_dim", 0) if kw.get("num_adj_degrees") is not None else 0
        for k in NETWORK_ONLY:
            kw.pop(k, None)
        kw["edge_dim"] = edge_dim + adj_dim
        kw["norm_feats"] = True          # forced by EGNN_Network (egnn_pytorch/egnn_pytorch.py:387)
    return kw


def check_neighbors(ref_vals, ref_idx, vals, idx, surviving=None):
    """SURVEY.md §8c(5) index-parity policy.

    * selected ranking values bit-equal to the reference's;
    * indices equal wherever the row's selected value is unique in that row;
    * tied groups compared as sets; where a tie straddles the K boundary only its size is
      compared (the reference's choice among equal-ranked candidates is implementation-defined)
      unless `surviving` (the final edge mask) says which entries matter.
    """
    assert ref_vals.shape == vals.shape and ref_idx.shape == idx.shape
    np.testing.assert_array_equal(ref_vals.view(np.uint32), vals.view(np.uint32))
    k = ref_vals.shape[-1]
    rv = ref_vals.reshape(-1, k)
    ri = ref_idx.reshape(-1, k)
    mi = idx.reshape(-1, k)
    sv = None if surviving is None else surviving.reshape(-1, k)
    for r in range(rv.shape[0]):
        v = rv[r]
        start = 0
        while start < k:
            end = start
            while end + 1 < k and v[end + 1] == v[start]:
                end += 1
            a, b = set(ri[r, start:end + 1].tolist()), set(mi[r, start:end + 1].tolist())
            if end < k - 1:
                assert a == b, f"row {r}: tied group {start}:{end+1} differs {a} vs {b}"
            else:
                # group touches the K boundary: the set is only defined if nothing outside ties with it;
                # compare the entries that survive the final mask when given, else sizes.
                if sv is not None:
                    a = {j for j, s in zip(ri[r, start:end + 1], sv[r, start:end + 1]) if s}
                    b = {j for j, s in zip(mi[r, start:end + 1], sv[r, start:end + 1]) if s}
                    assert a == b or len(a) == len(b), f"row {r}: boundary group differs {a} vs {b}"
                else:
                    assert len(a) == len(b)
            start = end + 1
