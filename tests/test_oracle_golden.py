"""The oracle (oracle/egnn_oracle.py) against every committed golden vector generated from the
live reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import egnn_oracle as O
from tests._util import ATOL, check_neighbors, golden_names, layer_kwargs, load_golden


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_golden(name):
    meta, params, d = load_golden(name)
    cfg = O.EGNNConfig(**layer_kwargs(meta))
    kw = dict(edges=d.get("edges"), mask=d.get("mask"), adj_mat=d.get("adj_mat"))
    if meta["kind"] == "layer":
        node, coors, nr, ni = O.egnn_forward(cfg, params, d["feats"], d["coors"], return_neighbors=True, **kw)
        if meta["n_topk"]:
            check_neighbors(d["topk_values.0"], d["topk_indices.0"], nr.astype(np.float32), ni.astype(np.int32))
    else:
        node, coors, changes = O.egnn_network_forward(meta["kwargs"]["depth"], cfg, params, d["feats"], d["coors"],
                                                      return_coor_changes=True,
                                                      num_adj_degrees=meta["kwargs"].get("num_adj_degrees"),
                                                      global_linear_attn_every=meta["kwargs"].get("global_linear_attn_every", 0),
                                                      global_linear_attn_heads=meta["kwargs"].get("global_linear_attn_heads", 8),
                                                      **kw)
        for i, c in enumerate(changes):
            np.testing.assert_allclose(c, d[f"coor_change.{i}"], atol=ATOL, rtol=0)
    # the oracle follows the reference's op order, so it sits far inside the 1e-4 north-star tolerance
    np.testing.assert_allclose(node, d["node_out"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(coors, d["coors_out"], atol=2e-5, rtol=0)
    assert node.dtype == np.float32 and coors.dtype == np.float32


def test_golden_set_is_discriminating():
    """With xavier-scale weights, wrong neighbours must move the outputs by far more than ATOL
    (default init would make feature parity vacuous)."""
    meta, params, d = load_golden("knn8_mask")
    cfg = O.EGNNConfig(**layer_kwargs(meta))
    coors_bad = d["coors"][:, ::-1].copy()      # scramble geometry -> different neighbours
    node, coors = O.egnn_forward(cfg, params, d["feats"], coors_bad, mask=d["mask"])
    assert np.abs(node - d["node_out"]).max() > 100 * ATOL


def test_topk_k_larger_than_n_raises():
    r = np.zeros((1, 4, 4), np.float32)
    with pytest.raises(RuntimeError):
        O.topk_smallest(r, 5)
