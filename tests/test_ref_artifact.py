"""oracle/_ref (the reference itself, byte-compiled by oracle/build_ref.py) is what bench.py's `cpu_baseline` times.
Checks that the artefact imports from oracle/_ref -- not from /root/reference -- and that it agrees with the numpy oracle
and a golden file, i.e. that the thing being timed is the thing parity is pinned to."""
import os
import sys

import numpy as np
import pytest
import torch

from tests._util import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, ROOT)
    from oracle import build_ref
    if not build_ref.build():
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return build_ref.import_reference()


def test_artifact_is_sourceless_and_in_tree(ref):
    assert os.path.abspath(ref.__file__).startswith(os.path.join(ROOT, "oracle", "_ref"))
    assert ref.__file__.endswith(".pyc")
    listed = open(os.path.join(ROOT, ".gitignore")).read().split()
    assert "oracle/_ref/" in listed                                   # outputs only, never committed
    ignore = os.path.join(ROOT, ".gpurunignore")
    assert not os.path.exists(ignore) or "oracle/_ref" not in open(ignore).read()     # ... but it travels to the GPU box


@pytest.mark.parametrize("name", ["knn8_mask", "all_flags"])
def test_artifact_reproduces_golden(ref, name):
    meta, params, d = load_golden(name)
    layer = ref.EGNN(**meta["kwargs"]).eval()
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    t = lambda k: None if k not in d else torch.from_numpy(d[k])
    with torch.no_grad():
        node, co = layer(t("feats"), t("coors"), t("edges"), t("mask"), t("adj_mat"))
    np.testing.assert_array_equal(node.numpy(), d["node_out"])
    np.testing.assert_array_equal(co.numpy(), d["coors_out"])
