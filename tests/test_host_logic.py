"""CPU-side checks of the shipped host code: the C-ABI library loads and exports every symbol the header
declares, the drop-in modules keep the reference's constructor / state_dict contract, the weight re-layout
is algebraically exact, and the product path refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from oracle import egnn_oracle as O
from tests._util import golden_names, layer_kwargs, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_loads_and_exports_header_symbols():
    from egnn_pytorch_amd import _abi
    lib = _abi.load()                                   # no compute call: loading needs no GPU
    header = open(os.path.join(ROOT, "include", "egnn_hip.h")).read()
    declared = set(re.findall(r"\b(egnn_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_abi.SYMBOLS), declared ^ set(_abi.SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.egnn_abi_version() == _abi.ABI_VERSION
    assert lib.egnn_padded_hidden(2050) == 2080 and lib.egnn_padded_hidden(64) == 64
    from egnn_pytorch_amd import _weights
    for s_ in range(1, 17):                             # host mirror of the kernel's instantiation table
        assert lib.egnn_edge_mfmas(s_) == _weights.edge_mfmas(s_) and 4 * _weights.edge_mfmas(s_) >= 3 * s_
    assert b"out of range" in lib.egnn_error_string(-5)


def test_edge_args_struct_matches_header():
    """Field order / count of the ctypes mirror against `struct egnn_edge_args` in the header."""
    from egnn_pytorch_amd import _abi
    header = open(os.path.join(ROOT, "include", "egnn_hip.h")).read()
    body = header[header.index("typedef struct egnn_edge_args {"):header.index("} egnn_edge_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part)[-1])
    assert names == [f[0] for f in _abi.EdgeArgs._fields_]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from egnn_pytorch_amd import _abi
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setenv("EGNN_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_abi.EGNNHipError):
        _abi.load()


def test_no_cpu_fallback():
    from egnn_pytorch_amd import EGNN
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        EGNN(dim=8)(torch.randn(1, 4, 8), torch.randn(1, 4, 3))


def test_autograd_mode_is_decided_before_no_grad():
    """ADVICE r1 (superseded by autograd support): whether a graph is recorded is decided BEFORE any no_grad block -- with
    grad mode on and something requiring grad the call goes through autograd.EGNNFunction, under no_grad it does not.  (On
    the CPU both end at the no-CPU-fallback error; the GPU behaviour is in tests/test_autograd.py.)"""
    from egnn_pytorch_amd import EGNN, autograd
    layer = EGNN(dim=8)
    x = torch.randn(1, 4, 8, requires_grad=True)
    assert autograd.wants_grad(layer, x) and autograd.wants_grad(layer, torch.randn(1, 4, 8))
    with torch.no_grad():
        assert not autograd.wants_grad(layer, x)
    for p in layer.parameters():
        p.requires_grad_(False)
    assert not autograd.wants_grad(layer, torch.randn(1, 4, 8)) and autograd.wants_grad(layer, x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(x, torch.randn(1, 4, 3))


def test_product_code_never_imports_oracle():
    pkg = os.path.join(ROOT, "egnn_pytorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("checks it against the CPU oracle", ""), f


@pytest.mark.parametrize("name", golden_names())
def test_state_dict_roundtrip_with_reference_keys(name):
    """The reference's state_dict (stored in the golden file) loads strictly into the drop-in module."""
    from egnn_pytorch_amd import EGNN, EGNN_Network
    meta, params, _ = load_golden(name)
    net = EGNN(**meta["kwargs"]) if meta["kind"] == "layer" else EGNN_Network(**meta["kwargs"])
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    res = net.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert list(net.state_dict().keys()) == list(sd.keys())            # same registration order too
    for k, v in net.state_dict().items():
        assert v.shape == sd[k].shape


def test_global_attention_block_matches_oracle():
    """The induced-set attention block (device-agnostic tensor plumbing) against the oracle's
    restatement of egnn_pytorch.py:81-144, with the reference's parameters from the golden file."""
    from egnn_pytorch_amd.attention import GlobalLinearAttention
    from oracle import egnn_oracle as O
    meta, params, d = load_golden("net_global_attn")
    kw = meta["kwargs"]
    blk = GlobalLinearAttention(dim=kw["dim"], heads=kw["global_linear_attn_heads"],
                                dim_head=kw["global_linear_attn_dim_head"]).eval()
    pre = "layers.0.0."
    blk.load_state_dict({k[len(pre):]: torch.from_numpy(v) for k, v in params.items() if k.startswith(pre)}, strict=True)
    x, mask = d["feats"], d["mask"]
    tok = np.broadcast_to(params["global_tokens"][None], (x.shape[0],) + params["global_tokens"].shape).copy()
    with torch.no_grad():
        got_x, got_q = blk(torch.from_numpy(x), torch.from_numpy(tok), mask=torch.from_numpy(mask))
    ref_x, ref_q = O.global_linear_attention(params, pre, x, tok, kw["global_linear_attn_heads"], mask=mask)
    np.testing.assert_allclose(got_x.numpy(), ref_x, atol=2e-5, rtol=0)
    np.testing.assert_allclose(got_q.numpy(), ref_q, atol=2e-5, rtol=0)


def test_global_attention_fully_masked_graph_stays_finite():
    """a graph whose mask row is all False (a fully padded batch entry) -- the reference fills the logits with
    -finfo.max and softmaxes to a uniform distribution (egnn_pytorch.py:102-107); outputs stay finite and equal the
    oracle's."""
    from egnn_pytorch_amd.attention import GlobalLinearAttention
    from oracle import egnn_oracle as O
    meta, params, d = load_golden("net_global_attn")
    kw = meta["kwargs"]
    blk = GlobalLinearAttention(dim=kw["dim"], heads=kw["global_linear_attn_heads"],
                                dim_head=kw["global_linear_attn_dim_head"]).eval()
    pre = "layers.0.0."
    blk.load_state_dict({k[len(pre):]: torch.from_numpy(v) for k, v in params.items() if k.startswith(pre)}, strict=True)
    x, mask = d["feats"], d["mask"].copy()
    mask[0, :] = False
    tok = np.broadcast_to(params["global_tokens"][None], (x.shape[0],) + params["global_tokens"].shape).copy()
    with torch.no_grad():
        got_x, got_q = blk(torch.from_numpy(x), torch.from_numpy(tok), mask=torch.from_numpy(mask))
    assert torch.isfinite(got_x).all() and torch.isfinite(got_q).all()
    ref_x, ref_q = O.global_linear_attention(params, pre, x, tok, kw["global_linear_attn_heads"], mask=mask)
    np.testing.assert_allclose(got_x.numpy(), ref_x, atol=2e-5, rtol=0)
    np.testing.assert_allclose(got_q.numpy(), ref_q, atol=2e-5, rtol=0)


def test_constructor_contract():
    from egnn_pytorch_amd import EGNN, EGNN_Network
    with pytest.raises(AssertionError):
        EGNN(dim=8, m_pool_method="max")
    with pytest.raises(AssertionError):
        EGNN(dim=8, update_feats=False, update_coors=False)
    with pytest.raises(AssertionError):
        EGNN_Network(depth=1, dim=8, num_adj_degrees=0)
    layer = EGNN(dim=16)
    w = layer.edge_mlp[0].weight
    assert abs(float(w.std()) - 1e-3) < 3e-4                        # init_: N(0, init_eps)
    net = EGNN_Network(depth=2, dim=8, num_nearest_neighbors=3, norm_coors=True)
    assert all(l[1].norm_feats for l in net.layers)                 # forced on, egnn_pytorch.py:387


@pytest.mark.parametrize("kw", [dict(dim=20, m_dim=8, edge_dim=3, fourier_features=2, soft_edges=True),
                                dict(dim=64), dict(dim=33, edge_dim=1), dict(dim=24, m_dim=32), dict(dim=16, m_dim=40)])
def test_weight_relayout_is_exact(kw):
    """Evaluate the factorised / padded / fragment-ordered weights with plain numpy on random edges and
    compare with the oracle's unfactorised Linear(cat(h_i, h_j, scal)): same numbers to fp32 round-off."""
    from egnn_pytorch_amd import EGNN, _weights
    torch.manual_seed(0)
    layer = EGNN(**kw)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.randn_like(p) * 0.2)
    w = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in _weights.pack(layer).items()}
    dim, m = kw["dim"], kw.get("m_dim", 16)
    s = w["S"]
    rng = np.random.default_rng(0)
    hi = rng.standard_normal((7, dim)).astype(np.float32)
    hj = rng.standard_normal((7, dim)).astype(np.float32)
    sc = rng.standard_normal((7, s)).astype(np.float32)
    sd = {k: v.detach().numpy() for k, v in layer.state_dict().items()}
    x_ref = np.concatenate([hi, hj, sc], -1) @ sd["edge_mlp.0.weight"].T + sd["edge_mlp.0.bias"]
    hp, h = w["Hp"], w["H"]
    assert hp % 32 == 0 and hp >= h and w["Ws"].shape == (s, hp)
    # the kernel's variable is y = -log2(e) * x ...
    pi = hi @ w["Wcat"][:hp].T + w["bcat"][:hp]
    pj = hj @ w["Wcat"][hp:].T + w["bcat"][hp:]
    y = pi + pj + sc @ w["Ws"]
    np.testing.assert_allclose(y[:, :h] / -np.log2(np.e), x_ref, atol=3e-5)
    assert np.all(y[:, h:] == 0)
    # the scalar columns as first-layer MFMA fragments: five fp16 products per (scalar, hidden unit) rebuild W * s
    nt = 4 * _weights.edge_mfmas(s)
    assert w["Wst"].dtype == np.float16 and w["Wst"].shape == (hp, nt, 2) and 3 * s <= nt
    plain = _weights.swizzle_terms(torch.from_numpy(np.ascontiguousarray(w["Wst"]))).numpy()    # (an involution: undoes the pair swap of units 8 .. 15)
    assert np.all(plain[:, 3 * s:] == 0)
    tab = plain[:, :3 * s].astype(np.float64).reshape(hp, s, 3, 2)     # (hidden unit, scalar, kind, hi|lo)
    sp = sc.astype(np.float64) * w["ws_inv_scale"]                    # what the kernel splits: s' = s / ws_scale
    s1 = (sp / 1024).astype(np.float16).astype(np.float64)
    r = sp - 1024 * s1
    rh = r.astype(np.float16).astype(np.float64)
    rl = (r - rh).astype(np.float16).astype(np.float64)
    ys = np.einsum("es,hs->eh", s1, tab[:, :, 0].sum(-1)) + np.einsum("es,hs->eh", rh, tab[:, :, 1].sum(-1)) \
        + np.einsum("es,hs->eh", rl, tab[:, :, 2, 0])
    np.testing.assert_allclose(ys, sc.astype(np.float64) @ w["Ws"].astype(np.float64), atol=2e-6)
    # ... hidden = y / (1 + 2^y) = SiLU(x) / (-ln 2), contracted with hi + lo fp16 fragments of -ln2 * scale * W2
    nb = _weights.m_blocks(m)
    mp = 16 * nb
    w2h = w["W2h"].astype(np.float64)                                 # (Hp/32, NB, 2, 64, 8)
    assert w["W2h"].dtype == np.float16 and w2h.shape == (hp // 32, nb, 2, 64, 8)
    # [step][nb][lane = 16 g + c][t = 4 hb + r] holds channel 16 nb + c, hidden unit 32 step + 16 hb + 4 g + r
    unfrag = lambda f: f.reshape(hp // 32, nb, 4, 16, 2, 4).transpose(1, 3, 0, 4, 2, 5).reshape(mp, hp)
    w2 = (unfrag(w2h[:, :, 0]) + unfrag(w2h[:, :, 1])) * w["w2_inv_scale"]
    # 22 significant bits for elements near the tensor's max; lo of much smaller elements falls into fp16
    # subnormals (spacing 2^-24 of the scaled max), i.e. the absolute error stays at fp32 level of the max
    w2_true = -np.log(2.0) * sd["edge_mlp.3.weight"].astype(np.float64)
    np.testing.assert_allclose(w2[:m, :h], w2_true, rtol=4e-7, atol=2.0 ** -24 * np.abs(w2_true).max())
    assert np.all(w2[m:] == 0) and np.all(w2[:, h:] == 0)
    lg = np.log2(w["w2_inv_scale"])
    assert lg == np.round(lg) and 1.0 <= np.abs(unfrag(w2h[:, :, 0])).max() < 2.0      # power-of-two range scale
    hid = y.astype(np.float64) / (1.0 + np.exp2(y.astype(np.float64)))
    m_ref = O.silu(O.silu(x_ref) @ sd["edge_mlp.3.weight"].T + sd["edge_mlp.3.bias"])
    m_new = O.silu((hid @ w2.T + w["b2"]).astype(np.float32))
    np.testing.assert_allclose(m_new[:, :m], m_ref, atol=2e-5)
    assert np.all(m_new[:, m:] == 0)
    np.testing.assert_array_equal(w["W3"][:4 * m, :m], sd["coors_mlp.0.weight"])
    assert w["W3"].shape == (64 * nb, 16 * nb) and np.all(w["W3"][4 * m:] == 0)


def test_packed_weights_cache_tracks_parameter_updates():
    from egnn_pytorch_amd import EGNN
    layer = EGNN(dim=8)
    a = layer.packed_weights()
    assert layer.packed_weights() is a
    with torch.no_grad():
        layer.edge_mlp[0].weight.add_(1.0)
    b = layer.packed_weights()
    assert b is not a and not torch.equal(a["Wcat"], b["Wcat"])


def test_packed_weights_cache_tracks_replaced_submodules():
    """A whole submodule replaced between two forwards (the old module's own _parameters never change): the packed tables must follow."""
    from egnn_pytorch_amd import EGNN
    torch.manual_seed(0)
    layer = EGNN(dim=8, norm_feats=True)
    a = layer.packed_weights()
    assert layer.packed_weights() is a
    new_first = torch.nn.Linear(layer.edge_mlp[0].in_features, layer.edge_mlp[0].out_features)
    layer.edge_mlp[0] = new_first                                   # nn.Sequential.__setitem__
    b = layer.packed_weights()
    assert b is not a and not torch.equal(a["Wcat"], b["Wcat"])
    assert layer.packed_weights() is b
    norm = torch.nn.LayerNorm(8)
    with torch.no_grad():
        norm.weight.fill_(3.0)
    layer.node_norm = norm                                          # nn.Module.__setattr__
    c = layer.packed_weights()
    assert c is not b and float(c["gamma"][0]) == 3.0
    assert layer.packed_weights() is c


def test_packed_tile_layout_matches_header_formula():
    """_weights.pack_tiles / unpack_tiles against the offset formula documented in include/egnn_hip.h."""
    from egnn_pytorch_amd import _weights
    r, kp = 96, 64
    x = torch.arange(r * kp, dtype=torch.float32).reshape(r, kp)
    flat = _weights.pack_tiles(x)
    assert flat.numel() == r * kp
    rng = np.random.default_rng(0)
    for _ in range(500):
        row, k = int(rng.integers(r)), int(rng.integers(kp))
        off = ((((row >> 5) * (kp // 16) + (k >> 4)) * 32 + (row & 31)) * 2 + (((k >> 3) & 1) ^ (((row & 31) >> 3) & 1))) * 8 + (k & 7)
        assert float(flat[off]) == float(x[row, k])
    assert torch.equal(_weights.unpack_tiles(flat, r, kp), x)
    hi, lo, inv, w_rows = _weights.split_f16(torch.randn(130, 70))
    assert w_rows == 256 and hi.numel() == 256 * 96 and hi.dtype == torch.float16
    back = (_weights.unpack_tiles(hi, 256, 96).float() + _weights.unpack_tiles(lo, 256, 96).float()) * inv
    assert float(back[130:].abs().max()) == 0 and float(back[:, 70:].abs().max()) == 0


@pytest.mark.parametrize("kw", [dict(dim=20, m_dim=8, edge_dim=3, fourier_features=2, soft_edges=True, norm_feats=True, norm_coors=True),
                                dict(dim=64, num_nearest_neighbors=8), dict(dim=33, edge_dim=1, update_coors=False),
                                dict(dim=16, update_feats=False), dict(dim=512, num_nearest_neighbors=32),
                                dict(dim=24, m_dim=32, soft_edges=True), dict(dim=24, m_dim=40, edge_dim=2),
                                dict(dim=16, m_dim=64, fourier_features=1)])
def test_c_weight_packer_matches_python(kw):
    """egnn_pack_weights_host (C, host) == egnn_pytorch_amd/_weights.py::pack (torch): every re-laid-out tensor and every
    power-of-two scale, bit for bit -- a binding without torch gets exactly the weights the Python module computes with."""
    from egnn_pytorch_amd import EGNN, _ops, _weights
    torch.manual_seed(1)
    layer = EGNN(**kw)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    w = _weights.pack(layer)
    desc, info, blob = _ops.pack_weights_c(layer)
    assert (info.H, info.Hp, info.S) == (w["H"], w["Hp"], w["S"]) and 4 * info.NM == w["Wst"].shape[1]

    def piece(off, ref):
        ref = ref.contiguous()
        nbytes = ref.numel() * ref.element_size()
        got = blob[off:off + nbytes].view(ref.dtype).view(ref.shape)
        assert torch.equal(got, ref), off

    hi, lo, inv, rows = w["Wcat_split"]
    assert info.wcat_rows == rows and info.wcat_inv_scale == inv
    piece(info.wcat_hi, hi); piece(info.wcat_lo, lo); piece(info.bcat, w["bcat"])
    piece(info.wst, w["Wst"]); assert info.ws_inv_scale == w["ws_inv_scale"]
    piece(info.w2h, w["W2h"]); assert info.w2_inv_scale == w["w2_inv_scale"]
    piece(info.b2, w["b2"])
    if "gate_w" in w:
        piece(info.gate_w, w["gate_w"]); piece(info.gate_b, w["gate_b"])
    if "W3h" in w:
        piece(info.w3h, w["W3h"]); assert info.w3_inv_scale == w["w3_inv_scale"]
        piece(info.b3, w["b3"]); piece(info.w4, w["W4"]); piece(info.b4, w["b4"])
    if "coors_scale" in w:
        piece(info.coors_scale, w["coors_scale"])
    if "W5_split" in w:
        for key, o_hi, o_lo, o_b, inv_c, rows_c in (("W5", info.w5_hi, info.w5_lo, info.b5, info.w5_inv_scale, info.w5_rows),
                                                    ("W6", info.w6_hi, info.w6_lo, info.b6, info.w6_inv_scale, info.w6_rows)):
            hi, lo, inv, rows = w[key + "_split"]
            assert rows_c == rows and inv_c == inv
            piece(o_hi, hi); piece(o_lo, lo); piece(o_b, w["b" + key[1]])
    if "gamma" in w:
        piece(info.gamma, w["gamma"]); piece(info.beta, w["beta"])


def test_workspace_bytes_and_descriptor_checks():
    from ctypes import byref
    from egnn_pytorch_amd import EGNN, _abi
    lib = _abi.load()
    d = _abi.layer_desc(EGNN(dim=512, num_nearest_neighbors=32))
    need = lib.egnn_workspace_bytes(byref(d), 64, 1024, 32)
    rows = 64 * 1024
    hp = lib.egnn_padded_hidden(2 * (2 * 512 + 1))
    assert need >= rows * 2 * hp * 4 + 2 * rows * 32 * 4                       # at least P and the neighbour list
    assert need < 2 * (rows * 2 * hp * 4)                                      # ... and not wildly more (P dominates)
    assert lib.egnn_workspace_bytes(byref(d), 0, 1024, 32) == 0
    bad = _abi.layer_desc(EGNN(dim=8, m_dim=16))
    bad.m_dim = 65                                                             # beyond the four accumulator tiles
    assert lib.egnn_packed_weights_bytes(byref(bad)) == 0 and lib.egnn_workspace_bytes(byref(bad), 1, 4, 4) == 0


def test_host_read_and_lazy_destination_lists_on_the_cpu():
    """_ops.HostRead on host tensors (the CPU tests of the backward run through it): the values themselves, no event; max-|x| bit
    patterns decode to floats; DestLists cuts its entry list to length on first use, from the tile count it was given as a HostRead."""
    from egnn_pytorch_amd import _ops
    bits = torch.tensor([1.5, 0.0, float("inf")], dtype=torch.float32).view(torch.int32)
    hr = _ops.HostRead(bits)
    assert hr.ev is None and hr.floats() == [1.5, 0.0, float("inf")]
    assert _ops.bits_to_floats(hr) == [1.5, 0.0, float("inf")] and _ops.bits_to_floats(bits) == [1.5, 0.0, float("inf")]
    assert _ops.HostRead(torch.tensor([7, 9], dtype=torch.int64)).ints() == [7, 9]
    assert _ops.absmax_async(torch.tensor([[-3.0, 2.0]])).floats()[0] == 3.0
    ent = torch.arange(1024, dtype=torch.int32)
    dl = _ops.DestLists(ent, torch.zeros(3, dtype=torch.int64), torch.zeros(4, dtype=torch.int64), torch.zeros(3, dtype=torch.int64),
                        _ops.HostRead(torch.tensor([20], dtype=torch.int64)))
    assert dl.ent.numel() == 384 and dl.ent.numel() == 384            # 20 tiles x 16 entries, whole 128-entry rounds; resolved once
    assert _ops.DestLists(ent[:256], None, None, None).ent.numel() == 256


def test_copies_and_pickles_of_a_module_leave_the_kernel_side_caches_behind():
    """copy.deepcopy / pickle / torch.save of an EGNN carry the reference's state only: the re-laid-out weights, the C entry's blob and
    the cached parameter list (device tensors, ctypes structs, references to the ORIGINAL's Parameter objects) are rebuilt by the copy."""
    import copy
    import io
    import pickle
    from egnn_pytorch_amd import EGNN, _weights
    layer = EGNN(dim=16, num_nearest_neighbors=4, norm_feats=True)
    _weights.version_key(layer)                                   # (fills the parameter-list cache)
    layer.__dict__["_c_packed"] = (("key",), None)
    layer._packed, layer._packed_key = {"stale": 1}, ("key",)
    for make in (copy.deepcopy, lambda m: pickle.loads(pickle.dumps(m)),
                 lambda m: (lambda b: (torch.save(m, b), b.seek(0), torch.load(b, weights_only=False))[2])(io.BytesIO())):
        twin = make(layer)
        assert twin._packed is None and twin._packed_key is None
        assert not any(k in twin.__dict__ for k in ("_c_packed", "_param_cache", "_param_cache_mods", "_param_cache_tree", "_param_cache_counts"))
        assert all(torch.equal(a, b) for a, b in zip(layer.state_dict().values(), twin.state_dict().values()))
        # the copy's key is built from ITS parameters
        assert all(p is q for p, q in zip(twin.__dict__.get("_param_cache") or (_weights.version_key(twin) and twin.__dict__["_param_cache"]),
                                          twin.parameters()))
    assert layer.__dict__["_c_packed"] == (("key",), None) and layer._packed == {"stale": 1}      # (the original keeps its caches)


@pytest.mark.parametrize("kw", [dict(dim=24, num_nearest_neighbors=6), dict(dim=32, edge_dim=3, fourier_features=2, soft_edges=True, norm_coors=True,
                                                                               norm_feats=True), dict(dim=16, update_feats=False),
                                dict(dim=16, update_coors=False, m_dim=40), dict(dim=64, norm_feats=True)])
def test_blob_assembled_from_the_torch_packer_equals_the_c_host_packer(kw):
    """_ops.pack_weights_blob (the module's own re-laid-out tensors copied to the offsets of egnn_packed_layout: what the one-call inference
    forward runs on) == egnn_pack_weights_host's blob, byte for byte, and the same info struct."""
    from egnn_pytorch_amd import EGNN, _abi, _ops, _weights
    torch.manual_seed(2)
    layer = EGNN(**kw)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.randn_like(p) * 0.2)
    desc_c, info_c, blob_c = _ops.pack_weights_c(layer)
    desc_t, info_t, blob_t = _ops.pack_weights_blob(layer, _weights.pack(layer), torch.device("cpu"))
    assert bytes(desc_c) == bytes(desc_t)
    for name, _ in _abi.PackedInfo._fields_:
        assert getattr(info_c, name) == getattr(info_t, name), name
    assert blob_c.numel() == blob_t.numel() and torch.equal(blob_c, blob_t)
