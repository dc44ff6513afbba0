"""The C-ABI library loads on a GPU-less host and exports every symbol include/regtgcn.h declares;
host-side argument validation works without touching a GPU."""
import ctypes
import os
import re

import pytest
import torch

import regtgcn_amd as R
from regtgcn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "regtgcn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(regt_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = R.load_library()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/regtgcn.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype in _lib.SIGNATURES"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.regt_abi_version() == _lib.ABI_VERSION


def test_host_side_validation_without_gpu():
    lib = R.load_library()
    d = _lib.Dims(10, 6, 7, 256, 5, 1, 128, 1, 0.01)            # F = 7 is not a multiple of 4
    assert lib.regt_workspace_bytes(ctypes.byref(d), 1, 0) == 0
    assert b"multiple of 4" in lib.regt_last_error()
    d = _lib.Dims(104, 6, 8, 256, 5, 1, 128, 1, 0.01)
    assert lib.regt_workspace_bytes(ctypes.byref(d), 1, 0) > 104 * 6 * 256 * 4 * 9
    assert lib.regt_workspace_bytes(ctypes.byref(d), 1, 1) > lib.regt_workspace_bytes(ctypes.byref(d), 1, 0)
    rc = lib.regt_spmm_csr(None, None, None, None, None, 1, 1, 4, None)
    assert rc != 0 and b"NULL" in lib.regt_last_error()
    assert lib.regt_graph_workspace_bytes(1000, 100) > 0
    assert lib.regt_wgrad_slab_floats(1000, 256, 256, 1) >= 256 * 256 + 256


def test_developer_options_are_host_state_and_every_documented_name_is_known():
    """regt_set_option touches no GPU: every option name the header documents is accepted, returns the previous value and can be
    restored; an unknown name is refused with -1 and a message."""
    lib = R.load_library()
    src = open(os.path.join(ROOT, "include", "regtgcn.h")).read()
    doc = src[src.index("Developer switches"):src.index("int32_t regt_set_option")]
    names = sorted(set(re.findall(r'"([a-z_0-9]+)"', doc)))
    assert {"xbf", "fused_bwd", "spmm_rows", "dgrad1_gen", "tgcn_collapse", "wgrad_ring", "wgrad_tile", "wgrad_ring256", "wgrad_pairs",
            "wgrad_wave", "wgrad_bnw64"} <= set(names), names
    for n in names:
        prev = lib.regt_set_option(n.encode(), 1)
        assert prev >= 0, (n, lib.regt_last_error())
        assert lib.regt_set_option(n.encode(), prev) >= 0          # restored (the value just set is returned)
        assert lib.regt_set_option(n.encode(), prev) == prev
    assert lib.regt_set_option(b"no_such_option", 1) == -1 and b"unknown option" in lib.regt_last_error()


def test_modules_refuse_cpu_tensors_and_keep_reference_layout():
    m = R.RegionalTemporalGCN(node_features=8, num_nodes=104, periods=6, output_dim=1)
    sd = torch.load(os.path.join(ROOT, "tests", "golden", "ref_ckpt_in6_out1_epoch50.pt"), map_location="cpu", weights_only=True)
    assert list(sd.keys()) == list(m.state_dict().keys())
    assert all(tuple(sd[k].shape) == tuple(v.shape) for k, v in m.state_dict().items())
    m.load_state_dict(sd, strict=True)
    with pytest.raises(R.RegtError):
        m(torch.zeros(104, 8, 6), torch.zeros(2, 0, dtype=torch.long))
    # the reference's keyword names bind like its positional parameters (models/RegionalTemporalGCN.py:25-26): a complete keyword
    # call gets as far as the device check, a wrong name is Python's TypeError
    z, e = torch.zeros(104, 8, 6), torch.zeros(2, 0, dtype=torch.long)
    kw = {f"{r}edge_index": e for r in ("IA", "KS", "KY", "OH", "WI")}
    kw.update({f"{r}edge_attr": torch.zeros(0) for r in ("IA", "KS", "KY", "OH", "WI")})
    with pytest.raises(R.RegtError):
        m(x=z, edge_index=e, **kw)
    with pytest.raises(TypeError, match="unexpected keyword"):
        m(z, e, **kw, edge_weight=None)
    with pytest.raises(TypeError, match="missing 1 required"):
        m(z, e, **{k: v for k, v in kw.items() if k != "WIedge_attr"})
    t = R.TemporalGCN(node_features=8, periods=6, output_dim=3)
    assert "tgnn.linear.weight" in t.state_dict() and tuple(t.state_dict()["tgnn.linear.weight"].shape) == (256, 64)
    assert not any(k.startswith("tgnn._weight_att") for k in t.state_dict())
    import inspect
    assert list(inspect.signature(t.forward).parameters) == ["x", "edge_index", "edge_attr"]   # keyword call, run.py:188


def test_entry_points_reject_bad_arguments_before_touching_the_gpu():
    """Every model-level entry point validates dims / pointers / alignment on the host and reports through
    regt_last_error (the reference raises Python exceptions at the same places: wrong shapes, missing tensors)."""
    lib = R.load_library()
    B = ctypes.byref
    good = _lib.Dims(104, 6, 8, 256, 5, 1, 128, 1, 0.01)
    g, p, gr = _lib.Graph(), _lib.Params(), _lib.Grads()
    one = ctypes.c_void_p(256)                                    # a non-NULL, 16-byte aligned dummy that is never dereferenced

    rc = lib.regt_forward(None, B(g), B(p), one, one, one, one, 1 << 30, None)
    assert rc != 0 and b"dims is NULL" in lib.regt_last_error()
    rc = lib.regt_forward(B(good), B(g), B(p), one, one, one, one, 1 << 30, None)
    assert rc != 0 and b"graph incomplete" in lib.regt_last_error()
    g.rowptr = g.col = g.val = g.node_region = 256
    rc = lib.regt_forward(B(good), B(g), B(p), one, one, one, one, 1 << 30, None)
    assert rc != 0 and b"required tensor pointer is NULL" in lib.regt_last_error()
    bad_t = _lib.Dims(104, 256, 8, 256, 5, 1, 128, 1, 0.01)
    rc = lib.regt_forward(B(bad_t), B(g), B(p), one, one, one, one, 1 << 30, None)
    assert rc != 0 and b"exceeds 255 periods" in lib.regt_last_error()
    rc = lib.regt_cell_forward(B(good), B(g), B(p), one, one, one, one, one, 1 << 30, None)
    assert rc != 0 and b"regional must be 0" in lib.regt_last_error()
    rc = lib.regt_backward(B(good), B(g), B(p), None, one, None, one, None, one, 1 << 30, None)
    assert rc != 0                                                 # params incomplete / grads NULL
    rc = lib.regt_spmm_csr(one, one, one, one, one, 8, 8, 6, None)
    assert rc != 0 and b"multiple of 4" in lib.regt_last_error()
    rc = lib.regt_spmm_dual(one, one, one, one, one, one, one, 8, 46, None)
    assert rc != 0 and b"multiple of 4 floats" in lib.regt_last_error()
    rc = lib.regt_spmm_dual(one, one, one, one, one, one, one, 8, 2052, None)
    assert rc != 0 and b"neither a multiple of 32 floats nor at most 2048" in lib.regt_last_error()
    rc = lib.regt_linear(one, 8, 4, 8, one, 8, 4, None, 7, 0.0, one, 4, None)
    assert rc != 0 and b"act must be" in lib.regt_last_error()
    rc = lib.regt_wgrad(one, 4, one, 8, 16, 4, 8, one, 8, None, None, None)
    assert rc != 0 and b"bad argument" in lib.regt_last_error()
    assert lib.regt_set_gemm_mode(0) in (0, 1)


def test_zero_hidden_models_keep_reference_layout():
    """GraphSAGETemporalGCN / GATTemporal: state_dict keys in the order the reference's modules register them (the same
    order oracle.init_params emits, which oracle/make_goldens.py asserts against the imported reference classes)."""
    from oracle import model as M
    for name, cls in (("GraphSAGETemporalGCN", R.GraphSAGETemporalGCN), ("GATTemporal", R.GATTemporal)):
        m = cls(node_features=8, num_nodes=104, periods=6, output_dim=3)
        p = M.init_params(name, 8, 6, 3, num_nodes=104, seed=0)
        assert list(p.keys()) == list(m.state_dict().keys()), name
        assert all(tuple(p[k].shape) == tuple(v.shape) for k, v in m.state_dict().items()), name
        m.load_state_dict(p, strict=True)
        import inspect
        assert list(inspect.signature(m.forward).parameters) == ["x", "edge_index", "edge_attr"]
        with pytest.raises(R.RegtError):
            m(torch.zeros(104, 8, 6), torch.zeros(2, 0, dtype=torch.long), None)
    with pytest.raises(NotImplementedError):
        R.TGCN(8, 16, baseblock="transformer")


def test_reference_launch_line_parses():
    """The argument string of the reference's own launch script (scripts/RegionalTemporalGCN.sh:1, copied here as a string) is
    accepted by the run.py counterpart, with run.py's defaults for everything it leaves out (run.py:22-45)."""
    from regtgcn_amd import train
    line = "--num_timesteps_in 6 --num_timesteps_out 1 --tr 0.2 --model RegionalTemporalGCN --tf occrate --dataloading_type 2 --epochs 50 --decomp_type regional"
    a = train.build_parser().parse_args(line.split())
    assert (a.num_timesteps_in, a.num_timesteps_out, a.tr, a.model, a.tf, a.dataloading_type, a.epochs, a.decomp_type) == \
        (6, 1, 0.2, "RegionalTemporalGCN", "occrate", 2, 50, "regional")
    d = train.build_parser().parse_args([])
    assert (d.seed, d.lr, d.decay, d.momentum, d.bs, d.tf, d.model, d.num_timesteps_in, d.num_timesteps_out, d.checkpoint_path) == \
        (42, 1e-3, 1e-4, 0.9, 32, "available", "TemporalGCN", 8, 4, "../checkpoints/")
    for flags in ("--train_ratio 0.5 --batch_size 8 --train_feature available --edge_cut random --dataset_path ./dataset --is_preprocessed --logs",
                  "--model GraphSAGETemporalGCN", "--model GAT", "--model RandomTemporalGCN --decomp_type random",
                  "--is_pretrained --pretrained_model m.pt --pretrained_model_epoch 10"):
        train.build_parser().parse_args(flags.split())


def test_etl_target_column_follows_train_feature():
    from regtgcn_amd import etl
    assert etl.feature_columns("occrate")[-1] == 12 and etl.feature_columns("AVAILABLE")[-1] == 11
    assert etl.feature_columns("occrate")[:-1] == etl.feature_columns("available")[:-1]
    with pytest.raises(ValueError):
        etl.feature_columns("speed")
