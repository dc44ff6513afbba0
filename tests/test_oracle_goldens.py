"""Pin the oracle's orchestration restatement (oracle/model.py, oracle/loop.py) to the golden
vectors produced by the reference's own model files (oracle/make_goldens.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, check_grads_against_golden, load_npz, region_lists
from oracle import loop as oloop
from oracle import model as M

TOL = 2e-6   # same arithmetic, same op order: only last-bit differences are expected


def _leaf(params):
    return {k: v.clone().requires_grad_(True) for k, v in params.items()}


def test_cell_golden():
    g = load_npz("golden_cell.npz")
    p = {k[3:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("p__")}
    ei, ew = torch.from_numpy(g["edge_index"]), torch.from_numpy(g["edge_weight"])
    x, h = torch.from_numpy(g["x"]), torch.from_numpy(g["h"])
    for tag, w in (("unit", None), ("weighted", ew)):
        pl = _leaf(p)
        out = M.tgcn_cell(pl, "", x, ei, w, h)
        np.testing.assert_allclose(out.detach().numpy(), g[f"out_{tag}"], atol=TOL)
        (out ** 2).sum().backward()
        for name, t in pl.items():
            np.testing.assert_allclose(t.grad.numpy(), g[f"g_{tag}__" + name.replace(".", "__")], atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("tag", ["in6_out1", "in12_out1", "in12_out3", "in6_out3", "ckpt"])
def test_regt_golden(tpims, tag):
    g = load_npz(f"golden_regt_{tag}.npz")
    t_in, t_out, w0 = int(g["t_in"]), int(g["t_out"]), int(g["window"])
    n = tpims["node_data"].shape[0]
    if tag == "ckpt":
        p = torch.load(os.path.join(GOLDEN, "ref_ckpt_in6_out1_epoch50.pt"), map_location="cpu", weights_only=True)
    else:
        p = M.init_params("RegionalTemporalGCN", 8, t_in, t_out, num_nodes=n, seed=int(g["seed"]))
    chk = float(sum(v.double().abs().sum() for v in p.values()))
    assert abs(chk - float(g["param_checksum"][0])) < 1e-6 * chk, "seeded parameter stream drifted"
    p = _leaf(p)
    x = tpims["node_data"][:, :, w0:w0 + t_in].contiguous()
    y = tpims["node_data"][:, -1, w0 + t_in:w0 + t_in + t_out].contiguous()
    ri, rw = region_lists(tpims)
    pred, hidden = M.regional_temporal_gcn(p, x, tpims["edge_index"], ri, rw)
    np.testing.assert_allclose(pred.detach().numpy(), g["pred"], atol=TOL)
    np.testing.assert_allclose(hidden.detach().numpy(), g["hidden"], atol=TOL)
    loss = torch.mean((pred - y) ** 2)
    assert abs(float(loss.detach()) - float(g["loss"][0])) < 1e-6
    loss.backward()
    check_grads_against_golden(g, {k: v.grad for k, v in p.items()}, atol=2e-6)
    for name in M.UNUSED_PARAMS:
        assert p[name].grad is None


@pytest.mark.parametrize("tag", ["in6_out1", "in12_out3"])
def test_temporal_gcn_golden(tpims, tag):
    g = load_npz(f"golden_tgcn_{tag}.npz")
    t_in, t_out, w0 = int(g["t_in"]), int(g["t_out"]), int(g["window"])
    p = _leaf(M.init_params("TemporalGCN", 8, t_in, t_out, seed=int(g["seed"])))
    x = tpims["node_data"][:, :, w0:w0 + t_in].contiguous()
    y = tpims["node_data"][:, -1, w0 + t_in:w0 + t_in + t_out].contiguous()
    pred, hidden = M.temporal_gcn(p, x, tpims["edge_index"], tpims["edge_attr"])
    np.testing.assert_allclose(pred.detach().numpy(), g["pred"], atol=TOL)
    np.testing.assert_allclose(hidden.detach().numpy(), g["hidden"], atol=TOL)
    loss = torch.mean((pred - y) ** 2)
    assert abs(float(loss.detach()) - float(g["loss"][0])) < 1e-6
    loss.backward()
    check_grads_against_golden(g, {k: v.grad for k, v in p.items()}, atol=2e-6)
    for name in M.UNUSED_PARAMS_TEMPORAL:
        assert p[name].grad is None


@pytest.mark.parametrize("tag", ["in6_out1", "in12_out3"])
def test_conv_stacked_golden(tpims, tag):
    """the oracle's ConvStackedTemporalGCN against the reference module's outputs."""
    g = load_npz(f"golden_convstack_{tag}.npz")
    t_in, t_out, w0 = int(g["t_in"]), int(g["t_out"]), int(g["window"])
    p0 = M.init_params("ConvStackedTemporalGCN", 8, t_in, t_out, seed=int(g["seed"]))
    chk = float(sum(v.double().abs().sum() for v in p0.values()))
    assert abs(chk - float(g["param_checksum"][0])) < 1e-6 * chk, "seeded parameter stream drifted"
    p = _leaf(p0)
    x = tpims["node_data"][:, :, w0:w0 + t_in].contiguous()
    y = tpims["node_data"][:, -1, w0 + t_in:w0 + t_in + t_out].contiguous()
    pred, hidden = M.conv_stacked_temporal_gcn(p, x, tpims["edge_index"], tpims["edge_attr"])
    assert hidden.shape == (x.shape[0], M.CONVSTACK_HIDDEN)
    np.testing.assert_allclose(pred.detach().numpy(), g["pred"], atol=TOL)
    np.testing.assert_allclose(hidden.detach().numpy(), g["hidden"], atol=TOL)
    loss = torch.mean((pred - y) ** 2)
    assert abs(float(loss.detach()) - float(g["loss"][0])) < 1e-6
    loss.backward()
    check_grads_against_golden(g, {k: v.grad for k, v in p.items()}, atol=2e-6)
    for name in M.UNUSED_PARAMS_CONVSTACK:
        assert p[name].grad is None


@pytest.mark.parametrize("name,short", [("GraphSAGETemporalGCN", "sage"), ("GATTemporal", "gat")])
@pytest.mark.parametrize("tag", ["in6_out1", "in12_out3"])
def test_zero_hidden_models_golden(tpims, name, short, tag):
    """SURVEY 8(f) rank 4, second half: the oracle's GraphSAGE / GAT models against the reference modules' outputs."""
    g = load_npz(f"golden_{short}_{tag}.npz")
    t_in, t_out, w0 = int(g["t_in"]), int(g["t_out"]), int(g["window"])
    n = tpims["node_data"].shape[0]
    p0 = M.init_params(name, 8, t_in, t_out, num_nodes=n, seed=int(g["seed"]))
    chk = float(sum(v.double().abs().sum() for v in p0.values()))
    assert abs(chk - float(g["param_checksum"][0])) < 1e-6 * chk, "seeded parameter stream drifted"
    p = _leaf(p0)
    x = tpims["node_data"][:, :, w0:w0 + t_in].contiguous()
    y = tpims["node_data"][:, -1, w0 + t_in:w0 + t_in + t_out].contiguous()
    fwd = M.graphsage_temporal_gcn if short == "sage" else M.gat_temporal
    pred, hidden = fwd(p, x, tpims["edge_index"], tpims["edge_attr"])
    np.testing.assert_allclose(pred.detach().numpy(), g["pred"], atol=TOL)
    np.testing.assert_allclose(hidden.detach().numpy(), g["hidden"], atol=TOL)
    loss = torch.mean((pred - y) ** 2)
    assert abs(float(loss.detach()) - float(g["loss"][0])) < 1e-6
    loss.backward()
    check_grads_against_golden(g, {k: v.grad for k, v in p.items()}, atol=2e-6)
    pre = "tgnn." if short == "sage" else "gat."
    # the reset gate multiplies the zero hidden state: its parameters receive an all-zero gradient (not None)
    assert float(p[f"{pre}_base_tgcn.linear_r.weight"].grad.abs().max()) == 0.0
    assert float(p[f"{pre}_base_tgcn.linear_z.weight"].grad[:, 256:].abs().max()) == 0.0     # H-half columns see H = 0
    if short == "sage":
        for k in M.UNUSED_PARAMS_SAGE:
            assert p[k].grad is None


def test_loop_golden(tpims):
    g = load_npz("golden_loop.npz")
    t_in, t_out = int(g["t_in"]), int(g["t_out"])
    n_train, n_test, epochs = int(g["n_train"]), int(g["n_test"]), int(g["epochs"])
    n = tpims["node_data"].shape[0]
    names = [str(s) for s in g["names"]]
    p0 = M.init_params("RegionalTemporalGCN", 8, t_in, t_out, num_nodes=n, seed=int(g["seed"]))
    p = {k: p0[k].clone().requires_grad_(True) for k in names}   # reference named_parameters() order
    opt = torch.optim.RMSprop(list(p.values()), lr=1e-3, weight_decay=1e-4)
    xs, ys = oloop.make_windows(tpims["node_data"][:, :, :t_in + t_out + n_train + n_test - 1], t_in, t_out)
    assert len(xs) == n_train + n_test
    ri, rw = region_lists(tpims)
    fwd = lambda prm, x: M.regional_temporal_gcn(prm, x.contiguous(), tpims["edge_index"], ri, rw)
    losses, metrics = [], []
    for ep in range(epochs):
        last, all_l = oloop.train_epoch(p, fwd, xs[:n_train], ys[:n_train], opt)
        losses += all_l
        assert last == all_l[-1]
        metrics.append(oloop.evaluate(p, fwd, xs[n_train:], ys[n_train:]))
        sums = [float(p[k].detach().double().sum()) for k in names]
        np.testing.assert_allclose(sums, g["param_sums"][ep], atol=5e-4, rtol=1e-5)
    np.testing.assert_allclose(losses, g["losses"], atol=2e-6)
    np.testing.assert_allclose(np.array(metrics), g["metrics"], atol=2e-6)


@pytest.mark.parametrize("t_in,t_out", [(6, 1), (12, 3), (24, 12)])
def test_window_construction_matches_the_reference_get(tpims, t_in, t_out):
    """golden_windows.npz was written by the reference's own TruckParkingDataset2.get() (load_dataset.py:442-471, run under a
    StaticGraphTemporalSignal stand-in: oracle/make_window_golden.py) on the fixture's node data: the package's window builder
    (data.snapshot_windows, what train.py / evaluate.py feed the model), the oracle's (loop.make_windows) and the stacked
    device-side store of the batched loop (train.WindowStore) reproduce every window -- count, three whole windows, and two
    checksums per window over all of them -- exactly (windows are slices: no arithmetic)."""
    import regtgcn_amd as R
    g = load_npz("golden_windows.npz")
    tag = f"in{t_in}_out{t_out}"
    count, pick = int(g[f"{tag}_count"]), [int(i) for i in g[f"{tag}_pick"]]

    def sums(arrs):
        out = np.zeros((len(arrs), 2))
        for i, a in enumerate(arrs):
            a = a.double().numpy()
            w = np.arange(1, a.size + 1, dtype=np.float64).reshape(a.shape)
            out[i] = (a.sum(), (a * w).sum() / a.size)
        return out

    for name, (xs, ys) in (("data.snapshot_windows", R.data.snapshot_windows(tpims["node_data"], t_in, t_out)),
                           ("oracle.loop.make_windows", oloop.make_windows(tpims["node_data"], t_in, t_out))):
        assert len(xs) == len(ys) == count, name
        for k, i in enumerate(pick):
            assert tuple(xs[i].shape) == g[f"{tag}_features"][k].shape, name
            assert xs[i].is_contiguous() or name.startswith("oracle")          # the package hands the kernels contiguous (N,F,T) snapshots
            np.testing.assert_array_equal(xs[i].numpy(), g[f"{tag}_features"][k], err_msg=name)
            np.testing.assert_array_equal(ys[i].numpy(), g[f"{tag}_targets"][k], err_msg=name)
        np.testing.assert_allclose(sums(xs), g[f"{tag}_feature_sums"], rtol=1e-12, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(sums(ys), g[f"{tag}_target_sums"], rtol=1e-12, atol=1e-12, err_msg=name)
    xs, ys = R.data.snapshot_windows(tpims["node_data"], t_in, t_out)
    store = R.train.WindowStore(xs, ys)
    assert len(store) == count
    xb, yb = store.batch(pick[1], 2) if pick[1] + 2 <= count else store.batch(pick[1], 1)
    n = tpims["node_data"].shape[0]
    np.testing.assert_array_equal(xb[:n].numpy(), g[f"{tag}_features"][1])          # a batch is B windows stacked along the nodes
    np.testing.assert_array_equal(yb[:n].numpy(), g[f"{tag}_targets"][1])
    if xb.shape[0] == 2 * n:
        np.testing.assert_array_equal(xb[n:].numpy(), xs[pick[1] + 1].numpy())
