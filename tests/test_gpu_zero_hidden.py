"""GraphSAGE / GAT base blocks of the TGCN cell (second half) on the GPU: the HIP modules
GraphSAGETemporalGCN / GATTemporal against the golden vectors recorded from the reference's own modules and against the
oracle on synthetic graphs; the new operator entry points (regt_mean_csr, regt_gat_forward/backward, regt_cell0_*) against
dense formulas and torch autograd.  Tolerance: north_star's 1e-5 (fp32)."""
import numpy as np
import pytest
import torch

from conftest import check_grads_against_golden, load_npz
from oracle import graph_ops as G
from oracle import model as M

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def R():
    import regtgcn_amd
    regtgcn_amd.load_library()
    return regtgcn_amd


def _graph(n, e, seed, loops=True):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, n - 2, (e,), generator=g)            # the last two nodes receive no edge
    if loops:
        src[3], dst[3] = 5, 5                                    # a self loop
        src[7], dst[7] = src[8], dst[8]                          # a duplicate edge
    return torch.stack([src, dst])


def test_mean_csr_matches_dense_mean_operator(R):
    n = 50
    ei = _graph(n, 300, 1)
    rp, col, val = R.graph.mean_csr(ei.cuda(), n)
    dense = torch.zeros(n, n, dtype=torch.float64)
    rp_h, col_h, val_h = rp.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
    for i in range(n):
        for k in range(rp_h[i], rp_h[i + 1]):
            dense[i, col_h[k]] += float(val_h[k])
    np.testing.assert_allclose(dense.numpy(), G.dense_mean_operator(ei, n).numpy(), atol=1e-6)
    assert rp_h[n] == ei.shape[1] and rp_h[n - 1] == rp_h[n]      # every listed edge kept; isolated rows empty


@pytest.mark.parametrize("n,e,t,f", [(40, 200, 3, 8), (300, 3000, 12, 32), (64, 400, 5, 12), (100, 900, 2, 64)])
def test_gat_aggregate_forward_and_score_gradients(R, n, e, t, f):
    """regt_gat_forward / regt_gat_backward against torch autograd of the dense attention formula, period by period."""
    ei = _graph(n, e, n)
    g = torch.Generator().manual_seed(f)
    x = torch.rand(n, f, t, generator=g)
    us = (torch.randn(f, generator=g) * 0.7).requires_grad_(True)
    ud = (torch.randn(f, generator=g) * 0.7).requires_grad_(True)
    go = torch.randn(n, t, f, generator=g)
    cnt = torch.zeros(n, n)
    for k in range(ei.shape[1]):
        s, d = int(ei[0, k]), int(ei[1, k])
        if s != d:
            cnt[d, s] += 1.0
    cnt = cnt + torch.eye(n)
    outs = []
    for tt in range(t):
        xt = x[:, :, tt]
        score = torch.nn.functional.leaky_relu((xt @ ud).view(-1, 1) + (xt @ us).view(1, -1), 0.2)
        w = cnt * torch.exp(score - score.max(dim=1, keepdim=True).values)
        outs.append((w / w.sum(dim=1, keepdim=True)) @ xt)
    want = torch.stack(outs, dim=1)                               # (N, T, F)
    (want * go).sum().backward()
    pat = R.graph.prepare_attention_pattern(ei.cuda(), n)
    xp = R.ops.pack_x(x.cuda())
    usc, udc = us.detach().cuda().requires_grad_(True), ud.detach().cuda().requires_grad_(True)
    got = R.functional.GatAggregateFunction.apply(xp, usc, udc, pat, 0.2)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), atol=TOL)
    (got * go.cuda()).sum().backward()
    scale = max(1.0, float(us.grad.abs().max()), float(ud.grad.abs().max()))
    np.testing.assert_allclose(usc.grad.cpu().numpy(), us.grad.numpy(), atol=2e-5 * scale, rtol=1e-4)
    np.testing.assert_allclose(udc.grad.cpu().numpy(), ud.grad.numpy(), atol=2e-5 * scale, rtol=1e-4)
    # bit-reproducible (no float atomics in either pass)
    usc2, udc2 = us.detach().cuda().requires_grad_(True), ud.detach().cuda().requires_grad_(True)
    got2 = R.functional.GatAggregateFunction.apply(xp, usc2, udc2, pat, 0.2)
    (got2 * go.cuda()).sum().backward()
    assert torch.equal(got, got2) and torch.equal(usc.grad, usc2.grad) and torch.equal(udc.grad, udc2.grad)


@pytest.mark.parametrize("name,short", [("GraphSAGETemporalGCN", "sage"), ("GATTemporal", "gat")])
@pytest.mark.parametrize("tag", ["in6_out1", "in12_out3"])
def test_zero_hidden_models_match_reference_goldens(R, tpims, name, short, tag):
    g = load_npz(f"golden_{short}_{tag}.npz")
    t_in, t_out, w0 = int(g["t_in"]), int(g["t_out"]), int(g["window"])
    n = tpims["node_data"].shape[0]
    p = M.init_params(name, 8, t_in, t_out, num_nodes=n, seed=int(g["seed"]))
    x = tpims["node_data"][:, :, w0:w0 + t_in].contiguous()
    y = tpims["node_data"][:, -1, w0 + t_in:w0 + t_in + t_out].contiguous()
    mod = getattr(R, name)(node_features=8, num_nodes=n, periods=t_in, output_dim=t_out)
    mod.load_state_dict(p, strict=True)
    mod = mod.cuda()
    pred, hidden = mod(x.cuda(), tpims["edge_index"].cuda(), tpims["edge_attr"].cuda())       # positional call of run.py:214
    loss = torch.mean((pred - y.cuda()) ** 2)
    loss.backward()
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g["pred"], atol=TOL)
    np.testing.assert_allclose(hidden.detach().cpu().numpy(), g["hidden"], atol=TOL)
    assert abs(float(loss.detach()) - float(g["loss"][0])) < 1e-6
    grads = {k: (None if q.grad is None else q.grad.cpu()) for k, q in mod.named_parameters()}
    check_grads_against_golden(g, grads, atol=TOL, rtol=1e-4)
    pre = "tgnn." if short == "sage" else "gat."
    assert float(grads[f"{pre}_base_tgcn.linear_r.weight"].abs().max()) == 0.0      # zeros, not None: as in the reference
    if short == "sage":
        for k in M.UNUSED_PARAMS_SAGE:
            assert grads[k] is None


@pytest.mark.parametrize("name", ["GraphSAGETemporalGCN", "GATTemporal"])
@pytest.mark.parametrize("n,e,f,t,o,hidden", [(1500, 15000, 32, 12, 1, 256), (400, 3000, 8, 6, 3, 256), (700, 5000, 12, 5, 2, 132),
                                              (300, 2000, 7, 4, 1, 256), (250, 1500, 6, 3, 2, 256)])     # F = 7, 6: padded feature rows
def test_zero_hidden_models_match_oracle_on_synthetic_graphs(R, name, n, e, f, t, o, hidden):
    ei = _graph(n, e, n + f)
    gen = torch.Generator().manual_seed(n)
    x, y = torch.rand(n, f, t, generator=gen), torch.rand(n, o, generator=gen)
    p = M.init_params(name, f, t, o, num_nodes=n, seed=5, hidden=hidden)
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    fwd = M.graphsage_temporal_gcn if name == "GraphSAGETemporalGCN" else M.gat_temporal
    pred_o, hid_o = fwd(po, x, ei)
    torch.mean((pred_o - y) ** 2).backward()
    mod = getattr(R, name)(node_features=f, num_nodes=n, periods=t, output_dim=o, hidden_channels=hidden)
    mod.load_state_dict(p, strict=True)
    mod = mod.cuda()
    pred, hid = mod(x.cuda(), ei.cuda(), None)
    torch.mean((pred - y.cuda()) ** 2).backward()
    assert float((pred.detach().cpu() - pred_o.detach()).abs().max()) < TOL
    assert float((hid.detach().cpu() - hid_o.detach()).abs().max()) < TOL
    for k, q in mod.named_parameters():
        if po[k].grad is None:
            assert q.grad is None, k
            continue
        np.testing.assert_allclose(q.grad.cpu().numpy(), po[k].grad.numpy(), atol=TOL, rtol=1e-4, err_msg=k)


@pytest.mark.parametrize("name", ["GraphSAGETemporalGCN", "GATTemporal"])
def test_dead_reset_gate_gets_zero_gradients_from_either_output(R, name):
    """A loss built from ``hidden`` alone still hands the reset gate's parameters zeros (not None), as the reference's autograd does."""
    n, e, f, t, o = 200, 1500, 8, 4, 1
    ei = _graph(n, e, 3)
    p = M.init_params(name, f, t, o, num_nodes=n, seed=5)
    mod = getattr(R, name)(node_features=f, num_nodes=n, periods=t, output_dim=o)
    mod.load_state_dict(p, strict=True)
    mod = mod.cuda()
    _, hid = mod(torch.rand(n, f, t).cuda(), ei.cuda(), None)
    hid.sum().backward()
    pre = "tgnn." if name == "GraphSAGETemporalGCN" else "gat."
    g = dict(mod.named_parameters())[f"{pre}_base_tgcn.linear_r.weight"].grad
    assert g is not None and float(g.abs().max()) == 0.0
    gh = dict(mod.named_parameters())["linear1.weight"].grad                    # the head is not on hidden's path: no gradient, or
    assert gh is None or float(gh.abs().max()) == 0.0                           # (an autograd Function materialises it) zeros


@pytest.mark.parametrize("name", ["GraphSAGETemporalGCN", "GATTemporal", "ConvStackedTemporalGCN"])
def test_snapshot_batch_of_the_other_models_equals_per_snapshot_runs(R, name):
    """--snap_batch for the models whose operators are not the RegT-GCN ones: SAGE's mean, GAT's softmax over in-neighbours and the
    stacked GCNConv are all local to a node's in-neighbours, so B disjoint copies of the graph (prepare_graph(copies=B)) are B
    independent snapshots: predictions of every copy and the summed gradients equal the per-snapshot runs."""
    n, e, f, t, o, b = 300, 2500, 8, 6, 2, 3
    ei = _graph(n, e, 77)
    gen = torch.Generator().manual_seed(9)
    xs = [torch.rand(n, f, t, generator=gen) for _ in range(b)]
    ys = [torch.rand(n, o, generator=gen) for _ in range(b)]
    if name == "ConvStackedTemporalGCN":
        p = M.init_params(name, f, t, o, seed=5)
        for layer in range(2, 6):
            p[f"tgnn.conv{layer}.lin.weight"] *= 0.5
        mod = R.ConvStackedTemporalGCN(f, t, o)
        w = (torch.rand(ei.shape[1], generator=gen) * 100 + 1).cuda()
        prep = lambda copies: mod.prepare_graph(ei.cuda(), w, n, copies=copies)
    else:
        p = M.init_params(name, f, t, o, num_nodes=n, seed=5)
        mod = getattr(R, name)(node_features=f, num_nodes=n, periods=t, output_dim=o)
        prep = lambda copies: mod.prepare_graph(ei.cuda(), n, copies=copies)
    mod.load_state_dict(p, strict=True)
    mod = mod.cuda()
    g1, gb = prep(1), prep(b)
    preds = []
    for x, y in zip(xs, ys):
        pred, _ = mod.forward_prepared(x.cuda(), g1)
        (((pred - y.cuda()) ** 2).sum() / (n * o)).backward()
        preds.append(pred.detach())
    want = {k: q.grad.clone() for k, q in mod.named_parameters() if q.grad is not None}
    mod.zero_grad(set_to_none=True)
    xb, yb = torch.cat(xs).cuda(), torch.cat(ys).cuda()
    pred, _ = mod.forward_prepared(xb, gb)
    (((pred - yb) ** 2).sum() / (n * o)).backward()
    scale = max(1.0, float(pred.abs().max()))
    for i in range(b):
        assert float((pred[i * n:(i + 1) * n].detach() - preds[i]).abs().max()) < 2e-5 * scale
    for k, q in mod.named_parameters():
        if k not in want:
            assert q.grad is None, k
            continue
        np.testing.assert_allclose(q.grad.cpu().numpy(), want[k].cpu().numpy(), rtol=2e-4, atol=2e-5 * max(1.0, float(want[k].abs().max())), err_msg=k)
