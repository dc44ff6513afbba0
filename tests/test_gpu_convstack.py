"""ConvStackedTemporalGCN on the HIP path: the reference module's golden vectors on the TPIMS
fixture, the oracle on a larger synthetic graph (directed: A_hat is not symmetric, the transposed CSR matters), the
wide aggregation and its transpose on their own, and state_dict compatibility."""
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
    import regtgcn_amd as R
    R.load_library()
    return R


def test_state_dict_layout_matches_reference(R):
    mod = R.ConvStackedTemporalGCN(8, 6, 1)
    want = M.init_params("ConvStackedTemporalGCN", 8, 6, 1, seed=0)
    got = mod.state_dict()
    assert set(got) == set(want)
    for k, v in want.items():
        assert tuple(got[k].shape) == tuple(v.shape), k
    mod.load_state_dict(want, strict=True)


# collapse = True: the five activation-free conv layers as ONE contraction on [A^5 x | A^j 1] (nn.ConvStackedTemporalGCN);
# False: layer by layer, the learned hidden state aggregated at width T*512 (the reference's evaluation order)
@pytest.mark.parametrize("collapse", [True, False], ids=["collapsed", "layerwise"])
@pytest.mark.parametrize("tag", ["in6_out1", "in12_out3"])
def test_convstack_matches_reference_goldens(R, tpims, tag, collapse):
    g = load_npz(f"golden_convstack_{tag}.npz")
    t_in, t_out, w0 = int(g["t_in"]), int(g["t_out"]), int(g["window"])
    p = M.init_params("ConvStackedTemporalGCN", 8, t_in, t_out, seed=int(g["seed"]))
    x = tpims["node_data"][:, :, w0:w0 + t_in].contiguous()
    y = tpims["node_data"][:, -1, w0 + t_in:w0 + t_in + t_out].contiguous()
    mod = R.ConvStackedTemporalGCN(8, t_in, t_out)
    mod.load_state_dict(p, strict=True)
    mod = mod.cuda()
    mod.collapse = collapse
    pred, hidden = mod(x.cuda(), tpims["edge_index"].cuda(), tpims["edge_attr"].cuda())
    loss = torch.mean((pred - y.cuda()) ** 2)
    loss.backward()
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g["pred"], atol=TOL)
    np.testing.assert_allclose(hidden.detach().cpu().numpy(), g["hidden"], atol=TOL)
    assert abs(float(loss.detach()) - float(g["loss"][0])) < 1e-6
    grads = {k: (None if q.grad is None else q.grad.cpu()) for k, q in mod.named_parameters()}
    check_grads_against_golden(g, grads, atol=TOL, rtol=1e-4)
    for name in M.UNUSED_PARAMS_CONVSTACK:
        assert grads[name] is None


@pytest.mark.parametrize("collapse", [True, False], ids=["collapsed", "layerwise"])
def test_convstack_matches_oracle_on_directed_synthetic_graph(R, collapse):
    n, e, f, t, o = 700, 5000, 8, 6, 2
    g = R.data.synthetic_regional_graph(n, e, 3, seed=21)          # directed edge list
    (x, y), = R.data.synthetic_snapshots(n, f, t, o, 1, seed=21)
    p = M.init_params("ConvStackedTemporalGCN", f, t, o, seed=22)
    for layer in range(2, 6):                                       # keep five un-normalised 512-wide layers O(1)
        p[f"tgnn.conv{layer}.lin.weight"] *= 0.5
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    pred_o, hid_o = M.conv_stacked_temporal_gcn(po, x, g.edge_index, g.edge_attr)
    torch.mean((pred_o - y) ** 2).backward()
    mod = R.ConvStackedTemporalGCN(f, t, o)
    mod.load_state_dict(p, strict=True)
    mod = mod.cuda()
    mod.collapse = collapse
    pred, hidden = mod(x.cuda(), g.edge_index.cuda(), g.edge_attr.cuda())
    torch.mean((pred - y.cuda()) ** 2).backward()
    scale = max(1.0, float(hid_o.detach().abs().max()))
    assert float((pred.detach().cpu() - pred_o.detach()).abs().max()) < TOL * scale
    assert float((hidden.detach().cpu() - hid_o.detach()).abs().max()) < TOL * scale
    for k, q in mod.named_parameters():
        if k in M.UNUSED_PARAMS_CONVSTACK:
            assert q.grad is None
            continue
        want = po[k].grad
        np.testing.assert_allclose(q.grad.cpu().numpy(), want.numpy(), atol=TOL * max(1.0, float(want.abs().max())), rtol=1e-4, err_msg=k)


@pytest.mark.parametrize("n,e,width", [(300, 2500, 6144), (5000, 40000, 3072), (64, 200, 2052)])
def test_wide_aggregation_and_its_transpose(R, n, e, width):
    """Rows wider than 2048 floats (T*512) through regt_spmm_csr, and A_hat^T through the transposed CSR."""
    g = R.data.synthetic_regional_graph(n, e, 2, seed=n)
    op = R.graph.prepare_gcn_operator(g.edge_index.cuda(), g.edge_attr.cuda(), n)
    h = torch.randn(n, width, generator=torch.Generator().manual_seed(1))
    src, dst, w = G.gcn_norm_edges(g.edge_index, g.edge_attr, n, torch.float64)
    dense = torch.zeros(n, n, dtype=torch.float64).index_put_((dst, src), w, accumulate=True)
    y = R.ops.spmm_csr(op.rowptr, op.col, op.val, h.cuda()).cpu().double()
    yt = R.ops.spmm_csr(op.t_rowptr, op.t_col, op.t_val, h.cuda()).cpu().double()
    assert float((y - dense @ h.double()).abs().max()) < 2e-5
    assert float((yt - dense.t() @ h.double()).abs().max()) < 2e-5
