"""Opt-in bf16x3 split arithmetic (regt_set_gemm_mode(1), gemm_split.h): same parity bars as the fp32-MFMA default.

The split represents every fp32 operand exactly as three bf16 pieces and keeps the six partial products of weight
>= 2^-16; these tests hold it to the tolerances of the default path -- against float64 for the GEMM entry point and
against the reference's golden vectors / the oracle for the whole forward + backward."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, check_grads_against_golden, load_npz, region_lists
from oracle import model as M

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture()
def split_mode():
    import regtgcn_amd as R
    lib = R.load_library()
    prev = lib.regt_set_gemm_mode(1)
    yield R
    lib.regt_set_gemm_mode(prev)


@pytest.mark.parametrize("m,k,n,act", [(4096, 288, 512, 0), (1000, 100, 36, 1), (130, 2048, 256, 2), (7, 4, 4, 0)])
def test_split_linear_is_as_accurate_as_fp32_mfma(split_mode, m, k, n, act):
    R = split_mode
    lib = R.load_library()
    g = torch.Generator().manual_seed(m + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / max(1.0, k ** 0.5)
    b = torch.randn(n, generator=g)
    want = a.double() @ w.double().t() + b.double()
    if act == 1:
        want = torch.nn.functional.leaky_relu(want, 0.01)
    elif act == 2:
        want = torch.relu(want)
    got = R.ops.linear(a.cuda(), w.cuda(), b.cuda(), act).cpu()
    err_split = float((got.double() - want).abs().max())
    lib.regt_set_gemm_mode(0)
    err_fp32 = float((R.ops.linear(a.cuda(), w.cuda(), b.cuda(), act).cpu().double() - want).abs().max())
    lib.regt_set_gemm_mode(1)
    assert err_split < 2e-5                         # the bar of test_linear_matches_torch_fp32
    assert err_split <= 2.0 * err_fp32 + 1e-7       # and no worse than the fp32 matrix pipe itself


def test_split_handles_extreme_magnitudes(split_mode):
    """bf16 shares fp32's exponent range: the split neither overflows nor flushes where fp32 does not."""
    R = split_mode
    a = torch.tensor([[1e30, 3e-30, 1.0, 65504.0]] * 3)           # all products positive: no cancellation
    w = torch.tensor([[1e-30, 1e30, 1.0, 1.0 / 65504.0], [2.0, 0.0, 1.0, 0.0]])
    got = R.ops.linear(a.cuda(), w.cuda()).cpu().double()
    want = a.double() @ w.double().t()
    assert torch.isfinite(got).all()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-6)


@pytest.mark.parametrize("tag", ["in6_out1", "in12_out3", "ckpt"])
def test_split_regt_matches_reference_goldens(split_mode, tpims, tag):
    R = split_mode
    g = load_npz(f"golden_regt_{tag}.npz")
    t_in, t_out, w0 = int(g["t_in"]), int(g["t_out"]), int(g["window"])
    n = tpims["node_data"].shape[0]
    if tag == "ckpt":
        p = torch.load(os.path.join(GOLDEN, "ref_ckpt_in6_out1_epoch50.pt"), map_location="cpu", weights_only=True)
    else:
        p = M.init_params("RegionalTemporalGCN", 8, t_in, t_out, num_nodes=n, seed=int(g["seed"]))
    x = tpims["node_data"][:, :, w0:w0 + t_in].contiguous()
    y = tpims["node_data"][:, -1, w0 + t_in:w0 + t_in + t_out].contiguous()
    mod = R.RegionalTemporalGCN(8, n, t_in, t_out)
    mod.load_state_dict(p)
    mod = mod.cuda()
    ri, rw = region_lists(tpims)
    pred, hidden = mod(x.cuda(), tpims["edge_index"].cuda(), *[t.cuda() for t in ri], *[t.cuda() for t in rw])
    loss = torch.mean((pred - y.cuda()) ** 2)
    loss.backward()
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g["pred"], atol=TOL)
    np.testing.assert_allclose(hidden.detach().cpu().numpy(), g["hidden"], atol=TOL)
    assert abs(float(loss.detach()) - float(g["loss"][0])) < 1e-6
    grads = {k: (None if q.grad is None else q.grad.cpu()) for k, q in mod.named_parameters()}
    check_grads_against_golden(g, grads, atol=TOL, rtol=1e-4)


def test_split_matches_oracle_on_synthetic_regional_graph(split_mode):
    R = split_mode
    n, e, regions, f, t, o = 1500, 15000, 8, 32, 12, 1
    g = R.data.synthetic_regional_graph(n, e, regions, seed=n)
    (x, y), = R.data.synthetic_snapshots(n, f, t, o, 1, seed=n)
    p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=3)
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    pred_o, hid_o = M.regional_temporal_gcn(po, x, g.edge_index, g.region_index, g.region_attr)
    torch.mean((pred_o - y) ** 2).backward()
    mod = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions)
    mod.load_state_dict(p, strict=True)
    mod = mod.cuda()
    pred, hidden = mod(x.cuda(), g.edge_index.cuda(), [i.cuda() for i in g.region_index], [a.cuda() for a in g.region_attr])
    torch.mean((pred - y.cuda()) ** 2).backward()
    assert float((pred.cpu() - pred_o).abs().max()) < TOL
    assert float((hidden.cpu() - hid_o).abs().max()) < TOL
    for k, q in mod.named_parameters():
        if k in M.UNUSED_PARAMS:
            continue
        np.testing.assert_allclose(q.grad.cpu().numpy(), po[k].grad.numpy(), atol=TOL, rtol=1e-4, err_msg=k)
