"""The algebra behind the HIP pipeline (aggregate-first + composed (C,F) weights) reproduces the
oracle's outputs and gradients within fp32 round-off (north_star tolerance 1e-5)."""
import numpy as np
import pytest
import torch

from conftest import region_lists
from fused_math import dense_ops, forward_fused
from oracle import model as M


@pytest.mark.parametrize("t_in,t_out", [(6, 1), (12, 3)])
def test_fused_formulation_matches_oracle_regional(tpims, t_in, t_out):
    n = tpims["node_data"].shape[0]
    x = tpims["node_data"][:, :, 3:3 + t_in].contiguous()
    y = tpims["node_data"][:, -1, 3 + t_in:3 + t_in + t_out]
    ri, rw = region_lists(tpims)
    p = {k: v.clone().requires_grad_(True) for k, v in
         M.init_params("RegionalTemporalGCN", 8, t_in, t_out, num_nodes=n, seed=9).items()}
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    pred_o, hid_o = M.regional_temporal_gcn(p, x, tpims["edge_index"], ri, rw)
    torch.mean((pred_o - y) ** 2).backward()
    a, ls = dense_ops(tpims["edge_index"], None, ri, rw, n, torch.float32)
    pred_f, hid_f = forward_fused(q, x, a, ls, regional=True)
    torch.mean((pred_f - y) ** 2).backward()
    assert float((pred_o - pred_f).abs().max()) < 1e-5
    assert float((hid_o - hid_f).abs().max()) < 1e-5
    for k in p:
        if k in M.UNUSED_PARAMS:
            continue
        np.testing.assert_allclose(q[k].grad.numpy(), p[k].grad.numpy(), atol=1e-5, rtol=1e-4, err_msg=k)


def test_fused_formulation_matches_oracle_temporal(tpims):
    t_in, t_out = 6, 1
    n = tpims["node_data"].shape[0]
    x = tpims["node_data"][:, :, 3:3 + t_in].contiguous()
    y = tpims["node_data"][:, -1, 3 + t_in:3 + t_in + t_out]
    p = {k: v.clone().requires_grad_(True) for k, v in M.init_params("TemporalGCN", 8, t_in, t_out, seed=9).items()}
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    pred_o, hid_o = M.temporal_gcn(p, x, tpims["edge_index"], tpims["edge_attr"])
    torch.mean((pred_o - y) ** 2).backward()
    a, ls = dense_ops(tpims["edge_index"], tpims["edge_attr"], [tpims["edge_index"]], [tpims["edge_attr"]], n, torch.float32)
    pred_f, hid_f = forward_fused(q, x, a, ls, regional=False)
    torch.mean((pred_f - y) ** 2).backward()
    assert float((pred_o - pred_f).abs().max()) < 1e-5
    assert float((hid_o - hid_f).abs().max()) < 1e-5
    for k in p:
        if k in M.UNUSED_PARAMS_TEMPORAL:
            continue
        np.testing.assert_allclose(q[k].grad.numpy(), p[k].grad.numpy(), atol=1e-5, rtol=1e-4, err_msg=k)
