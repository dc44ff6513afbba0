"""Property test of the whole forward + backward: arbitrary small shapes (node counts below one tile, a single period,
1..5 regions, feature widths 4..32, hidden widths that are not multiples of the tile, horizons 1..4) against the oracle."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import model as M

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def R():
    import regtgcn_amd as R
    R.load_library()
    return R


@settings(max_examples=14, deadline=None, derandomize=True)
@given(n=st.integers(5, 220), t=st.integers(1, 9), f=st.sampled_from([4, 8, 12, 32]), o=st.integers(1, 4),
       regions=st.integers(1, 5), hidden=st.sampled_from([8, 64, 100, 256]), mode=st.integers(0, 1), seed=st.integers(0, 10_000))
def test_arbitrary_small_shapes_match_oracle(R, n, t, f, o, regions, hidden, mode, seed):
    e = min(n * (n - 1), 6 * n)
    g = R.data.synthetic_regional_graph(n, e, regions, seed=seed, p_intra=0.8)
    (x, y), = R.data.synthetic_snapshots(n, f, t, o, 1, seed=seed)
    p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=seed + 1, hidden=hidden)
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    pred_o, hid_o = M.regional_temporal_gcn(po, x, g.edge_index, g.region_index, g.region_attr)
    torch.mean((pred_o - y) ** 2).backward()
    lib = R.load_library()
    prev = lib.regt_set_gemm_mode(mode)
    try:
        mod = R.RegionalTemporalGCN(f, n, t, o, num_regions=regions, hidden_channels=hidden)
        mod.load_state_dict(p, strict=True)
        mod = mod.cuda()
        pred, hid = mod(x.cuda(), g.edge_index.cuda(), [i.cuda() for i in g.region_index], [a.cuda() for a in g.region_attr])
        torch.mean((pred - y.cuda()) ** 2).backward()
    finally:
        lib.regt_set_gemm_mode(prev)
    assert float((pred.detach().cpu() - pred_o.detach()).abs().max()) < TOL
    assert float((hid.detach().cpu() - hid_o.detach()).abs().max()) < TOL
    for k, q in mod.named_parameters():
        if k in M.UNUSED_PARAMS:
            continue
        np.testing.assert_allclose(q.grad.cpu().numpy(), po[k].grad.numpy(), atol=TOL, rtol=1e-4, err_msg=k)
