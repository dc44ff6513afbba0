"""The regional embedding of the fp32 path as a kernel of its own (csrc/embed.hip, C = 256, F = 32) against the general GEMM core.

Same op site (RegionalTemporalGCN.py:136-148 in the composed-weight form), same operands, another summation order over k (the
16x16x4 fp32 MFMA instead of the 32x32x2 one): outputs and every gradient agree to fp32 rounding, and both agree with the
oracle inside tests/test_gpu_model.py's bars.  Shapes: several regions per graph (tiles that straddle a region boundary run one
pass per region), a row count that is no multiple of the 128-row tile, one region (no region lookup at all)."""
import pytest
import torch

from oracle import model as M
from test_gpu_model import _synthetic

pytestmark = pytest.mark.gpu


def _run(R, n, e, regions, t, o, embed):
    lib = R.load_library()
    lib.regt_set_option(b"embed_kernel", embed)
    f = 32
    ei, ri, rw, x = _synthetic(n, e, regions, f, t, seed=n)
    y = torch.rand(n, o, generator=torch.Generator().manual_seed(1))
    p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=3, hidden=256)
    mod = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions, hidden_channels=256)
    mod.load_state_dict(p, strict=True)
    mod = mod.cuda()
    pred, hid = mod(x.cuda(), ei.cuda(), [i.cuda() for i in ri], [a.cuda() for a in rw])
    (torch.mean((pred - y.cuda()) ** 2) + 1e-3 * hid.sum()).backward()
    return pred.detach(), hid.detach(), {k: q.grad.detach().clone() for k, q in mod.named_parameters() if q.grad is not None}


# (nodes, edges, regions, T, O): M = nodes * T >= 65 536 rows selects the kernel; 8003 * 12 is no multiple of 128
@pytest.mark.parametrize("n,e,regions,t,o", [(8003, 60000, 4, 12, 1), (6000, 40000, 1, 12, 2), (12000, 90000, 8, 6, 1)])
def test_embedding_kernel_equals_the_general_core(n, e, regions, t, o):
    import regtgcn_amd as R
    lib = R.load_library()
    prev = lib.regt_set_gemm_mode(0)
    try:
        p1, h1, g1 = _run(R, n, e, regions, t, o, 1)
        p0, h0, g0 = _run(R, n, e, regions, t, o, 0)
    finally:
        lib.regt_set_option(b"embed_kernel", 1)
        lib.regt_set_gemm_mode(prev)
    assert float(h1.abs().max()) > 0
    assert float((p1 - p0).abs().max()) <= 2e-6 * float(p0.abs().max()) + 1e-9
    assert float((h1 - h0).abs().max()) <= 2e-6 * float(h0.abs().max()) + 1e-9
    assert set(g1) == set(g0)
    for k in g0:
        # (the attention gradient is a sum of M x C products with heavy cancellation: 1e-4 of its scale, as in test_gpu_fused.py)
        rel = 3e-4 if k == "tgnn._attention" else 2e-5
        assert float((g1[k] - g0[k]).abs().max()) <= rel * float(g0[k].abs().max()) + 1e-10, k
