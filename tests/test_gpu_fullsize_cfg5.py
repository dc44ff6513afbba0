"""Parity at the full size of ONE rank's shard of BASELINE.json configs[4]: the global graph (1 000 000 nodes / 10 000 000
edges / 64 regions / F = 64) split over 8 ranks by ``dist.build_shard``; rank 0 owns 125 000 nodes (regions 0..7) and reads
~53 000 halo rows.  Op sites: models/RegionalTemporalGCN.py:131-149, models/utils.py:163-203.

* the aggregation of the shard operator (A_hat rows of the owned nodes with GLOBAL degrees, merged with the owned regional
  Laplacians) against scipy.sparse in float64 built from the global edge list -- independent of build_shard;
* fp32 forward + backward of the shard (packed own rows + halo rows from the global x) against the CPU ORACLE at T = 2.  The
  oracle cannot hold the global problem (its (N, R*C) concat is 65 GB per period at 64 regions), so it runs on a reduced problem
  that is exactly equivalent for the owned rows: nodes = owned + halo + one dummy source whose duplicate edges give every halo
  node its global in-degree (GCNConv normalises a source by its own degree), regions = the 8 owned ones + one empty region whose
  linear block is the SUM of the 56 foreign blocks (a node outside region r contributes (x W0 + b) Wlin_r there), loss over the
  owned rows only;
* T = 12 through the identical-periods property (hidden / prediction equal the T = 1 run, attention gradient 0) in fp32 and in
  the bf16 arithmetic, the latter also against the fp32 result within the derived 8 u bound of tests/test_gpu_bf16.py."""
import numpy as np
import pytest
import torch

from oracle import graph_ops as G
from oracle import model as M

pytestmark = pytest.mark.gpu
GN, GE, GR, F, O, WORLD = 1_000_000, 10_000_000, 64, 64, 1, 8
TOL = 1e-5
BF16_U = 2.0 ** -9


@pytest.fixture(scope="module")
def R():
    import regtgcn_amd as R
    R.load_library()
    return R


@pytest.fixture(scope="module")
def world(R):
    g = R.data.synthetic_regional_graph(GN, GE, GR, seed=42)
    rpg = GR // WORLD
    bounds = np.asarray(g.region_bounds[::rpg], dtype=np.int64)
    owner = [r // rpg for r in range(GR)]
    shard = R.dist.build_shard(g.edge_index, g.region_index, g.region_attr, GN, bounds, owner, 0, WORLD, torch.device("cuda"))
    return g, shard, int(bounds[1] - bounds[0]), rpg


def test_shard_aggregation_fullsize_against_scipy(R, world):
    import scipy.sparse as sp
    g, shard, n, rpg = world
    w = 64
    xg = torch.rand(GN, w, generator=torch.Generator().manual_seed(3))
    halo = torch.from_numpy(shard.topo.halo_ids())
    ext = torch.cat([xg[:n], xg[halo]]).cuda()
    pg = shard.graph
    ya, yl = R.ops.spmm_dual(pg.m_rowptr, pg.m_col, pg.m_val_a, pg.m_val_l, ext)
    assert ya.shape[0] == n
    s, d, wn = G.gcn_norm_edges(g.edge_index, None, GN, torch.float64)
    keep = (d < n).numpy()
    a_own = sp.coo_matrix((wn.numpy()[keep], (d.numpy()[keep], s.numpy()[keep])), shape=(n, GN)).tocsr()
    rows, cols, vals = [], [], []
    for ei, ew in zip(g.region_index[:rpg], g.region_attr[:rpg]):
        s, d, wl = G.cheb_norm_edges(ei, ew, n, torch.float64)
        rows.append(d.numpy()); cols.append(s.numpy()); vals.append(wl.numpy())
    lap = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
    xd = xg.double().numpy()
    assert np.abs(ya.cpu().double().numpy() - a_own @ xd).max() < TOL
    assert np.abs(yl.cpu().double().numpy() - lap @ xd[:n]).max() < TOL
    # the same rows as bf16 (the layout of the bf16 arithmetic): fp32 sums of the rounded inputs, one rounding at the end
    xb = ext.to(torch.bfloat16)
    ba, bl = R.ops.spmm_dual_bf16(pg.m_rowptr, pg.m_col, pg.m_val_a, pg.m_val_l, xb)
    halo_np = halo.numpy()
    xr = np.concatenate([xb[:n].float().cpu().double().numpy(), xb[n:].float().cpu().double().numpy()])
    idx = np.concatenate([np.arange(n), halo_np])
    xfull = np.zeros((GN, w))
    xfull[idx] = xr
    ref = a_own @ xfull
    assert np.abs(ba.float().cpu().double().numpy() - ref).max() <= 2.0 ** -8 * np.abs(ref).max() + 1e-6


def _reduced_problem(g, shard, n, rpg):
    """(edge_index', regional lists') of the oracle's equivalent problem for the owned rows (module docstring)."""
    halo = shard.topo.halo_ids()
    src, dst = g.edge_index[0].numpy(), g.edge_index[1].numpy()
    into_own = dst < n
    s_own = src[into_own]
    pos = np.searchsorted(halo, s_own)
    s_ext = np.where(s_own < n, s_own, n + np.minimum(pos, max(halo.size - 1, 0)))
    assert bool(((s_own < n) | (halo[np.minimum(pos, halo.size - 1)] == s_own)).all())
    indeg = np.bincount(dst, minlength=GN)[halo]
    dummy = n + halo.size
    d_src = np.full(int(indeg.sum()), dummy, dtype=np.int64)
    d_dst = np.repeat(n + np.arange(halo.size, dtype=np.int64), indeg)
    ei = torch.from_numpy(np.stack([np.concatenate([s_ext, d_src]), np.concatenate([dst[into_own], d_dst])]))
    ri = [t.clone() for t in g.region_index[:rpg]] + [torch.zeros(2, 0, dtype=torch.int64)]
    rw = [t.clone() for t in g.region_attr[:rpg]] + [torch.zeros(0)]
    return ei, ri, rw, halo, dummy + 1


def test_shard_forward_backward_fullsize_two_periods_against_oracle(R, world):
    g, shard, n, rpg = world
    t, C = 2, 256
    ei, ri, rw, halo, n_red = _reduced_problem(g, shard, n, rpg)
    gen = torch.Generator().manual_seed(7)
    x_red = torch.rand(n_red, F, t, generator=gen)
    x_red[-1] = 0
    y = torch.rand(n, O, generator=gen)
    p9 = M.init_params("RegionalTemporalGCN", F, t, O, num_nodes=n_red, num_regions=rpg + 1, seed=8)
    # the 64-region model of the shard: owned blocks as in the oracle's model, 56 foreign blocks that SUM to its ninth block
    p64 = {k: v.clone() for k, v in p9.items()}
    wl9 = p9["tgnn.linear.weight"]
    foreign = torch.randn(C, (GR - rpg - 1) * C, generator=gen) * 0.02
    last = wl9[:, rpg * C:] - foreign.view(C, GR - rpg - 1, C).sum(dim=1)
    p64["tgnn.linear.weight"] = torch.cat([wl9[:, :rpg * C], foreign, last], dim=1).contiguous()
    p64["tgnn._weight_att2"] = torch.zeros(n, 1)
    torch.set_num_threads(16)
    po = {k: v.clone().requires_grad_(True) for k, v in p9.items()}
    pred_o, hid_o = M.regional_temporal_gcn(po, x_red, ei, ri, rw)
    (((pred_o[:n] - y) ** 2).sum() / (n * O)).backward()
    mod = R.RegionalTemporalGCN(F, n, t, O, num_regions=GR)
    mod.load_state_dict(p64)
    mod = mod.cuda()
    ext = torch.empty(shard.topo.x_rows, t, F, device="cuda")
    ext[:n] = x_red[:n].permute(0, 2, 1).cuda()
    ext[n:] = x_red[n:n + halo.size].permute(0, 2, 1).cuda()
    pred, hid = mod.forward_packed(ext, shard.graph)
    (((pred - y.cuda()) ** 2).sum() / (n * O)).backward()
    assert float((pred.detach().cpu() - pred_o[:n].detach()).abs().max()) < TOL
    assert float((hid.detach().cpu() - hid_o[:n].detach()).abs().max()) < TOL
    for k, q in mod.named_parameters():
        if k in M.UNUSED_PARAMS:
            continue
        want, got = po[k].grad, q.grad.cpu()
        if k == "tgnn.linear.weight":
            g9 = want
            np.testing.assert_allclose(got[:, :rpg * C].numpy(), g9[:, :rpg * C].numpy(), rtol=2e-4, atol=1e-5 * float(g9.abs().max()), err_msg=k)
            shared = g9[:, rpg * C:]                      # every foreign block receives the term all regions share
            for r in (rpg, GR // 2, GR - 1):
                np.testing.assert_allclose(got[:, r * C:(r + 1) * C].numpy(), shared.numpy(), rtol=2e-4, atol=1e-5 * float(g9.abs().max()), err_msg=f"{k}[{r}]")
            continue
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-4, atol=1e-5 * float(want.abs().max()), err_msg=k)


@pytest.mark.parametrize("mode", [0, 2], ids=["fp32", "bf16"])
def test_shard_full_configuration_identical_periods_property(R, world, mode):
    g, shard, n, rpg = world
    lib = R.load_library()
    t = 12
    gen = torch.Generator().manual_seed(9)
    x1 = torch.rand(shard.topo.x_rows, 1, F, generator=gen)            # packed rows (own + halo), one period
    y = torch.rand(n, O, generator=gen).cuda()
    p12 = M.init_params("RegionalTemporalGCN", F, t, O, num_nodes=n, num_regions=GR, seed=10)
    p1 = {k: v.clone() for k, v in p12.items()}
    p1["tgnn._attention"] = p12["tgnn._attention"][:1].clone()

    def run(params, ext, periods):
        mod = R.RegionalTemporalGCN(F, n, periods, O, num_regions=GR)
        mod.load_state_dict(params)
        mod = mod.cuda()
        pred, hid = mod.forward_packed(ext.cuda(), shard.graph)
        (((pred - y) ** 2).sum() / (n * O)).backward()
        return pred.detach(), hid.detach(), {k: q.grad for k, q in mod.named_parameters() if q.grad is not None}

    x12 = x1.expand(shard.topo.x_rows, t, F).contiguous()
    prev = lib.regt_set_gemm_mode(mode)
    try:
        pred12, hid12, g12 = run(p12, x12, t)
        pred12b, hid12b, g12b = run(p12, x12, t)
        pred1, hid1, g1 = run(p1, x1, 1)
        if mode == 2:
            lib.regt_set_gemm_mode(0)
            pred_f, hid_f, g_f = run(p12, x12, t)
    finally:
        lib.regt_set_gemm_mode(prev)
    assert torch.equal(pred12, pred12b) and torch.equal(hid12, hid12b)                  # bit-reproducible
    assert all(torch.equal(g12[k], g12b[k]) for k in g12)
    # identical periods: the attention-weighted sum of identical cell outputs is that output.  fp32: to rounding; bf16: the
    # T = 1 and T = 12 runs round the same operands, what differs is the fp32 order of the sum over periods
    tol = TOL if mode == 0 else 2e-5
    assert float((hid12 - hid1).abs().max()) < tol
    # (bf16: the head rounds `hidden` to bf16 again -- a last-ulp difference of hidden that crosses a rounding boundary moves one
    # operand by 2^-8 relative: bounded by 2 u of the prediction's scale, seen on a few of the 125 000 rows)
    assert float((pred12 - pred1).abs().max()) < (tol if mode == 0 else 2 * BF16_U * float(pred1.abs().max()))
    assert float(g12["tgnn._attention"].abs().max()) < (1e-6 if mode == 0 else 1e-5)
    for k in g1:
        if k == "tgnn._attention":
            continue
        scale = max(float(g1[k].abs().max()), 1e-8)
        assert float((g12[k] - g1[k]).abs().max()) < (2e-4 if mode == 0 else 2e-2) * scale + 1e-9, k
    if mode == 2:       # against the fp32 arithmetic on the same shard: the derived bound of tests/test_gpu_bf16.py (8 u)
        for got, want in ((pred12, pred_f), (hid12, hid_f)):
            assert float((got - want).abs().max()) <= 8 * BF16_U * float(want.abs().max())
        for k in g_f:
            if k == "tgnn._attention":
                continue
            assert float((g12[k] - g_f[k]).norm()) <= 8 * BF16_U * float(g_f[k].norm()) + 1e-9, k


# ---- BASELINE configs[4] at full size on ONE GPU: the whole 1 M-node graph against its eight shards ---------------------------------
# The N = 1 anchor of the scaling curve (bench.py --workload cfg5full) and all eight ranks' shards (different halo counts, the last
# region block), executed one after the other as tests/test_gpu_fullsize.py does for configs[3]: every owned prediction / hidden row
# of a shard must equal the whole-graph run's, the gradients summed over the ranks (= the all-reduce) its gradients -- within the 8 u
# bar of the bf16 arithmetic (tile boundaries differ between the two runs, so operands can round the other way).
def test_configs4_whole_graph_on_one_gpu_equals_its_eight_shards_bf16(R):
    lib = R.load_library()
    t = 12
    g = R.data.synthetic_regional_graph(GN, GE, GR, seed=42)
    (x, y), = R.data.synthetic_snapshots(GN, F, t, O, 1, seed=31)
    p = M.init_params("RegionalTemporalGCN", F, t, O, num_nodes=GN, num_regions=GR, seed=32)
    prev = lib.regt_set_gemm_mode(2)
    try:
        def fresh():
            m = R.RegionalTemporalGCN(F, GN, t, O, num_regions=GR)
            m.load_state_dict(p)
            return m.cuda()

        full = fresh()
        pg = full.prepare_graph(g.edge_index.cuda(), [i.cuda() for i in g.region_index], [a.cuda() for a in g.region_attr])
        assert pg.region_sorted
        torch.cuda.reset_peak_memory_stats()
        pred_f, hid_f = full.forward_prepared(x.cuda(), pg)
        (((pred_f - y.cuda()) ** 2).sum() / (GN * O)).backward()
        torch.cuda.synchronize()
        peak_gb = torch.cuda.max_memory_allocated() / 1e9
        assert peak_gb < 250, peak_gb                                     # the whole configs[4] step fits one 288 GB part
        pred_f, hid_f = pred_f.detach(), hid_f.detach()
        assert bool(torch.isfinite(pred_f).all()) and bool(torch.isfinite(hid_f).all())
        grads_f = {k: q.grad.clone() for k, q in full.named_parameters() if q.grad is not None}
        del full, pg
        torch.cuda.empty_cache()

        sharded = fresh()
        rpg = GR // WORLD
        bounds = np.asarray(g.region_bounds[::rpg], dtype=np.int64)
        owner = [r // rpg for r in range(GR)]
        xp_glob = R.ops.pack_x(x.cuda()).view(GN, t * F)
        hscale, pscale = float(hid_f.abs().max()), float(pred_f.abs().max())
        worst_pred = worst_hid = 0.0
        halo_rows = []
        for rank in range(WORLD):
            sh = R.dist.build_shard(g.edge_index, g.region_index, g.region_attr, GN, bounds, owner, rank, WORLD, "cuda")
            lo, hi = sh.topo.node_lo, sh.topo.node_hi
            assert sh.graph.region_lo == rank * rpg and sh.graph.region_hi == (rank + 1) * rpg
            halo_rows.append(sh.topo.halo_rows)
            xp = torch.empty(sh.topo.x_rows, t * F, device="cuda")
            xp[:hi - lo] = xp_glob[lo:hi]
            xp[hi - lo:] = xp_glob[torch.from_numpy(sh.topo.halo_ids()).cuda()]
            pred, hid = sharded.forward_packed(xp.view(sh.topo.x_rows, t, F), sh.graph)
            worst_pred = max(worst_pred, float((pred.detach() - pred_f[lo:hi]).abs().max()))
            worst_hid = max(worst_hid, float((hid.detach() - hid_f[lo:hi]).abs().max()))
            (((pred - y[lo:hi].cuda()) ** 2).sum() / (GN * O)).backward()              # .grad accumulates = all-reduce(sum)
            del sh, xp, pred, hid
        assert all(30_000 < h < 80_000 for h in halo_rows), halo_rows      # ~53 000 halo rows per 125 000-node shard
        assert worst_hid < 8 * BF16_U * hscale and worst_pred < 8 * BF16_U * pscale, (worst_pred, worst_hid, hscale, pscale)
        grads_s = {k: q.grad for k, q in sharded.named_parameters() if q.grad is not None}
        assert set(grads_s) == set(grads_f)
        for k in grads_f:
            a, b = grads_s[k].double().cpu(), grads_f[k].double().cpu()
            assert float((a - b).norm()) <= 8 * BF16_U * float(b.norm()) + 1e-9, k
    finally:
        lib.regt_set_gemm_mode(prev)
