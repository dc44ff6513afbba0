"""GPU parity of the whole hot path (module -> autograd Function -> C ABI -> HIP kernels) against the
CPU oracle and against the golden vectors produced by the reference's own model files.

Tolerance: north_star asks for 1e-5 (fp32) against the reference CPU path; used as written."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, check_grads_against_golden, load_npz, region_lists
from oracle import model as M

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def R():
    import regtgcn_amd
    regtgcn_amd.load_library()
    return regtgcn_amd


@pytest.fixture(params=[0, 1], ids=["fp32mfma", "bf16x3split"])
def arith(request, R):
    """GEMM arithmetic for the parity tests that take this fixture: the default fp32 MFMA and the opt-in exact 3-way bf16
    split (regt_set_gemm_mode(1)) are held to the SAME goldens, oracle and 1e-5 -- so the driver's GPU run itself shows that
    the split "passes the same parity suite" (DESIGN.md 5a)."""
    lib = R.load_library()
    prev = lib.regt_set_gemm_mode(request.param)
    yield request.param
    lib.regt_set_gemm_mode(prev)


def _cuda_list(ts):
    return [t.cuda() for t in ts]


_ORACLE_CACHE = {}


def _oracle_once(key, fn):
    """The CPU oracle is the slow half of these tests: tests parametrized over the GEMM arithmetic share one oracle run."""
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = fn()
    return _ORACLE_CACHE[key]


def _run_regt(R, params, x, y, fx, num_regions=5):
    n, f, t = x.shape
    mod = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=y.shape[1], num_regions=num_regions)
    mod.load_state_dict(params, strict=True)
    mod = mod.cuda()
    ri, rw = region_lists(fx)
    pred, hidden = mod(x.cuda(), fx["edge_index"].cuda(), *_cuda_list(ri), *_cuda_list(rw))
    loss = torch.mean((pred - y.cuda()) ** 2)
    loss.backward()
    grads = {k: (None if p.grad is None else p.grad.cpu()) for k, p in mod.named_parameters()}
    return pred.detach().cpu(), hidden.detach().cpu(), float(loss.detach()), grads


@pytest.mark.parametrize("tag", ["in6_out1", "in12_out1", "in12_out3", "in6_out3", "ckpt"])
def test_regt_matches_reference_goldens(R, arith, tpims, tag):
    g = load_npz(f"golden_regt_{tag}.npz")
    t_in, t_out, w0 = int(g["t_in"]), int(g["t_out"]), int(g["window"])
    n = tpims["node_data"].shape[0]
    if tag == "ckpt":
        p = torch.load(os.path.join(GOLDEN, "ref_ckpt_in6_out1_epoch50.pt"), map_location="cpu", weights_only=True)
    else:
        p = M.init_params("RegionalTemporalGCN", 8, t_in, t_out, num_nodes=n, seed=int(g["seed"]))
    x = tpims["node_data"][:, :, w0:w0 + t_in].contiguous()
    y = tpims["node_data"][:, -1, w0 + t_in:w0 + t_in + t_out].contiguous()
    pred, hidden, loss, grads = _run_regt(R, p, x, y, tpims)
    np.testing.assert_allclose(pred.numpy(), g["pred"], atol=TOL)
    np.testing.assert_allclose(hidden.numpy(), g["hidden"], atol=TOL)
    assert abs(loss - float(g["loss"][0])) < 1e-6
    check_grads_against_golden(g, grads, atol=TOL, rtol=1e-4)
    for name in M.UNUSED_PARAMS:
        assert grads[name] is None


@pytest.mark.parametrize("tag", ["in6_out1", "in12_out3"])
def test_temporal_gcn_matches_reference_goldens(R, arith, tpims, tag):
    g = load_npz(f"golden_tgcn_{tag}.npz")
    t_in, t_out, w0 = int(g["t_in"]), int(g["t_out"]), int(g["window"])
    p = M.init_params("TemporalGCN", 8, t_in, t_out, seed=int(g["seed"]))
    x = tpims["node_data"][:, :, w0:w0 + t_in].contiguous()
    y = tpims["node_data"][:, -1, w0 + t_in:w0 + t_in + t_out].contiguous()
    mod = R.TemporalGCN(node_features=8, periods=t_in, output_dim=t_out)
    mod.load_state_dict(p, strict=True)
    mod = mod.cuda()
    pred, hidden = mod(x=x.cuda(), edge_index=tpims["edge_index"].cuda(), edge_attr=tpims["edge_attr"].cuda())
    loss = torch.mean((pred - y.cuda()) ** 2)
    loss.backward()
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g["pred"], atol=TOL)
    np.testing.assert_allclose(hidden.detach().cpu().numpy(), g["hidden"], atol=TOL)
    assert abs(float(loss.detach()) - float(g["loss"][0])) < 1e-6
    grads = {k: (None if q.grad is None else q.grad.cpu()) for k, q in mod.named_parameters()}
    check_grads_against_golden(g, grads, atol=TOL, rtol=1e-4)
    for name in M.UNUSED_PARAMS_TEMPORAL:
        assert grads[name] is None


def _synthetic(n, e, regions, f, t, seed):
    """Small instance of the bench generator: contiguous region blocks, 95% intra-region edges."""
    g = torch.Generator().manual_seed(seed)
    per = n // regions
    src = torch.randint(0, n, (e,), generator=g)
    blk = torch.clamp(src // per, max=regions - 1)
    intra = torch.rand(e, generator=g) < 0.95
    lo = blk * per
    hi = torch.where(blk == regions - 1, torch.full_like(lo, n), lo + per)
    dst_in = lo + (torch.rand(e, generator=g) * (hi - lo)).long()
    dst = torch.where(intra, dst_in, torch.randint(0, n, (e,), generator=g))
    keep = src != dst
    src, dst, intra = src[keep], dst[keep], intra[keep]
    w = torch.rand(src.numel(), generator=g) * 2925 + 75
    ei = torch.stack([src, dst])
    dblk = torch.clamp(dst // per, max=regions - 1)
    sblk = torch.clamp(src // per, max=regions - 1)
    ri, rw = [], []
    for r in range(regions):
        m = (sblk == r) & (dblk == r)
        ri.append(ei[:, m].contiguous())
        rw.append(w[m].contiguous())
    x = torch.rand(n, f, t, generator=g)
    return ei, ri, rw, x


@pytest.mark.parametrize("n,e,regions,f,t,o", [(1500, 15000, 8, 32, 12, 1), (777, 4000, 3, 8, 6, 3), (300, 2500, 1, 4, 5, 2),
                                              (2000, 16000, 64, 8, 6, 1),      # 64 regions = the 8-GPU global region count
                                              (1200, 9000, 4, 64, 12, 1),      # feat_dim 64 (BASELINE configs[4])
                                              (2048, 20000, 64, 64, 12, 1),    # feat_dim 64 AND 64 regions: the configs[4] shapes at fp32 1e-5
                                              (600, 4000, 3, 7, 5, 2), (500, 3000, 2, 10, 4, 1),    # F not a multiple of 4: padded staging
                                              # period counts at the ends of what the candidate kernel's 64-row halves see: one row per
                                              # node, and nodes of 48 rows that straddle halves and tiles (at most two partial sums each)
                                              (9000, 45000, 3, 32, 1, 1), (400, 3000, 2, 32, 48, 1),
                                              # more than 64 periods (a node then spans three 64-row blocks; results stay within 1e-5,
                                              # only bit-reproducibility of `hidden` is given up there), with widths 150 x 8 = 1200 that
                                              # are no multiple of 32 either
                                              (150, 1000, 2, 8, 150, 1)])
def test_regt_matches_oracle_on_synthetic_regional_graph(R, arith, n, e, regions, f, t, o):
    ei, ri, rw, x = _synthetic(n, e, regions, f, t, seed=n)
    y = torch.rand(n, o, generator=torch.Generator().manual_seed(1))
    p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=3)

    def run_oracle():
        po_ = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        pr, hd = M.regional_temporal_gcn(po_, x, ei, ri, rw)
        ls = torch.mean((pr - y) ** 2)
        ls.backward()
        return po_, pr.detach(), hd.detach(), ls.detach()

    po, pred_o, hid_o, loss_o = _oracle_once(("synth", n, e, regions, f, t, o), run_oracle)
    mod = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions)
    mod.load_state_dict(p, strict=True)
    mod = mod.cuda()
    pred, hidden = mod(x.cuda(), ei.cuda(), _cuda_list(ri), _cuda_list(rw))
    loss = torch.mean((pred - y.cuda()) ** 2)
    loss.backward()
    assert float((pred.cpu() - pred_o).abs().max()) < TOL
    assert float((hidden.cpu() - hid_o).abs().max()) < TOL
    assert abs(float(loss.detach()) - float(loss_o.detach())) < 1e-6
    for k, q in mod.named_parameters():
        if k in M.UNUSED_PARAMS:
            assert q.grad is None
            continue
        np.testing.assert_allclose(q.grad.cpu().numpy(), po[k].grad.numpy(), atol=TOL, rtol=1e-4, err_msg=k)


@pytest.mark.parametrize("n,e,regions,f,t,o", [(900, 6000, 5, 8, 6, 1), (3000, 30000, 3, 32, 12, 2)])
def test_overlapping_random_decomposition_matches_oracle(R, arith, n, e, regions, f, t, o):
    """The reference's 'random' decomposition (load_dataset.py:324-329): the edges of the full graph are dealt
    to R regional graphs at random, so every node has edges in several of them (general, non-disjoint mode)."""
    g = torch.Generator().manual_seed(n)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, n, (e,), generator=g)
    keep = src != dst
    ei = torch.stack([src[keep], dst[keep]])
    w = torch.rand(ei.shape[1], generator=g) * 2925 + 75
    part = torch.randint(0, regions, (ei.shape[1],), generator=g)
    ri = [ei[:, part == r].contiguous() for r in range(regions)]
    rw = [w[part == r].contiguous() for r in range(regions)]
    x = torch.rand(n, f, t, generator=g)
    y = torch.rand(n, o, generator=g)
    p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=4)

    def run_oracle():
        po_ = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        pr, hd = M.regional_temporal_gcn(po_, x, ei, ri, rw)
        torch.mean((pr - y) ** 2).backward()
        return po_, pr.detach(), hd.detach()

    po, pred_o, hid_o = _oracle_once(("overlap", n, e, regions, f, t, o), run_oracle)
    mod = R.RegionalTemporalGCN(f, n, t, o, num_regions=regions)
    mod.load_state_dict(p)
    mod = mod.cuda()
    pred, hidden = mod(x.cuda(), ei.cuda(), _cuda_list(ri), _cuda_list(rw))
    torch.mean((pred - y.cuda()) ** 2).backward()
    assert float((pred.detach().cpu() - pred_o.detach()).abs().max()) < TOL
    assert float((hidden.detach().cpu() - hid_o.detach()).abs().max()) < TOL
    for k, q in mod.named_parameters():
        if k in M.UNUSED_PARAMS:
            continue
        np.testing.assert_allclose(q.grad.cpu().numpy(), po[k].grad.numpy(), atol=TOL, rtol=1e-4, err_msg=k)


def test_forward_is_deterministic_and_graph_is_cached(R, tpims):
    n = tpims["node_data"].shape[0]
    p = M.init_params("RegionalTemporalGCN", 8, 6, 1, num_nodes=n, seed=11)
    mod = R.RegionalTemporalGCN(8, n, 6, 1)
    mod.load_state_dict(p)
    mod = mod.cuda()
    ri, rw = region_lists(tpims)
    x = tpims["node_data"][:, :, :6].contiguous().cuda()
    ei = tpims["edge_index"].cuda()
    ric, rwc = _cuda_list(ri), _cuda_list(rw)
    a = mod(x, ei, *ric, *rwc)
    b = mod(x, ei.clone(), *ric, *rwc)         # fresh edge_index tensor, same content -> fingerprint hit
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert len(mod._graphs._by_hash) == 1
    g = mod.prepare_graph(ei, ric, rwc)
    c = mod.forward_prepared(x, g)
    assert torch.equal(a[0], c[0])


def test_regional_forward_accepts_the_reference_keyword_names(R, tpims):
    """models/RegionalTemporalGCN.py:25-26 names its twelve parameters; a keyword call (all ten region arguments, or a positional
    prefix + the rest by name, as Python binds them) gives the positional call's result, and the binding errors are Python's."""
    n = tpims["node_data"].shape[0]
    p = M.init_params("RegionalTemporalGCN", 8, 6, 1, num_nodes=n, seed=11)
    mod = R.RegionalTemporalGCN(8, n, 6, 1)
    mod.load_state_dict(p)
    mod = mod.cuda()
    ri, rw = region_lists(tpims)
    x = tpims["node_data"][:, :, :6].contiguous().cuda()
    ei = tpims["edge_index"].cuda()
    ric, rwc = _cuda_list(ri), _cuda_list(rw)
    want = mod(x, ei, *ric, *rwc)
    names = ("IA", "KS", "KY", "OH", "WI")
    kw = {f"{r}edge_index": t for r, t in zip(names, ric)}
    kw.update({f"{r}edge_attr": t for r, t in zip(names, rwc)})
    got = mod(x=x, edge_index=ei, **kw)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    got = mod(x, ei, ric[0], ric[1], **{k: v for k, v in kw.items() if k not in ("IAedge_index", "KSedge_index")})
    assert torch.equal(got[0], want[0])
    with pytest.raises(TypeError, match="unexpected keyword"):
        mod(x, ei, **kw, edge_attr=rwc[0])
    with pytest.raises(TypeError, match="missing 1 required"):
        mod(x, ei, **{k: v for k, v in kw.items() if k != "OHedge_attr"})
    with pytest.raises(TypeError, match="multiple values"):
        mod(x, ei, ric[0], **kw)


def test_hidden_gradient_path(R, tpims):
    """A loss on the hidden output (second return value) reaches the parameters too."""
    n = tpims["node_data"].shape[0]
    p = M.init_params("RegionalTemporalGCN", 8, 6, 1, num_nodes=n, seed=12)
    ri, rw = region_lists(tpims)
    x = tpims["node_data"][:, :, :6].contiguous()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    pred_o, hid_o = M.regional_temporal_gcn(po, x, tpims["edge_index"], ri, rw)
    (pred_o.sum() + (hid_o ** 2).mean()).backward()
    mod = R.RegionalTemporalGCN(8, n, 6, 1)
    mod.load_state_dict(p)
    mod = mod.cuda()
    pred, hid = mod(x.cuda(), tpims["edge_index"].cuda(), *_cuda_list(ri), *_cuda_list(rw))
    (pred.sum() + (hid ** 2).mean()).backward()
    for k, q in mod.named_parameters():
        if k in M.UNUSED_PARAMS:
            continue
        np.testing.assert_allclose(q.grad.cpu().numpy(), po[k].grad.numpy(), atol=2e-5, rtol=1e-4, err_msg=k)


def test_region_sharded_path_matches_single_gpu(R):
    """Three region shards executed one after the other on the one GPU (halo rows copied by hand instead of
    exchanged): predictions, hidden rows and the summed gradients equal the unsharded run.  The middle rank owns a region
    block with region_lo > 0 and region_hi < R (all three gradient-composition tasks of tgnn.linear.weight)."""
    import numpy as np
    world, n_per, regions_per, f, t, o = 3, 1000, 3, 8, 6, 2
    n = n_per * world
    g = R.data.synthetic_regional_graph(n, 12000 * world, regions_per * world, seed=5, p_intra=0.85)
    (x, y), = R.data.synthetic_snapshots(n, f, t, o, 1, seed=5)
    p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions_per * world, seed=6)

    def fresh():
        m = R.RegionalTemporalGCN(f, n, t, o, num_regions=regions_per * world)
        m.load_state_dict(p)
        return m.cuda()

    full = fresh()
    pred_f, hid_f = full(x.cuda(), g.edge_index.cuda(), [i.cuda() for i in g.region_index], [a.cuda() for a in g.region_attr])
    (((pred_f - y.cuda()) ** 2).sum() / (n * o)).backward()
    shard_model = fresh()
    bounds = np.arange(world + 1, dtype=np.int64) * n_per
    region_owner = [r // regions_per for r in range(regions_per * world)]
    xp_glob = R.ops.pack_x(x.cuda()).view(n, t * f)
    for rank in range(world):
        sh = R.dist.build_shard(g.edge_index, g.region_index, g.region_attr, n, bounds, region_owner, rank, world, "cuda")
        lo, hi = sh.topo.node_lo, sh.topo.node_hi
        xp = torch.zeros(sh.topo.x_rows, t * f, device="cuda")
        xp[:n_per] = xp_glob[lo:hi]
        xp[n_per:] = xp_glob[torch.from_numpy(sh.topo.halo_ids()).cuda()]
        pred, hid = shard_model.forward_packed(xp.view(sh.topo.x_rows, t, f), sh.graph)
        assert float((pred - pred_f[lo:hi]).abs().max()) < 1e-6
        assert float((hid - hid_f[lo:hi]).abs().max()) < 1e-6
        (((pred - y[lo:hi].cuda()) ** 2).sum() / (n * o)).backward()          # grads accumulate = all-reduce(sum)
    for (k, a), (_, b) in zip(full.named_parameters(), shard_model.named_parameters()):
        if a.grad is None:
            assert b.grad is None
            continue
        np.testing.assert_allclose(b.grad.cpu().numpy(), a.grad.cpu().numpy(), atol=2e-6, rtol=1e-4, err_msg=k)


def test_training_loop_matches_reference_trajectory(R, arith, tpims):
    """run.py semantics (accumulate over snapshots, one RMSprop step per epoch, (rmse, mse) test) driven through the
    HIP module reproduce the trajectory recorded from the reference's own module (golden_loop.npz)."""
    g = load_npz("golden_loop.npz")
    t_in, t_out = int(g["t_in"]), int(g["t_out"])
    n_train, n_test, epochs = int(g["n_train"]), int(g["n_test"]), int(g["epochs"])
    n = tpims["node_data"].shape[0]
    mod = R.RegionalTemporalGCN(8, n, t_in, t_out)
    mod.load_state_dict(M.init_params("RegionalTemporalGCN", 8, t_in, t_out, num_nodes=n, seed=int(g["seed"])))
    mod = mod.cuda()
    ri, rw = region_lists(tpims)
    graph = mod.prepare_graph(tpims["edge_index"].cuda(), _cuda_list(ri), _cuda_list(rw))
    xs, ys = R.data.snapshot_windows(tpims["node_data"][:, :, :t_in + t_out + n_train + n_test - 1], t_in, t_out)
    xs, ys = _cuda_list(xs), _cuda_list(ys)
    opt = torch.optim.RMSprop(mod.parameters(), lr=1e-3, weight_decay=1e-4)
    names = [str(s) for s in g["names"]]
    losses, metrics = [], []
    for ep in range(epochs):
        last, all_l = R.train.train_epoch(mod, xs[:n_train], ys[:n_train], graph, opt)
        losses += [float(v) for v in all_l]
        assert float(last) == losses[-1]
        metrics.append(R.train.evaluate(mod, xs[n_train:], ys[n_train:], graph))
        named = dict(mod.named_parameters())
        sums = [float(named[k].detach().double().sum()) for k in names]
        np.testing.assert_allclose(sums, g["param_sums"][ep], atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(losses, g["losses"], atol=1e-5)
    np.testing.assert_allclose(np.array(metrics), g["metrics"], atol=1e-5)


def test_predict_metrics_with_shipped_checkpoint(R, tpims):
    """predict.py's MAE / RMSE / MAPE on the test split with the reference's trained checkpoint: HIP module vs oracle."""
    from oracle import loop as oloop
    n = tpims["node_data"].shape[0]
    p = torch.load(os.path.join(GOLDEN, "ref_ckpt_in6_out1_epoch50.pt"), map_location="cpu", weights_only=True)
    xs, ys = R.data.snapshot_windows(tpims["node_data"][:, :, :30], 6, 1)
    (_, _), (vx, vy) = R.train.split(xs, ys, 0.2)
    ri, rw = region_lists(tpims)
    want = oloop.predict_metrics(p, lambda prm, x: M.regional_temporal_gcn(prm, x.contiguous(), tpims["edge_index"], ri, rw), vx, vy)
    mod = R.RegionalTemporalGCN(8, n, 6, 1)
    mod.load_state_dict(p)
    mod = mod.cuda()
    graph = mod.prepare_graph(tpims["edge_index"].cuda(), _cuda_list(ri), _cuda_list(rw))
    got = R.evaluate.predict_metrics(mod, _cuda_list(vx), _cuda_list(vy), graph)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-6)
    assert 0.05 < got[1] < 0.2          # RMSE in the band of the paper's 0.086


def test_cpu_tensors_are_refused(R):
    mod = R.RegionalTemporalGCN(8, 10, 6, 1)
    with pytest.raises(R.RegtError):
        mod(torch.zeros(10, 8, 6), torch.zeros(2, 0, dtype=torch.long))


@pytest.mark.parametrize("hidden,mode", [(64, 0), (132, 0), (132, 1), (320, 1)])
def test_other_hidden_widths_match_oracle(R, hidden, mode):
    """The reference fixes C = 256; the kernels do not (column tiles that are partial, K loops that end mid-slab)."""
    n, e, regions, f, t, o = 500, 4000, 3, 8, 5, 2
    ei, ri, rw, x = _synthetic(n, e, regions, f, t, seed=hidden)
    y = torch.rand(n, o, generator=torch.Generator().manual_seed(2))
    p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=4, hidden=hidden)
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    pred_o, hid_o = M.regional_temporal_gcn(po, x, ei, ri, rw)
    torch.mean((pred_o - y) ** 2).backward()
    lib = R.load_library()
    prev = lib.regt_set_gemm_mode(mode)
    try:
        mod = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions, hidden_channels=hidden)
        mod.load_state_dict(p, strict=True)
        mod = mod.cuda()
        pred, hidden_out = mod(x.cuda(), ei.cuda(), _cuda_list(ri), _cuda_list(rw))
        torch.mean((pred - y.cuda()) ** 2).backward()
    finally:
        lib.regt_set_gemm_mode(prev)
    assert float((pred.detach().cpu() - pred_o.detach()).abs().max()) < TOL
    assert float((hidden_out.detach().cpu() - hid_o.detach()).abs().max()) < TOL
    for k, q in mod.named_parameters():
        if k in M.UNUSED_PARAMS:
            continue
        np.testing.assert_allclose(q.grad.cpu().numpy(), po[k].grad.numpy(), atol=TOL, rtol=1e-4, err_msg=k)


@pytest.mark.parametrize("model_name", ["RegionalTemporalGCN", "TemporalGCN"])
def test_fused_train_step_equals_autograd_path(R, tpims, model_name):
    """functional.FusedTrainStep (forward + loss gradient + backward through the C ABI, no autograd) accumulates exactly
    the gradients of the module path and reports the same losses."""
    t_in, t_out, k = 6, 2, 3
    n = tpims["node_data"].shape[0]
    xs, ys = R.data.snapshot_windows(tpims["node_data"][:, :, :t_in + t_out + k - 1], t_in, t_out)
    xs, ys = _cuda_list(xs), _cuda_list(ys)
    ri, rw = region_lists(tpims)

    def build():
        if model_name == "RegionalTemporalGCN":
            m = R.RegionalTemporalGCN(8, n, t_in, t_out)
            m.load_state_dict(M.init_params(model_name, 8, t_in, t_out, num_nodes=n, seed=31))
            m = m.cuda()
            return m, m.prepare_graph(tpims["edge_index"].cuda(), _cuda_list(ri), _cuda_list(rw))
        m = R.TemporalGCN(8, t_in, t_out)
        m.load_state_dict(M.init_params(model_name, 8, t_in, t_out, seed=31))
        m = m.cuda()
        return m, m.prepare_graph(tpims["edge_index"].cuda(), tpims["edge_attr"].cuda(), n)

    ref, g_ref = build()
    losses_ref = []
    for x, y in zip(xs, ys):
        pred, _ = ref.forward_prepared(x, g_ref)
        loss = torch.mean((pred - y) ** 2)
        loss.backward()
        losses_ref.append(float(loss.detach()))
    fused, g_f = build()
    stepper = R.functional.FusedTrainStep(fused, g_f, 8, t_in)
    losses = [float(stepper(x, y)) for x, y in zip(xs, ys)]
    np.testing.assert_allclose(losses, losses_ref, rtol=1e-6)
    for (k_, a), (_, b) in zip(ref.named_parameters(), fused.named_parameters()):
        if a.grad is None:
            assert b.grad is None, k_
            continue
        assert torch.equal(a.grad, b.grad), k_                      # same kernels, same accumulation order
    opt = torch.optim.RMSprop(fused.parameters(), lr=1e-3, weight_decay=1e-4)
    opt.step()
    stepper.zero_grad()
    assert all(float(p.grad.abs().max()) == 0.0 for p in fused.parameters() if p.grad is not None)
    float(stepper(xs[0], ys[0]))                                    # parameters updated in place: the stepper keeps working


def test_hipgraph_replay_path_matches_goldens():
    """REGT_HIPGRAPH=1 (opt-in, read at first use -> needs its own process): the captured forward / backward launch
    sequences replay to the same numbers as eager launches, and the third call really is a replay."""
    import subprocess
    import sys
    script = r'''
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import regtgcn_amd as R
from regtgcn_amd import _lib
from oracle import model as M
d = np.load("tests/golden/tpims_fixture.npz"); fx = {k: torch.from_numpy(d[k]) for k in d.files if d[k].ndim > 0}
g = np.load("tests/golden/golden_regt_in6_out1.npz")
regs = ("IA", "KS", "KY", "OH", "WI")
n, t_in, t_out = fx["node_data"].shape[0], 6, 1
mod = R.RegionalTemporalGCN(8, n, t_in, t_out)
mod.load_state_dict(M.init_params("RegionalTemporalGCN", 8, t_in, t_out, num_nodes=n, seed=int(g["seed"])))
mod = mod.cuda()
graph = mod.prepare_graph(fx["edge_index"].cuda(), [fx[f"edge_{r}_index"].cuda() for r in regs], [fx[f"edge_{r}_attr"].cuda() for r in regs])
x = fx["node_data"][:, :, :t_in].contiguous().cuda(); y = fx["node_data"][:, -1, t_in:t_in + t_out].contiguous().cuda()
for it in range(3):
    mod.zero_grad()
    pred, hid = mod.forward_prepared(x, graph)
    torch.mean((pred - y) ** 2).backward()
    assert np.abs(pred.detach().cpu().numpy() - g["pred"]).max() < 1e-5, it
    assert np.abs(hid.detach().cpu().numpy() - g["hidden"]).max() < 1e-5, it
    gw = mod.linear1.weight.grad.cpu().numpy()
    assert np.abs(gw.sum(1) - g["grow__linear1__weight"]).max() < 1e-4, it
st = (ctypes.c_int64 * 6)(); _lib.load().regt_graph_stats(st)
assert st[2] >= 1 and st[5] >= 1, list(st)
print("OK", list(st))
'''
    env = dict(os.environ, REGT_HIPGRAPH="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", script], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


@pytest.mark.parametrize("extra", ["", "--model GraphSAGETemporalGCN", "--model GAT", "--model RandomTemporalGCN --decomp_type random",
                                   "--snap_batch 16", "--model RandomTemporalGCN --decomp_type random --snap_batch 5",
                                   "--model TemporalGCN --snap_batch 8", "--model GraphSAGETemporalGCN --snap_batch 4", "--model GAT --snap_batch 4"],
                         ids=["reference_line", "graphsage", "gat", "random_decomposition", "snap_batch16", "random_snap_batch5", "tgcn_snap_batch8",
                              "graphsage_snap_batch4", "gat_snap_batch4"])
def test_reference_launch_line_trains_on_the_fixture(R, tmp_path, capsys, extra):
    """scripts/RegionalTemporalGCN.sh:1's argument string (copied as a string; --epochs cut to 1) drives the run.py counterpart on
    the TPIMS fixture: epochs + 1 iterations (run.py:230), the run.py:236 line per epoch, a checkpoint with the reference's file
    name under pretrained/<tf>/<model>/ (run.py:138, 243) that strict-loads into a fresh module."""
    from regtgcn_amd import train
    line = "--num_timesteps_in 6 --num_timesteps_out 1 --tr 0.2 --model RegionalTemporalGCN --tf occrate --dataloading_type 2 --epochs 1 --decomp_type regional"
    argv = line.split() + extra.split() + ["--fixture", os.path.join(GOLDEN, "tpims_fixture.npz"), "--out_dir", str(tmp_path)]
    train.main(argv)
    a = train.build_parser().parse_args(argv)
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("Train Loss:")]
    assert len(lines) == 2 and all("Test RMSE:" in ln and "MAE:" in ln for ln in lines)
    ck = tmp_path / "occrate" / a.model / "model_in6_out1_epoch0.pt"
    assert ck.exists()
    sd = torch.load(str(ck), map_location="cpu", weights_only=True)
    n = np.load(os.path.join(GOLDEN, "tpims_fixture.npz"))["node_data"].shape[0]
    if a.model == "TemporalGCN":
        R.TemporalGCN(8, 6, 1).load_state_dict(sd, strict=True)
    else:
        cls = {"RegionalTemporalGCN": R.RegionalTemporalGCN, "RandomTemporalGCN": R.RegionalTemporalGCN, "GraphSAGETemporalGCN": R.GraphSAGETemporalGCN,
               "GAT": R.GATTemporal}[a.model]
        cls(8, n, 6, 1).load_state_dict(sd, strict=True)
    assert all(bool(torch.isfinite(v).all()) for v in sd.values())


def test_gradient_accumulation_inside_backward_equals_autograd_accumulation():
    """functional.set_grad_accumulation_in_backward: the model's backward adds into .grad itself (one multi-tensor add) instead of
    autograd's one add per parameter -- run.py:178-194 accumulates over the epoch's snapshots.  Same values bit for bit, for
    grads that exist (zero_grad(set_to_none=False)) and for grads that do not (first step after zero_grad())."""
    import regtgcn_amd as R
    n, e, regions, f, t, o = 900, 7000, 3, 8, 6, 2
    ei, ri, rw, _ = _synthetic(n, e, regions, f, t, seed=4)
    snaps = R.data.synthetic_snapshots(n, f, t, o, 3, seed=4)
    p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=5)
    out = {}
    for flag in (False, True):
        prev = R.functional.set_grad_accumulation_in_backward(flag)
        try:
            mod = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions)
            mod.load_state_dict(p, strict=True)
            mod = mod.cuda()
            graph = mod.prepare_graph(ei.cuda(), [i.cuda() for i in ri], [a.cuda() for a in rw])
            for x, y in snaps:                                    # three snapshots accumulate
                pred, _h = mod.forward_prepared(x.cuda(), graph)
                torch.mean((pred - y.cuda()) ** 2).backward()
            out[flag] = {k: q.grad.clone() for k, q in mod.named_parameters() if q.grad is not None}
        finally:
            R.functional.set_grad_accumulation_in_backward(prev)
    assert set(out[False]) == set(out[True]) and len(out[True]) > 15
    for k in out[False]:
        assert torch.equal(out[False][k], out[True][k]), k


@pytest.mark.parametrize("n,o,gc", [(104, 1, None), (100_000, 1, None), (4097, 3, 50_000)])
def test_mse_loss_function(n, o, gc):
    """functional.mse_loss (regt_mse_loss_grad: value + gradient in one kernel, fixed summation order) vs torch.mean((out - y)**2),
    run.py:180 -- with the global element count of a region shard."""
    import regtgcn_amd as R
    g = torch.Generator().manual_seed(n)
    pred = torch.randn(n, o, generator=g).cuda().requires_grad_(True)
    y = torch.rand(n, o, generator=g).cuda()
    cnt = n * o if gc is None else gc
    loss = R.functional.mse_loss(pred, y, gc)
    loss.backward()
    want = ((pred.detach().double() - y.double()) ** 2).sum() / cnt
    assert abs(float(loss) - float(want)) <= 2e-6 * float(want)
    np.testing.assert_allclose(pred.grad.cpu().numpy(), (2.0 * (pred.detach() - y) / cnt).cpu().numpy(), rtol=1e-6, atol=1e-12)
    again = R.functional.mse_loss(pred.detach(), y, gc)
    assert float(again) == float(loss)                           # fixed summation order: bit-reproducible


@pytest.mark.parametrize("mode,f", [(0, 32), (2, 64)])
def test_training_steps_are_bit_reproducible_with_the_side_stream(mode, f):
    """Twenty forward + backward passes on one snapshot: outputs and every gradient identical each time.  The weight compositions,
    the head's weight gradients and the attention-gradient tail run on the library's side stream (forked / joined by events inside
    regt_forward / regt_backward): a missing join would show up here as a run-to-run difference.  fp32 path and the fused bf16 path."""
    import regtgcn_amd as R
    lib = R.load_library()
    prev = lib.regt_set_gemm_mode(mode)
    try:
        n, e, regions, t, o = 6000, 50000, 8, 12, 1
        ei, ri, rw, x = _synthetic(n, e, regions, f, t, seed=9)
        y = torch.rand(n, o, generator=torch.Generator().manual_seed(2)).cuda()
        p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=6)
        mod = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions)
        mod.load_state_dict(p, strict=True)
        mod = mod.cuda()
        graph = mod.prepare_graph(ei.cuda(), [i.cuda() for i in ri], [a.cuda() for a in rw])
        xs = x.cuda()
        first = None
        for it in range(20):
            mod.zero_grad(set_to_none=True)
            pred, hidden = mod.forward_prepared(xs, graph)
            R.functional.mse_loss(pred, y).backward()
            snap = [pred.detach().clone(), hidden.detach().clone()] + [q.grad.clone() for q in mod.parameters() if q.grad is not None]
            if first is None:
                first = snap
            else:
                assert len(snap) == len(first)
                for k, (a_, b_) in enumerate(zip(snap, first)):
                    assert torch.equal(a_, b_), (it, k)
    finally:
        lib.regt_set_gemm_mode(prev)


def _one_step(R, mod, graph, xs, y):
    mod.zero_grad(set_to_none=True)
    pred, hidden = mod.forward_prepared(xs, graph)
    R.functional.mse_loss(pred, y).backward()
    return [pred.detach().clone(), hidden.detach().clone()] + [q.grad.clone() for q in mod.parameters() if q.grad is not None]


def test_arithmetic_is_per_call_two_models_of_one_process():
    """regt_dims.arith (ABI v6; `model.arithmetic` on the modules): a bf16 model and an fp32 model of the same process, their calls
    interleaved, give bit for bit what each gives alone under the process-wide switch -- and never touch that switch."""
    import regtgcn_amd as R
    lib = R.load_library()
    n, e, regions, f, t, o = 3000, 24000, 8, 64, 12, 1
    ei, ri, rw, x = _synthetic(n, e, regions, f, t, seed=4)
    y = torch.rand(n, o, generator=torch.Generator().manual_seed(3)).cuda()
    p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=5)

    def fresh(arith=None):
        m = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions)
        m.load_state_dict(p, strict=True)
        m.arithmetic = arith
        return m.cuda()

    ref = {}
    base = fresh()
    graph = base.prepare_graph(ei.cuda(), [i.cuda() for i in ri], [a.cuda() for a in rw])
    xs = x.cuda()
    for mode in (0, 2):                                # what the process-wide switch gives
        prev = lib.regt_set_gemm_mode(mode)
        try:
            ref[mode] = _one_step(R, base, graph, xs, y)
        finally:
            lib.regt_set_gemm_mode(prev)
    assert not torch.equal(ref[0][1], ref[2][1])       # the two arithmetics do differ
    before = lib.regt_set_gemm_mode(0)
    lib.regt_set_gemm_mode(before)
    m_bf, m_fp = fresh("bf16"), fresh("fp32")
    for _ in range(2):
        pb, hb = m_bf.forward_prepared(xs, graph)      # forward bf16, forward fp32, backward bf16, backward fp32
        pf, hf = m_fp.forward_prepared(xs, graph)
        m_bf.zero_grad(set_to_none=True); m_fp.zero_grad(set_to_none=True)
        R.functional.mse_loss(pb, y).backward()
        R.functional.mse_loss(pf, y).backward()
        got_b = [pb.detach(), hb.detach()] + [q.grad for q in m_bf.parameters() if q.grad is not None]
        got_f = [pf.detach(), hf.detach()] + [q.grad for q in m_fp.parameters() if q.grad is not None]
        assert all(torch.equal(a_, b_) for a_, b_ in zip(got_b, ref[2])) and len(got_b) == len(ref[2])
        assert all(torch.equal(a_, b_) for a_, b_ in zip(got_f, ref[0])) and len(got_f) == len(ref[0])
    now = lib.regt_set_gemm_mode(before)
    assert now == before                               # process default untouched
    with pytest.raises(ValueError):
        fresh("fp8").forward_prepared(xs, graph)


def test_two_launch_streams_have_side_streams_of_their_own():
    """Two models driven from two torch streams, steps interleaved on the host: each launch stream forks / joins a side stream of its
    own (api.hip: one per (device, stream)), so both reproduce their single-stream results bit for bit."""
    import regtgcn_amd as R
    R.load_library()
    n, e, regions, f, t, o = 4000, 30000, 8, 32, 12, 1
    ei, ri, rw, x = _synthetic(n, e, regions, f, t, seed=14)
    y = torch.rand(n, o, generator=torch.Generator().manual_seed(2)).cuda()
    mods, graphs = [], []
    for seed in (6, 7):
        p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=seed)
        m = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions)
        m.load_state_dict(p, strict=True)
        m = m.cuda()
        mods.append(m)
        graphs.append(m.prepare_graph(ei.cuda(), [i.cuda() for i in ri], [a.cuda() for a in rw]))
    xs = x.cuda()
    want = [_one_step(R, m, g, xs, y) for m, g in zip(mods, graphs)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(5):
        got = [None, None]
        preds = [None, None]
        for k in (0, 1):
            with torch.cuda.stream(streams[k]):
                mods[k].zero_grad(set_to_none=True)
                preds[k] = mods[k].forward_prepared(xs, graphs[k])
        for k in (1, 0):
            with torch.cuda.stream(streams[k]):
                R.functional.mse_loss(preds[k][0], y).backward()
        torch.cuda.synchronize()
        for k in (0, 1):
            got[k] = [preds[k][0].detach(), preds[k][1].detach()] + [q.grad for q in mods[k].parameters() if q.grad is not None]
            assert len(got[k]) == len(want[k]) and all(torch.equal(a_, b_) for a_, b_ in zip(got[k], want[k])), k


# ---- snapshot batching (train.train_epoch_batched / evaluate_batched / evaluate.predict_metrics_batched) ------------------------------
def _tpims_model(R, tpims, t_in, t_out, seed):
    n = tpims["node_data"].shape[0]
    mod = R.RegionalTemporalGCN(8, n, t_in, t_out)
    mod.load_state_dict(M.init_params("RegionalTemporalGCN", 8, t_in, t_out, num_nodes=n, seed=seed))
    mod = mod.cuda()
    ri, rw = region_lists(tpims)
    ei, ric, rwc = tpims["edge_index"].cuda(), _cuda_list(ri), _cuda_list(rw)
    graphs = R.train.BatchedGraphs(lambda b: mod.prepare_graph(ei, ric, rwc, copies=b))
    return mod, graphs


def test_batched_training_loop_matches_reference_trajectory(R, tpims):
    """The golden run.py trajectory (golden_loop.npz: 3 train snapshots accumulate, one RMSprop step per epoch, (rmse, mse) test) with
    ALL THREE train snapshots in one forward / backward on the block-diagonal graph of three copies (--snap_batch 3), and the test
    snapshots two per forward: same losses, metrics and parameter sums within the per-snapshot loop's tolerances."""
    g = load_npz("golden_loop.npz")
    t_in, t_out = int(g["t_in"]), int(g["t_out"])
    n_train, n_test, epochs = int(g["n_train"]), int(g["n_test"]), int(g["epochs"])
    assert n_train == 3
    mod, graphs = _tpims_model(R, tpims, t_in, t_out, int(g["seed"]))
    xs, ys = R.data.snapshot_windows(tpims["node_data"][:, :, :t_in + t_out + n_train + n_test - 1], t_in, t_out)
    xs, ys = _cuda_list(xs), _cuda_list(ys)
    train, test = R.train.WindowStore(xs[:n_train], ys[:n_train]), R.train.WindowStore(xs[n_train:], ys[n_train:])
    opt = torch.optim.RMSprop(mod.parameters(), lr=1e-3, weight_decay=1e-4)
    names = [str(s) for s in g["names"]]
    losses, metrics = [], []
    for ep in range(epochs):
        last, all_l = R.train.train_epoch_batched(mod, train, graphs, opt, 3)
        losses += [float(v) for v in all_l]
        assert float(last) == losses[-1] and len(all_l) == n_train
        metrics.append(R.train.evaluate_batched(mod, test, graphs, 2))
        named = dict(mod.named_parameters())
        sums = [float(named[k].detach().double().sum()) for k in names]
        np.testing.assert_allclose(sums, g["param_sums"][ep], atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(losses, g["losses"], atol=1e-5)
    np.testing.assert_allclose(np.array(metrics), g["metrics"], atol=1e-5)


@pytest.mark.parametrize("model_name", ["RegionalTemporalGCN", "TemporalGCN"])
def test_snapshot_batch_equals_per_snapshot_accumulation(R, tpims, model_name):
    """B = 4 (+ a short last batch of 3) against the per-snapshot loop on 7 snapshots: per-snapshot losses, accumulated gradients
    and the predict.py metrics.  Regional model and the weighted-GCN baseline."""
    t_in, t_out, count = 12, 1, 7
    n = tpims["node_data"].shape[0]
    xs, ys = R.data.snapshot_windows(tpims["node_data"][:, :, :t_in + t_out + count - 1], t_in, t_out)
    xs, ys = _cuda_list(xs), _cuda_list(ys)
    ei = tpims["edge_index"].cuda()
    if model_name == "RegionalTemporalGCN":
        mod, graphs = _tpims_model(R, tpims, t_in, t_out, 31)
    else:
        ea = tpims["edge_attr"].cuda()
        mod = R.TemporalGCN(8, t_in, t_out)
        mod.load_state_dict(M.init_params("TemporalGCN", 8, t_in, t_out, seed=31))
        mod = mod.cuda()
        graphs = R.train.BatchedGraphs(lambda b: mod.prepare_graph(ei, ea, n, copies=b))
    prev = R.functional.set_grad_accumulation_in_backward(True)
    try:
        want_l = []
        for x, y in zip(xs, ys):
            out, _ = mod.forward_prepared(x, graphs.get(1))
            loss = R.functional.mse_loss(out, y)
            loss.backward()
            want_l.append(float(loss))
        want_g = {k: q.grad.clone() for k, q in mod.named_parameters() if q.grad is not None}
        mod.zero_grad(set_to_none=True)
        store = R.train.WindowStore(xs, ys)
        got_l = []
        for i in range(0, count, 4):
            b = min(4, count - i)
            x, y = store.batch(i, b)
            out, _ = mod.forward_prepared(x, graphs.get(b))
            R.functional.mse_loss(out, y, n * t_out).backward()
            got_l += [float(v) for v in ((out.detach() - y) ** 2).view(b, -1).mean(dim=1)]
    finally:
        R.functional.set_grad_accumulation_in_backward(prev)
    np.testing.assert_allclose(got_l, want_l, rtol=2e-5, atol=1e-7)
    for k, w in want_g.items():
        got = dict(mod.named_parameters())[k].grad
        np.testing.assert_allclose(got.cpu().numpy(), w.cpu().numpy(), rtol=1e-4, atol=2e-6 * max(1.0, float(w.abs().max())), err_msg=k)
    a = R.evaluate.predict_metrics(mod, xs, ys, graphs.get(1))
    b_ = R.evaluate.predict_metrics_batched(mod, store, graphs, 4)
    np.testing.assert_allclose(b_, a, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(R.train.evaluate_batched(mod, store, graphs, 4), R.train.evaluate(mod, xs, ys, graphs.get(1)), rtol=2e-5)


def test_call_flags_are_per_call_switches():
    """regt_dims.flags (ABI v6): REGT_DIMS_NO_SIDE_STREAM keeps every kernel of a call on the launch stream -- identical results; under
    the bf16 arithmetic REGT_DIMS_NO_BF16_ROWS / _NO_FUSED_BWD select what regt_set_option("xbf" / "fused_bwd", 0) selects process-wide:
    bit-identical to that path, and (the fused kernels reproduce the three-launch arithmetic) to the default one except for the
    attention gradient's summation order."""
    import regtgcn_amd as R
    from regtgcn_amd import _lib
    lib = R.load_library()
    n, e, regions, f, t, o = 3000, 24000, 8, 64, 12, 1
    ei, ri, rw, x = _synthetic(n, e, regions, f, t, seed=4)
    y = torch.rand(n, o, generator=torch.Generator().manual_seed(3)).cuda()
    p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=5)

    def run(arith, flags, option=None):
        m = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions)
        m.load_state_dict(p, strict=True)
        m.arithmetic, m.call_flags = arith, flags
        m = m.cuda()
        graph = m.prepare_graph(ei.cuda(), [i.cuda() for i in ri], [a.cuda() for a in rw])
        prev = lib.regt_set_option(option[0], option[1]) if option else None
        try:
            return _one_step(R, m, graph, x.cuda(), y)
        finally:
            if option:
                lib.regt_set_option(option[0], prev)

    base = run("fp32", 0)
    assert all(torch.equal(a_, b_) for a_, b_ in zip(run("fp32", _lib.DIMS_NO_SIDE_STREAM), base))
    for flag, opt in ((_lib.DIMS_NO_FUSED_BWD, b"fused_bwd"), (_lib.DIMS_NO_BF16_ROWS, b"xbf")):
        per_call, process_wide = run("bf16", flag), run("bf16", 0, (opt, 0))
        assert len(per_call) == len(process_wide) and all(torch.equal(a_, b_) for a_, b_ in zip(per_call, process_wide)), opt
    assert lib.regt_set_option(b"xbf", 1) == 1 and lib.regt_set_option(b"fused_bwd", 1) == 1      # process defaults untouched

    # functional.FusedTrainStep (train.py --fused_step) passes the model's switches on as well: with NO_BF16_ROWS the accumulated
    # gradients are those of the autograd path under the same flag, bit for bit
    def fused_step(flags):
        m = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions)
        m.load_state_dict(p, strict=True)
        m.arithmetic, m.call_flags = "bf16", flags
        m = m.cuda()
        graph = m.prepare_graph(ei.cuda(), [i.cuda() for i in ri], [a.cuda() for a in rw])
        st = R.functional.FusedTrainStep(m, graph, f, t)
        assert st.dims.flags == flags
        st(x.cuda(), y)
        torch.cuda.synchronize()
        return [q.grad.detach().clone() for q in st.params]

    g_flag, g_plain = fused_step(_lib.DIMS_NO_BF16_ROWS), fused_step(0)
    assert all(bool(torch.isfinite(a_).all()) for a_ in g_flag)
    assert not all(torch.equal(a_, b_) for a_, b_ in zip(g_flag, g_plain))      # (x is rounded at another point: another path did run)


def test_big_tile_shapes_take_the_round4_kernels():
    """Guard against a silent fallback: at a shape with >= 128 GEMM tiles the fp32 backward runs the generated-operand candidate data
    gradient (no cell_bwd stage), TemporalGCN runs the collapsed gates (a wgrad_P01 stage, no dgrad_gates / wgrad_Uzr), and the
    small TPIMS shape keeps the small-tile kernels with cell_bwd (library per-stage profile, regt_profile_*)."""
    import ctypes
    import regtgcn_amd as R
    from regtgcn_amd import _lib
    lib = R.load_library()

    def stages(mod, graph, x, y):
        _one_step(R, mod, graph, x, y)                       # warm-up
        lib.regt_profile_enable(1)
        _one_step(R, mod, graph, x, y)
        torch.cuda.synchronize()
        lib.regt_profile_enable(0)
        buf = (ctypes.c_char * 16384)()
        _lib.check(lib.regt_profile_collect(buf, 16384), "regt_profile_collect")
        return {ln.split()[0] for ln in buf.value.decode().splitlines()}

    n, e, regions, f, t, o = 3000, 24000, 8, 32, 12, 1
    ei, ri, rw, x = _synthetic(n, e, regions, f, t, seed=4)
    y = torch.rand(n, o, generator=torch.Generator().manual_seed(3)).cuda()
    m = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions).cuda()
    g = m.prepare_graph(ei.cuda(), [i.cuda() for i in ri], [a.cuda() for a in rw])
    s = stages(m, g, x.cuda(), y)
    assert "dgrad_candidate" in s and "cell_bwd" not in s and "dgrad_gates" in s and "wgrad_Uzr" in s, s
    tg = R.TemporalGCN(node_features=f, periods=t, output_dim=o).cuda()
    gt = tg.prepare_graph(ei.cuda(), torch.rand(ei.shape[1]).cuda() + 1.0, n)
    s = stages(tg, gt, x.cuda(), y)
    assert "wgrad_P01" in s and "dgrad_gates" not in s and "wgrad_Uzr" not in s and "cell_bwd" not in s, s
    ns = 104                                                 # TPIMS size: 10 row tiles -> small-tile kernels, two-launch cell backward
    eis, ris, rws, xs = _synthetic(ns, 400, 5, 8, 12, seed=5)
    ys = torch.rand(ns, 1, generator=torch.Generator().manual_seed(3)).cuda()
    ms = R.RegionalTemporalGCN(node_features=8, num_nodes=ns, periods=12, output_dim=1, num_regions=5).cuda()
    gs = ms.prepare_graph(eis.cuda(), [i.cuda() for i in ris], [a.cuda() for a in rws])
    assert "cell_bwd" in stages(ms, gs, xs.cuda(), ys)


@pytest.mark.parametrize("arith_mode", [0, 2])
def test_weight_gradient_slabs_fit_at_hidden_128_and_many_rows(arith_mode):
    """The backward picks its row-chunk counts per launch (768 chunks of dUh at C = 128 in fp32); the slab regions of the workspace
    are sized for them (wgrad_chunk_bound): ~400 k rows at hidden 128, F = 8 used to fail with 'weight-gradient slab ... exceeds the
    workspace region'.  Also two runs agree bit for bit (deterministic partial-slab reduction at these chunk counts)."""
    import regtgcn_amd as R
    lib = R.load_library()
    n, e, regions, f, t, o, hidden = 34000, 200000, 4, 8, 12, 1, 128
    ei, ri, rw, x = _synthetic(n, e, regions, f, t, seed=11)
    y = torch.rand(n, o, generator=torch.Generator().manual_seed(5)).cuda()
    prev = lib.regt_set_gemm_mode(arith_mode)
    try:
        torch.manual_seed(0)
        mod = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions, hidden_channels=hidden).cuda()
        g = mod.prepare_graph(ei.cuda(), [i.cuda() for i in ri], [a.cuda() for a in rw])
        runs = []
        for _ in range(2):
            mod.zero_grad(set_to_none=True)
            pred, _h = mod.forward_prepared(x.cuda(), g)
            torch.mean((pred - y) ** 2).backward()
            runs.append({k: q.grad.detach().clone() for k, q in mod.named_parameters() if q.grad is not None})
        torch.cuda.synchronize()
    finally:
        lib.regt_set_gemm_mode(prev)
    for k in runs[0]:
        assert bool(torch.isfinite(runs[0][k]).all()), k
        assert torch.equal(runs[0][k], runs[1][k]), k
