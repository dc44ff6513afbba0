"""GPU parity of the op-site kernels (through the C ABI) against the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import region_lists
from oracle import graph_ops as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import regtgcn_amd
    regtgcn_amd.load_library()
    return regtgcn_amd


def _csr_to_dense(rowptr, col, val, n):
    rowptr, col, val = rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy().astype(np.float64)
    d = np.zeros((len(rowptr) - 1, n), dtype=np.float64)
    for i in range(len(rowptr) - 1):
        for p in range(rowptr[i], rowptr[i + 1]):
            d[i, col[p]] += val[p]
    return d


def _rand_graph(n, e, seed, loops=True):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, n, (e,), generator=g)
    if loops and e > 4:
        src[0] = dst[0] = 1
        src[1] = dst[1] = 1           # two self loops on node 1: the last listed weight wins
        src[2], dst[2] = src[3], dst[3]   # duplicate edge
    w = torch.rand(e, generator=g) * 100 + 1
    return torch.stack([src, dst]), w


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("n,e,seed", [(9, 25, 0), (104, 400, 1), (1000, 12000, 2), (5, 0, 3)])
def test_gcn_csr_matches_oracle(R, n, e, seed, weighted):
    ei, w = _rand_graph(n, e, seed)
    ew = w if weighted else None
    rp, col, val = R.graph.gcn_csr(ei.cuda(), None if ew is None else ew.cuda(), n)
    got = _csr_to_dense(rp, col, val, n)
    want = G.dense_gcn_operator(ei, ew, n, torch.float64).numpy()
    np.testing.assert_allclose(got, want, atol=2e-7, rtol=1e-6)
    # bit-exact against the fp32 edge-ordered restatement (same op order)
    s, d, wn = G.gcn_norm_edges(ei, ew, n, torch.float32)
    ref = np.zeros((n, n), dtype=np.float64)
    np.add.at(ref, (d.numpy(), s.numpy()), wn.numpy().astype(np.float64))
    np.testing.assert_allclose(got, ref, atol=0, rtol=2e-7)
    # rows are sorted by destination and keep edge order: deterministic rebuild
    rp2, col2, val2 = R.graph.gcn_csr(ei.cuda(), None if ew is None else ew.cuda(), n)
    assert torch.equal(rp, rp2) and torch.equal(col, col2) and torch.equal(val, val2)


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("n,e,seed", [(9, 25, 0), (104, 400, 1), (1000, 12000, 2)])
def test_cheb_operator_matches_oracle(R, n, e, seed, weighted):
    ei, w = _rand_graph(n, e, seed)
    ew = w if weighted else None
    wt = R.graph.cheb_edge_weights(ei.cuda(), None if ew is None else ew.cuda(), n)
    rp, col, val = R.graph.raw_csr(ei.cuda(), wt, n)
    got = _csr_to_dense(rp, col, val, n)
    want = G.dense_cheb_operator(ei, ew, n, torch.float64).numpy()
    np.testing.assert_allclose(got, want, atol=2e-7, rtol=1e-6)
    assert np.all(np.diag(got) == 0)


def test_graph_rejects_bad_indices(R):
    ei = torch.tensor([[0, 7], [1, 2]]).cuda()
    with pytest.raises(ValueError):
        R.graph.gcn_csr(ei, None, 5)
    with pytest.raises(ValueError):
        R.graph.cheb_edge_weights(torch.tensor([[0, 1], [1, 0]]).cuda(), torch.tensor([1.0, -2.0]).cuda(), 3)


@pytest.mark.parametrize("width", [4, 32, 48, 96, 128, 256, 384, 768, 1024])
def test_spmm_matches_oracle(R, width):
    n, e = 777, 9000
    ei, w = _rand_graph(n, e, 5)
    rp, col, val = R.graph.gcn_csr(ei.cuda(), w.cuda(), n)
    x = torch.randn(n, width)
    got = R.ops.spmm_csr(rp, col, val, x.cuda()).cpu()
    s, d, wn = G.gcn_norm_edges(ei, w, n, torch.float32)
    want = G.propagate(s, d, wn, x, n)
    assert float((got - want).abs().max()) < 2e-6
    want64 = G.dense_gcn_operator(ei, w, n, torch.float64) @ x.double()
    assert float((got.double() - want64).abs().max()) < 5e-6


@pytest.mark.parametrize("width", [160, 192])      # 160: 128-byte panels; 192 (a multiple of 64 floats): 256-byte panels
def test_spmm_panel_variant_large_graph(R, width):
    """Large enough (X > 24 MB, width a multiple of 32) to take the XCD-aware column-panel kernels."""
    n, e = 41003, 300000
    ei, w = _rand_graph(n, e, 9)
    rp, col, val = R.graph.gcn_csr(ei.cuda(), w.cuda(), n)
    x = torch.randn(n, width, generator=torch.Generator().manual_seed(9))
    got = R.ops.spmm_csr(rp, col, val, x.cuda()).cpu()
    s, d, wn = G.gcn_norm_edges(ei, w, n, torch.float64)
    want = G.propagate(s, d, wn, x.double(), n)
    # fp32 sums in edge order vs float64: a few ulp of the largest row sum
    assert float((got.double() - want).abs().max()) < 4e-6 * max(1.0, float(want.abs().max()) / 8)
    again = R.ops.spmm_csr(rp, col, val, x.cuda()).cpu()
    assert torch.equal(got, again)


def test_spmm_dual_matches_two_single_operator_passes(R):
    n, width = 9000, 96
    g = R.data.synthetic_regional_graph(n, 80000, 4, seed=2, p_intra=0.9)
    pg = R.prepare_graph(g.edge_index.cuda(), None, [t.cuda() for t in g.region_index], [t.cuda() for t in g.region_attr], n)
    x = torch.randn(n, width).cuda()
    stacked = R.ops.spmm_csr(pg.rowptr, pg.col, pg.val, x)
    ya, yl = R.ops.spmm_dual(pg.m_rowptr, pg.m_col, pg.m_val_a, pg.m_val_l, x)
    assert float((ya - stacked[:n]).abs().max()) < 2e-6
    assert float((yl - stacked[n:]).abs().max()) < 2e-6
    # the merged pattern is the union: no more entries than the two operators together, at least as many as A_hat
    assert pg.nnz_gcn <= pg.m_col.numel() <= pg.nnz_gcn + pg.nnz_cheb


def test_spmm_empty_rows_and_hub(R):
    # node 0 receives every edge (hub), nodes 5.. receive none
    n = 300
    src = torch.arange(1, n)
    ei = torch.stack([src, torch.zeros_like(src)])
    wt = torch.rand(n - 1) + 0.5
    rp, col, val = R.graph.raw_csr(ei.cuda(), wt.cuda(), n)
    x = torch.randn(n, 96)
    got = R.ops.spmm_csr(rp, col, val, x.cuda()).cpu()
    want = torch.zeros(n, 96)
    want[0] = (wt.double().view(-1, 1) * x[1:].double()).sum(0).float()
    assert float((got - want).abs().max()) < 1e-4      # 299-term fp32 sum in edge order
    assert float(got[1:].abs().max()) == 0.0


def test_pack_x(R):
    x = torch.randn(37, 8, 12)
    assert torch.equal(R.ops.pack_x(x.cuda()).cpu(), x.permute(0, 2, 1).contiguous())


@pytest.mark.parametrize("m,k,n", [(1, 1, 1), (104, 8, 256), (624, 256, 512), (1000, 36, 130), (129, 257, 65), (4096, 256, 256)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_matches_torch_fp32(R, m, k, n, act):
    g = torch.Generator().manual_seed(m * 31 + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / max(1.0, k ** 0.5)
    b = torch.randn(n, generator=g)
    got = R.ops.linear(a.cuda(), w.cuda(), b.cuda(), act).cpu()
    want = a.double() @ w.double().t() + b.double()
    if act == 1:
        want = torch.nn.functional.leaky_relu(want, 0.01)
    elif act == 2:
        want = torch.relu(want)
    # fp32 MFMA = k-ordered fp32 fma chain: error ~ 1e-7 * sum|a*w|
    assert float((got.double() - want).abs().max()) < 2e-5


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("act", [0, 1, 3, 4])
def test_linear_full_tiles_every_row_every_run(R, mode, act):
    """Many full 128 x 128 tiles next to a partial row tile, repeated: the branch-free epilogue of full tiles (buffer stores,
    one per thread and row slot) must give every element, every time.  A 16-byte buffer store with an SGPR soffset whose
    data registers the next VALU instruction reused came out with that instruction's value now and then on gfx950 (first
    dword of a row = 1 + e^-x instead of sigmoid(x)); the stores now carry the row offset in the vector offset."""
    lib = R.load_library()
    lib.regt_set_gemm_mode(mode)
    try:
        g = torch.Generator().manual_seed(7 + act)
        m, n, k = 36000, 256, 64
        a = torch.randn(m, k, generator=g).cuda()
        w = (torch.randn(n, k, generator=g) * 0.1).cuda()
        b = torch.randn(n, generator=g).cuda()
        z = a.double() @ w.double().t() + b.double()
        want = {0: z, 1: torch.nn.functional.leaky_relu(z, 0.01), 3: torch.sigmoid(z), 4: torch.tanh(z)}[act].float()
        first = None
        for _ in range(5):
            got = R.ops.linear(a, w, b, act)
            assert float((got - want).abs().max()) < 2e-5
            if first is None:
                first = got.clone()
            else:
                assert torch.equal(got, first)
    finally:
        lib.regt_set_gemm_mode(0)


def test_linear_asymmetric_identity(R):
    # A = I with an asymmetric weight catches a transposed C write (cdna guide section 3)
    k = 64
    w = torch.arange(96 * k, dtype=torch.float32).reshape(96, k) / 100.0
    got = R.ops.linear(torch.eye(k).cuda(), w.cuda()).cpu()
    assert torch.equal(got, w.t().contiguous())


@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (624, 256, 256), (5000, 512, 256), (3000, 256, 8), (700, 128, 33), (40000, 256, 32)])
def test_wgrad_matches_torch(R, m, n, k):
    g = torch.Generator().manual_seed(m + n + k)
    d = torch.randn(m, n, generator=g)
    a = torch.randn(m, k, generator=g)
    dw, db = R.ops.wgrad(d.cuda(), a.cuda())
    want = d.double().t() @ a.double()
    scale = max(1.0, m ** 0.5)
    assert float((dw.cpu().double() - want).abs().max()) < 3e-5 * scale
    assert float((db.cpu().double() - d.double().sum(0)).abs().max()) < 3e-5 * scale
    dw2, db2 = R.ops.wgrad(d.cuda(), a.cuda())
    assert torch.equal(dw, dw2) and torch.equal(db, db2)   # slab reduction is deterministic


def test_prepared_graph_tpims(R, tpims):
    ri, rw = region_lists(tpims)
    n = tpims["node_data"].shape[0]
    g = R.prepare_graph(tpims["edge_index"].cuda(), None, [t.cuda() for t in ri], [t.cuda() for t in rw], n)
    d = _csr_to_dense(g.rowptr, g.col, g.val, n)
    np.testing.assert_allclose(d[:n], G.dense_gcn_operator(tpims["edge_index"], None, n).numpy(), atol=2e-7)
    lsum = sum(G.dense_cheb_operator(i, w, n).numpy() for i, w in zip(ri, rw))
    np.testing.assert_allclose(d[n:], lsum, atol=2e-7)
    bounds = [0, 44, 62, 75, 93, 104]
    for r in range(5):
        assert np.all(g.node_region_host[bounds[r]:bounds[r + 1]] == r)


def test_overlapping_regions_take_the_general_layout(R):
    a = torch.tensor([[0, 1], [1, 0]]).cuda()
    b = torch.tensor([[2, 1], [1, 2]]).cuda()      # node 1 also receives an edge in region b
    g = R.prepare_graph(a, None, [a, b], [None, None], 3)
    assert g.overlap and g.rowptr.numel() == 3 * 3 + 1 and g.m_rowptr is None


@pytest.mark.parametrize("n,e,regions,f,t,o,model,hidden", [
    (3000, 24000, 8, 32, 12, 1, "regt", 256), (1409, 9000, 3, 16, 12, 2, "regt", 256), (2200, 15000, 1, 8, 6, 1, "tgcn", 256),
    (20000, 90000, 4, 8, 1, 1, "regt", 256),        # T = 1: every row its own node (128 row-table entries per tile)
    (500, 3000, 2, 8, 48, 1, "regt", 256),          # T = 48: nodes straddle tiles
    (300, 2000, 2, 8, 150, 1, "regt", 256),         # T = 150 > 128: a node longer than a tile
    (1500, 9000, 3, 8, 12, 1, "regt", 512),         # C = 512: four column tiles, four partial attention dots per row
    (6000, 40000, 3, 8, 12, 1, "regt", 128)])       # C = 128: one column tile (the writer is the only workgroup of a row tile)
@pytest.mark.parametrize("mode", [0, 1], ids=["fp32mfma", "bf16x3split"])
def test_generated_candidate_dgrad_equals_the_two_launch_path(n, e, regions, f, t, o, model, hidden, mode):
    """fp32 backward with dhp generated inside the candidate data gradient (gemm_dgrad1_gen_kernel: no cell_bwd pass) against the
    two-launch path (regt_set_option("dgrad1_gen", 0)): every gradient that flows through dhp / dzp / drp / dh is BIT-identical
    (same element-wise helpers, same GEMM order); the attention gradient is summed in another fixed order (<= 1e-5 of its scale).
    Full tiles, a tail tile (1409 * 12 rows), a node-boundary inside a tile, the TemporalGCN variant."""
    import regtgcn_amd as R
    from oracle import model as M
    from test_gpu_model import _synthetic
    lib = R.load_library()
    ei, ri, rw, x = _synthetic(n, e, regions, f, t, seed=n)
    y = torch.rand(n, o, generator=torch.Generator().manual_seed(1)).cuda()
    if model == "regt":
        p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=3, hidden=hidden)
        mod = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions, hidden_channels=hidden)
        mod.load_state_dict(p, strict=True)
        mod = mod.cuda()
        graph = mod.prepare_graph(ei.cuda(), [i.cuda() for i in ri], [a.cuda() for a in rw])
    else:
        p = M.init_params("TemporalGCN", f, t, o, seed=3)
        mod = R.TemporalGCN(node_features=f, periods=t, output_dim=o)
        mod.load_state_dict(p, strict=True)
        mod = mod.cuda()
        w = torch.rand(ei.shape[1], generator=torch.Generator().manual_seed(2)) * 100 + 1
        graph = mod.prepare_graph(ei.cuda(), w.cuda(), n)
    xs = x.cuda()
    res = {}
    prev_mode = lib.regt_set_gemm_mode(mode)          # the generated-operand kernel exists for both fp32-storage arithmetics
    for gen in (1, 0, 1):
        prev = lib.regt_set_option(b"dgrad1_gen", gen)
        try:
            mod.zero_grad(set_to_none=True)
            pred, hidden = mod.forward_prepared(xs, graph)
            (R.functional.mse_loss(pred, y) + (hidden ** 2).mean()).backward()
            got = {k: q.grad.clone() for k, q in mod.named_parameters() if q.grad is not None}
        finally:
            lib.regt_set_option(b"dgrad1_gen", prev)
        if gen in res and t <= 64:
            assert all(torch.equal(got[k], res[gen][k]) for k in got)          # the generated path is bit-reproducible
        res[gen] = got
    lib.regt_set_gemm_mode(prev_mode)
    assert set(res[0]) == set(res[1])
    for k in res[0]:
        if k == "tgnn._attention":
            scale = float(res[0][k].abs().max())
            assert float((res[0][k] - res[1][k]).abs().max()) <= 2e-5 * scale + 1e-9
        elif t > 64:
            # (a node of more than 64 rows meets three or more partial sums in the forward's hidden state: not bit-reproducible
            # between ANY two runs, include/regtgcn.h -- the two backward paths then see different dOH in the last bits)
            assert float((res[0][k] - res[1][k]).abs().max()) <= 1e-5 * max(float(res[0][k].abs().max()), 1e-12), k
        else:
            assert torch.equal(res[0][k], res[1][k]), k


@pytest.mark.parametrize("n,e,regions,f,t,model", [(3000, 24000, 8, 32, 12, "regt"), (1409, 9000, 3, 32, 5, "regt"), (2500, 20000, 1, 32, 6, "tgcn"),
                                                    (900, 6000, 2, 24, 12, "regt")])
def test_sixty_four_wide_weight_gradient_tile_equals_two_of_thirty_two(n, e, regions, f, t, model):
    """fp32: the fused dA0 | dA_r gradient ds^T [x | L~ x] (and TemporalGCN's dzr^T [x | L~ x]) at F = 32 is 64 columns wide: one
    64-column tile whose two-part right-hand side splits INSIDE the tile (wgrad_kernel<64>) instead of two 32-column tiles that
    each read ds.  Same products, same k order per output element: every gradient bit-identical.  F = 24: not a case of it."""
    import regtgcn_amd as R
    from oracle import model as M
    from test_gpu_model import _synthetic
    lib = R.load_library()
    ei, ri, rw, x = _synthetic(n, e, regions, f, t, seed=n)
    y = torch.rand(n, 1, generator=torch.Generator().manual_seed(1)).cuda()
    if model == "regt":
        p = M.init_params("RegionalTemporalGCN", f, t, 1, num_nodes=n, num_regions=regions, seed=3)
        mod = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=1, num_regions=regions)
        mod.load_state_dict(p, strict=True)
        mod = mod.cuda()
        graph = mod.prepare_graph(ei.cuda(), [i.cuda() for i in ri], [a.cuda() for a in rw])
    else:
        p = M.init_params("TemporalGCN", f, t, 1, seed=3)
        mod = R.TemporalGCN(node_features=f, periods=t, output_dim=1)
        mod.load_state_dict(p, strict=True)
        mod = mod.cuda()
        graph = mod.prepare_graph(ei.cuda(), None, n)
    res = {}
    for on in (1, 0):
        prev = lib.regt_set_option(b"wgrad_bnw64", on)
        try:
            mod.zero_grad(set_to_none=True)
            pred, hidden = mod.forward_prepared(x.cuda(), graph)
            (R.functional.mse_loss(pred, y) + (hidden ** 2).mean()).backward()
            res[on] = {k: q.grad.clone() for k, q in mod.named_parameters() if q.grad is not None}
        finally:
            lib.regt_set_option(b"wgrad_bnw64", prev)
    assert set(res[0]) == set(res[1])
    bad = {k: float((res[0][k] - res[1][k]).abs().max()) for k in res[0] if not torch.equal(res[0][k], res[1][k])}
    assert not bad, bad


@pytest.mark.parametrize("n,e,f,t,o", [(3000, 24000, 32, 12, 1), (1409, 9000, 8, 6, 2), (104, 400, 8, 12, 1)])
def test_temporal_gcn_collapsed_gates_equal_the_uncollapsed_form(n, e, f, t, o):
    """TemporalGCN's hidden input has no activation (models/TemporalGCN.py:88), so the gates' use of it folds into x and L~ x
    (api.hip: FMT_TCOLLAPSE -- gate GEMM at K = 3F, no K = 2C data gradient, no (2C x C) weight gradient in the backward pass).
    Same function, reassociated: outputs within 2e-6, every gradient within 2e-5 of its scale of the uncollapsed form
    (regt_set_option("tgcn_collapse", 0)); F = 32 (two-part right-hand side), F = 8 (one launch per part), a TPIMS-sized graph."""
    import regtgcn_amd as R
    from oracle import model as M
    from test_gpu_model import _synthetic
    lib = R.load_library()
    ei, ri, rw, x = _synthetic(n, e, 2, f, t, seed=n)
    y = torch.rand(n, o, generator=torch.Generator().manual_seed(1)).cuda()
    p = M.init_params("TemporalGCN", f, t, o, seed=3)
    mod = R.TemporalGCN(node_features=f, periods=t, output_dim=o)
    mod.load_state_dict(p, strict=True)
    mod = mod.cuda()
    w = torch.rand(ei.shape[1], generator=torch.Generator().manual_seed(2)) * 100 + 1
    graph = mod.prepare_graph(ei.cuda(), w.cuda(), n)
    xs = x.cuda()
    res = {}
    for col in (1, 0):
        prev = lib.regt_set_option(b"tgcn_collapse", col)
        try:
            mod.zero_grad(set_to_none=True)
            pred, hidden = mod.forward_prepared(xs, graph)
            (R.functional.mse_loss(pred, y) + (hidden ** 2).mean()).backward()
            res[col] = (pred.detach().clone(), hidden.detach().clone(), {k: q.grad.clone() for k, q in mod.named_parameters() if q.grad is not None})
        finally:
            lib.regt_set_option(b"tgcn_collapse", prev)
    assert float((res[1][0] - res[0][0]).abs().max()) < 2e-6 * max(1.0, float(res[0][0].abs().max()))
    assert float((res[1][1] - res[0][1]).abs().max()) < 2e-6 * max(1.0, float(res[0][1].abs().max()))
    assert set(res[1][2]) == set(res[0][2])
    for k, g0 in res[0][2].items():
        scale = max(float(g0.abs().max()), 1e-12)
        assert float((res[1][2][k] - g0).abs().max()) <= 2e-5 * scale + 1e-9, (k, float((res[1][2][k] - g0).abs().max()), scale)
