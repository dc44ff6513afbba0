"""bf16 rows of x / A_hat x / L~ x and the fused forward kernel of the bf16 arithmetic (REGT_GEMM_MODE=bf16, csrc/fused.hip).

The fused kernel replaces three launches (regional embedding, gates, candidate: RegionalTemporalGCN.py:136-148,
models/utils.py:168-188) and is built to reproduce their arithmetic exactly: same bf16 MFMA operands, same k order, same fp32
gate math, same rounding points, same per-node summation order.  So the checks here are BIT-FOR-BIT: with a snapshot that is
bf16-representable (the three-launch path then rounds x, A_hat x and L~ x at LDS staging to the very values the bf16-row path
stores), forward outputs and every gradient of the two paths must be identical.  The oracle-level tolerance of the mode
itself is tests/test_gpu_bf16.py's business."""
import numpy as np
import pytest
import torch

from fused_math import bf16_round
from oracle import model as M
from test_gpu_model import _synthetic

pytestmark = pytest.mark.gpu
DEFAULT_WGRAD_TILE = 256


@pytest.fixture()
def bf16_mode():
    import regtgcn_amd as R
    lib = R.load_library()
    prev = lib.regt_set_gemm_mode(2)
    # one launch per weight gradient here: the paired launches (default with the ring kernel) only exist on the bf16-row layout and
    # chunk the rows of dGh / dGzr differently -- the xbf = 0 / 1 comparisons below are about the forward / data-gradient kernels
    lib.regt_set_option(b"wgrad_pairs", 0)
    yield R
    lib.regt_set_gemm_mode(prev)
    lib.regt_set_option(b"xbf", 1)
    lib.regt_set_option(b"fused_bwd", 1)
    lib.regt_set_option(b"fused_rows", 1)
    lib.regt_set_option(b"spmm_rows", 0)
    lib.regt_set_option(b"wgrad_ring", 6)
    lib.regt_set_option(b"wgrad_tile", DEFAULT_WGRAD_TILE)
    lib.regt_set_option(b"wgrad_pairs", 2)
    lib.regt_set_option(b"wgrad_wave", 1)
    lib.regt_set_option(b"wgrad_ring256", 2)


def test_pack_x_bf16_rounds_to_nearest_even_and_leaves_halo_rows_alone():
    import regtgcn_amd as R
    x = torch.rand(777, 64, 12, device="cuda") * 3 - 1
    out = torch.full((777 + 9, 12, 64), 7.0, dtype=torch.bfloat16, device="cuda")
    R.ops.pack_x_bf16_into(x, out)
    assert torch.equal(out[:777], x.permute(0, 2, 1).contiguous().to(torch.bfloat16))
    assert bool((out[777:] == 7.0).all())


@pytest.mark.parametrize("n,e,regions,w,extra", [(5000, 40000, 4, 768, 0), (20000, 150000, 8, 384, 333), (300, 2000, 2, 64, 5),
                                                  (9, 30, 2, 128, 0)])
def test_bf16_row_aggregation_is_the_rounded_fp32_aggregation(n, e, regions, w, extra):
    """regt_spmm_dual_bf16 on bf16 rows == regt_spmm_dual on the same values as fp32, rounded once: same CSR order, fp32 sums."""
    import regtgcn_amd as R
    ei, ri, rw, _ = _synthetic(n, e, regions, 4, 1, seed=n)
    g = R.prepare_graph(ei.cuda(), None, [i.cuda() for i in ri], [a.cuda() for a in rw], n)
    x = torch.randn(n + extra, w, device="cuda").to(torch.bfloat16)
    lib = R.load_library()
    try:            # the opt-in row-block kernel and the default panel kernel sum in the same order
        lib.regt_set_option(b"spmm_rows", 1)
        ra, rl = R.ops.spmm_dual_bf16(g.m_rowptr, g.m_col, g.m_val_a, g.m_val_l, x)
    finally:
        lib.regt_set_option(b"spmm_rows", 0)
    ya, yl = R.ops.spmm_dual_bf16(g.m_rowptr, g.m_col, g.m_val_a, g.m_val_l, x)
    assert torch.equal(ra, ya) and torch.equal(rl, yl)
    if w % 32 == 0:
        fa, fl = R.ops.spmm_dual(g.m_rowptr, g.m_col, g.m_val_a, g.m_val_l, x[:n].float().contiguous())
    # reference sums in float64 from the CSR itself (also covers halo columns: none here, but x has extra rows)
    rp, col = g.m_rowptr.cpu().long(), g.m_col.cpu().long()
    rows = torch.repeat_interleave(torch.arange(n), rp[1:] - rp[:-1])
    xd = x.float().cpu().double()
    for got, val in ((ya, g.m_val_a), (yl, g.m_val_l)):
        want = torch.zeros(n, w, dtype=torch.float64).index_add_(0, rows, val.cpu().double()[:, None] * xd[col])
        err = (got.float().cpu().double() - want).abs()
        assert bool((err <= 2.0 ** -8 * want.abs() + 1e-6).all())           # one bf16 rounding of an fp32 sum
    if w % 32 == 0:
        assert torch.equal(ya, fa.to(torch.bfloat16)) and torch.equal(yl, fl.to(torch.bfloat16))


@pytest.mark.parametrize("n,e,regions,w", [(20000, 200000, 8, 384), (4096, 30000, 3, 96 * 4)])
def test_row_block_kernel_equals_panel_kernel_bit_for_bit(n, e, regions, w):
    import regtgcn_amd as R
    lib = R.load_library()
    ei, ri, rw, _ = _synthetic(n, e, regions, 4, 1, seed=n + 1)
    g = R.prepare_graph(ei.cuda(), None, [i.cuda() for i in ri], [a.cuda() for a in rw], n)
    x = torch.randn(n, w, device="cuda")
    try:
        lib.regt_set_option(b"spmm_rows", 1)
        a1, l1 = R.ops.spmm_dual(g.m_rowptr, g.m_col, g.m_val_a, g.m_val_l, x)
        lib.regt_set_option(b"spmm_rows", 0)
        a0, l0 = R.ops.spmm_dual(g.m_rowptr, g.m_col, g.m_val_a, g.m_val_l, x)
    finally:
        lib.regt_set_option(b"spmm_rows", 0)
    assert torch.equal(a1, a0) and torch.equal(l1, l0)


def _run(R, n, e, regions, f, t, o, xbf, seed=0, fused_bwd=1, hidden=256, fused_rows=1):
    lib = R.load_library()
    lib.regt_set_option(b"xbf", xbf)
    lib.regt_set_option(b"fused_bwd", fused_bwd)
    lib.regt_set_option(b"fused_rows", fused_rows)
    ei, ri, rw, x = _synthetic(n, e, regions, f, t, seed=n + seed)
    x = bf16_round(x)
    y = torch.rand(n, o, generator=torch.Generator().manual_seed(1))
    p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=3, hidden=hidden)
    mod = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions, hidden_channels=hidden)
    mod.load_state_dict(p, strict=True)
    mod = mod.cuda()
    pred, hid = mod(x.cuda(), ei.cuda(), [i.cuda() for i in ri], [a.cuda() for a in rw])
    (torch.mean((pred - y.cuda()) ** 2) + 1e-3 * hid.sum()).backward()
    grads = {k: q.grad.detach().clone() for k, q in mod.named_parameters() if q.grad is not None}
    return pred.detach(), hid.detach(), grads


# (nodes, edges, regions, F, T, O): 64 regions = configs[4]; O = 3; T = 5 and a row count that is no multiple of 64 (tail tile,
# tiles that start inside a node); T = 48 (a node spans two 64-row tiles); one region per ~100 nodes (tiles with two regions)
FUSED_SHAPES = [(2048, 20000, 64, 64, 12, 1), (1200, 9000, 4, 64, 12, 3), (701, 5000, 3, 64, 5, 1), (400, 3000, 2, 64, 48, 1),
                (600, 4000, 6, 64, 1, 1), (1500, 15000, 8, 32, 12, 1), (450, 3000, 3, 32, 6, 2)]       # F = 32: the cfg-3 width


@pytest.mark.parametrize("n,e,regions,f,t,o", FUSED_SHAPES)
def test_fused_forward_equals_three_launch_path_bit_for_bit(bf16_mode, n, e, regions, f, t, o):
    R = bf16_mode
    p1, h1, g1 = _run(R, n, e, regions, f, t, o, 1)
    p0, h0, g0 = _run(R, n, e, regions, f, t, o, 0)
    worst = {"pred": float((p1 - p0).abs().max()), "hidden": float((h1 - h0).abs().max())}
    assert worst["pred"] == 0.0 and worst["hidden"] == 0.0, worst
    if f == 32:
        # F = 32: the (C x F) gradients dGh / dGzr run on another kernel when A_hat x is stored as bf16 rows (bf16-pipe kernel, its
        # own row chunks) than when it is fp32 (skinny fp32-MFMA kernel): same products, another summation order
        for k in g0:
            assert float((g1[k] - g0[k]).abs().max()) <= 2e-4 * float(g0[k].abs().max()) + 1e-9, k
    else:
        worst.update({k: float((g1[k] - g0[k]).abs().max()) for k in g0})
        bad = {k: v for k, v in worst.items() if v != 0.0}
        assert not bad, bad
    assert float(h1.abs().max()) > 0 and all(bool(torch.isfinite(v).all()) for v in g1.values())


@pytest.mark.parametrize("n,e,regions,f,t,o", FUSED_SHAPES[:3] + [(1500, 15000, 8, 32, 12, 1), (30000, 250000, 8, 64, 12, 1)])
def test_row_owning_forward_with_the_short_ring_equals_the_default_form(bf16_mode, n, e, regions, f, t, o):
    """fused_rows = 2: two workgroups of four waves per CU, a ring of 8 slices each instead of 16 -- the hand-counted vmcnt waits of
    csrc/fused_rows.hip have a quarter of the slack there (a request is read four steps after it was issued, not twelve; this form
    is what showed that the count reached two slices too far back).  Same 16-row blocks, same arithmetic: every output and gradient
    identical, run after run -- a wait that is too weak shows up as a difference that comes and goes."""
    R = bf16_mode
    lib = R.load_library()
    try:
        p0, h0, g0 = _run(R, n, e, regions, f, t, o, 1, fused_rows=1)
        for rep in range(3):
            p1, h1, g1 = _run(R, n, e, regions, f, t, o, 1, seed=0, fused_rows=2)
            assert torch.equal(p1, p0) and torch.equal(h1, h0), rep
            for k in g0:
                assert torch.equal(g1[k], g0[k]), (rep, k)
    finally:
        lib.regt_set_option(b"fused_rows", 1)


# the fused backward kernel does not depend on F, nor on the forward being the fused one ((333, ..., 32, 7): T F % 64 != 0)
@pytest.mark.parametrize("n,e,regions,f,t,o", FUSED_SHAPES + [(333, 2500, 2, 32, 7, 2)])
def test_fused_backward_equals_three_launch_backward_bit_for_bit(bf16_mode, n, e, regions, f, t, o):
    """cell_bwd + dgrad_candidate + dgrad_gates as one kernel (csrc/fused.hip, fused_bwd_kernel): same operands, same k order, same
    rounding points -> every gradient identical, except the attention gradient, whose per-row dots are summed in another
    (fixed) order."""
    R = bf16_mode
    p1, h1, g1 = _run(R, n, e, regions, f, t, o, 1, fused_bwd=1)
    p0, h0, g0 = _run(R, n, e, regions, f, t, o, 1, fused_bwd=0)
    assert torch.equal(p1, p0) and torch.equal(h1, h0)
    assert set(g1) == set(g0)
    worst = {k: float((g1[k] - g0[k]).abs().max()) for k in g0 if k != "tgnn._attention"}
    bad = {k: v for k, v in worst.items() if v != 0.0}
    assert not bad, bad
    a1, a0 = g1["tgnn._attention"], g0["tgnn._attention"]
    assert float((a1 - a0).abs().max()) <= 1e-4 * float(a0.abs().max()) + 1e-7, (a1, a0)
    assert all(bool(torch.isfinite(v).all()) for v in g1.values())


# rows per weight-gradient chunk vary with the shape: chunks of fewer half slabs than the ring is deep (600 x 1), chunks that end
# inside a ring turn, a tail chunk, 20 000 x 12 rows (chunks of hundreds of half slabs)
@pytest.mark.parametrize("ring,tile,pairs", [(4, 128, 1), (6, 128, 0), (8, 128, 1), (6, 256, 1), (6, 256, 0), (6, 1024 + 256, 1)])
@pytest.mark.parametrize("n,e,regions,f,t,o", FUSED_SHAPES[:5] + [(20000, 150000, 8, 64, 12, 1), (37, 200, 2, 64, 3, 1)])
def test_ring_weight_gradient_equals_the_one_ahead_kernel_bit_for_bit(bf16_mode, ring, tile, pairs, n, e, regions, f, t, o):
    """wgrad_bf16_ring_kernel<D, MI> (bf16-stored operands requested D half slabs ahead through a register ring; 128- or 256-row
    output tiles; descriptors that end with the row chunk instead of a row test per step) walks the same half slabs in the same
    order with the same MFMA as wgrad_split_kernel<1, true, true>: identical slabs, so identical gradients -- with the gradients
    of a left operand paired in one launch (dhp^T [q | A_hat x], dzr^T [h | A_hat x]) and one launch each."""
    R = bf16_mode
    lib = R.load_library()
    lib.regt_set_option(b"wgrad_pairs", pairs)
    lib.regt_set_option(b"wgrad_wave", 0)              # the same row chunks on both sides (the one-wave chunking follows the ring kernel)
    lib.regt_set_option(b"wgrad_ring", ring)
    lib.regt_set_option(b"wgrad_tile", tile & 1023)
    lib.regt_set_option(b"wgrad_ring256", 4 if tile > 1024 else 2)         # (1024 + 256: the 256-row tile with a ring of 4)
    p1, h1, g1 = _run(R, n, e, regions, f, t, o, 1)
    lib.regt_set_option(b"wgrad_ring", 0)
    p0, h0, g0 = _run(R, n, e, regions, f, t, o, 1)
    assert torch.equal(p1, p0) and torch.equal(h1, h0) and set(g1) == set(g0)
    bad = {k: float((g1[k] - g0[k]).abs().max()) for k in g0 if not torch.equal(g1[k], g0[k])}
    assert not bad, bad
    assert all(bool(torch.isfinite(v).all()) for v in g1.values())


@pytest.mark.parametrize("hidden", [128, 384, 512])
def test_ring_weight_gradient_other_hidden_widths(bf16_mode, hidden):
    """C = 128 (one 128-row tile for dhp, a 256-row tile for dz|dr), 384 (no 256-row tiling of C; 768 = three for 2C), 512: the
    ring kernel's tile choice follows the output's row count; paired launches need the [C | F] split on a column-tile boundary."""
    R = bf16_mode
    lib = R.load_library()
    shape = (3000, 24000, 4, 64, 12, 1)
    lib.regt_set_option(b"wgrad_pairs", 1)
    lib.regt_set_option(b"wgrad_wave", 0)
    lib.regt_set_option(b"wgrad_ring", 6)
    p1, h1, g1 = _run(R, *shape, 1, hidden=hidden)
    lib.regt_set_option(b"wgrad_ring", 0)
    p0, h0, g0 = _run(R, *shape, 1, hidden=hidden)
    assert torch.equal(p1, p0) and torch.equal(h1, h0) and set(g1) == set(g0)
    bad = {k: float((g1[k] - g0[k]).abs().max()) for k in g0 if not torch.equal(g1[k], g0[k])}
    assert not bad, bad
    assert float(g1["tgnn._base_tgcn.linear_h.weight"].abs().max()) > 0


def test_one_wave_row_chunking_of_the_paired_weight_gradients(bf16_mode):
    """The paired ring-kernel launches cut the rows into as many chunks as fill the GPU once (api.hip / wgrad_ring_chunking) instead
    of the layout's ~128: the same products summed over other chunk boundaries -- gradients within 2e-5 of their scale, the
    rest of the step untouched.  40 000 x 12 rows: 170 / 85 chunks of 2 848 / 5 664 rows instead of 128 of 3 776."""
    R = bf16_mode
    lib = R.load_library()
    lib.regt_set_option(b"wgrad_pairs", 1)
    shape = (40000, 300000, 8, 64, 12, 1)
    lib.regt_set_option(b"wgrad_wave", 1)
    p1, h1, g1 = _run(R, *shape, 1)
    p1b, h1b, g1b = _run(R, *shape, 1)
    lib.regt_set_option(b"wgrad_wave", 0)
    p0, h0, g0 = _run(R, *shape, 1)
    assert torch.equal(p1, p0) and torch.equal(h1, h0) and set(g1) == set(g0)
    assert all(torch.equal(g1[k], g1b[k]) for k in g1)                       # reproducible
    changed = [k for k in g0 if not torch.equal(g1[k], g0[k])]
    assert changed, "the one-wave chunking did not take effect"
    for k in g0:
        assert float((g1[k] - g0[k]).abs().max()) <= 2e-5 * float(g0[k].abs().max()) + 1e-9, k


@pytest.mark.parametrize("n,e,regions,f,t,o", [(300, 2400, 3, 64, 100, 1), (130, 900, 2, 32, 255, 2)])
def test_fused_kernels_with_windows_longer_than_a_tile(bf16_mode, n, e, regions, f, t, o):
    """T > 64: a node's rows span three or more 64-row tiles, so its hidden-state row receives more than two atomically added
    partial sums (in either path) -- their order is the one thing not fixed, hence a tolerance instead of bit equality here."""
    R = bf16_mode
    p1, h1, g1 = _run(R, n, e, regions, f, t, o, 1, fused_bwd=1)
    p0, h0, g0 = _run(R, n, e, regions, f, t, o, 0, fused_bwd=0)
    for a, b in ((p1, p0), (h1, h0)):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-7
    for k in g0:
        # a last-bit difference of the hidden state can flip the bf16 rounding of a head operand: bf16-level tolerance for gradients
        assert float((g1[k] - g0[k]).abs().max()) <= 2.0 ** -7 * float(g0[k].abs().max()) + 1e-7, k
    assert all(bool(torch.isfinite(v).all()) for v in g1.values())


def test_packed_bf16_rows_equal_packed_fp32_rows(bf16_mode):
    """Region-shard entry: bf16 rows packed by the caller (regt_pack_x_bf16 -> regt_forward_packed_bf16) give the results of the
    fp32 packed rows (converted inside regt_forward_packed) bit for bit; halo rows are present but unread here."""
    R = bf16_mode
    lib = R.load_library()
    lib.regt_set_option(b"xbf", 1)
    n, e, regions, f, t, o = 1500, 12000, 5, 64, 12, 1
    ei, ri, rw, x = _synthetic(n, e, regions, f, t, seed=5)
    y = torch.rand(n, o).cuda()
    p = M.init_params("RegionalTemporalGCN", f, t, o, num_nodes=n, num_regions=regions, seed=3)
    outs = []
    for dt in (torch.float32, torch.bfloat16):
        mod = R.RegionalTemporalGCN(node_features=f, num_nodes=n, periods=t, output_dim=o, num_regions=regions)
        mod.load_state_dict(p, strict=True)
        mod = mod.cuda()
        graph = mod.prepare_graph(ei.cuda(), [i.cuda() for i in ri], [a.cuda() for a in rw])
        ext = torch.zeros(n + 40, t, f, dtype=dt, device="cuda")
        (R.ops.pack_x_into if dt == torch.float32 else R.ops.pack_x_bf16_into)(x.cuda(), ext)
        pred, hidden = mod.forward_packed(ext, graph)
        torch.mean((pred - y) ** 2).backward()
        outs.append((pred.detach(), hidden.detach(), {k: q.grad.clone() for k, q in mod.named_parameters() if q.grad is not None}))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        assert torch.equal(outs[0][2][k], outs[1][2][k]), k
