"""Known-answer tests for the restated PyG operators (oracle/graph_ops.py) -- SURVEY.md 8(c) G1.

The reference holds no test for these third-party operators; they are pinned here against
dense-matrix statements of the published formulas and against hand-computed values."""
import math

import numpy as np
import pytest
import torch

from oracle import graph_ops as G


def _random_graph(n, e, seed, with_loop=True, with_dup=True):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, n, (e,), generator=g)
    if with_loop:
        src[0] = dst[0] = 2
    if with_dup:
        src[1], dst[1] = src[2], dst[2]
    w = torch.rand(e, generator=g, dtype=torch.float64) * 10 + 0.1
    return torch.stack([src, dst]), w


def test_gcn_hand_computed_path_graph():
    # 0 -> 1 -> 2, unit weights.  in-degrees incl. loop: d = [1, 2, 2]
    ei = torch.tensor([[0, 1], [1, 2]])
    src, dst, w = G.gcn_norm_edges(ei, None, 3, torch.float64)
    dense = torch.zeros(3, 3, dtype=torch.float64)
    dense.index_put_((dst, src), w, accumulate=True)
    expect = torch.tensor([[1.0, 0, 0],
                           [1 / math.sqrt(2), 0.5, 0],
                           [0, 0.5, 0.5]], dtype=torch.float64)
    assert torch.allclose(dense, expect, atol=1e-14)


def test_cheb_hand_computed_weighted():
    # 0 -> 1 (w=4), 1 -> 0 (w=9), 2 isolated.  out-degree s = [4, 9, 0]
    ei = torch.tensor([[0, 1], [1, 0]])
    ew = torch.tensor([4.0, 9.0], dtype=torch.float64)
    src, dst, w = G.cheb_norm_edges(ei, ew, 3, torch.float64)
    dense = torch.zeros(3, 3, dtype=torch.float64)
    dense.index_put_((dst, src), w, accumulate=True)
    # L~[1,0] = -4/(2*3), L~[0,1] = -9/(3*2); diagonal exactly zero, isolated row/col zero
    expect = torch.tensor([[0, -1.5, 0], [-2.0 / 3.0, 0, 0], [0, 0, 0]], dtype=torch.float64)
    assert torch.allclose(dense, expect, atol=1e-14)


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gcn_matches_dense_operator(seed, weighted):
    n, e = 9, 25
    ei, w = _random_graph(n, e, seed)
    ew = w if weighted else None
    x = torch.randn(n, 5, dtype=torch.float64)
    src, dst, wn = G.gcn_norm_edges(ei, ew, n, torch.float64)
    got = G.propagate(src, dst, wn, x, n)
    want = G.dense_gcn_operator(ei, ew, n) @ x
    assert torch.allclose(got, want, atol=1e-12)


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_cheb_matches_dense_operator(seed, weighted):
    n, e = 9, 25
    ei, w = _random_graph(n, e, seed)
    ei[:, ei[0] == 7] = torch.tensor([[3], [4]])   # node 7: no out-edges -> deg 0 -> inf -> 0
    ew = w if weighted else None
    x = torch.randn(n, 5, dtype=torch.float64)
    src, dst, wn = G.cheb_norm_edges(ei, ew, n, torch.float64)
    got = G.propagate(src, dst, wn, x, n)
    want = G.dense_cheb_operator(ei, ew, n) @ x
    assert torch.allclose(got, want, atol=1e-12)
    # scaled Laplacian has an exactly-zero diagonal (2L/lambda_max - I with lambda_max = 2)
    loops = src == dst
    assert torch.all(wn[loops] == 0)


def test_gcn_existing_loop_weight_is_kept_last_wins():
    ei = torch.tensor([[0, 1, 1, 0], [1, 1, 1, 0]])
    ew = torch.tensor([2.0, 5.0, 7.0, 3.0], dtype=torch.float64)
    src, dst, w = G.gcn_norm_edges(ei, ew, 2, torch.float64)
    # kept edge 0->1 (2); loops: node0 -> 3, node1 -> 7 (last listed); deg = [3, 9]
    dense = torch.zeros(2, 2, dtype=torch.float64)
    dense.index_put_((dst, src), w, accumulate=True)
    expect = torch.tensor([[1.0, 0.0], [2 / math.sqrt(27), 7.0 / 9.0]], dtype=torch.float64)
    assert torch.allclose(dense, expect, atol=1e-14)


def test_conv_layers_compose():
    n, f, c = 7, 4, 6
    ei, w = _random_graph(n, 15, 3)
    x = torch.randn(n, f, dtype=torch.float64)
    w0, w1, wl = (torch.randn(c, f, dtype=torch.float64) for _ in range(3))
    b = torch.randn(c, dtype=torch.float64)
    cheb = G.cheb_conv(x, ei, w, w0, w1, b)
    want = x @ w0.t() + (G.dense_cheb_operator(ei, w, n) @ x) @ w1.t() + b
    assert torch.allclose(cheb, want, atol=1e-12)
    gcn = G.gcn_conv(x, ei, None, wl, b)
    want = G.dense_gcn_operator(ei, None, n) @ (x @ wl.t()) + b
    assert torch.allclose(gcn, want, atol=1e-12)


# ---- SAGEConv / GATConv restatements (base blocks 'graphsage' / 'gat' of the TGCN cell) ----------------

def test_sage_conv_is_mean_of_listed_in_edges():
    # node 3 has a self loop (counted as a neighbour), edge 2->0 is listed twice (counted twice), node 6 has no in-edge (mean 0)
    ei = torch.tensor([[0, 1, 2, 3, 3, 5, 2, 4, 4], [1, 2, 0, 3, 0, 2, 0, 1, 1]])
    n, f, c = 7, 5, 6
    x = torch.randn(n, f, dtype=torch.float64)
    wl, wr = torch.randn(c, f, dtype=torch.float64), torch.randn(c, f, dtype=torch.float64)
    bl = torch.randn(c, dtype=torch.float64)
    got = G.sage_conv(x, ei, wl, bl, wr)
    a = G.dense_mean_operator(ei, n)
    assert torch.allclose(a[0], torch.tensor([0, 0, 2 / 3, 1 / 3, 0, 0, 0], dtype=torch.float64))       # 2->0 twice, 3->0 once
    assert torch.allclose(a[3], torch.tensor([0, 0, 0, 1.0, 0, 0, 0], dtype=torch.float64))             # own loop only
    assert float(a[6].abs().sum()) == 0.0
    assert torch.allclose(got, (a @ x) @ wl.t() + bl + x @ wr.t(), atol=1e-12)


def test_gat_conv_matches_dense_attention_formula():
    ei = torch.tensor([[0, 1, 2, 3, 3, 5, 2, 4, 4], [1, 2, 0, 3, 0, 2, 0, 1, 1]])    # a self loop (dropped, re-added once) and a duplicate
    n, f, c = 7, 5, 6
    x = torch.randn(n, f, dtype=torch.float64)
    w = torch.randn(c, f, dtype=torch.float64)
    a_s, a_d = torch.randn(1, 1, c, dtype=torch.float64), torch.randn(1, 1, c, dtype=torch.float64)
    b = torch.randn(c, dtype=torch.float64)
    got = G.gat_conv(x, ei, w, a_s, a_d, b)
    alpha = G.dense_gat_attention(x, ei, w, a_s, a_d)
    assert torch.allclose(alpha.sum(dim=1), torch.ones(n, dtype=torch.float64))
    assert float(alpha[6, 6]) == 1.0                       # a node without in-edges attends to itself only
    assert torch.allclose(got, alpha @ (x @ w.t()) + b, atol=1e-12)
    # hand-checkable two-node case: node 1 hears node 0 and itself
    x2 = torch.tensor([[1.0], [2.0]], dtype=torch.float64)
    one = torch.ones(1, 1, dtype=torch.float64)
    out = G.gat_conv(x2, torch.tensor([[0], [1]]), one, one.view(1, 1, 1), one.view(1, 1, 1), torch.zeros(1, dtype=torch.float64))
    e01, e11 = 1.0 + 2.0, 2.0 + 2.0                        # a_s[j] + a_d[i], positive -> leaky_relu is the identity
    a01 = math.exp(e01) / (math.exp(e01) + math.exp(e11))
    assert abs(float(out[1, 0]) - (a01 * 1.0 + (1 - a01) * 2.0)) < 1e-12 and abs(float(out[0, 0]) - 1.0) < 1e-12
