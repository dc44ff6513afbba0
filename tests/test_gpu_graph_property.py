"""Property test of the graph preparation kernels: arbitrary small multigraphs (self loops, repeated edges, isolated
nodes, hubs, empty edge lists) -- the normalised operators built on the GPU equal the oracle's dense operators, and
the merged two-weight operator reproduces both of its parts."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import graph_ops as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import regtgcn_amd as R
    R.load_library()
    return R


def _dense(rowptr, col, val, n_cols):
    rowptr, col, val = rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy().astype(np.float64)
    d = np.zeros((len(rowptr) - 1, n_cols))
    for i in range(len(rowptr) - 1):
        for p in range(rowptr[i], rowptr[i + 1]):
            d[i, col[p]] += val[p]
    return d


@st.composite
def graphs(draw):
    n = draw(st.integers(1, 24))
    e = draw(st.integers(0, 80))
    node = st.integers(0, n - 1)
    # skew towards collisions: few distinct endpoints when `narrow`
    narrow = draw(st.booleans())
    pick = st.integers(0, min(n - 1, 2)) if narrow else node
    src = draw(st.lists(pick, min_size=e, max_size=e))
    dst = draw(st.lists(pick if draw(st.booleans()) else node, min_size=e, max_size=e))
    w = draw(st.lists(st.floats(0.25, 3000.0, allow_nan=False, width=32), min_size=e, max_size=e))
    weighted = draw(st.booleans())
    ei = torch.tensor([src, dst], dtype=torch.int64).reshape(2, e)
    return n, ei, torch.tensor(w, dtype=torch.float32), weighted


@settings(max_examples=40, deadline=None, derandomize=True)
@given(graphs())
def test_operators_on_arbitrary_multigraphs(R, g):
    n, ei, w, weighted = g
    ew = w if weighted else None
    dev_w = None if ew is None else ew.cuda()
    rp, col, val = R.graph.gcn_csr(ei.cuda(), dev_w, n)
    np.testing.assert_allclose(_dense(rp, col, val, n), G.dense_gcn_operator(ei, ew, n, torch.float64).numpy(), atol=3e-7, rtol=2e-6)
    wt = R.graph.cheb_edge_weights(ei.cuda(), dev_w, n)
    rl, cl, vl = R.graph.raw_csr(ei.cuda(), wt, n)
    lap = _dense(rl, cl, vl, n)
    np.testing.assert_allclose(lap, G.dense_cheb_operator(ei, ew, n, torch.float64).numpy(), atol=3e-7, rtol=2e-6)
    # merged operator: union pattern, both weights per entry
    m_rp, m_col, m_a, m_l = R.graph.merge_operators(rp, col, val, rl, cl, vl, n)
    np.testing.assert_allclose(_dense(m_rp, m_col, m_a, n), _dense(rp, col, val, n), atol=1e-7)
    np.testing.assert_allclose(_dense(m_rp, m_col, m_l, n), lap, atol=1e-7)
    # and the aggregation itself on a random right-hand side
    x = torch.randn(n, 32, generator=torch.Generator().manual_seed(n))
    ya, yl = R.ops.spmm_dual(m_rp, m_col, m_a, m_l, x.cuda())
    np.testing.assert_allclose(ya.cpu().double().numpy(), _dense(rp, col, val, n) @ x.double().numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(yl.cpu().double().numpy(), lap @ x.double().numpy(), atol=2e-5, rtol=1e-5)
