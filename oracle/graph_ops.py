"""CPU restatement of the two PyG graph operators on the RegT-GCN hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  torch_geometric is not vendored in the
reference and not installed here; the functions below restate its *published* behaviour
(PyG 2.3-2.6) at the reference's call sites:

* GCNConv   -- models/utils.py:107-113 (ctor), :169/:175/:181 (calls)
* ChebConv  -- models/RegionalTemporalGCN.py:77-80 (ctor), :136-140 (calls);
               models/TemporalGCN.py:65-69, :88

Conventions: ``edge_index[0]`` is the message *source* (``row``), ``edge_index[1]`` the
*target* (``col``); messages are summed at the target.  Everything is written as plain
functions over tensors so that the same code serves fp32 (the reference dtype) and fp64
(tight checks).
"""
from __future__ import annotations

import torch

__all__ = [
    "gcn_norm_edges", "cheb_norm_edges", "propagate", "gcn_conv", "cheb_conv",
    "dense_gcn_operator", "dense_cheb_operator", "sage_conv", "gat_conv", "dense_mean_operator", "dense_gat_attention",
]


def _inv_sqrt_zero_inf(deg: torch.Tensor) -> torch.Tensor:
    dis = deg.pow(-0.5)
    return torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)


def gcn_norm_edges(edge_index: torch.Tensor, edge_weight, num_nodes: int, dtype=torch.float32):
    """PyG ``gcn_norm(add_self_loops=True, improved=False)``.

    1. unit weights when ``edge_weight`` is None;
    2. ``add_remaining_self_loops``: existing self-loop edges are dropped and every node
       gets exactly one loop -- weight 1.0, or the weight of the node's own (last listed)
       pre-existing loop;
    3. degree = sum of weights arriving at each *target*;
    4. ``w' = deg^-1/2[src] * w * deg^-1/2[dst]`` with ``inf -> 0``.
    Returns ``(src, dst, w')`` with the N loop edges appended after the kept edges.
    """
    src, dst = edge_index[0].long(), edge_index[1].long()
    w = torch.ones(src.numel(), dtype=dtype) if edge_weight is None else edge_weight.to(dtype)
    keep = src != dst
    loop_w = torch.ones(num_nodes, dtype=dtype)
    if (~keep).any():
        # sequential assignment: the last listed loop of a node wins
        for e in torch.nonzero(~keep).flatten().tolist():
            loop_w[src[e]] = w[e]
    ar = torch.arange(num_nodes, dtype=torch.long)
    src2 = torch.cat([src[keep], ar])
    dst2 = torch.cat([dst[keep], ar])
    w2 = torch.cat([w[keep], loop_w])
    deg = torch.zeros(num_nodes, dtype=dtype).index_add_(0, dst2, w2)
    dis = _inv_sqrt_zero_inf(deg)
    return src2, dst2, dis[src2] * w2 * dis[dst2]


def cheb_norm_edges(edge_index: torch.Tensor, edge_weight, num_nodes: int, dtype=torch.float32):
    """PyG ``ChebConv.__norm__(normalization='sym', lambda_max=None)``.

    ``get_laplacian``: drop self loops, unit weights if None, degree = sum of weights
    leaving each *source*, ``-deg^-1/2[src] w deg^-1/2[dst]`` plus N loops of +1;
    then ``lambda_max = 2 * max(weight)``, ``w <- 2 w / lambda_max`` (``inf -> 0``) and
    ``-1`` on the loops.  Returns ``(src, dst, w~)`` including the (zero-weight) loops so
    the restatement stays literal.
    """
    src, dst = edge_index[0].long(), edge_index[1].long()
    w = torch.ones(src.numel(), dtype=dtype) if edge_weight is None else edge_weight.to(dtype)
    keep = src != dst
    src, dst, w = src[keep], dst[keep], w[keep]
    deg = torch.zeros(num_nodes, dtype=dtype).index_add_(0, src, w)
    dis = _inv_sqrt_zero_inf(deg)
    lap = -(dis[src] * w * dis[dst])
    ar = torch.arange(num_nodes, dtype=torch.long)
    src2 = torch.cat([src, ar])
    dst2 = torch.cat([dst, ar])
    w2 = torch.cat([lap, torch.ones(num_nodes, dtype=dtype)])
    lam = 2.0 * w2.max()
    w2 = (2.0 * w2) / lam
    w2 = torch.where(torch.isinf(w2), torch.zeros_like(w2), w2)
    is_loop = src2 == dst2
    w2 = w2 - is_loop.to(dtype)
    return src2, dst2, w2


def propagate(src, dst, w, x: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """``out[dst_e] += w_e * x[src_e]`` -- gather, scale, scatter-add (MessagePassing, aggr='add')."""
    msg = w.view(-1, 1) * x.index_select(0, src)
    return torch.zeros(num_nodes, x.shape[1], dtype=x.dtype).index_add_(0, dst, msg)


def gcn_conv(x, edge_index, edge_weight, lin_weight, bias):
    """GCNConv.forward: normalise, ``x @ W^T``, propagate at width out_channels, add bias."""
    n = x.shape[0]
    src, dst, w = gcn_norm_edges(edge_index, edge_weight, n, x.dtype)
    return propagate(src, dst, w, x @ lin_weight.t(), n) + bias


def cheb_conv(x, edge_index, edge_weight, w0, w1, bias):
    """ChebConv(K=2).forward: ``x W0^T + (L~ x) W1^T + b`` with the propagate at width in_channels."""
    n = x.shape[0]
    src, dst, w = cheb_norm_edges(edge_index, edge_weight, n, x.dtype)
    tx1 = propagate(src, dst, w, x, n)
    return x @ w0.t() + tx1 @ w1.t() + bias


def sage_conv(x, edge_index, lin_l_weight, lin_l_bias, lin_r_weight):
    """PyG ``SAGEConv(aggr='mean', root_weight=True, normalize=False, project=False)`` -- base block 'graphsage' of the
    reference's TGCN cell (models/utils.py:99-100; called as ``conv(X, edge_index, None)``, the third positional being
    SAGEConv's ``size``).  ``out_i = lin_l(mean_{j -> i} x_j) + lin_r(x_i)``: the mean runs over every listed in-edge (self
    loops and duplicates as listed; ``scatter(..., reduce='mean')`` = sum / max(count, 1), so isolated targets get 0);
    ``lin_l`` carries the bias, ``lin_r`` has none."""
    n = x.shape[0]
    src, dst = edge_index[0].long(), edge_index[1].long()
    s = torch.zeros(n, x.shape[1], dtype=x.dtype).index_add_(0, dst, x.index_select(0, src))
    cnt = torch.zeros(n, dtype=x.dtype).index_add_(0, dst, torch.ones(src.numel(), dtype=x.dtype))
    mean = s / cnt.clamp(min=1).view(-1, 1)
    return mean @ lin_l_weight.t() + lin_l_bias + x @ lin_r_weight.t()


def gat_conv(x, edge_index, lin_weight, att_src, att_dst, bias, negative_slope: float = 0.2):
    """PyG ``GATConv(heads=1, concat=True, negative_slope=0.2, dropout=0, add_self_loops=True, edge_dim=None, bias=True)`` --
    base block 'gat' of the reference's TGCN cell (models/utils.py:97-98; called as ``conv(X, edge_index, None)``, the third
    positional being GATConv's ``edge_attr``).

    ``x' = x W^T``; ``a_s = <x', att_src>``, ``a_d = <x', att_dst>`` per node; self loops are removed and one loop per node is
    added; ``e_ij = leaky_relu(a_s[j] + a_d[i])`` for every edge j -> i; ``alpha = softmax`` over the in-edges of i
    (``exp(e - max) / (sum + 1e-16)``, torch_geometric.utils.softmax); ``out_i = sum_j alpha_ij x'_j + bias``."""
    n = x.shape[0]
    xs = x @ lin_weight.t()
    a_s = (xs * att_src.reshape(1, -1)).sum(-1)
    a_d = (xs * att_dst.reshape(1, -1)).sum(-1)
    src, dst = edge_index[0].long(), edge_index[1].long()
    keep = src != dst
    ar = torch.arange(n, dtype=torch.long)
    src, dst = torch.cat([src[keep], ar]), torch.cat([dst[keep], ar])
    e = torch.nn.functional.leaky_relu(a_s[src] + a_d[dst], negative_slope)
    emax = torch.full((n,), float("-inf"), dtype=x.dtype).scatter_reduce(0, dst, e.detach(), "amax", include_self=True)
    ex = (e - emax[dst]).exp()
    den = torch.zeros(n, dtype=x.dtype).index_add_(0, dst, ex) + 1e-16
    alpha = ex / den[dst]
    out = torch.zeros(n, xs.shape[1], dtype=x.dtype).index_add_(0, dst, alpha.view(-1, 1) * xs.index_select(0, src))
    return out + bias


# ---- dense-matrix statements of the same operators (used for known-answer tests) --------------

def dense_mean_operator(edge_index, num_nodes, dtype=torch.float64):
    """Row-normalised in-adjacency: ``A[i, j]`` = (number of listed edges j -> i) / (number of listed edges into i)."""
    a = torch.zeros(num_nodes, num_nodes, dtype=dtype)
    for k in range(edge_index.shape[1]):
        a[int(edge_index[1, k]), int(edge_index[0, k])] += 1.0
    deg = a.sum(dim=1)
    return a / torch.where(deg > 0, deg, torch.ones_like(deg)).view(-1, 1)


def dense_gat_attention(x, edge_index, lin_weight, att_src, att_dst, negative_slope=0.2):
    """alpha as a dense (N, N) matrix straight from the GAT paper's formula over the edge multiset (A minus loops) + I."""
    n = x.shape[0]
    xs = x @ lin_weight.t()
    cnt = torch.zeros(n, n, dtype=x.dtype)
    for k in range(edge_index.shape[1]):
        s, d = int(edge_index[0, k]), int(edge_index[1, k])
        if s != d:
            cnt[d, s] += 1.0
    cnt = cnt + torch.eye(n, dtype=x.dtype)
    score = torch.nn.functional.leaky_relu((xs @ att_dst.reshape(-1)).view(-1, 1) + (xs @ att_src.reshape(-1)).view(1, -1), negative_slope)
    w = cnt * torch.exp(score - score.max())
    return w / w.sum(dim=1, keepdim=True)


def dense_gcn_operator(edge_index, edge_weight, num_nodes, dtype=torch.float64):
    """A_hat = D^-1/2 (A + I) D^-1/2 built as a dense matrix straight from the formula;
    ``A[i, j]`` = total weight of edges j -> i, loops replaced as in ``gcn_norm_edges``."""
    a = torch.zeros(num_nodes, num_nodes, dtype=dtype)
    loop = torch.ones(num_nodes, dtype=dtype)
    e = edge_index.shape[1]
    for k in range(e):
        s, d = int(edge_index[0, k]), int(edge_index[1, k])
        wk = 1.0 if edge_weight is None else float(edge_weight[k])
        if s == d:
            loop[s] = wk
        else:
            a[d, s] += wk
    a = a + torch.diag(loop)
    deg = a.sum(dim=1)
    dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
    return dis.view(-1, 1) * a * dis.view(1, -1)


def dense_cheb_operator(edge_index, edge_weight, num_nodes, dtype=torch.float64):
    """L~ = -D^-1/2 A D^-1/2 (zero diagonal) with D the weighted *out*-degree."""
    a = torch.zeros(num_nodes, num_nodes, dtype=dtype)
    e = edge_index.shape[1]
    for k in range(e):
        s, d = int(edge_index[0, k]), int(edge_index[1, k])
        if s != d:
            a[d, s] += 1.0 if edge_weight is None else float(edge_weight[k])
    deg = a.sum(dim=0)  # column sums = weight leaving each source
    dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
    return -(dis.view(-1, 1) * a * dis.view(1, -1))
