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
    "dense_gcn_operator", "dense_cheb_operator",
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


# ---- dense-matrix statements of the same operators (used for known-answer tests) --------------

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
