"""Torch statement of the *fused formulation* the HIP pipeline implements.

Test helper only: it mirrors, stage by stage and with hand-derived backward formulas, what the
HIP kernels compute, so a failing GPU parity test can be bisected per stage.  It is checked
against the oracle in tests/test_fused_math.py (CPU) -- that test is what validates the algebra
(aggregate-first, composed weights, no sparse backward).
"""
from __future__ import annotations

import torch

from oracle import graph_ops as G

LRELU = 0.01


def dense_ops(edge_index, edge_weight, region_index, region_weight, n, dtype):
    """A_hat (N,N) and the list of L~_r (N,N) as dense matrices."""
    src, dst, w = G.gcn_norm_edges(edge_index, edge_weight, n, dtype)
    a = torch.zeros(n, n, dtype=dtype).index_put_((dst, src), w, accumulate=True)
    ls = []
    for ei, ew in zip(region_index, region_weight):
        s, d, wl = G.cheb_norm_edges(ei, ew, n, dtype)
        ls.append(torch.zeros(n, n, dtype=dtype).index_put_((d, s), wl, accumulate=True))
    return a, ls


def compose(p, num_regions, regional=True, prefix="tgnn."):
    """Composed weights: everything that multiplies an input of width F is folded to (C,F)."""
    C = p[f"{prefix}conv.bias"].numel()
    w0, w1, bc = p[f"{prefix}conv.lins.0.weight"], p[f"{prefix}conv.lins.1.weight"], p[f"{prefix}conv.bias"]
    out = {}
    if regional:
        wl = p[f"{prefix}linear.weight"]
        blocks = [wl[:, r * C:(r + 1) * C] for r in range(num_regions)]
        wsum = sum(blocks)
        out["A0"] = wsum @ w0
        out["Ar"] = [b @ w1 for b in blocks]
        out["b"] = wsum @ bc + p[f"{prefix}linear.bias"]
    else:
        out["A0"], out["Ar"], out["b"] = w0, [w1], bc
    for k in "zrh":
        u = p[f"{prefix}_base_tgcn.linear_{k}.weight"]
        u1, u2 = u[:, :C], u[:, C:]
        out[f"G{k}"] = u1 @ p[f"{prefix}_base_tgcn.conv_{k}.lin.weight"]
        out[f"c{k}"] = u1 @ p[f"{prefix}_base_tgcn.conv_{k}.bias"] + p[f"{prefix}_base_tgcn.linear_{k}.bias"]
        out[f"U{k}"] = u2
    return out


def bf16_round(t):
    """Round-to-nearest-even to bf16 and back: what REGT_GEMM_MODE=bf16 does to every operand of a big GEMM."""
    return t.to(torch.bfloat16).to(torch.float32)


def forward_fused(p, x, a_hat, l_list, regional=True, rnd=None, store=None, round_x=False):
    """x (N,F,T).  Returns (pred, hidden) through the composed-weight formulation using autograd.

    ``rnd``: optional rounding applied to both operands of every activation x weight contraction that the HIP path
    runs on the matrix cores (``bf16_round`` emulates REGT_GEMM_MODE=bf16; accumulation, SpMM, compositions, gate
    math and the skinny last head layer stay fp32, as in the kernels).  ``store``: rounding applied to the M x C
    activations the pipeline keeps in HBM between kernels (h, Z; ``bf16_round`` when that mode stores them as bf16):
    whatever reads them later -- the GRU blend here -- sees the stored value.  ``round_x``: the snapshot itself is rounded
    before it is aggregated (bf16 rows of x / A_hat x / L~ x, the cfg-5 layout of SURVEY 8(d): where the fused forward kernel
    applies the packing kernel rounds x once and the aggregation reads and writes bf16 rows)."""
    q = (lambda v: v) if rnd is None else rnd
    st = (lambda v: v) if store is None else store
    n, f, t = x.shape
    R = len(l_list)
    w = compose(p, R, regional)
    xp = x.permute(0, 2, 1)                              # (N,T,F) packed rows
    if round_x:
        xp = q(xp)
    ax = torch.einsum("ij,jtf->itf", a_hat, xp)
    pre = q(xp) @ q(w["A0"]).t() + w["b"]
    for r in range(R):
        pre = pre + q(torch.einsum("ij,jtf->itf", l_list[r], xp)) @ q(w["Ar"][r]).t()
    h = st(torch.nn.functional.leaky_relu(pre, LRELU) if regional else pre)
    z = torch.sigmoid(q(h) @ q(w["Uz"]).t() + q(ax) @ q(w["Gz"]).t() + w["cz"])
    r_ = torch.sigmoid(q(h) @ q(w["Ur"]).t() + q(ax) @ q(w["Gr"]).t() + w["cr"])
    ht = torch.tanh(q(h * r_) @ q(w["Uh"]).t() + q(ax) @ q(w["Gh"]).t() + w["ch"])
    hn = st(z) * h + (1 - st(z)) * ht
    probs = torch.softmax(p["tgnn._attention"], dim=0)
    hidden = (hn * probs.view(1, t, 1)).sum(dim=1)
    y = q(torch.relu(hidden)) @ q(p["linear1.weight"]).t() + p["linear1.bias"]
    y = torch.relu(y) @ p["linear2.weight"].t() + p["linear2.bias"]
    return y, hidden
