"""CPU restatement of the reference's model orchestration, op for op in its order.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parameters are passed as a plain
``dict[str, Tensor]`` keyed by the reference's own ``state_dict`` names, so a reference
checkpoint can be fed in unchanged.

Follows:
* TGCN cell .................. models/utils.py:163-203
* RegionalA3TGCN.forward ..... models/RegionalTemporalGCN.py:114-149
* RegionalTemporalGCN.forward  models/RegionalTemporalGCN.py:25-39
* A3TGCN.forward ............. models/TemporalGCN.py:75-91
* TemporalGCN.forward ........ models/TemporalGCN.py:21-32

This is the "eager-faithful" path: one Python iteration per period, one ChebConv call
per region per period, GCNConv as lin-then-propagate at width 256, normalisations
recomputed on every call (``cached=False`` in the reference).  It is what ``bench.py``
times as ``cpu_baseline`` (kind "port").
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from .graph_ops import cheb_conv, gat_conv, gcn_conv, sage_conv

Params = Dict[str, torch.Tensor]
HIDDEN = 256      # models/RegionalTemporalGCN.py:14, models/TemporalGCN.py:12
HEAD_HIDDEN = 128  # models/RegionalTemporalGCN.py:19


def tgcn_cell(p: Params, prefix: str, x, edge_index, edge_weight, h):
    """models/utils.py:190-203.  ``h`` is the incoming hidden state (never None on the hot path)."""
    def gate(name, hidden_in):
        conv = gcn_conv(x, edge_index, edge_weight,
                        p[f"{prefix}conv_{name}.lin.weight"], p[f"{prefix}conv_{name}.bias"])
        cat = torch.cat([conv, hidden_in], dim=1)
        return cat @ p[f"{prefix}linear_{name}.weight"].t() + p[f"{prefix}linear_{name}.bias"]

    z = torch.sigmoid(gate("z", h))                  # :168-172
    r = torch.sigmoid(gate("r", h))                  # :174-178
    h_tilde = torch.tanh(gate("h", h * r))           # :180-184
    return z * h + (1 - z) * h_tilde                 # :186-188


def regional_a3tgcn(p: Params, x, edge_index, region_index: Sequence[torch.Tensor],
                    region_weight: Sequence[torch.Tensor], edge_weight=None, prefix="tgnn."):
    """models/RegionalTemporalGCN.py:131-149; any number of regions (the reference hard-codes 5)."""
    periods = x.shape[2]
    probs = torch.softmax(p[f"{prefix}_attention"], dim=0)
    w0, w1, cb = (p[f"{prefix}conv.lins.0.weight"], p[f"{prefix}conv.lins.1.weight"], p[f"{prefix}conv.bias"])
    acc = 0
    for t in range(periods):
        xt = x[:, :, t]
        per_region = [cheb_conv(xt, ei, ew, w0, w1, cb) for ei, ew in zip(region_index, region_weight)]
        h = torch.cat(per_region, dim=1) @ p[f"{prefix}linear.weight"].t() + p[f"{prefix}linear.bias"]
        h = F.leaky_relu(h)
        acc = acc + probs[t] * tgcn_cell(p, f"{prefix}_base_tgcn.", xt, edge_index, edge_weight, h)
    return acc


def a3tgcn(p: Params, x, edge_index, edge_weight, prefix="tgnn."):
    """models/TemporalGCN.py:82-91: ChebConv on the full weighted graph gives the cell's hidden input."""
    periods = x.shape[2]
    probs = torch.softmax(p[f"{prefix}_attention"], dim=0)
    w0, w1, cb = (p[f"{prefix}conv.lins.0.weight"], p[f"{prefix}conv.lins.1.weight"], p[f"{prefix}conv.bias"])
    acc = 0
    for t in range(periods):
        xt = x[:, :, t]
        h = cheb_conv(xt, edge_index, edge_weight, w0, w1, cb)
        acc = acc + probs[t] * tgcn_cell(p, f"{prefix}_base_tgcn.", xt, edge_index, edge_weight, h)
    return acc


def tgcn_cell_block(p: Params, prefix: str, x, edge_index, h, block: str):
    """models/utils.py:163-203 with base block 'graphsage' or 'gat' (:93-100); ``h`` None = zeros (:163-166)."""
    if h is None:
        h = torch.zeros(x.shape[0], p[f"{prefix}linear_z.bias"].numel(), dtype=x.dtype)

    def conv(name):
        q = f"{prefix}conv_{name}."
        if block == "graphsage":
            return sage_conv(x, edge_index, p[q + "lin_l.weight"], p[q + "lin_l.bias"], p[q + "lin_r.weight"])
        return gat_conv(x, edge_index, p[q + "lin.weight"], p[q + "att_src"], p[q + "att_dst"], p[q + "bias"])

    def gate(name, hidden_in):
        cat = torch.cat([conv(name), hidden_in], dim=1)
        return cat @ p[f"{prefix}linear_{name}.weight"].t() + p[f"{prefix}linear_{name}.bias"]

    z = torch.sigmoid(gate("z", h))
    r = torch.sigmoid(gate("r", h))
    h_tilde = torch.tanh(gate("h", h * r))
    return z * h + (1 - z) * h_tilde


def zero_hidden_a3tgcn(p: Params, x, edge_index, block: str, prefix: str):
    """models/GraphSAGETemporalGCN.py:88-96 / models/GATTemporal.py:73-82: ``self._base_tgcn(X[:, :, period], edge_index, H)``
    hands H (None) to TGCN.forward's ``edge_weight`` parameter, so the cell runs on a zero hidden state in every period."""
    periods = x.shape[2]
    probs = torch.softmax(p[f"{prefix}_attention"], dim=0)
    acc = 0
    for t in range(periods):
        acc = acc + probs[t] * tgcn_cell_block(p, f"{prefix}_base_tgcn.", x[:, :, t], edge_index, None, block)
    return acc


def graphsage_temporal_gcn(p: Params, x, edge_index, edge_attr=None):
    """GraphSAGETemporalGCN.forward (models/GraphSAGETemporalGCN.py:24-43) -> (prediction, hidden)."""
    hidden = zero_hidden_a3tgcn(p, x, edge_index, "graphsage", "tgnn.")
    return head(p, hidden), hidden


def gat_temporal(p: Params, x, edge_index, edge_attr=None):
    """GATTemporal.forward (models/GATTemporal.py:21-34) -> (prediction, hidden)."""
    hidden = zero_hidden_a3tgcn(p, x, edge_index, "gat", "gat.")
    return head(p, hidden), hidden


CONVSTACK_HIDDEN = 512   # models/ConvStackedTemporalGCN.py:13
CONVSTACK_HEAD_HIDDEN = 256   # models/ConvStackedTemporalGCN.py:16


def conv_stacked_a3tgcn(p: Params, x, edge_index, edge_weight, prefix="tgnn."):
    """models/ConvStackedTemporalGCN.py:115-126: five stacked GCNConv (no activation in between) on the weighted
    full graph give the cell's hidden input; ``tgnn.linear`` (512*5 -> 512, :100) is never called."""
    periods = x.shape[2]
    probs = torch.softmax(p[f"{prefix}_attention"], dim=0)
    acc = 0
    for t in range(periods):
        xt = x[:, :, t]
        h = xt
        for layer in range(1, 6):
            h = gcn_conv(h, edge_index, edge_weight, p[f"{prefix}conv{layer}.lin.weight"], p[f"{prefix}conv{layer}.bias"])
        acc = acc + probs[t] * tgcn_cell(p, f"{prefix}_base_tgcn.", xt, edge_index, edge_weight, h)
    return acc


def conv_stacked_temporal_gcn(p: Params, x, edge_index, edge_attr):
    """ConvStackedTemporalGCN.forward -> (prediction (N,O), hidden (N,512))."""
    hidden = conv_stacked_a3tgcn(p, x, edge_index, edge_attr)
    return head(p, hidden), hidden


def head(p: Params, hidden):
    """relu -> linear1 -> relu -> linear2 (models/RegionalTemporalGCN.py:35-38)."""
    y = torch.relu(hidden) @ p["linear1.weight"].t() + p["linear1.bias"]
    return torch.relu(y) @ p["linear2.weight"].t() + p["linear2.bias"]


def regional_temporal_gcn(p: Params, x, edge_index, region_index, region_weight):
    """RegionalTemporalGCN.forward -> (prediction (N,O), hidden (N,256))."""
    hidden = regional_a3tgcn(p, x, edge_index, region_index, region_weight)
    return head(p, hidden), hidden


def temporal_gcn(p: Params, x, edge_index, edge_attr):
    """TemporalGCN.forward -> (prediction (N,O), hidden (N,256))."""
    hidden = a3tgcn(p, x, edge_index, edge_attr)
    return head(p, hidden), hidden


# ---- parameter construction (names/shapes = the shipped checkpoints; init is distributional) ----

def _glorot(gen, out_f, in_f, dtype):
    a = math.sqrt(6.0 / (in_f + out_f))
    return (torch.rand(out_f, in_f, generator=gen, dtype=dtype) * 2 - 1) * a


def _linear_init(gen, out_f, in_f, dtype):
    b = 1.0 / math.sqrt(in_f)
    return ((torch.rand(out_f, in_f, generator=gen, dtype=dtype) * 2 - 1) * b,
            (torch.rand(out_f, generator=gen, dtype=dtype) * 2 - 1) * b)


def init_params(model: str, node_features: int, periods: int, output_dim: int, num_nodes: Optional[int] = None,
                num_regions: int = 5, seed: int = 0, dtype=torch.float32, hidden: int = HIDDEN,
                bias_scale: float = 0.1) -> Params:
    """Seeded parameters with the reference's state_dict names and shapes.

    PyG zero-initialises its conv biases; here they get small random values
    (``bias_scale``) so that parity tests exercise the bias paths too.
    """
    g = torch.Generator().manual_seed(seed)
    C = hidden
    p: Params = {}
    p["tgnn._attention"] = torch.rand(periods, generator=g, dtype=dtype)
    if model == "ConvStackedTemporalGCN":
        C = CONVSTACK_HIDDEN if hidden == HIDDEN else hidden
        for k in "zrh":
            p[f"tgnn._base_tgcn.conv_{k}.bias"] = torch.randn(C, generator=g, dtype=dtype) * bias_scale
            p[f"tgnn._base_tgcn.conv_{k}.lin.weight"] = _glorot(g, C, node_features, dtype)
            w, b = _linear_init(g, C, 2 * C, dtype)
            p[f"tgnn._base_tgcn.linear_{k}.weight"], p[f"tgnn._base_tgcn.linear_{k}.bias"] = w, b
        for layer in range(1, 6):
            p[f"tgnn.conv{layer}.bias"] = torch.randn(C, generator=g, dtype=dtype) * bias_scale
            p[f"tgnn.conv{layer}.lin.weight"] = _glorot(g, C, node_features if layer == 1 else C, dtype)
        p["tgnn.linear.weight"], p["tgnn.linear.bias"] = _linear_init(g, C, 5 * C, dtype)     # dead layer, :100
        hh = CONVSTACK_HEAD_HIDDEN if hidden == HIDDEN else HEAD_HIDDEN
        p["linear1.weight"], p["linear1.bias"] = _linear_init(g, hh, C, dtype)
        p["linear2.weight"], p["linear2.bias"] = _linear_init(g, output_dim, hh, dtype)
        return p
    if model in ("GraphSAGETemporalGCN", "GATTemporal"):
        pre = "tgnn." if model == "GraphSAGETemporalGCN" else "gat."
        p = {f"{pre}_attention": p["tgnn._attention"]}
        if model == "GraphSAGETemporalGCN":
            assert num_nodes is not None
            p[f"{pre}_weight_att1"] = torch.randn(C, 1, generator=g, dtype=dtype) * 0.1
            p[f"{pre}_weight_att2"] = torch.randn(num_nodes, 1, generator=g, dtype=dtype) * 0.1
            p[f"{pre}_bias_att1"] = torch.randn(1, 1, generator=g, dtype=dtype)
            p[f"{pre}_bias_att2"] = torch.randn(1, 1, generator=g, dtype=dtype)
        for k in "zrh":
            q = f"{pre}_base_tgcn.conv_{k}."
            if model == "GraphSAGETemporalGCN":
                p[q + "lin_l.weight"] = _glorot(g, C, node_features, dtype)
                p[q + "lin_l.bias"] = torch.randn(C, generator=g, dtype=dtype) * bias_scale
                p[q + "lin_r.weight"] = _glorot(g, C, node_features, dtype)
            else:
                a = math.sqrt(6.0 / (1 + C))
                p[q + "att_src"] = (torch.rand(1, 1, C, generator=g, dtype=dtype) * 2 - 1) * a
                p[q + "att_dst"] = (torch.rand(1, 1, C, generator=g, dtype=dtype) * 2 - 1) * a
                p[q + "bias"] = torch.randn(C, generator=g, dtype=dtype) * bias_scale
                p[q + "lin.weight"] = _glorot(g, C, node_features, dtype)
            w, b = _linear_init(g, C, 2 * C, dtype)
            p[f"{pre}_base_tgcn.linear_{k}.weight"], p[f"{pre}_base_tgcn.linear_{k}.bias"] = w, b
        if model == "GraphSAGETemporalGCN":        # never-called GCNConv, models/GraphSAGETemporalGCN.py:61-64
            p[f"{pre}conv.bias"] = torch.randn(C, generator=g, dtype=dtype) * bias_scale
            p[f"{pre}conv.lin.weight"] = _glorot(g, C, node_features, dtype)
        p["linear1.weight"], p["linear1.bias"] = _linear_init(g, HEAD_HIDDEN, C, dtype)
        p["linear2.weight"], p["linear2.bias"] = _linear_init(g, output_dim, HEAD_HIDDEN, dtype)
        return p
    if model == "RegionalTemporalGCN":
        assert num_nodes is not None
        p["tgnn._weight_att1"] = torch.randn(C, 1, generator=g, dtype=dtype) * 0.1
        p["tgnn._weight_att2"] = torch.randn(num_nodes, 1, generator=g, dtype=dtype) * 0.1
        p["tgnn._bias_att1"] = torch.randn(1, 1, generator=g, dtype=dtype)
        p["tgnn._bias_att2"] = torch.randn(1, 1, generator=g, dtype=dtype)
    for k in "zrh":
        p[f"tgnn._base_tgcn.conv_{k}.bias"] = torch.randn(C, generator=g, dtype=dtype) * bias_scale
        p[f"tgnn._base_tgcn.conv_{k}.lin.weight"] = _glorot(g, C, node_features, dtype)
        w, b = _linear_init(g, C, 2 * C, dtype)
        p[f"tgnn._base_tgcn.linear_{k}.weight"], p[f"tgnn._base_tgcn.linear_{k}.bias"] = w, b
    p["tgnn.conv.bias"] = torch.randn(C, generator=g, dtype=dtype) * bias_scale
    p["tgnn.conv.lins.0.weight"] = _glorot(g, C, node_features, dtype)
    p["tgnn.conv.lins.1.weight"] = _glorot(g, C, node_features, dtype)
    if model == "RegionalTemporalGCN":
        p["tgnn.linear.weight"], p["tgnn.linear.bias"] = _linear_init(g, C, num_regions * C, dtype)
    elif model == "TemporalGCN":
        p["tgnn.linear.weight"], p["tgnn.linear.bias"] = _linear_init(g, C, 64, dtype)  # dead layer, TemporalGCN.py:70
    else:
        raise ValueError(model)
    p["linear1.weight"], p["linear1.bias"] = _linear_init(g, HEAD_HIDDEN, C, dtype)
    p["linear2.weight"], p["linear2.bias"] = _linear_init(g, output_dim, HEAD_HIDDEN, dtype)
    return p


# parameters that never receive a gradient in the reference (dead code paths):
UNUSED_PARAMS = ("tgnn._weight_att1", "tgnn._weight_att2", "tgnn._bias_att1", "tgnn._bias_att2")
UNUSED_PARAMS_TEMPORAL = ("tgnn.linear.weight", "tgnn.linear.bias")
UNUSED_PARAMS_CONVSTACK = ("tgnn.linear.weight", "tgnn.linear.bias")
UNUSED_PARAMS_SAGE = UNUSED_PARAMS + ("tgnn.conv.bias", "tgnn.conv.lin.weight")
