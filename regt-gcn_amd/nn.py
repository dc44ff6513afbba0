"""Drop-in ``nn.Module`` surface of the RegT-GCN hot path.

Same constructor / ``forward()`` signatures, parameter names and ``state_dict`` layout as the
reference classes, so the reference's ``run.py`` / ``predict.py`` and its shipped checkpoints work
unchanged:

* :class:`RegionalTemporalGCN`  <-> models/RegionalTemporalGCN.py:9-39  (+ RegionalA3TGCN :42-149)
* :class:`TemporalGCN`          <-> models/TemporalGCN.py:7-32          (+ A3TGCN :35-91)
* :class:`ConvStackedTemporalGCN` <-> models/ConvStackedTemporalGCN.py:8-33 (+ ConvStackedA3TGCN :35-126)
* :class:`TGCN`                 <-> models/utils.py:69-203 (parameter container of the GRU cell)

The modules only *hold* parameters; all arithmetic runs in libregtgcn_hip.so through
:class:`regt-gcn_amd.functional.RegTGCNFunction`.  There is no CPU implementation here: calling
``forward`` with CPU tensors raises.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from . import ops
from .functional import (HEAD_HIDDEN, PARAM_NAMES_CELL, AggregateFunction, Cell0Function, CellFunction, GatAggregateFunction,
                         LinearFunction, RegTGCNFunction, ZeroGradAnchor, param_names)
from .graph import (AttentionPattern, GcnOperator, MeanOperator, PreparedGraph, fingerprint, prepare_attention_pattern,
                    prepare_gcn_operator, prepare_graph, prepare_mean_operator)

HIDDEN = 256          # out_channels=256, models/RegionalTemporalGCN.py:14 / models/TemporalGCN.py:12
LEAKY_SLOPE = 0.01    # F.leaky_relu default, models/RegionalTemporalGCN.py:143


class _PygLinear(nn.Module):
    """Linear parameter holder named like torch_geometric's Linear (``.weight`` (out,in) glorot; optional zero ``.bias``)."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        a = math.sqrt(6.0 / (in_channels + out_channels))
        nn.init.uniform_(self.weight, -a, a)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))


class _GCNConvParams(nn.Module):
    """Parameters of a PyG GCNConv: ``bias`` (C,) zero-init and ``lin.weight`` (C,F)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.lin = _PygLinear(in_channels, out_channels)


class _ChebConvParams(nn.Module):
    """Parameters of a PyG ChebConv(K=2): ``bias`` and ``lins.{0,1}.weight``."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.lins = nn.ModuleList([_PygLinear(in_channels, out_channels) for _ in range(2)])


class _SAGEConvParams(nn.Module):
    """Parameters of a PyG SAGEConv(aggr='mean', root_weight=True): ``lin_l.weight/.bias`` (neighbour mean) and ``lin_r.weight``."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.lin_l = _PygLinear(in_channels, out_channels, bias=True)
        self.lin_r = _PygLinear(in_channels, out_channels)


class _GATConvParams(nn.Module):
    """Parameters of a PyG GATConv(heads=1): ``att_src``, ``att_dst`` (1,1,C), ``bias`` (C), ``lin.weight`` (C,F) -- the layout
    of PyG >= 2.5, where an int ``in_channels`` shares one projection ``lin`` (2.0-2.4 call it lin_src / lin_dst)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        a = math.sqrt(6.0 / (1 + out_channels))
        self.att_src = nn.Parameter(torch.empty(1, 1, out_channels).uniform_(-a, a))
        self.att_dst = nn.Parameter(torch.empty(1, 1, out_channels).uniform_(-a, a))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.lin = _PygLinear(in_channels, out_channels)


class TGCN(nn.Module):
    """Parameter layout of the reference's T-GCN GRU cell (models/utils.py:75-161) for its three base blocks: 'gcn'
    (GCNConv, the hot path), 'graphsage' (SAGEConv) and 'gat' (GATConv) (:93-100)."""

    def __init__(self, in_channels: int, out_channels: int, baseblock: str = "gcn", improved: bool = False,
                 cached: bool = False, add_self_loops: bool = True):
        super().__init__()
        if improved or not add_self_loops:
            raise NotImplementedError("only improved=False, add_self_loops=True is on the hot path")
        blocks = {"gcn": _GCNConvParams, "graphsage": _SAGEConvParams, "gat": _GATConvParams}
        if baseblock not in blocks:
            raise NotImplementedError("Current baseblock %s is not supported." % (baseblock))     # models/utils.py:102
        conv = blocks[baseblock]
        self.in_channels, self.out_channels, self.baseblock = in_channels, out_channels, baseblock
        self.conv_z = conv(in_channels, out_channels)
        self.linear_z = nn.Linear(2 * out_channels, out_channels)
        self.conv_r = conv(in_channels, out_channels)
        self.linear_r = nn.Linear(2 * out_channels, out_channels)
        self.conv_h = conv(in_channels, out_channels)
        self.linear_h = nn.Linear(2 * out_channels, out_channels)


class _GraphCache:
    """Prepared graphs keyed first by tensor identity (free), then by content fingerprint (one 8-byte readback).

    An identity entry keeps strong references to its key tensors: while the entry lives their storage cannot be freed and
    handed to a new tensor with other edges at the same address, so (data_ptr, shape, _version) really identifies the
    content (an in-place edit bumps ``_version``).  The identity table is small (a loader that builds a fresh ``edge_index``
    per snapshot, run.py:172, goes through the fingerprint every time) so that it pins at most a few edge lists."""

    MAX_IDENT = 8

    def __init__(self):
        self._by_id: Dict[tuple, tuple] = {}            # key -> (graph, key tensors)
        self._by_hash: Dict[tuple, PreparedGraph] = {}

    @staticmethod
    def _ident(tensors):
        return tuple((t.data_ptr(), tuple(t.shape), t._version) if t is not None else None for t in tensors)

    def get(self, tensors: Sequence[Optional[torch.Tensor]], num_nodes: int, build):
        key = (num_nodes,) + self._ident(tensors)
        hit = self._by_id.get(key)
        if hit is not None and all(a is b for a, b in zip(hit[1], tensors)):
            return hit[0]
        hkey = (num_nodes, fingerprint(tensors))
        g = self._by_hash.get(hkey)
        if g is None:
            g = build()
            self._by_hash[hkey] = g
        if len(self._by_id) >= self.MAX_IDENT:
            self._by_id.clear()
        self._by_id[key] = (g, tuple(tensors))
        return g


def _need_cuda(x: torch.Tensor):
    if not x.is_cuda:
        raise _lib.RegtError("this package only runs on an MI355X through libregtgcn_hip.so; got a CPU tensor "
                             "(use oracle/ for CPU checks)")


class RegionalA3TGCN(nn.Module):
    """Parameter layout of models/RegionalTemporalGCN.py:42-88 (attention over periods + regional ChebConv + TGCN)."""

    def __init__(self, in_channels: int, out_channels: int, num_nodes: int, periods: int, improved: bool = False,
                 cached: bool = False, add_self_loops: bool = True, num_regions: int = 5):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_nodes, self.periods, self.num_regions = num_nodes, periods, num_regions
        self._attention = nn.Parameter(torch.empty(periods))
        # registered-but-unused parameters of the reference's dead attention() (:84-87, :91-111)
        self._weight_att1 = nn.Parameter(torch.normal(0.0, 0.1, size=(out_channels, 1)))
        self._weight_att2 = nn.Parameter(torch.normal(0.0, 0.1, size=(num_nodes, 1)))
        self._bias_att1 = nn.Parameter(torch.normal(0.0, 1.0, size=(1, 1)))
        self._bias_att2 = nn.Parameter(torch.normal(0.0, 1.0, size=(1, 1)))
        self._base_tgcn = TGCN(in_channels, out_channels, improved=improved, cached=cached, add_self_loops=add_self_loops)
        self.conv = _ChebConvParams(in_channels, out_channels)
        self.linear = nn.Linear(out_channels * num_regions, out_channels)
        nn.init.uniform_(self._attention)


class A3TGCN(nn.Module):
    """Parameter layout of models/TemporalGCN.py:35-73 (incl. its dead ``linear`` (64 -> C), :70)."""

    def __init__(self, in_channels: int, out_channels: int, periods: int, improved: bool = False, cached: bool = False,
                 add_self_loops: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.periods = in_channels, out_channels, periods
        self._base_tgcn = TGCN(in_channels, out_channels)
        self.conv = _ChebConvParams(in_channels, out_channels)
        self.linear = nn.Linear(64, out_channels)
        self._attention = nn.Parameter(torch.empty(periods))
        nn.init.uniform_(self._attention)


class _FusedModel(nn.Module):
    regional: bool

    def _params_in_order(self) -> List[torch.Tensor]:
        # walking named_parameters() costs ~70 us per call -- a fifth of a TPIMS-scale step's host time; the Parameter
        # objects are stable (load_state_dict / .to() / optimisers update them in place), so look them up once
        cached = self.__dict__.get("_ordered_params")
        if cached is None or any(p is not q for p, q in zip(cached[0], cached[1]())):
            named = dict(self.named_parameters())
            plist = [named[n] for n in param_names(self.regional)]
            owners = [self._owner_of(n) for n in param_names(self.regional)]
            getter = lambda: [m._parameters[k] for m, k in owners]      # noqa: E731  (catches a replaced Parameter object)
            cached = (plist, getter)
            self.__dict__["_ordered_params"] = cached
        return cached[0]

    def _owner_of(self, name: str):
        mod = self
        *path, leaf = name.split(".")
        for part in path:
            mod = mod._modules[part]
        return mod, leaf

    # GEMM arithmetic of THIS model's calls (regt_dims.arith): None = the process default (regt_set_gemm_mode / REGT_GEMM_MODE),
    # "fp32", "bf16x3" (exact split) or "bf16" (reduced precision, BASELINE configs[4]).  Per call, no process state: two models
    # of one process may differ.
    arithmetic = None
    call_flags = 0        # regt_dims.flags of this model's calls: _lib.DIMS_NO_BF16_ROWS | DIMS_NO_FUSED_BWD | DIMS_NO_SIDE_STREAM (A/B switches)

    def _run(self, x: torch.Tensor, graph: PreparedGraph, packed: bool = False):
        arith, flags = _lib.arith_code(self.arithmetic), int(self.call_flags)
        return RegTGCNFunction.apply(x, graph, self.regional, LEAKY_SLOPE, (packed, arith, flags) if (arith or flags) else packed,
                                     *self._params_in_order())

    def forward_packed(self, x_packed_ext: torch.Tensor, graph: PreparedGraph):
        """Region-sharded entry: ``x_packed_ext`` (x_rows, T, F) = own packed rows + gathered halo rows (dist.py)."""
        _need_cuda(x_packed_ext)
        return self._run(x_packed_ext, graph, packed=True)


class RegionalTemporalGCN(_FusedModel):
    """RegT-GCN.  ``forward`` keeps the reference's 12-positional-argument form (5 named regions,
    models/RegionalTemporalGCN.py:25-26) and also accepts any number R of regions as
    ``forward(x, edge_index, idx_1..idx_R, attr_1..attr_R)`` or ``forward(x, edge_index, [idx...], [attr...])``.
    Returns ``(prediction (N, output_dim), hidden (N, 256))``."""

    regional = True

    def __init__(self, node_features: int, num_nodes: int, periods: int, output_dim: int, num_regions: int = 5,
                 hidden_channels: int = HIDDEN):
        super().__init__()
        self.tgnn = RegionalA3TGCN(in_channels=node_features, out_channels=hidden_channels, num_nodes=num_nodes,
                                   periods=periods, num_regions=num_regions)
        self.output_dim = output_dim
        self.linear1 = nn.Linear(hidden_channels, HEAD_HIDDEN)
        self.linear2 = nn.Linear(HEAD_HIDDEN, output_dim)
        self.relu = nn.ReLU()
        self.num_nodes, self.num_regions = num_nodes, num_regions
        self._graphs = _GraphCache()

    def prepare_graph(self, edge_index, region_index: Sequence[torch.Tensor], region_attr: Sequence[torch.Tensor],
                      num_nodes: Optional[int] = None, copies: int = 1) -> PreparedGraph:
        """Normalise + sort the static graph once; pass the result to :meth:`forward_prepared`.  ``copies`` = B builds the
        block-diagonal graph of B snapshots (train.train_epoch_batched): forward_prepared then takes x of shape (B*N, F, T)."""
        n = self.num_nodes if num_nodes is None else num_nodes
        return prepare_graph(edge_index, None, list(region_index), list(region_attr), n, copies)

    def forward_prepared(self, x: torch.Tensor, graph: PreparedGraph):
        _need_cuda(x)
        return self._run(x, graph)

    # the reference's parameter names, in its order (models/RegionalTemporalGCN.py:25-26): accepted as keywords too
    REF_KEYWORDS = ("IAedge_index", "KSedge_index", "KYedge_index", "OHedge_index", "WIedge_index",
                    "IAedge_attr", "KSedge_attr", "KYedge_attr", "OHedge_attr", "WIedge_attr")

    def forward(self, x, edge_index, *regions, **named):
        if named:
            # keyword call with the reference's names (any prefix of the ten may still be positional, as in Python's own binding)
            unknown = [k for k in named if k not in self.REF_KEYWORDS]
            if unknown:
                raise TypeError(f"forward() got an unexpected keyword argument '{unknown[0]}'")
            if self.num_regions != 5:
                raise TypeError("the reference's keyword names cover its 5 regions; this model was built for "
                                f"{self.num_regions} -- pass the region tensors positionally or as two lists")
            if len(regions) > len(self.REF_KEYWORDS) or (len(regions) == 2 and isinstance(regions[0], (list, tuple))):
                raise TypeError("forward(): keyword region arguments cannot be mixed with the list form")
            bound = list(regions)
            for k in self.REF_KEYWORDS[len(regions):]:
                if k not in named:
                    raise TypeError(f"forward() missing 1 required positional argument: '{k}'")
                bound.append(named[k])
            for k in self.REF_KEYWORDS[:len(regions)]:
                if k in named:
                    raise TypeError(f"forward() got multiple values for argument '{k}'")
            regions = tuple(bound)
        _need_cuda(x)
        if len(regions) == 2 and isinstance(regions[0], (list, tuple)):
            idx, attr = list(regions[0]), list(regions[1])
        else:
            if len(regions) != 2 * self.num_regions:
                raise TypeError(f"forward() expects {self.num_regions} regional edge_index tensors followed by "
                                f"{self.num_regions} edge_attr tensors, got {len(regions)} tensors")
            idx, attr = list(regions[:self.num_regions]), list(regions[self.num_regions:])
        if len(idx) != self.num_regions:
            raise TypeError(f"model was built for {self.num_regions} regions, got {len(idx)}")
        n = x.shape[0]
        flat: List[Optional[torch.Tensor]] = [edge_index, None]
        for i, a in zip(idx, attr):
            flat += [i, a]
        graph = self._graphs.get(flat, n, lambda: prepare_graph(edge_index, None, idx, attr, n))
        return self._run(x, graph)


class TemporalGCN(_FusedModel):
    """A3T-GCN baseline sharing the cell.  ``forward(x, edge_index, edge_attr)`` (models/TemporalGCN.py:21)."""

    regional = False

    def __init__(self, node_features: int, periods: int, output_dim: int, hidden_channels: int = HIDDEN):
        super().__init__()
        self.tgnn = A3TGCN(in_channels=node_features, out_channels=hidden_channels, periods=periods)
        self.output_dim = output_dim
        self.linear1 = nn.Linear(hidden_channels, HEAD_HIDDEN)
        self.linear2 = nn.Linear(HEAD_HIDDEN, output_dim)
        self.relu = nn.ReLU()
        self._graphs = _GraphCache()

    def prepare_graph(self, edge_index, edge_attr, num_nodes: int, copies: int = 1) -> PreparedGraph:
        return prepare_graph(edge_index, edge_attr, [edge_index], [edge_attr], num_nodes, copies)

    def forward_prepared(self, x: torch.Tensor, graph: PreparedGraph):
        _need_cuda(x)
        return self._run(x, graph)

    def forward(self, x, edge_index, edge_attr):
        _need_cuda(x)
        n = x.shape[0]
        graph = self._graphs.get([edge_index, edge_attr], n, lambda: prepare_graph(edge_index, edge_attr, [edge_index], [edge_attr], n))
        return self._run(x, graph)


class ConvStackedA3TGCN(nn.Module):
    """Parameter layout of models/ConvStackedTemporalGCN.py:35-103: TGCN cell, five GCNConv, a never-called
    ``linear`` (512*5 -> 512, :100) and the attention over periods."""

    def __init__(self, in_channels: int, out_channels: int, periods: int, improved: bool = False, cached: bool = False,
                 add_self_loops: bool = True):
        super().__init__()
        if improved or not add_self_loops:
            raise NotImplementedError("only improved=False, add_self_loops=True is on the hot path")
        self.in_channels, self.out_channels, self.periods = in_channels, out_channels, periods
        self._base_tgcn = TGCN(in_channels=in_channels, out_channels=out_channels)
        self.conv1 = _GCNConvParams(in_channels, out_channels)
        self.conv2 = _GCNConvParams(out_channels, out_channels)
        self.conv3 = _GCNConvParams(out_channels, out_channels)
        self.conv4 = _GCNConvParams(out_channels, out_channels)
        self.conv5 = _GCNConvParams(out_channels, out_channels)
        self.linear = nn.Linear(out_channels * 5, out_channels)      # dead layer of the reference, kept for state_dict parity
        self._attention = nn.Parameter(torch.empty(periods))
        nn.init.uniform_(self._attention)


class ConvStackedTemporalGCN(nn.Module):
    """Stacked-GCNConv baseline.  ``forward(x, edge_index, edge_attr)`` ->
    ``(prediction (N, output_dim), hidden (N, 512))`` (models/ConvStackedTemporalGCN.py:21-33).

    Per period the reference runs conv1..conv5 = A_hat (h W^T) + b without activation and feeds the result to the TGCN
    cell as its hidden input.  Here all periods go through each layer at once on node-major rows (node*T + t):
    layer 1 aggregates the *input* (width T*F, no sparse backward), layers 2-5 aggregate the learned hidden state at
    width T*512 with the CSR of A_hat forward and of A_hat^T backward; the dense parts are fp32-MFMA GEMMs and the
    cell + attention + head is ``regt_cell_forward`` / ``regt_cell_backward``."""

    HIDDEN = 512          # models/ConvStackedTemporalGCN.py:13
    HEAD = 256            # :16

    def __init__(self, node_features: int, periods: int, output_dim: int):
        super().__init__()
        self.tgnn = ConvStackedA3TGCN(in_channels=node_features, out_channels=self.HIDDEN, periods=periods)
        self.linear1 = nn.Linear(self.HIDDEN, self.HEAD)
        self.linear2 = nn.Linear(self.HEAD, output_dim)
        self.relu = nn.ReLU()
        self.output_dim = output_dim
        self._graphs = _GraphCache()

    def prepare_graph(self, edge_index, edge_attr, num_nodes: int, copies: int = 1) -> GcnOperator:
        return prepare_gcn_operator(edge_index, edge_attr, num_nodes, copies)

    # conv1 .. conv5 are applied WITHOUT an activation in between (models/ConvStackedTemporalGCN.py:116-120), and the aggregation
    # (acts on the node axis) commutes with the linear maps (act on the feature axis), so the five layers collapse exactly:
    #     h5 = (A^5 x) (W5 W4 W3 W2 W1)^T + sum_k (A^(5-k) 1) (W5 .. W_(k+1) b_k)^T,        A = A_hat
    # -- five aggregations of the INPUT at width T*F (no sparse backward: x is data) and of the all-ones vector (once per graph),
    # ONE (M x 512) GEMM with K = F + 8 on the rows [A^5 x | A^0 1 .. A^4 1], and a chain of 512 x 512 weight products that
    # autograd differentiates (LinearFunction: regt_linear / regt_wgrad).  The reference-faithful layer-by-layer form -- four
    # aggregations of the learned hidden state at width T*512 forward (A_hat) and backward (A_hat^T), 27 GB of gathered rows
    # each at the cfg-3 shape -- stays available as ``collapse = False`` / :meth:`forward_layerwise` (and is what pins the wide
    # aggregation kernels in tests/test_gpu_convstack.py); same parameters, same gradients up to fp32 reassociation.
    collapse = True

    def forward_prepared(self, x: torch.Tensor, op: GcnOperator):
        _need_cuda(x)
        if not self.collapse:
            return self.forward_layerwise(x, op)
        n, f, t = x.shape
        sd = dict(self.named_parameters())
        u = ops.pack_x(x).view(n, t * f)                                        # (N, T*F)
        for _ in range(5):
            u = ops.spmm_csr(op.rowptr, op.col, op.val, u)                      # A^5 x: input data, no backward
        chain = op.__dict__.get("_ones_chain")
        if chain is None:                                                       # [A^0 1 .. A^4 1 | 0 0 0] per node, once per graph
            cols, a = [], torch.ones(n, 4, device=x.device)
            for _ in range(5):
                cols.append(a[:, :1])
                a = ops.spmm_csr(op.rowptr, op.col, op.val, a)
            chain = torch.cat(cols + [torch.zeros(n, 3, device=x.device)], dim=1).contiguous()
            op.__dict__["_ones_chain"] = chain
        feat = torch.cat([u.view(n, t, f), chain[:, None, :].expand(n, t, 8)], dim=2).reshape(n * t, f + 8)
        # P_k = W5 .. W_(k+1),  c_k = P_k b_k  (k = 5 .. 1),  W_all = P_1 W1
        P = sd["tgnn.conv5.lin.weight"]
        cs = [sd["tgnn.conv5.bias"].view(1, -1)]                                # c_5 = b_5
        for k in (4, 3, 2, 1):
            cs.append(LinearFunction.apply(sd[f"tgnn.conv{k}.bias"].view(1, -1), P, None))          # (P b_k)^T as a row
            wk = sd[f"tgnn.conv{k}.lin.weight"]
            P = _compose(P, wk)                                                 # P @ W_k
        zero = torch.zeros(3, self.HIDDEN, device=x.device)
        wcat = torch.cat([P.t()] + cs + [zero], dim=0).t().contiguous()         # (512, F + 8): [W_all | c_5 c_4 c_3 c_2 c_1 | 0 0 0]
        h = LinearFunction.apply(feat, wcat, None)                              # (M, 512)
        return CellFunction.apply(x, h, op, *[sd[k] for k in PARAM_NAMES_CELL])

    def forward_layerwise(self, x: torch.Tensor, op: GcnOperator):
        """The reference's evaluation order: every layer aggregates its (learned) input at width T*512."""
        _need_cuda(x)
        n, f, t = x.shape
        c = self.HIDDEN
        sd = dict(self.named_parameters())
        xp = ops.pack_x(x)                                                      # (N, T, F)
        ax = ops.spmm_csr(op.rowptr, op.col, op.val, xp.view(n, t * f))          # A_hat x: input data, no backward
        h = LinearFunction.apply(ax.view(n * t, f), sd["tgnn.conv1.lin.weight"], sd["tgnn.conv1.bias"])
        for layer in range(2, 6):
            s = AggregateFunction.apply(h.view(n, t * c), op)
            h = LinearFunction.apply(s.view(n * t, c), sd[f"tgnn.conv{layer}.lin.weight"], sd[f"tgnn.conv{layer}.bias"])
        return CellFunction.apply(x, h, op, *[sd[k] for k in PARAM_NAMES_CELL])

    def forward(self, x, edge_index, edge_attr):
        _need_cuda(x)
        n = x.shape[0]
        op = self._graphs.get([edge_index, edge_attr], n, lambda: prepare_gcn_operator(edge_index, edge_attr, n))
        return self.forward_prepared(x, op)


# ---- GraphSAGE / GAT base blocks of the cell (second half) ---------------------------------------------
# Both reference models call ``self._base_tgcn(X[:, :, period], edge_index, H)``: TGCN.forward's third positional parameter is
# ``edge_weight``, so H stays None and the cell starts from zeros in every period (models/utils.py:163-166).  With H = 0 the
# reset gate only multiplies zeros: Z = sigmoid(U_z1 conv_z(x) + u_z), H~ = tanh(U_h1 conv_h(x) + u_h), H' = (1 - Z) H~
# -- every contraction has K = F (or 2F), the sparse operators act on the input, and the whole model is one aggregation, two
# skinny MFMA GEMMs with sigmoid / tanh epilogues, a blend + attention sum and the head (regt_cell0_forward).

def _compose(u1: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """U1 (C,C) @ W (C,K) on the matrix cores with autograd (regt_linear / regt_wgrad)."""
    return LinearFunction.apply(u1, w.t().contiguous(), None)


def _compose_bias(u1: torch.Tensor, b: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """U1 b + u."""
    return LinearFunction.apply(b.view(1, -1), u1, None).view(-1) + u


class GraphSAGE(nn.Module):
    """Parameter layout of models/GraphSAGETemporalGCN.py:46-72: TGCN cell with SAGEConv gates, a never-called GCNConv
    ``conv`` (:61-64), the attention over periods and the dead ``_weight_att*`` / ``_bias_att*`` of the A3T-GCN template."""

    def __init__(self, in_channels: int, out_channels: int, num_nodes: int, periods: int):
        super().__init__()
        self.in_channels, self.out_channels, self.num_nodes, self.periods = in_channels, out_channels, num_nodes, periods
        self._attention = nn.Parameter(torch.empty(periods))
        self._weight_att1 = nn.Parameter(torch.normal(0.0, 0.1, size=(out_channels, 1)))
        self._weight_att2 = nn.Parameter(torch.normal(0.0, 0.1, size=(num_nodes, 1)))
        self._bias_att1 = nn.Parameter(torch.normal(0.0, 1.0, size=(1, 1)))
        self._bias_att2 = nn.Parameter(torch.normal(0.0, 1.0, size=(1, 1)))
        self._base_tgcn = TGCN(in_channels, out_channels, baseblock="graphsage")
        self.conv = _GCNConvParams(in_channels, out_channels)
        nn.init.uniform_(self._attention)


class GAT(nn.Module):
    """Parameter layout of models/GATTemporal.py:37-64: TGCN cell with GATConv gates and the attention over periods."""

    def __init__(self, in_channels: int, out_channels: int, num_nodes: int, periods: int):
        super().__init__()
        self.in_channels, self.out_channels, self.num_nodes, self.periods = in_channels, out_channels, num_nodes, periods
        self._attention = nn.Parameter(torch.empty(periods))
        self._base_tgcn = TGCN(in_channels, out_channels, baseblock="gat")
        self.relu = nn.ReLU()
        nn.init.uniform_(self._attention)


def _pad_features(x: torch.Tensor):
    """(N, F, T) -> the same with F rounded up to a multiple of 4 by zero feature rows, and the number of rows added.  Any node
    feature width is accepted, as in the reference; zero features under zero weight columns change nothing, and autograd slices
    the gradients of the padded weights back."""
    fpad = (-x.shape[1]) % 4
    return (torch.nn.functional.pad(x, (0, 0, 0, fpad)).contiguous() if fpad else x), fpad


def _pad_cols(w: torch.Tensor, fpad: int) -> torch.Tensor:
    return torch.nn.functional.pad(w, (0, fpad)) if fpad else w


class _ZeroHiddenModel(nn.Module):
    cell_attr: str

    def _cell(self) -> TGCN:
        return getattr(self, self.cell_attr)._base_tgcn

    def _run_cell0(self, a_z, a_h, gz, gh, cz, ch, num_nodes: int):
        inner = getattr(self, self.cell_attr)
        pred, hidden = Cell0Function.apply(a_z, a_h, gz, gh, cz, ch, inner._attention, self.linear1.weight, self.linear1.bias,
                                           self.linear2.weight, self.linear2.bias, num_nodes)
        cell = self._cell()
        dead = list(cell.conv_r.parameters()) + list(cell.linear_r.parameters())     # reset gate: multiplied by H = 0
        return ZeroGradAnchor.apply(pred, hidden, *dead)


class GraphSAGETemporalGCN(_ZeroHiddenModel):
    """GraphSAGE baseline (models/GraphSAGETemporalGCN.py:8-43).  ``forward(x, edge_index, edge_attr)`` ->
    ``(prediction (N, output_dim), hidden (N, 256))``; ``edge_attr`` is accepted and ignored, as in the reference (:93-95)."""

    cell_attr = "tgnn"

    def __init__(self, node_features: int, num_nodes: int, periods: int, output_dim: int, hidden_channels: int = HIDDEN):
        super().__init__()
        self.tgnn = GraphSAGE(node_features, hidden_channels, num_nodes, periods)
        self.output_dim = output_dim
        self.linear1 = nn.Linear(hidden_channels, HEAD_HIDDEN)
        self.linear2 = nn.Linear(HEAD_HIDDEN, output_dim)
        self.relu = nn.ReLU()
        self._graphs = _GraphCache()

    def prepare_graph(self, edge_index, num_nodes: int, copies: int = 1) -> MeanOperator:
        return prepare_mean_operator(edge_index, num_nodes, copies)

    def forward_prepared(self, x: torch.Tensor, op: MeanOperator):
        _need_cuda(x)
        x, fpad = _pad_features(x)            # the kernels read 16-byte feature rows: zero columns for x and for the weights
        n, f, t = x.shape
        c = self.tgnn.out_channels
        cell = self.tgnn._base_tgcn
        xp = ops.pack_x(x)                                                        # (N, T, F)
        ax = ops.spmm_csr(op.rowptr, op.col, op.val, xp.view(n, t * f))           # mean over in-neighbours: input data, no backward
        a = torch.cat([ax.view(n * t, f), xp.view(n * t, f)], dim=1)              # [mean-neighbour x | x]  (M, 2F)
        gs, cs = [], []
        for conv, lin in ((cell.conv_z, cell.linear_z), (cell.conv_h, cell.linear_h)):
            u1 = lin.weight[:, :c].contiguous()
            gs.append(torch.cat([_compose(u1, _pad_cols(conv.lin_l.weight, fpad)), _compose(u1, _pad_cols(conv.lin_r.weight, fpad))], dim=1))
            cs.append(_compose_bias(u1, conv.lin_l.bias, lin.bias))
        return self._run_cell0(a, a, gs[0], gs[1], cs[0], cs[1], n)

    def forward(self, x, edge_index, edge_attr=None):
        _need_cuda(x)
        n = x.shape[0]
        op = self._graphs.get([edge_index, None], n, lambda: prepare_mean_operator(edge_index, n))
        return self.forward_prepared(x, op)


class GATTemporal(_ZeroHiddenModel):
    """GAT baseline (models/GATTemporal.py:7-34).  ``forward(x, edge_index, edge_attr)`` -> ``(prediction, hidden (N, 256))``;
    ``edge_attr`` is accepted and ignored, as in the reference (:78-80)."""

    cell_attr = "gat"
    NEGATIVE_SLOPE = 0.2          # GATConv default

    def __init__(self, node_features: int, num_nodes: int, periods: int, output_dim: int, hidden_channels: int = HIDDEN):
        super().__init__()
        self.gat = GAT(node_features, hidden_channels, num_nodes, periods)
        self.output_dim = output_dim
        self.linear1 = nn.Linear(hidden_channels, HEAD_HIDDEN)
        self.linear2 = nn.Linear(HEAD_HIDDEN, output_dim)
        self.relu = nn.ReLU()
        self._graphs = _GraphCache()

    def prepare_graph(self, edge_index, num_nodes: int, copies: int = 1) -> AttentionPattern:
        return prepare_attention_pattern(edge_index, num_nodes, copies)

    def forward_prepared(self, x: torch.Tensor, pat: AttentionPattern):
        _need_cuda(x)
        x, fpad = _pad_features(x)
        n, f, t = x.shape
        c = self.gat.out_channels
        cell = self.gat._base_tgcn
        xp = ops.pack_x(x)
        ins, gs, cs = [], [], []
        for conv, lin in ((cell.conv_z, cell.linear_z), (cell.conv_h, cell.linear_h)):
            w = _pad_cols(conv.lin.weight, fpad)                                   # (C, F)
            u_src = LinearFunction.apply(conv.att_src.view(1, c), w.t().contiguous(), None).view(f)      # W^T att_src
            u_dst = LinearFunction.apply(conv.att_dst.view(1, c), w.t().contiguous(), None).view(f)
            ins.append(GatAggregateFunction.apply(xp, u_src, u_dst, pat, self.NEGATIVE_SLOPE).view(n * t, f))
            u1 = lin.weight[:, :c].contiguous()
            gs.append(_compose(u1, w))
            cs.append(_compose_bias(u1, conv.bias, lin.bias))
        return self._run_cell0(ins[0], ins[1], gs[0], gs[1], cs[0], cs[1], n)

    def forward(self, x, edge_index, edge_attr=None):
        _need_cuda(x)
        n = x.shape[0]
        pat = self._graphs.get([edge_index, None], n, lambda: prepare_attention_pattern(edge_index, n))
        return self.forward_prepared(x, pat)
