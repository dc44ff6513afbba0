"""Prepared (normalised, destination-sorted) graph operators for the RegT-GCN hot path.

The reference builds its conv layers with ``cached=False`` (models/RegionalTemporalGCN.py:54,73)
so PyG re-derives 5 Laplacians + 3 GCN normalisations in every period of every forward; the graph
is static, so here that work happens once, on the GPU (csrc/graph.hip), and is cached.

A :class:`PreparedGraph` holds the stacked CSR the pipeline consumes:
rows ``[0, N)``  -> ``A_hat = D^-1/2 (A+I) D^-1/2`` of the full graph (GCNConv, models/utils.py:169-181),
rows ``[N, 2N)`` -> the merged regional scaled Laplacians ``L~_r`` (ChebConv K=2,
RegionalTemporalGCN.py:136-140), which is valid when every node receives regional edges from at
most one regional graph ("node-disjoint regions", true for the reference's regional decomposition).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib


@dataclass
class PreparedGraph:
    num_nodes: int
    num_regions: int
    rowptr: torch.Tensor        # (2N+1,) int32
    col: torch.Tensor           # (nnz,) int32
    val: torch.Tensor           # (nnz,) float32
    node_region: torch.Tensor   # (N,) int32
    node_region_host: np.ndarray
    nnz_gcn: int
    nnz_cheb: int
    # merged operator (one entry per distinct (row, col) with both weights) for the single-gather SpMM
    region_lo: int = 0             # region ids [region_lo, region_hi) own rows of this graph; 0, 0 = all (dist.build_shard
    region_hi: int = 0             # sets the owned block of a region shard)
    m_rowptr: Optional[torch.Tensor] = None
    m_col: Optional[torch.Tensor] = None
    m_val_a: Optional[torch.Tensor] = None
    m_val_l: Optional[torch.Tensor] = None
    overlap: bool = False       # True: (1+R)*N stacked rows, one Laplacian per region (regions share nodes)
    _chunks: Dict[int, Tuple[torch.Tensor, torch.Tensor, int]] = field(default_factory=dict)

    @property
    def device(self):
        return self.rowptr.device

    @property
    def region_sorted(self) -> bool:
        """Region ids non-decreasing by node (regt_graph.region_sorted): the nodes of a region are contiguous."""
        if "_region_sorted" not in self.__dict__:
            nr = self.node_region_host
            self.__dict__["_region_sorted"] = bool(len(nr) < 2 or np.all(nr[1:] >= nr[:-1]))
        return self.__dict__["_region_sorted"]

    def chunks_for(self, periods: int):
        """(chunk_tab (n,2) int32, chunk_region (n,) int32, n): row ranges of (node*T+t) rows inside one region."""
        if periods not in self._chunks:
            tab, reg = region_chunks(self.node_region_host, periods)
            self._chunks[periods] = (torch.from_numpy(tab).to(self.device), torch.from_numpy(reg).to(self.device), len(reg))
        return self._chunks[periods]


def region_chunks(node_region: np.ndarray, periods: int):
    """Split the (node*T + t) row space into chunks that never straddle a region boundary."""
    n = len(node_region)
    m = n * periods
    kc = max(128, ((m + 511) // 512 + 31) // 32 * 32)    # same granularity as the other skinny gradients
    change = np.flatnonzero(np.diff(node_region)) + 1
    starts = np.concatenate([[0], change]).astype(np.int64)
    ends = np.concatenate([change, [n]]).astype(np.int64)
    tab, reg = [], []
    for s, e in zip(starts, ends):
        r0, r1 = s * periods, e * periods
        while r0 < r1:
            r2 = min(r0 + kc, r1)
            tab.append((r0, r2))
            reg.append(node_region[s])
            r0 = r2
    return np.asarray(tab, dtype=np.int32).reshape(-1, 2), np.asarray(reg, dtype=np.int32)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check_edges(edge_index: torch.Tensor, weight: Optional[torch.Tensor], what: str):
    if edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_index.dtype != torch.int64:
        raise ValueError(f"{what}: edge_index must be a (2,E) int64 tensor, got {tuple(edge_index.shape)} {edge_index.dtype}")
    if not edge_index.is_cuda:
        raise _lib.RegtError(f"{what}: edge_index must live on the GPU (no CPU path)")
    if weight is not None:
        if weight.dtype != torch.float32 or weight.numel() != edge_index.shape[1] or weight.device != edge_index.device:
            raise ValueError(f"{what}: edge weights must be float32 (E,) on the same device")


def _raise_flags(flags: torch.Tensor, what: str):
    f = int(flags.item())
    if f & 1:
        raise ValueError(f"{what}: edge_index contains node ids outside [0, num_nodes)")
    if f & 2:
        raise ValueError(f"{what}: negative edge weights are not supported (degree^-1/2 normalisation)")


def gcn_csr(edge_index: torch.Tensor, edge_weight: Optional[torch.Tensor], num_nodes: int):
    """A_hat as (rowptr, col, val) -- GCNConv's gcn_norm, computed once."""
    lib = _lib.load()
    _check_edges(edge_index, edge_weight, "gcn_csr")
    dev = edge_index.device
    ei = edge_index.contiguous()
    ew = None if edge_weight is None else edge_weight.contiguous()
    e = ei.shape[1]
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    col = torch.empty(e + num_nodes, dtype=torch.int32, device=dev)
    val = torch.empty(e + num_nodes, dtype=torch.float32, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = lib.regt_graph_workspace_bytes(e, num_nodes)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.check(lib.regt_gcn_csr(_lib.ptr(ei), _lib.ptr(ew), e, num_nodes, _lib.ptr(rowptr), _lib.ptr(col),
                                _lib.ptr(val), _lib.ptr(flags), _lib.ptr(ws), wsb, _stream()), "regt_gcn_csr")
    _raise_flags(flags, "gcn_csr")
    nnz = int(rowptr[-1].item())
    return rowptr, col[:nnz], val[:nnz]


def gcn_dis(edge_index: torch.Tensor, edge_weight: Optional[torch.Tensor], num_nodes: int) -> torch.Tensor:
    """D^-1/2 of gcn_norm (in-degree + self loop) per node, in the arithmetic of :func:`gcn_csr`."""
    lib = _lib.load()
    _check_edges(edge_index, edge_weight, "gcn_dis")
    dev = edge_index.device
    ei = edge_index.contiguous()
    ew = None if edge_weight is None else edge_weight.contiguous()
    e = ei.shape[1]
    dis = torch.empty(num_nodes, dtype=torch.float32, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = lib.regt_graph_workspace_bytes(e, num_nodes)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.check(lib.regt_gcn_dis(_lib.ptr(ei), _lib.ptr(ew), e, num_nodes, _lib.ptr(dis), _lib.ptr(flags), _lib.ptr(ws), wsb, _stream()),
               "regt_gcn_dis")
    _raise_flags(flags, "gcn_dis")
    return dis


def cheb_edge_weights(edge_index: torch.Tensor, edge_weight: Optional[torch.Tensor], num_nodes: int) -> torch.Tensor:
    """Per-edge scaled-Laplacian weights (ChebConv.__norm__ with lambda_max=None), computed once."""
    lib = _lib.load()
    _check_edges(edge_index, edge_weight, "cheb_edge_weights")
    dev = edge_index.device
    ei = edge_index.contiguous()
    ew = None if edge_weight is None else edge_weight.contiguous()
    e = ei.shape[1]
    out = torch.empty(e, dtype=torch.float32, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = lib.regt_graph_workspace_bytes(e, num_nodes)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.check(lib.regt_cheb_edge_weights(_lib.ptr(ei), _lib.ptr(ew), e, num_nodes, _lib.ptr(out), _lib.ptr(flags),
                                          _lib.ptr(ws), wsb, _stream()), "regt_cheb_edge_weights")
    _raise_flags(flags, "cheb_edge_weights")
    return out


def raw_csr(edge_index: torch.Tensor, edge_value: torch.Tensor, num_nodes: int):
    lib = _lib.load()
    _check_edges(edge_index, edge_value, "raw_csr")
    dev = edge_index.device
    ei = edge_index.contiguous()
    e = ei.shape[1]
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    col = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    val = torch.empty(max(e, 1), dtype=torch.float32, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = lib.regt_graph_workspace_bytes(e, num_nodes)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.check(lib.regt_raw_csr(_lib.ptr(ei), _lib.ptr(edge_value.contiguous()), e, num_nodes, _lib.ptr(rowptr),
                                _lib.ptr(col), _lib.ptr(val), _lib.ptr(flags), _lib.ptr(ws), wsb, _stream()), "regt_raw_csr")
    _raise_flags(flags, "raw_csr")
    nnz = int(rowptr[-1].item())
    return rowptr, col[:nnz], val[:nnz]


def mean_csr(edge_index: torch.Tensor, num_nodes: int):
    """Mean-aggregation operator of SAGEConv (every listed in-edge, weight 1 / in-degree) as (rowptr, col, val)."""
    lib = _lib.load()
    _check_edges(edge_index, None, "mean_csr")
    dev = edge_index.device
    ei = edge_index.contiguous()
    e = ei.shape[1]
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    col = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    val = torch.empty(max(e, 1), dtype=torch.float32, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = lib.regt_graph_workspace_bytes(e, num_nodes)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.check(lib.regt_mean_csr(_lib.ptr(ei), e, num_nodes, _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val), _lib.ptr(flags),
                                 _lib.ptr(ws), wsb, _stream()), "regt_mean_csr")
    _raise_flags(flags, "mean_csr")
    nnz = int(rowptr[-1].item())
    return rowptr, col[:nnz], val[:nnz]


@dataclass
class MeanOperator:
    """SAGEConv's mean-neighbour operator of one static graph (input aggregation only: no transposed copy needed)."""
    num_nodes: int
    rowptr: torch.Tensor
    col: torch.Tensor
    val: torch.Tensor


@dataclass
class AttentionPattern:
    """Sparsity pattern of GATConv's attention (in-edges without self loops + one self loop per node) and its transpose."""
    num_nodes: int
    rowptr: torch.Tensor
    col: torch.Tensor
    t_rowptr: torch.Tensor
    t_col: torch.Tensor


def prepare_mean_operator(edge_index: torch.Tensor, num_nodes: int, copies: int = 1) -> MeanOperator:
    """``copies`` > 1: the block-diagonal operator of that many disjoint copies of the graph (snapshot batching: the mean over a
    node's in-neighbours is local to its copy)."""
    if copies > 1:
        edge_index, _ = replicate_edges(edge_index, None, copies, num_nodes)
        num_nodes *= copies
    rp, col, val = mean_csr(edge_index, num_nodes)
    return MeanOperator(num_nodes, rp, col, val)


def prepare_attention_pattern(edge_index: torch.Tensor, num_nodes: int, copies: int = 1) -> AttentionPattern:
    if copies > 1:      # (the attention softmax runs over a node's in-neighbours: local to its copy)
        edge_index, _ = replicate_edges(edge_index, None, copies, num_nodes)
        num_nodes *= copies
    rp, col, val = gcn_csr(edge_index, None, num_nodes)          # the pattern of gcn_norm = remove + add self loops
    t = transpose_csr(rp, col, val, num_nodes, num_nodes)
    return AttentionPattern(num_nodes, rp, col, t[0], t[1])


def fingerprint(tensors: Sequence[Optional[torch.Tensor]]) -> int:
    """Content hash of a list of (edge_index, weight) style tensors -- one device pass + one 8-byte readback."""
    lib = _lib.load()
    total = 0
    it = iter(tensors)
    for ei in it:
        w = next(it)
        out = torch.zeros(1, dtype=torch.int64, device=ei.device)
        _lib.check(lib.regt_graph_fingerprint(_lib.ptr(ei.contiguous()), _lib.ptr(None if w is None else w.contiguous()),
                                              ei.shape[1], _lib.ptr(out), _stream()), "regt_graph_fingerprint")
        total = (total * 1000003 + (int(out.item()) & 0xFFFFFFFFFFFFFFFF) + ei.shape[1]) & 0xFFFFFFFFFFFFFFFF
    return total


class OverlappingRegions(ValueError):
    """A node receives edges in more than one regional graph (e.g. the reference's 'random' decomposition)."""


def node_regions(region_index: Sequence[torch.Tensor], num_nodes: int) -> np.ndarray:
    """Region that owns each node's Laplacian row; raises OverlappingRegions if a node receives edges in two
    regional graphs."""
    owner = np.full(num_nodes, -1, dtype=np.int32)
    for r, ei in enumerate(region_index):
        e = ei.detach().cpu().numpy()
        dst = np.unique(e[1][e[0] != e[1]])
        clash = owner[dst] >= 0
        if clash.any():
            raise OverlappingRegions("node %d receives edges in regions %d and %d" %
                                     (int(dst[clash][0]), int(owner[dst[clash][0]]), r))
        owner[dst] = r
    # nodes without regional in-edges have an all-zero Laplacian row: attach them to the previous
    # node's region so that region runs (and wgrad chunks) stay long.
    have = owner >= 0
    idx = np.where(have, np.arange(num_nodes), -1)
    np.maximum.accumulate(idx, out=idx)            # index of the last owned node at or before i
    filled = np.where(idx >= 0, owner[np.maximum(idx, 0)], 0)
    return filled.astype(np.int32)


def merge_operators(rp_a, col_a, val_a, rp_l, col_l, val_l, num_nodes: int):
    """Union of the two CSR patterns with both weights per entry (index bookkeeping on the device, once per graph).

    Entries of one row are ordered by source id; an (i, j) pair present in both operators becomes one entry, a pair
    present in one of them gets weight 0 in the other.  Duplicate edges of one operator are summed."""
    dev = rp_a.device
    n = num_nodes
    nx = int(max(int(col_a.max()) if col_a.numel() else 0, int(col_l.max()) if col_l.numel() else 0)) + 1
    rows_a = torch.repeat_interleave(torch.arange(n, device=dev), (rp_a[1:] - rp_a[:-1]).long())
    rows_l = torch.repeat_interleave(torch.arange(n, device=dev), (rp_l[1:] - rp_l[:-1]).long())
    keys = torch.cat([rows_a * nx + col_a.long(), rows_l * nx + col_l.long()])
    uniq, inverse = torch.unique(keys, sorted=True, return_inverse=True)
    na = col_a.numel()
    va = torch.zeros(uniq.numel(), dtype=torch.float32, device=dev).index_add_(0, inverse[:na], val_a)
    vl = torch.zeros(uniq.numel(), dtype=torch.float32, device=dev).index_add_(0, inverse[na:], val_l)
    m_row = torch.div(uniq, nx, rounding_mode="floor")
    m_col = (uniq - m_row * nx).to(torch.int32)
    counts = torch.bincount(m_row, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return rowptr.contiguous(), m_col.contiguous(), va.contiguous(), vl.contiguous()


@dataclass
class GcnOperator:
    """A_hat of one static graph as a destination-sorted CSR, plus the CSR of its transpose (the backward of
    ``A_hat @ H`` w.r.t. a learned H is ``A_hat^T @ dY``; A_hat is symmetric only for symmetric edge lists)."""
    num_nodes: int
    rowptr: torch.Tensor
    col: torch.Tensor
    val: torch.Tensor
    t_rowptr: torch.Tensor
    t_col: torch.Tensor
    t_val: torch.Tensor


def transpose_csr(rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, num_rows: int, num_cols: int):
    """CSR of the transposed operator (index bookkeeping on the device, once per graph).  Entries of one output row
    are ordered by their original row id, so the summation order of the transposed SpMM is fixed too."""
    dev = rowptr.device
    rows = torch.repeat_interleave(torch.arange(num_rows, device=dev), (rowptr[1:] - rowptr[:-1]).long())
    key = col.long() * num_rows + rows                       # sort by (col, row)
    order = torch.argsort(key, stable=True)
    t_col = rows[order].to(torch.int32)
    t_val = val[order].contiguous()
    counts = torch.bincount(col.long(), minlength=num_cols)
    t_rowptr = torch.zeros(num_cols + 1, dtype=torch.int32, device=dev)
    t_rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return t_rowptr.contiguous(), t_col.contiguous(), t_val


def prepare_gcn_operator(edge_index: torch.Tensor, edge_weight: Optional[torch.Tensor], num_nodes: int, copies: int = 1) -> GcnOperator:
    if copies > 1:
        edge_index, edge_weight = replicate_edges(edge_index, edge_weight, copies, num_nodes)
        num_nodes *= copies
    rp, col, val = gcn_csr(edge_index, edge_weight, num_nodes)
    t = transpose_csr(rp, col, val, num_nodes, num_nodes)
    return GcnOperator(num_nodes, rp, col, val, t[0], t[1], t[2])


def replicate_edges(edge_index: torch.Tensor, weight: Optional[torch.Tensor], copies: int, num_nodes: int):
    """``copies`` disjoint copies of one graph: copy b holds the nodes [b*N, (b+1)*N).  Both normalisations of the path are
    degree-local (gcn_norm, ChebConv.__norm__ with lambda_max = 2), so the operators of the replicated graph are block diagonal
    with the single-graph operator in every block: B snapshots of the one static graph run as ONE problem of B*N nodes
    (the per-snapshot loop of run.py:170-192 / :208-216 is additive over snapshots)."""
    if copies == 1:
        return edge_index, weight
    off = (torch.arange(copies, device=edge_index.device, dtype=edge_index.dtype) * num_nodes).view(copies, 1, 1)
    ei = (edge_index.unsqueeze(0) + off).permute(1, 0, 2).reshape(2, -1).contiguous()
    return ei, (None if weight is None else weight.repeat(copies).contiguous())


def prepare_graph(edge_index: torch.Tensor, gcn_weight: Optional[torch.Tensor], region_index: Sequence[torch.Tensor],
                  region_weight: Sequence[Optional[torch.Tensor]], num_nodes: int, copies: int = 1) -> PreparedGraph:
    """Build the stacked [A_hat; L~] operator.

    ``copies`` > 1: the block-diagonal operator of ``copies`` disjoint copies of the graph (snapshot batching,
    :func:`replicate_edges`); the result has ``copies * num_nodes`` nodes.

    ``gcn_weight`` is None for RegT-GCN (the cell is called with edge_weight=None,
    RegionalTemporalGCN.py:146-148) and the distance weights for TemporalGCN (TemporalGCN.py:89-90).
    """
    if len(region_index) != len(region_weight) or len(region_index) == 0:
        raise ValueError("need one weight tensor (or None) per regional edge_index")
    if copies > 1:
        edge_index, gcn_weight = replicate_edges(edge_index, gcn_weight, copies, num_nodes)
        rep = [replicate_edges(ei, ew, copies, num_nodes) for ei, ew in zip(region_index, region_weight)]
        region_index, region_weight = [r[0] for r in rep], [r[1] for r in rep]
        num_nodes *= copies
    rp_a, col_a, val_a = gcn_csr(edge_index, gcn_weight, num_nodes)
    w_all = [cheb_edge_weights(ei, ew, num_nodes) for ei, ew in zip(region_index, region_weight)]
    try:
        owner = node_regions(region_index, num_nodes)
    except OverlappingRegions:
        # general mode: one Laplacian per region, (1+R)*N stacked rows; the pipeline then sums
        # (L~_r x) A_r^T over all regions instead of selecting one region per node
        rps, cols, vals, off = [rp_a], [col_a], [val_a], int(col_a.numel())
        for ei, w in zip(region_index, w_all):
            rp_r, col_r, val_r = raw_csr(ei, w, num_nodes)
            rps.append(rp_r[1:] + off)
            cols.append(col_r)
            vals.append(val_r)
            off += int(col_r.numel())
        zeros = np.zeros(num_nodes, dtype=np.int32)
        return PreparedGraph(num_nodes=num_nodes, num_regions=len(region_index), rowptr=torch.cat(rps).contiguous(),
                             col=torch.cat(cols).contiguous(), val=torch.cat(vals).contiguous(),
                             node_region=torch.from_numpy(zeros).to(edge_index.device), node_region_host=zeros,
                             nnz_gcn=int(col_a.numel()), nnz_cheb=off - int(col_a.numel()), overlap=True)
    ei_all = torch.cat([ei for ei in region_index], dim=1)
    rp_l, col_l, val_l = raw_csr(ei_all, torch.cat(w_all), num_nodes)
    nnz_a = int(col_a.numel())
    rowptr = torch.cat([rp_a, rp_l[1:] + nnz_a]).contiguous()
    col = torch.cat([col_a, col_l]).contiguous()
    val = torch.cat([val_a, val_l]).contiguous()
    if col.numel() == 0:
        raise ValueError("graph has no edges and no nodes")
    m = merge_operators(rp_a, col_a, val_a, rp_l, col_l, val_l, num_nodes)
    return PreparedGraph(num_nodes=num_nodes, num_regions=len(region_index), rowptr=rowptr, col=col, val=val,
                         node_region=torch.from_numpy(owner).to(edge_index.device), node_region_host=owner,
                         nnz_gcn=nnz_a, nnz_cheb=int(col_l.numel()), m_rowptr=m[0], m_col=m[1], m_val_a=m[2], m_val_l=m[3])
