"""Region-sharded multi-GPU execution: one process per GPU, RCCL (torch.distributed 'nccl') over xGMI.

The reference is single-process; this layer is new design:

* nodes are partitioned by region blocks -- rank g owns a contiguous node range; all dense work
  (GEMMs, gates, head, loss partial sums) and the regional ChebConv aggregation are local because
  regional edges never leave a region;
* the only cross-rank dependency is the full-graph GCN aggregation ``A_hat x`` for in-edges whose
  source lives on another rank.  ``x`` is input data, so ONE exchange of packed *halo rows* per snapshot
  suffices (a personalised all-to-all: each rank receives exactly the rows its in-edges read), and there
  is no sparse backward traffic at all (A_hat x is constant w.r.t. the parameters);
* gradients are summed with one flat-buffer all-reduce per optimiser step (the reference steps the
  optimiser once per epoch, run.py:194).

``ShardTopology`` is pure index logic (numpy, runs anywhere -- covered by gloo tests on CPU);
``build_shard`` adds the GPU graph preparation.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .graph import PreparedGraph, cheb_edge_weights, gcn_csr, gcn_dis, merge_operators, node_regions, raw_csr


@dataclass
class ShardTopology:
    rank: int
    world: int
    node_lo: int
    node_hi: int
    need: List[np.ndarray]         # need[s]: sorted global ids owned by rank s that in-edges of this rank read
    send: List[np.ndarray]         # send[r]: sorted global ids owned by this rank that rank r reads (= r's need[rank])

    @property
    def n_local(self) -> int:
        return self.node_hi - self.node_lo

    @property
    def halo_rows(self) -> int:
        return int(sum(a.size for a in self.need))

    @property
    def x_rows(self) -> int:
        """Rows of the extended packed input: own nodes, then the halo rows grouped by owner rank."""
        return self.n_local + self.halo_rows

    @property
    def recv_splits(self) -> List[int]:
        return [int(a.size) for a in self.need]

    @property
    def send_splits(self) -> List[int]:
        return [int(a.size) for a in self.send]

    def halo_ids(self) -> np.ndarray:
        """Global node id held by each halo row, in row order."""
        return np.concatenate(self.need) if self.world > 1 else np.zeros(0, dtype=np.int64)

    def send_index(self) -> np.ndarray:
        """Local row ids to send, grouped by destination rank."""
        if self.world == 1:
            return np.zeros(0, dtype=np.int64)
        return np.concatenate(self.send) - self.node_lo

    def remap_columns(self, cols: np.ndarray, owner_bounds: np.ndarray) -> np.ndarray:
        """Global source ids -> row ids of the extended packed input of this rank."""
        cols = np.asarray(cols, dtype=np.int64)
        out = np.empty_like(cols)
        local = (cols >= self.node_lo) & (cols < self.node_hi)
        out[local] = cols[local] - self.node_lo
        rem = ~local
        if rem.any():
            halo = self.halo_ids()                  # ascending: owners are contiguous ranges in rank order
            pos = np.searchsorted(halo, cols[rem])
            if (pos >= halo.size).any() or (halo[np.minimum(pos, halo.size - 1)] != cols[rem]).any():
                raise RuntimeError("halo source is not in this rank's need list")
            out[rem] = self.n_local + pos
        return out


def shard_topology(edge_index: np.ndarray, owner_bounds: np.ndarray, rank: int, world: int) -> ShardTopology:
    """``owner_bounds`` (world+1,) contiguous node ownership.  Every rank derives the same (reader, owner)
    lists from the global edge list, so send and receive sizes agree without a handshake."""
    src, dst = np.asarray(edge_index[0], dtype=np.int64), np.asarray(edge_index[1], dtype=np.int64)
    so = np.searchsorted(owner_bounds, src, side="right") - 1
    do = np.searchsorted(owner_bounds, dst, side="right") - 1
    empty = np.zeros(0, dtype=np.int64)
    need = [np.unique(src[(do == rank) & (so == s)]) if s != rank else empty for s in range(world)]
    send = [np.unique(src[(so == rank) & (do == r)]) if r != rank else empty for r in range(world)]
    return ShardTopology(rank, world, int(owner_bounds[rank]), int(owner_bounds[rank + 1]), need, send)


def topology_from_sources(sources: np.ndarray, owner_bounds: np.ndarray, rank: int, world: int, group=None) -> ShardTopology:
    """The same topology from THIS rank's in-edges alone.  ``sources``: global ids of the sources of the edges whose target the
    rank owns (any order, duplicates allowed).  ``need`` follows locally; ``send`` is what the other ranks need from this one:
    one all-gather of the list lengths and one all-to-all of the id lists at graph preparation (a rank never walks the global
    edge list).  Works with 'gloo' on CPU tensors and with 'nccl' (RCCL) on the current GPU."""
    lo, hi = int(owner_bounds[rank]), int(owner_bounds[rank + 1])
    src = np.unique(np.asarray(sources, dtype=np.int64))
    src = src[(src < lo) | (src >= hi)]
    owner = np.searchsorted(owner_bounds, src, side="right") - 1
    empty = np.zeros(0, dtype=np.int64)
    need = [src[owner == s] if s != rank else empty for s in range(world)]
    if world == 1:
        return ShardTopology(rank, world, lo, hi, need, [empty])
    if not dist.is_initialized() or dist.get_world_size(group) != world:
        raise RuntimeError("topology_from_sources needs a process group of `world` ranks (the send lists come from the peers)")
    send = exchange_need_lists(need, world, group)
    send[rank] = empty
    return ShardTopology(rank, world, lo, hi, need, send)


def exchange_need_lists(need: List[np.ndarray], world: int, group=None) -> List[np.ndarray]:
    """``need[s]`` = ids this rank wants from rank s  ->  ``send[r]`` = ids rank r wants from this rank: one all-gather of the
    list lengths and one all-to-all of the concatenated id lists (int64; device tensors under 'nccl', CPU tensors under 'gloo')."""
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    rank = dist.get_rank(group)
    counts = torch.tensor([a.size for a in need], dtype=torch.int64, device=dev)
    table = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(table, counts, group=group)                        # table[r][s] = how many ids rank r needs from rank s
    send_counts = [int(table[r][rank]) for r in range(world)]
    out = torch.empty(sum(send_counts), dtype=torch.int64, device=dev)
    empty = np.zeros(0, dtype=np.int64)
    inp = torch.from_numpy(np.concatenate(need) if sum(a.size for a in need) else empty).to(dev)
    dist.all_to_all_single(out, inp, send_counts, [int(a.size) for a in need], group=group)
    out = out.cpu().numpy()
    offs = np.concatenate([[0], np.cumsum(send_counts)])
    return [out[offs[r]:offs[r + 1]].copy() for r in range(world)]


# Rehearsal aid (bench.py --force-shard-path): issue the collectives even in a 1-rank group, so that the exact RCCL calls
# of the N > 1 flow can be exercised on a one-GPU box.
_FORCE_COLLECTIVES = os.environ.get("REGT_DIST_FORCE") == "1"


def exchange_boundary_rows(xp_ext: torch.Tensor, topo: ShardTopology, send_idx: torch.Tensor, group=None):
    """Fill rows [n_local, x_rows) of ``xp_ext`` (in place) with the halo rows, one all-to-all: every rank sends
    each peer exactly the rows that peer's in-edges read.  xGMI is point-to-point, so the personalised exchange
    maps onto the links directly and moves ~world x fewer bytes than all-gathering every boundary row to everyone.

    ``xp_ext``: (x_rows, W) with the rank's own packed rows already in [0, n_local).  Works with the
    'nccl' (RCCL) backend on GPU tensors and with 'gloo' on CPU tensors (tests)."""
    if topo.world == 1 and not _FORCE_COLLECTIVES:
        return xp_ext
    send = xp_ext.index_select(0, send_idx)                       # (sum send_splits, W)
    recv = xp_ext[topo.n_local:]                                   # (halo_rows, W), contiguous view
    if xp_ext.is_cuda and dist.get_backend(group) == "gloo":
        # functional rehearsal of the multi-rank flow on a box without RCCL peers: stage through the host
        host = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(host, send.cpu(), topo.recv_splits, topo.send_splits, group=group)
        recv.copy_(host)
    else:
        dist.all_to_all_single(recv, send, topo.recv_splits, topo.send_splits, group=group)
    return xp_ext


class HaloPipeline:
    """Pack + halo exchange of snapshot i+1 on a side stream while the compute stream works on snapshot i.

    The exchanged rows are input data (no dependence on the parameters), so the exchange can run a whole step
    ahead; two extended input buffers alternate.  Usage per step::

        buf = pipe.acquire(slot)            # compute stream waits for the slot's exchange
        pipe.submit(1 - slot, x_next)       # side stream: pack x_next, all-to-all its halo rows
        ... forward_packed(buf) / backward ...
        pipe.release(slot)                  # the slot may be overwritten once this point is reached
    """

    def __init__(self, shard: "Shard", periods: int, features: int, device, group=None, dtype=torch.float32):
        """``dtype`` torch.bfloat16: pack and exchange the rows as bf16 (REGT_GEMM_MODE=bf16 -- x only ever feeds bf16 matrix-core
        operands there): half the bytes on the xGMI links, and ``forward_packed`` takes the buffer without a conversion."""
        from . import ops
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("HaloPipeline: dtype must be float32 or bfloat16")
        self._pack = ops.pack_x_into if dtype == torch.float32 else ops.pack_x_bf16_into
        self.shard, self.group = shard, group
        self.T, self.F = periods, features
        self.stream = torch.cuda.Stream(device)
        rows = shard.topo.x_rows
        self.buf = [torch.empty(rows, periods, features, dtype=dtype, device=device) for _ in range(2)]
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.free = [torch.cuda.Event() for _ in range(2)]
        for e in self.free:
            e.record(torch.cuda.current_stream(device))
        # optional timing of pack + exchange on the side stream (bench.py's N > 1 line): event pairs of the last submits
        self.timed = False
        self._spans = []

    def submit(self, slot: int, x: torch.Tensor):
        topo = self.shard.topo
        self.stream.wait_event(self.free[slot])
        self.stream.wait_stream(torch.cuda.current_stream(x.device))      # x may have been produced just now
        with torch.cuda.stream(self.stream):
            if self.timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
            self._pack(x, self.buf[slot])
            exchange_boundary_rows(self.buf[slot].view(topo.x_rows, self.T * self.F), topo, self.shard.send_idx, self.group)
            if self.timed:
                e1.record(self.stream)
                self._spans.append((e0, e1))
                del self._spans[:-64]
            self.ready[slot].record(self.stream)

    def exchange_ms(self):
        """Mean / max device time of pack + halo exchange over the timed submits (side stream: off the critical path); synchronises."""
        if not self._spans:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._spans]
        return {"mean_ms": sum(ms) / len(ms), "max_ms": max(ms), "samples": len(ms)}

    def acquire(self, slot: int) -> torch.Tensor:
        torch.cuda.current_stream(self.buf[slot].device).wait_event(self.ready[slot])
        return self.buf[slot]

    def release(self, slot: int):
        self.free[slot].record(torch.cuda.current_stream(self.buf[slot].device))


def allreduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum all-reduce; a GPU tensor under the 'gloo' rehearsal backend is staged through the host."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _FORCE_COLLECTIVES):
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_gradients(params: Sequence[torch.nn.Parameter], group=None):
    """One flat-buffer sum all-reduce of all gradients (3-19 MB here: latency-bound on xGMI, so one call)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _FORCE_COLLECTIVES):
        return
    flat = allreduce_sum(torch.cat([g.reshape(-1) for g in grads]), group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


@dataclass
class Shard:
    topo: ShardTopology
    graph: PreparedGraph           # local stacked operator; A_hat columns index the extended input
    send_idx: torch.Tensor         # (sum send_splits,) int64 on the device


def _own_rows_operator(edge_index: torch.Tensor, gcn_weight: Optional[torch.Tensor], num_nodes: int, owner_bounds: np.ndarray,
                       rank: int, world: int, device, group=None):
    """A_hat rows of the owned nodes without normalising the global graph: the rank keeps the edges whose
    TARGET it owns (one vectorised filter on the device), normalises that reduced graph -- owned nodes see all of their
    in-edges there, so their degrees and rows come out right, halo sources get degree 1 -- and multiplies the entries of halo
    columns by the sources' true D^-1/2, which every rank publishes for its own nodes with one all-reduce of a length-N
    vector (each rank fills its slice; 4 MB at 1M nodes).  With unit edge weights (RegT-GCN: edge_weight=None) the result is
    bit-identical to slicing the globally normalised operator: (d_j * 1) * d_i and (1 * 1) * d_i * d_j round the same."""
    lo, hi = int(owner_bounds[rank]), int(owner_bounds[rank + 1])
    n_local = hi - lo
    ei = edge_index.to(device)
    mine = (ei[1] >= lo) & (ei[1] < hi)
    src, dst = ei[0][mine], ei[1][mine] - lo
    w = None if gcn_weight is None else gcn_weight.to(device)[mine].contiguous()
    local = (src >= lo) & (src < hi)
    topo = topology_from_sources(src[~local].cpu().numpy(), owner_bounds, rank, world, group)
    halo = torch.from_numpy(topo.halo_ids()).to(device)
    pos = torch.searchsorted(halo, src) if halo.numel() else torch.zeros_like(src)
    red = torch.stack([torch.where(local, src - lo, n_local + pos), dst]).contiguous()
    rp, col, val = gcn_csr(red, w, topo.x_rows)
    e = int(rp[n_local].item())
    rp_a, col_a, val_a = rp[:n_local + 1].contiguous(), col[:e].contiguous(), val[:e].clone()
    if world > 1:
        dis = torch.zeros(num_nodes, dtype=torch.float32, device=device)
        dis[lo:hi] = gcn_dis(red, w, topo.x_rows)[:n_local]
        allreduce_sum(dis, group)
        is_halo = col_a >= n_local
        val_a[is_halo] = val_a[is_halo] * dis[halo[(col_a[is_halo] - n_local).long()]]
    return topo, rp_a, col_a, val_a


def build_shard(edge_index: torch.Tensor, region_index: Sequence[torch.Tensor], region_attr: Sequence[torch.Tensor],
                num_nodes: int, owner_bounds: np.ndarray, region_owner: Sequence[int], rank: int, world: int,
                device, gcn_weight: Optional[torch.Tensor] = None, group=None, method: Optional[str] = None) -> Shard:
    """GPU graph preparation for one rank.  ``edge_index`` etc. are the GLOBAL graph (host tensors);
    ``region_owner[r]`` is the rank that owns region r (its nodes lie inside that rank's range).

    ``method`` "own_rows" (default whenever the process group has ``world`` ranks and the GCN operator has unit weights): every
    rank normalises only the edges into its own nodes and the ranks exchange degrees and need lists
    (:func:`_own_rows_operator`).  It issues COLLECTIVES (an all-gather + all-to-all of the need lists, one all-reduce of the
    degree vector): every rank of ``group`` must call build_shard, or the job deadlocks.  With a weighted GCN operator
    (``gcn_weight`` given -- not the RegT-GCN path, which normalises with unit weights) the rescaled halo entries
    ``(1 * w * d_i) * d_j`` round differently in the last bit from the global build's ``(d_j * w) * d_i``; the default is
    therefore "global" there, so that a sharded run stays bit-identical to the single-GPU operator.  "global": the rank walks and
    normalises the whole edge list by itself -- the form a single process needs when it plays one rank of a larger job
    (``bench.py --workload cfg5shard``, tests), and the reference the own-rows form is compared with bit for bit."""
    if method is None:
        grouped = world == 1 or (dist.is_initialized() and dist.get_world_size(group) == world)
        method = os.environ.get("REGT_SHARD_BUILD") or ("own_rows" if grouped and gcn_weight is None else "global")
    if method not in ("own_rows", "global"):
        raise ValueError("build_shard: method must be 'own_rows' or 'global'")
    if method == "own_rows":
        topo, rp_a, col_a, val_a = _own_rows_operator(edge_index, gcn_weight, num_nodes, owner_bounds, rank, world, device, group)
        lo, hi = topo.node_lo, topo.node_hi
    else:
        topo = shard_topology(edge_index.cpu().numpy(), owner_bounds, rank, world)
        lo, hi = topo.node_lo, topo.node_hi
        # normalise the global edge list on this rank's GPU, then keep the rows of the owned nodes and remap their columns to the
        # extended input -- all on the device (the slice bounds are the only values read back)
        rp, col, val = gcn_csr(edge_index.to(device), None if gcn_weight is None else gcn_weight.to(device), num_nodes)
        b, e = (int(v) for v in rp[[lo, hi]].tolist())
        rp_a = (rp[lo:hi + 1] - b).to(torch.int32).contiguous()
        cols = col[b:e].long()
        local = (cols >= lo) & (cols < hi)
        halo = torch.from_numpy(topo.halo_ids()).to(device)               # ascending global ids of the halo rows
        pos = torch.searchsorted(halo, cols.clamp(min=0)) if halo.numel() else torch.zeros_like(cols)
        if halo.numel():
            hit = halo[pos.clamp(max=halo.numel() - 1)] == cols
        else:
            hit = torch.zeros_like(local)
        if not bool((local | hit).all()):
            raise RuntimeError("halo source is not in this rank's need list")
        col_a = torch.where(local, cols - lo, topo.n_local + pos).to(torch.int32).contiguous()
        val_a = val[b:e].contiguous()
    # regional Laplacians of the owned regions, in local ids
    mine = [r for r in range(len(region_index)) if region_owner[r] == rank]
    if not mine:
        raise ValueError(f"rank {rank} owns no region: a region shard needs at least one region per rank "
                         f"({len(region_index)} regions over {world} ranks)")
    loc_idx = [(region_index[r] - lo).to(device) for r in mine]
    loc_w = [region_attr[r].to(device) for r in mine]
    n_loc = hi - lo
    w_all = [cheb_edge_weights(ei, ew, n_loc) for ei, ew in zip(loc_idx, loc_w)]
    # region ids stay GLOBAL: the region linear layer (tgnn.linear, (C, R_global*C)) is replicated on
    # every rank and its blocks are addressed by global region id
    owner = np.asarray(mine, dtype=np.int32)[node_regions([t.cpu() for t in loc_idx], n_loc)]
    rp_l, col_l, val_l = raw_csr(torch.cat(loc_idx, dim=1), torch.cat(w_all), n_loc)
    nnz_a = int(col_a.numel())
    m = merge_operators(rp_a, col_a, val_a, rp_l, col_l, val_l, n_loc)
    contiguous = mine == list(range(mine[0], mine[-1] + 1)) if mine else False
    graph = PreparedGraph(num_nodes=n_loc, num_regions=len(region_index), region_lo=mine[0] if contiguous else 0,
                          region_hi=mine[-1] + 1 if contiguous else 0, m_rowptr=m[0], m_col=m[1], m_val_a=m[2], m_val_l=m[3],
                          rowptr=torch.cat([rp_a, rp_l[1:] + nnz_a]).contiguous(),
                          col=torch.cat([col_a, col_l]).contiguous(), val=torch.cat([val_a, val_l]).contiguous(),
                          node_region=torch.from_numpy(owner).to(device), node_region_host=owner,
                          nnz_gcn=nnz_a, nnz_cheb=int(col_l.numel()))
    return Shard(topo, graph, torch.from_numpy(topo.send_index()).to(device))
