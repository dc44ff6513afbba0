"""autograd bridge: one Function = one regt_forward / regt_backward pair of the C ABI."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib
from .graph import PreparedGraph

HEAD_HIDDEN = 128

# canonical parameter order handed to the Function (state_dict names of the reference)
GATES = ("z", "r", "h")
PARAM_NAMES_COMMON = (
    ["tgnn._attention"]
    + [f"tgnn._base_tgcn.conv_{k}.lin.weight" for k in GATES]
    + [f"tgnn._base_tgcn.conv_{k}.bias" for k in GATES]
    + [f"tgnn._base_tgcn.linear_{k}.weight" for k in GATES]
    + [f"tgnn._base_tgcn.linear_{k}.bias" for k in GATES]
    + ["tgnn.conv.lins.0.weight", "tgnn.conv.lins.1.weight", "tgnn.conv.bias"]
)
PARAM_NAMES_REGION = ["tgnn.linear.weight", "tgnn.linear.bias"]
PARAM_NAMES_HEAD = ["linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias"]


# parameters whose last dimension is the node-feature width F
F_WIDE_PARAMS = tuple([f"tgnn._base_tgcn.conv_{k}.lin.weight" for k in GATES] + ["tgnn.conv.lins.0.weight", "tgnn.conv.lins.1.weight"])


def param_names(regional: bool) -> List[str]:
    return PARAM_NAMES_COMMON + (PARAM_NAMES_REGION if regional else []) + PARAM_NAMES_HEAD


def _fill(struct, tensors: Dict[str, Optional[torch.Tensor]], regional: bool):
    g = lambda n: None if tensors.get(n) is None else tensors[n].data_ptr()
    struct.attention = g("tgnn._attention")
    for i, k in enumerate(GATES):
        struct.conv_lin_w[i] = g(f"tgnn._base_tgcn.conv_{k}.lin.weight")
        struct.conv_bias[i] = g(f"tgnn._base_tgcn.conv_{k}.bias")
        struct.gate_w[i] = g(f"tgnn._base_tgcn.linear_{k}.weight")
        struct.gate_b[i] = g(f"tgnn._base_tgcn.linear_{k}.bias")
    struct.cheb_w0 = g("tgnn.conv.lins.0.weight")
    struct.cheb_w1 = g("tgnn.conv.lins.1.weight")
    struct.cheb_bias = g("tgnn.conv.bias")
    struct.region_w = g("tgnn.linear.weight") if regional else None
    struct.region_b = g("tgnn.linear.bias") if regional else None
    struct.head1_w = g("linear1.weight")
    struct.head1_b = g("linear1.bias")
    struct.head2_w = g("linear2.weight")
    struct.head2_b = g("linear2.bias")
    return struct


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _WorkspacePool:
    """The forward leaves its activations in a multi-GB workspace that the backward reads.  Blocks are recycled
    here (keyed by device and size) instead of going back to the allocator after every step: a step then never
    depends on allocator behaviour for its one large buffer, and two forwards in flight simply use two blocks.  Blocks are
    keyed by the launch stream as well (like the caching allocator's own free lists): work enqueued on another stream may still be
    using a block that the host has already released."""

    def __init__(self):
        self._free = {}
        self._stream_of = {}

    def acquire(self, nbytes: int, device) -> torch.Tensor:
        stream = _stream()
        lst = self._free.get((device, stream, nbytes))
        ws = lst.pop() if lst else torch.empty(nbytes, dtype=torch.uint8, device=device)
        self._stream_of[ws.data_ptr()] = stream
        return ws

    def release(self, ws: torch.Tensor):
        stream = self._stream_of.pop(ws.data_ptr(), None)
        lst = self._free.setdefault((ws.device, stream, ws.numel()), [])
        if len(lst) < 2:
            lst.append(ws)


_POOL = _WorkspacePool()


class _WsHandle:
    """Returns the block to the pool when the autograd node that owns it goes away."""

    def __init__(self, ws):
        self.ws = ws

    def __del__(self):
        try:
            _POOL.release(self.ws)
        except Exception:  # interpreter shutdown
            pass


def _graph_struct(graph: PreparedGraph, periods: int) -> _lib.Graph:
    cache = graph.__dict__.setdefault("_struct_cache", {})       # the struct only holds pointers into the graph's own tensors
    if periods in cache:
        return cache[periods]
    tab, reg, n = graph.chunks_for(periods)
    g = _lib.Graph()
    g.rowptr, g.col, g.val = graph.rowptr.data_ptr(), graph.col.data_ptr(), graph.val.data_ptr()
    g.node_region = graph.node_region.data_ptr()
    g.chunk_tab, g.chunk_region, g.n_chunks = tab.data_ptr(), reg.data_ptr(), n
    if graph.m_rowptr is not None:
        g.m_rowptr, g.m_col = graph.m_rowptr.data_ptr(), graph.m_col.data_ptr()
        g.m_val_a, g.m_val_l = graph.m_val_a.data_ptr(), graph.m_val_l.data_ptr()
    g.overlap = 1 if graph.overlap else 0
    g.region_lo, g.region_hi = graph.region_lo, graph.region_hi
    g.region_sorted = 1 if graph.region_sorted else 0
    cache[periods] = g
    return g


# run.py accumulates the gradients of all snapshots of an epoch before one optimiser step (run.py:178-194): with autograd doing the
# accumulation every backward ends in one `p.grad += g` kernel per parameter (21 launches of ~6 us each).  With this switch on,
# RegTGCNFunction.backward adds its gradients to the existing `.grad` tensors itself -- ONE multi-tensor add -- and returns None
# for the parameters (autograd then has nothing left to accumulate); a parameter without `.grad` receives this call's gradient
# tensor directly.  Same values bit for bit (`tests/test_gpu_model.py`).  Off by default: `torch.autograd.grad(...)` and tensor
# hooks on the parameters would see no gradient; `train.py` and `bench.py` (plain `loss.backward()` loops) switch it on.
_ACCUMULATE_IN_BACKWARD = False


def set_grad_accumulation_in_backward(flag: bool) -> bool:
    """Returns the previous setting."""
    global _ACCUMULATE_IN_BACKWARD
    prev, _ACCUMULATE_IN_BACKWARD = _ACCUMULATE_IN_BACKWARD, bool(flag)
    return prev


class RegTGCNFunction(torch.autograd.Function):
    """(x, *params) -> (pred (N,O), hidden (N,C)); whole-model forward and backward in HIP."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, graph: PreparedGraph, regional: bool, slope: float, packed,
                *params: torch.Tensor):
        """``packed`` False: x is the reference's (N,F,T) snapshot.  True: x is the extended packed input
        (x_rows >= N, T, F) of the region-sharded path (own rows first, halo rows after; see dist.py).
        ``packed`` may also be a tuple ``(packed, arith, flags)``: the per-call GEMM arithmetic (``_lib.ARITH_*``) and
        ``_lib.DIMS_*`` switches of regt_dims (0, 0 = the process defaults)."""
        arith = flags = 0
        if isinstance(packed, tuple):
            packed, arith, flags = packed
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.RegtError("RegT-GCN forward needs CUDA/HIP tensors: there is no CPU path in this package")
        xbf = packed and x.dtype == torch.bfloat16        # region shard that packs and exchanges its rows as bf16 (REGT_GEMM_MODE=bf16)
        if (x.dtype != torch.float32 and not xbf) or x.dim() != 3:
            raise ValueError(f"x must be float32 (N,F,T) (or packed bfloat16 (x_rows,T,F)), got {x.dtype} {tuple(x.shape)}")
        names = param_names(regional)
        if len(params) != len(names):
            raise ValueError(f"expected {len(names)} parameter tensors, got {len(params)}")
        ctx.leaf_params = params                 # the caller's tensors (before any padding): whose .grad an accumulating backward updates
        ctx.set_materialize_grads(False)         # an output the loss does not use arrives as None in backward, not as a tensor of zeros
                                                 # (N x C floats filled and read back for nothing: `hidden` in every training loop)
        for n_, p_ in zip(names, params):
            if p_.dtype != torch.float32 or not p_.is_cuda or not p_.is_contiguous():
                raise ValueError(f"parameter {n_} must be a contiguous float32 CUDA tensor")
        x = x.contiguous()
        # The kernels read 16-byte feature rows (F a multiple of 4; the reference uses F = 8).  Any other width is staged
        # padded: zero feature columns in x and zero weight columns for them change nothing in the forward, and their
        # gradient columns (ds^T x_pad = 0) are sliced off in the backward.
        f_real = x.shape[2] if packed else x.shape[1]
        f_pad = (-f_real) % 4
        if f_pad and xbf:
            raise ValueError("bf16 packed input needs a feature width that is a multiple of 4")
        if f_pad:
            x = torch.nn.functional.pad(x, (0, f_pad) if packed else (0, 0, 0, f_pad)).contiguous()
            params = tuple(torch.nn.functional.pad(p_, (0, f_pad)).contiguous() if n_ in F_WIDE_PARAMS else p_
                           for n_, p_ in zip(names, params))
        if packed:
            x_rows, T, F = x.shape
            N = graph.num_nodes
            if x_rows < N:
                raise ValueError(f"packed input has {x_rows} rows but the shard owns {N} nodes")
        else:
            N, F, T = x.shape
            x_rows = N
            if N != graph.num_nodes:
                raise ValueError(f"x has {N} nodes but the prepared graph has {graph.num_nodes}")
        # shape validation, the dims / parameter-pointer structs and the workspace size only depend on (shapes, parameter
        # addresses): remembered per graph, so a steady-state step skips ~60 us of Python (TPIMS-scale steps are host-bound)
        plan_key = (N, T, F, x_rows, regional, float(slope), arith, flags, tuple((p_.data_ptr(), tuple(p_.shape)) for p_ in params))
        plans = graph.__dict__.setdefault("_plan_cache", {})
        plan = plans.get(plan_key)
        if plan is None:
            tens = dict(zip(names, params))
            Cdim = tens["tgnn.conv.bias"].numel()
            O = tens["linear2.weight"].shape[0]
            H1 = tens["linear1.weight"].shape[0]
            R = graph.num_regions
            expect = {"tgnn._attention": (T,), "tgnn.conv.lins.0.weight": (Cdim, F), "tgnn.conv.lins.1.weight": (Cdim, F),
                      "linear1.weight": (H1, Cdim), "linear2.weight": (O, H1)}
            for k in GATES:
                expect[f"tgnn._base_tgcn.conv_{k}.lin.weight"] = (Cdim, F)
                expect[f"tgnn._base_tgcn.linear_{k}.weight"] = (Cdim, 2 * Cdim)
            if regional:
                expect["tgnn.linear.weight"] = (Cdim, R * Cdim)
            for k, shp in expect.items():
                if tuple(tens[k].shape) != shp:
                    raise ValueError(f"parameter {k} has shape {tuple(tens[k].shape)}, expected {shp}")
            dims = _lib.Dims(N, T, F, Cdim, R, O, H1, 1 if regional else 0, float(slope), int(arith), int(flags))
            gs = _graph_struct(graph, T)
            wsb = lib.regt_workspace_bytes(C.byref(dims), gs.n_chunks, gs.overlap)
            if wsb == 0:
                _lib.check(1, "regt_workspace_bytes")
            if len(plans) > 16:
                plans.clear()
            plan = plans[plan_key] = (dims, gs, wsb, _fill(_lib.Params(), tens, regional), Cdim, O)
        dims, gs, wsb, ps, Cdim, O = plan
        handle = _WsHandle(_POOL.acquire(wsb, x.device))
        ws = handle.ws
        pred = torch.empty(N, O, dtype=torch.float32, device=x.device)
        hidden = torch.empty(N, Cdim, dtype=torch.float32, device=x.device)
        if xbf:
            _lib.check(lib.regt_forward_packed_bf16(C.byref(dims), C.byref(gs), C.byref(ps), _lib.ptr(x), x_rows, _lib.ptr(pred),
                                                    _lib.ptr(hidden), _lib.ptr(ws), wsb, _stream()), "regt_forward_packed_bf16")
        elif packed:
            _lib.check(lib.regt_forward_packed(C.byref(dims), C.byref(gs), C.byref(ps), _lib.ptr(x), x_rows, _lib.ptr(pred),
                                               _lib.ptr(hidden), _lib.ptr(ws), wsb, _stream()), "regt_forward_packed")
        else:
            _lib.check(lib.regt_forward(C.byref(dims), C.byref(gs), C.byref(ps), _lib.ptr(x), _lib.ptr(pred),
                                        _lib.ptr(hidden), _lib.ptr(ws), wsb, _stream()), "regt_forward")
        ctx.graph, ctx.regional, ctx.dims, ctx.ws, ctx.wsb = graph, regional, dims, ws, wsb
        ctx.ws_handle = handle
        ctx.ps, ctx.gs = ps, gs                  # parameter / graph pointer structs: unchanged until backward
        ctx.xp = x if packed else None
        ctx.names = names
        ctx.f_real, ctx.f_pad = f_real, f_pad
        ctx.save_for_backward(hidden, *params)
        return pred, hidden

    @staticmethod
    def backward(ctx, dpred, dhidden):
        lib = _lib.load()
        hidden, *params = ctx.saved_tensors
        names, regional, dims = ctx.names, ctx.regional, ctx.dims
        tens = dict(zip(names, params))
        dev = hidden.device
        if dpred is None:
            dpred = torch.zeros(dims.N, dims.O, dtype=torch.float32, device=dev)
        dpred = dpred.contiguous()
        dhid = None if dhidden is None else dhidden.contiguous()
        # one allocation for all gradients (16-byte aligned views), not one per parameter
        sizes = [(p_.numel() + 3) & ~3 for p_ in params]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        grads, off = {}, 0
        for n_, p_, sz in zip(names, params, sizes):
            grads[n_] = flat[off:off + p_.numel()].view_as(p_)
            off += sz
        gr = _fill(_lib.Grads(), grads, regional)
        _lib.check(lib.regt_backward(C.byref(dims), C.byref(ctx.gs), C.byref(ctx.ps), C.byref(gr), _lib.ptr(dpred), _lib.ptr(dhid),
                                     _lib.ptr(hidden), _lib.ptr(ctx.xp), _lib.ptr(ctx.ws), ctx.wsb, _stream()), "regt_backward")
        if ctx.f_pad:
            for n_ in F_WIDE_PARAMS:
                if n_ in grads:
                    grads[n_] = grads[n_][:, :ctx.f_real].contiguous()
        leaves = ctx.leaf_params
        if _ACCUMULATE_IN_BACKWARD and all(p_.is_leaf and p_.requires_grad for p_ in leaves):
            have, new = [], []
            for n_, p_ in zip(names, leaves):
                if p_.grad is None:
                    p_.grad = grads[n_]          # (a view of this call's buffer, which nothing else refers to)
                else:
                    have.append(p_.grad)
                    new.append(grads[n_])
            if have:
                torch._foreach_add_(have, new)
            return (None, None, None, None, None) + (None,) * len(names)
        return (None, None, None, None, None) + tuple(grads[n_] for n_ in names)


class MseLossFunction(torch.autograd.Function):
    """sum((pred - y)^2) / global_count and its gradient in one kernel (regt_mse_loss_grad; run.py:180 with the mean taken over the
    GLOBAL graph: a region shard passes the global element count)."""

    @staticmethod
    def forward(ctx, pred: torch.Tensor, y: torch.Tensor, global_count: int):
        lib = _lib.load()
        if not pred.is_cuda or pred.dtype != torch.float32 or y.dtype != torch.float32 or pred.shape != y.shape:
            raise ValueError("mse_loss: pred and y must be float32 CUDA tensors of one shape")
        pred, y = pred.contiguous(), y.contiguous()
        dpred = torch.empty_like(pred)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        _lib.check(lib.regt_mse_loss_grad(_lib.ptr(pred), _lib.ptr(y), _lib.ptr(dpred), _lib.ptr(loss), pred.numel(), int(global_count),
                                          _stream()), "regt_mse_loss_grad")
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None


def mse_loss(pred: torch.Tensor, y: torch.Tensor, global_count: Optional[int] = None) -> torch.Tensor:
    return MseLossFunction.apply(pred, y, pred.numel() if global_count is None else global_count)


class FusedTrainStep:
    """The per-snapshot body of run.py::train() (forward, ``mean((out - y)**2)``, backward with accumulating gradients,
    run.py:178-191) as three C-ABI calls -- regt_forward, regt_mse_loss_grad, regt_backward -- plus ONE axpy, without
    autograd.  Same kernels and arithmetic as :class:`RegTGCNFunction`; what goes away is the host work per step
    (autograd graph, 22 gradient allocations, 22 AccumulateGrad launches, the loss ops).  Measured at TPIMS size:
    0.60 -> 0.54 ms per step -- that regime is bound by the ~45 dependent small kernels themselves, not by the host.

    The parameters' ``.grad`` become views of one flat accumulator, so any torch optimizer works; clear them with
    :meth:`zero_grad` (or ``optimizer.zero_grad(set_to_none=False)``).  Build it after ``model.to(device)``."""

    def __init__(self, model, graph: PreparedGraph, num_features: int, periods: int, slope: float = 0.01,
                 loss_count: Optional[int] = None):
        """``loss_count``: the divisor of the squared-error sum (default: all N*O entries = the mean).  A batched graph of B
        snapshots (prepare_graph(copies=B)) passes the entries of ONE snapshot: the loss is then the SUM of the B snapshot means
        and the gradients are the sum run.py accumulates over those snapshots."""
        lib = _lib.load()
        self.loss_count = loss_count
        arith = _lib.arith_code(getattr(model, "arithmetic", None))
        self.lib, self.graph, self.regional = lib, graph, bool(model.regional)
        names = param_names(self.regional)
        named = dict(model.named_parameters())
        self.params = [named[n] for n in names]
        for n_, p_ in zip(names, self.params):
            if p_.dtype != torch.float32 or not p_.is_cuda or not p_.is_contiguous():
                raise ValueError(f"parameter {n_} must be a contiguous float32 CUDA tensor")
        dev = self.params[0].device
        tens = dict(zip(names, self.params))
        N, F, T = graph.num_nodes, num_features, periods
        Cdim = tens["tgnn.conv.bias"].numel()
        O, H1 = tens["linear2.weight"].shape[0], tens["linear1.weight"].shape[0]
        # (the model's per-call switches -- DIMS_NO_BF16_ROWS / _NO_FUSED_BWD / _NO_SIDE_STREAM -- apply here as under autograd)
        self.dims = _lib.Dims(N, T, F, Cdim, graph.num_regions, O, H1, 1 if self.regional else 0, float(slope), arith,
                              int(getattr(model, "call_flags", 0)))
        self.gs = _graph_struct(graph, T)
        self.wsb = lib.regt_workspace_bytes(C.byref(self.dims), self.gs.n_chunks, self.gs.overlap)
        if self.wsb == 0:
            _lib.check(1, "regt_workspace_bytes")
        self.ws = torch.empty(self.wsb, dtype=torch.uint8, device=dev)
        total = sum(p_.numel() for p_ in self.params)
        pad = [(-p_.numel()) % 4 for p_ in self.params]              # keep every view 16-byte aligned
        self.acc = torch.zeros(total + sum(pad), dtype=torch.float32, device=dev)
        self.step_grad = torch.empty_like(self.acc)
        views, off = {}, 0
        self._acc_views = []
        for n_, p_, extra in zip(names, self.params, pad):
            p_.grad = self.acc[off:off + p_.numel()].view_as(p_)
            self._acc_views.append(p_.grad)
            views[n_] = self.step_grad[off:off + p_.numel()].view_as(p_)
            off += p_.numel() + extra
        self.ps = _fill(_lib.Params(), tens, self.regional)
        self.gr = _fill(_lib.Grads(), views, self.regional)
        self._keep = (tens, views)
        self.pred = torch.empty(N, O, dtype=torch.float32, device=dev)
        self.hidden = torch.empty(N, Cdim, dtype=torch.float32, device=dev)
        self.dpred = torch.empty_like(self.pred)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.shape_x = (N, F, T)

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """One training step on snapshot (x (N,F,T), y (N,O)); returns the loss as a 1-element device tensor that the
        next call overwrites."""
        lib, st = self.lib, _stream()
        if tuple(x.shape) != self.shape_x or x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous():
            raise ValueError(f"x must be a contiguous float32 CUDA tensor of shape {self.shape_x}")
        if tuple(y.shape) != tuple(self.pred.shape) or y.dtype != torch.float32 or not y.is_contiguous():
            raise ValueError(f"y must be a contiguous float32 tensor of shape {tuple(self.pred.shape)}")
        for p_, view in zip(self.params, self._acc_views):
            if p_.grad is not view:
                # optimizer.zero_grad() with its default set_to_none=True dropped the views: the optimiser would skip every
                # parameter while this object kept accumulating into a detached buffer.  Re-attach (cleared, as requested).
                if p_.grad is None:
                    view.zero_()
                else:
                    view.copy_(p_.grad)
                p_.grad = view
        _lib.check(lib.regt_forward(C.byref(self.dims), C.byref(self.gs), C.byref(self.ps), _lib.ptr(x), _lib.ptr(self.pred),
                                    _lib.ptr(self.hidden), _lib.ptr(self.ws), self.wsb, st), "regt_forward")
        cnt = self.pred.numel()
        _lib.check(lib.regt_mse_loss_grad(_lib.ptr(self.pred), _lib.ptr(y), _lib.ptr(self.dpred), _lib.ptr(self.loss), cnt,
                                          self.loss_count or cnt, st),
                   "regt_mse_loss_grad")
        _lib.check(lib.regt_backward(C.byref(self.dims), C.byref(self.gs), C.byref(self.ps), C.byref(self.gr), _lib.ptr(self.dpred),
                                     None, _lib.ptr(self.hidden), None, _lib.ptr(self.ws), self.wsb, st), "regt_backward")
        self.acc.add_(self.step_grad)
        return self.loss

    def zero_grad(self):
        self.acc.zero_()


# ---- models whose embedding stage is not the regional one: cell on a caller-supplied hidden input ---------------------

PARAM_NAMES_CELL = (
    ["tgnn._attention"]
    + [f"tgnn._base_tgcn.conv_{k}.lin.weight" for k in GATES]
    + [f"tgnn._base_tgcn.conv_{k}.bias" for k in GATES]
    + [f"tgnn._base_tgcn.linear_{k}.weight" for k in GATES]
    + [f"tgnn._base_tgcn.linear_{k}.bias" for k in GATES]
    + PARAM_NAMES_HEAD
)


def _gcn_graph_struct(op) -> _lib.Graph:
    g = _lib.Graph()
    g.rowptr, g.col, g.val = op.rowptr.data_ptr(), op.col.data_ptr(), op.val.data_ptr()
    return g


class CellFunction(torch.autograd.Function):
    """(x (N,F,T), h_in (N*T, C), *cell params) -> (pred (N,O), hidden (N,C)): regt_cell_forward / regt_cell_backward."""

    @staticmethod
    def forward(ctx, x, h_in, op, *params):
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.RegtError("RegT-GCN forward needs CUDA/HIP tensors: there is no CPU path in this package")
        names = PARAM_NAMES_CELL
        if len(params) != len(names):
            raise ValueError(f"expected {len(names)} parameter tensors, got {len(params)}")
        x, h_in = x.contiguous(), h_in.contiguous()
        N, F, T = x.shape
        tens = dict(zip(names, params))
        Cdim = tens["tgnn._base_tgcn.conv_z.bias"].numel()
        O, H1 = tens["linear2.weight"].shape[0], tens["linear1.weight"].shape[0]
        if tuple(h_in.shape) != (N * T, Cdim) or h_in.dtype != torch.float32:
            raise ValueError(f"h_in must be float32 ({N * T}, {Cdim}), got {h_in.dtype} {tuple(h_in.shape)}")
        if N != op.num_nodes:
            raise ValueError(f"x has {N} nodes but the prepared operator has {op.num_nodes}")
        dims = _lib.Dims(N, T, F, Cdim, 1, O, H1, 0, 0.0)
        gs = _gcn_graph_struct(op)
        wsb = lib.regt_workspace_bytes(C.byref(dims), 0, 0)
        if wsb == 0:
            _lib.check(1, "regt_workspace_bytes")
        handle = _WsHandle(_POOL.acquire(wsb, x.device))
        pred = torch.empty(N, O, dtype=torch.float32, device=x.device)
        hidden = torch.empty(N, Cdim, dtype=torch.float32, device=x.device)
        ps = _fill(_lib.Params(), tens, False)
        _lib.check(lib.regt_cell_forward(C.byref(dims), C.byref(gs), C.byref(ps), _lib.ptr(x), _lib.ptr(h_in), _lib.ptr(pred),
                                         _lib.ptr(hidden), _lib.ptr(handle.ws), wsb, _stream()), "regt_cell_forward")
        ctx.op, ctx.dims, ctx.ws_handle, ctx.wsb = op, dims, handle, wsb
        ctx.save_for_backward(hidden, h_in, *params)
        ctx.set_materialize_grads(False)         # an unused output (hidden) arrives as None in backward, not as zeros
        return pred, hidden

    @staticmethod
    def backward(ctx, dpred, dhidden):
        lib = _lib.load()
        hidden, h_in, *params = ctx.saved_tensors
        names, dims = PARAM_NAMES_CELL, ctx.dims
        tens = dict(zip(names, params))
        dev = hidden.device
        if dpred is None:
            dpred = torch.zeros(dims.N, dims.O, dtype=torch.float32, device=dev)
        dpred = dpred.contiguous()
        dhid = None if dhidden is None else dhidden.contiguous()
        grads = {n_: torch.empty_like(p_) for n_, p_ in tens.items()}
        dh_in = torch.empty_like(h_in)
        gs = _gcn_graph_struct(ctx.op)
        ps = _fill(_lib.Params(), tens, False)
        gr = _fill(_lib.Grads(), grads, False)
        _lib.check(lib.regt_cell_backward(C.byref(dims), C.byref(gs), C.byref(ps), C.byref(gr), _lib.ptr(dpred), _lib.ptr(dhid),
                                          _lib.ptr(hidden), _lib.ptr(h_in), _lib.ptr(dh_in), _lib.ptr(ctx.ws_handle.ws), ctx.wsb,
                                          _stream()), "regt_cell_backward")
        return (None, dh_in, None) + tuple(grads[n_] for n_ in names)


class AggregateFunction(torch.autograd.Function):
    """Y = A_hat @ H for a learned H (N, W): forward pulls over the CSR of A_hat, backward over the CSR of A_hat^T."""

    @staticmethod
    def forward(ctx, h, op):
        from . import ops
        ctx.op = op
        return ops.spmm_csr(op.rowptr, op.col, op.val, h)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        op = ctx.op
        return ops.spmm_csr(op.t_rowptr, op.t_col, op.t_val, dy.contiguous()), None


class LinearFunction(torch.autograd.Function):
    """y = a @ w.T + b on the matrix cores; backward: da = dy @ w, (dw, db) = wgrad(dy, a)."""

    @staticmethod
    def forward(ctx, a, w, b):
        from . import ops
        ctx.save_for_backward(a, w)
        ctx.need_da = a.requires_grad
        ctx.has_bias = b is not None
        return ops.linear(a, w, b, 0)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        a, w = ctx.saved_tensors
        dy = dy.contiguous()
        da = ops.linear(dy, w.t().contiguous(), None, 0) if ctx.need_da else None
        dw, db = ops.wgrad(dy, a, with_bias=ctx.has_bias)
        return da, dw, db


class Cell0Function(torch.autograd.Function):
    """Zero-hidden TGCN cell + attention + head (regt_cell0_forward / regt_cell0_backward):
    (a_z (M,kz), a_h (M,kh), gz (C,kz), gh (C,kh), cz (C), ch (C), attention (T), linear1.weight, linear1.bias, linear2.weight,
    linear2.bias, num_nodes) -> (pred (N,O), hidden (N,C))."""

    @staticmethod
    def forward(ctx, a_z, a_h, gz, gh, cz, ch, att, l1w, l1b, l2w, l2b, num_nodes: int):
        lib = _lib.load()
        tens = [a_z, a_h, gz, gh, cz, ch, att, l1w, l1b, l2w, l2b]
        for t_ in tens:
            if not t_.is_cuda or t_.dtype != torch.float32:
                raise _lib.RegtError("zero-hidden cell needs float32 CUDA/HIP tensors: there is no CPU path in this package")
        a_z, a_h, gz, gh, cz, ch, att, l1w, l1b, l2w, l2b = [t_.contiguous() for t_ in tens]
        M, kz = a_z.shape
        kh = a_h.shape[1]
        T = att.numel()
        Cdim, O, H1 = gz.shape[0], l2w.shape[0], l1w.shape[0]
        if M != num_nodes * T or a_h.shape[0] != M or tuple(gz.shape) != (Cdim, kz) or tuple(gh.shape) != (Cdim, kh):
            raise ValueError("zero-hidden cell: inconsistent shapes")
        dims = _lib.Dims(num_nodes, T, max(kz, kh), Cdim, 1, O, H1, 0, 0.0)
        args = _lib.Cell0Args(a_z.data_ptr(), a_h.data_ptr(), kz, kh, gz.data_ptr(), gh.data_ptr(), cz.data_ptr(), ch.data_ptr(),
                              att.data_ptr(), l1w.data_ptr(), l1b.data_ptr(), l2w.data_ptr(), l2b.data_ptr())
        wsb = lib.regt_cell0_workspace_bytes(C.byref(dims), kz, kh)
        if wsb == 0:
            raise _lib.RegtError("regt_cell0_workspace_bytes: bad dims")
        handle = _WsHandle(_POOL.acquire(wsb, a_z.device))
        pred = torch.empty(num_nodes, O, dtype=torch.float32, device=a_z.device)
        hidden = torch.empty(num_nodes, Cdim, dtype=torch.float32, device=a_z.device)
        _lib.check(lib.regt_cell0_forward(C.byref(dims), C.byref(args), _lib.ptr(pred), _lib.ptr(hidden), _lib.ptr(handle.ws), wsb,
                                          _stream()), "regt_cell0_forward")
        ctx.dims, ctx.args, ctx.ws_handle, ctx.wsb = dims, args, handle, wsb
        ctx.save_for_backward(hidden, a_z, a_h, gz, gh, cz, ch, att, l1w, l1b, l2w, l2b)
        ctx.set_materialize_grads(False)         # an unused output (hidden) arrives as None in backward, not as zeros
        return pred, hidden

    @staticmethod
    def backward(ctx, dpred, dhidden):
        lib = _lib.load()
        hidden, a_z, a_h, gz, gh, cz, ch, att, l1w, l1b, l2w, l2b = ctx.saved_tensors
        dims = ctx.dims
        dev = hidden.device
        if dpred is None:
            dpred = torch.zeros(dims.N, dims.O, dtype=torch.float32, device=dev)
        dpred = dpred.contiguous()
        dhid = None if dhidden is None else dhidden.contiguous()
        need_az, need_ah = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d_az = torch.empty_like(a_z) if need_az else None
        d_ah = torch.empty_like(a_h) if need_ah else None
        g = [torch.empty_like(t_) for t_ in (gz, gh, cz, ch, att, l1w, l1b, l2w, l2b)]
        gr = _lib.Cell0Grads(_lib.ptr(d_az), _lib.ptr(d_ah), *[t_.data_ptr() for t_ in g])
        _lib.check(lib.regt_cell0_backward(C.byref(dims), C.byref(ctx.args), C.byref(gr), _lib.ptr(dpred), _lib.ptr(dhid),
                                           _lib.ptr(hidden), _lib.ptr(ctx.ws_handle.ws), ctx.wsb, _stream()), "regt_cell0_backward")
        return (d_az, d_ah, *g, None)


class GatAggregateFunction(torch.autograd.Function):
    """out (N,T,F) = attention-weighted neighbour sum of GATConv on packed input rows (regt_gat_forward); the backward returns
    the gradients of the two score vectors u_src, u_dst (F) -- the input rows are data."""

    @staticmethod
    def forward(ctx, xp, u_src, u_dst, pattern, slope: float):
        from . import ops
        out, stats = ops.gat_forward(pattern.rowptr, pattern.col, xp, u_src, u_dst, slope)
        ctx.pattern, ctx.slope = pattern, slope
        ctx.save_for_backward(xp, u_src.contiguous(), stats)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import ops
        xp, u_src, stats = ctx.saved_tensors
        p = ctx.pattern
        n, t, f = xp.shape
        dsd = ops.gat_backward(p.rowptr, p.col, p.t_rowptr, p.t_col, xp, u_src, dout.contiguous(), stats, ctx.slope)
        du, _ = ops.wgrad(dsd, xp.view(n * t, f), with_bias=False)          # (2, F) = dsd^T x
        return None, du[0].contiguous(), du[1].contiguous(), None, None


class ZeroGradAnchor(torch.autograd.Function):
    """Identity on ``(pred, hidden)`` that gives ``dead`` parameters an all-zero gradient: in the reference's GraphSAGE / GAT models
    the reset gate is computed and multiplied by the zero hidden state, so autograd hands its parameters zeros, not None --
    whichever of the two outputs the loss is built from."""

    @staticmethod
    def forward(ctx, pred, hidden, *dead):
        ctx.shapes = [(d.shape, d.device) for d in dead]
        ctx.set_materialize_grads(False)         # the unused output's gradient stays None on its way to the cell's backward
        return pred.view_as(pred), hidden.view_as(hidden)

    @staticmethod
    def backward(ctx, g_pred, g_hidden):
        return (g_pred, g_hidden, *[torch.zeros(s, dtype=torch.float32, device=dv) for s, dv in ctx.shapes])


def regt_gcn_forward(x, graph: PreparedGraph, params: Dict[str, torch.Tensor], regional: bool = True, slope: float = 0.01):
    """Functional entry: ``params`` keyed by the reference's state_dict names."""
    names = param_names(regional)
    return RegTGCNFunction.apply(x, graph, regional, slope, False, *[params[n] for n in names])
