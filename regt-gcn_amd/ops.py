"""Thin tensor-level wrappers over the op-site entry points of the C ABI (used by tests, bench and the
reference-side binding shown in INTEGRATION.md).  No arithmetic happens in Python."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.float32:
        raise _lib.RegtError(f"{name} must be a float32 CUDA tensor (no CPU path)")
    return t.contiguous()


def pack_x(x: torch.Tensor) -> torch.Tensor:
    """(N,F,T) -> (N,T,F)."""
    x = _f32c(x, "x")
    n, f, t = x.shape
    out = torch.empty(n, t, f, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().regt_pack_x(_lib.ptr(x), _lib.ptr(out), n, f, t, _stream()), "regt_pack_x")
    return out


def pack_x_into(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """(N,F,T) -> rows [0, N) of ``out`` (>= N rows of (T,F)); the remaining rows are left alone (halo slots)."""
    x = _f32c(x, "x")
    n, f, t = x.shape
    if (out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device or out.dim() != 3
            or out.shape[0] < n or tuple(out.shape[1:]) != (t, f)):
        raise ValueError(f"pack_x_into: out must be a contiguous fp32 (>= {n}, {t}, {f}) tensor on {x.device}")
    _lib.check(_lib.load().regt_pack_x(_lib.ptr(x), _lib.ptr(out), n, f, t, _stream()), "regt_pack_x")
    return out


def pack_x_bf16_into(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """(N,F,T) fp32 -> rows [0, N) of the bf16 tensor ``out`` (>= N rows of (T,F)), rounded to nearest even (REGT_GEMM_MODE=bf16:
    the packed rows a region shard exchanges and hands to ``forward_packed``); F % 8 == 0."""
    x = _f32c(x, "x")
    n, f, t = x.shape
    if (out.dtype != torch.bfloat16 or not out.is_contiguous() or out.device != x.device or out.dim() != 3
            or out.shape[0] < n or tuple(out.shape[1:]) != (t, f)):
        raise ValueError(f"pack_x_bf16_into: out must be a contiguous bf16 (>= {n}, {t}, {f}) tensor on {x.device}")
    _lib.check(_lib.load().regt_pack_x_bf16(_lib.ptr(x), _lib.ptr(out), n, f, t, _stream()), "regt_pack_x_bf16")
    return out


def spmm_dual_bf16(rowptr, col, val_a, val_l, x: torch.Tensor):
    """(A x, L x) as bf16 rows from bf16 rows ``x`` (x_rows >= N, W): fp32 accumulation, one rounding per element; W % 64 == 0."""
    if not x.is_cuda or x.dtype != torch.bfloat16 or x.dim() != 2:
        raise _lib.RegtError("x must be a 2-D bfloat16 CUDA tensor (no CPU path)")
    x = x.contiguous()
    n = rowptr.numel() - 1
    ya = torch.empty(n, x.shape[1], dtype=torch.bfloat16, device=x.device)
    yl = torch.empty_like(ya)
    _lib.check(_lib.load().regt_spmm_dual_bf16(_lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val_a), _lib.ptr(val_l), _lib.ptr(x),
                                               _lib.ptr(ya), _lib.ptr(yl), n, x.shape[0], x.shape[1], _stream()), "regt_spmm_dual_bf16")
    return ya, yl


def spmm_csr(rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, x: torch.Tensor, out: Optional[torch.Tensor] = None):
    """Y[r,:] = sum_e val[e] * X[col[e],:] for r in range(len(rowptr)-1)."""
    x = _f32c(x, "x")
    nrows = rowptr.numel() - 1
    width = x.shape[1]
    if out is None:
        out = torch.empty(nrows, width, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().regt_spmm_csr(_lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val), _lib.ptr(x), _lib.ptr(out),
                                         nrows, x.shape[0], width, _stream()), "regt_spmm_csr")
    return out


def spmm_dual(rowptr, col, val_a, val_l, x: torch.Tensor):
    """(A x, L x) from the merged two-weight CSR in one gather pass; x.shape[1] % 4 == 0."""
    x = _f32c(x, "x")
    n = rowptr.numel() - 1
    ya = torch.empty(n, x.shape[1], dtype=torch.float32, device=x.device)
    yl = torch.empty_like(ya)
    _lib.check(_lib.load().regt_spmm_dual(_lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val_a), _lib.ptr(val_l), _lib.ptr(x),
                                          _lib.ptr(ya), _lib.ptr(yl), n, x.shape[1], _stream()), "regt_spmm_dual")
    return ya, yl


def linear(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = 0, slope: float = 0.01):
    """act(a @ w.T + bias) on the fp32 matrix cores."""
    a, w = _f32c(a, "a"), _f32c(w, "w")
    m, k = a.shape
    n = w.shape[0]
    out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    b = None if bias is None else _f32c(bias, "bias")
    _lib.check(_lib.load().regt_linear(_lib.ptr(a), k, m, k, _lib.ptr(w), k, n, _lib.ptr(b), act, slope, _lib.ptr(out), n,
                                       _stream()), "regt_linear")
    return out


def wgrad(dout: torch.Tensor, a: torch.Tensor, with_bias: bool = True):
    """(dout.T @ a, dout.sum(0)) -- the weight / bias gradient of ``linear``."""
    dout, a = _f32c(dout, "dout"), _f32c(a, "a")
    m, n = dout.shape
    k = a.shape[1]
    lib = _lib.load()
    slab = torch.empty(lib.regt_wgrad_slab_floats(m, n, k, 1 if with_bias else 0), dtype=torch.float32, device=a.device)
    dw = torch.empty(n, k, dtype=torch.float32, device=a.device)
    db = torch.empty(n, dtype=torch.float32, device=a.device) if with_bias else None
    _lib.check(lib.regt_wgrad(_lib.ptr(dout), n, _lib.ptr(a), k, m, n, k, _lib.ptr(dw), k, _lib.ptr(db), _lib.ptr(slab),
                              _stream()), "regt_wgrad")
    return dw, db


def mse_loss_grad(pred: torch.Tensor, y: torch.Tensor, global_count: Optional[int] = None):
    """(loss, dloss/dpred) of ``mean((pred - y)**2)`` (run.py:180); mean over ``global_count`` entries."""
    pred, y = _f32c(pred, "pred"), _f32c(y, "y")
    dpred = torch.empty_like(pred)
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    cnt = pred.numel()
    _lib.check(_lib.load().regt_mse_loss_grad(_lib.ptr(pred), _lib.ptr(y), _lib.ptr(dpred), _lib.ptr(loss), cnt,
                                              cnt if global_count is None else global_count, _stream()), "regt_mse_loss_grad")
    return loss, dpred


def gat_forward(rowptr: torch.Tensor, col: torch.Tensor, xp: torch.Tensor, u_src: torch.Tensor, u_dst: torch.Tensor, slope: float = 0.2):
    """GATConv attention aggregation on packed input rows xp (N,T,F): returns (out (N,T,F), stats (N*T,4))."""
    xp, u_src, u_dst = _f32c(xp, "xp"), _f32c(u_src, "u_src"), _f32c(u_dst, "u_dst")
    n, t, f = xp.shape
    out = torch.empty_like(xp)
    stats = torch.empty(n * t, 4, dtype=torch.float32, device=xp.device)
    _lib.check(_lib.load().regt_gat_forward(_lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(xp), _lib.ptr(u_src), _lib.ptr(u_dst), slope,
                                            n, t, f, _lib.ptr(out), _lib.ptr(stats), _stream()), "regt_gat_forward")
    return out, stats


def gat_backward(rowptr, col, t_rowptr, t_col, xp: torch.Tensor, u_src: torch.Tensor, dout: torch.Tensor, stats: torch.Tensor,
                 slope: float = 0.2) -> torch.Tensor:
    """Score gradients dsd (N*T, 2) = (dL/ds, dL/dd) of gat_forward given dL/dout (N,T,F)."""
    xp, u_src, dout = _f32c(xp, "xp"), _f32c(u_src, "u_src"), _f32c(dout, "dout")
    n, t, f = xp.shape
    dsd = torch.empty(n * t, 2, dtype=torch.float32, device=xp.device)
    _lib.check(_lib.load().regt_gat_backward(_lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(t_rowptr), _lib.ptr(t_col), _lib.ptr(xp),
                                             _lib.ptr(u_src), slope, n, t, f, _lib.ptr(dout), _lib.ptr(stats), _lib.ptr(dsd), _stream()),
               "regt_gat_backward")
    return dsd
