"""ctypes binding of libregtgcn_hip.so (C ABI: include/regtgcn.h).

There is no fallback: if the library is missing or a call fails, this module raises.
PyTorch is used by the callers only to own device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# REGT_LIB_DIR: a developer build of the same library in another directory (build.py honours the same variable), e.g. the
# workgroup-trace build of tools/wg_trace.py; never a different implementation
LIB_PATH = os.path.join(os.environ.get("REGT_LIB_DIR") or os.path.join(HERE, "lib"), "libregtgcn_hip.so")
ABI_VERSION = 7

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
vp = C.c_void_p


class RegtError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [("N", C.c_int32), ("T", C.c_int32), ("F", C.c_int32), ("C", C.c_int32), ("R", C.c_int32),
                ("O", C.c_int32), ("H1", C.c_int32), ("regional", C.c_int32), ("lrelu_slope", C.c_float),
                # ABI v6, per-call configuration (zero = process defaults): GEMM arithmetic (ARITH_*), DIMS_* flag bits
                ("arith", C.c_int32), ("flags", C.c_uint32)]


ARITH_DEFAULT, ARITH_FP32, ARITH_BF16X3, ARITH_BF16 = 0, 1, 2, 3
ARITH_NAMES = {None: ARITH_DEFAULT, "default": ARITH_DEFAULT, "fp32": ARITH_FP32, "bf16x3": ARITH_BF16X3, "bf16": ARITH_BF16}
DIMS_NO_BF16_ROWS, DIMS_NO_FUSED_BWD, DIMS_NO_SIDE_STREAM = 1, 2, 4


def arith_code(arith) -> int:
    """``None`` / "default" / "fp32" / "bf16x3" / "bf16" (or the REGT_ARITH_* integer) -> regt_dims.arith."""
    if isinstance(arith, int) and not isinstance(arith, bool) and 0 <= arith <= 3:
        return arith
    try:
        return ARITH_NAMES[arith]
    except (KeyError, TypeError):
        raise ValueError(f"arithmetic must be one of {sorted(k for k in ARITH_NAMES if k)} or None, got {arith!r}") from None


class Graph(C.Structure):
    _fields_ = [("rowptr", vp), ("col", vp), ("val", vp), ("node_region", vp), ("chunk_tab", vp),
                ("chunk_region", vp), ("n_chunks", C.c_int32),
                ("m_rowptr", vp), ("m_col", vp), ("m_val_a", vp), ("m_val_l", vp), ("overlap", C.c_int32),
                ("region_lo", C.c_int32), ("region_hi", C.c_int32), ("region_sorted", C.c_int32)]


_PARAM_FIELDS = [("attention", vp), ("conv_lin_w", vp * 3), ("conv_bias", vp * 3), ("gate_w", vp * 3),
                 ("gate_b", vp * 3), ("cheb_w0", vp), ("cheb_w1", vp), ("cheb_bias", vp), ("region_w", vp),
                 ("region_b", vp), ("head1_w", vp), ("head1_b", vp), ("head2_w", vp), ("head2_b", vp)]


class Params(C.Structure):
    _fields_ = _PARAM_FIELDS


class Grads(C.Structure):
    _fields_ = _PARAM_FIELDS


class Cell0Args(C.Structure):
    _fields_ = [("a_z", vp), ("a_h", vp), ("kz", C.c_int32), ("kh", C.c_int32), ("gz", vp), ("gh", vp), ("cz", vp), ("ch", vp),
                ("attention", vp), ("head1_w", vp), ("head1_b", vp), ("head2_w", vp), ("head2_b", vp)]


class Cell0Grads(C.Structure):
    _fields_ = [("a_z", vp), ("a_h", vp), ("gz", vp), ("gh", vp), ("cz", vp), ("ch", vp), ("attention", vp),
                ("head1_w", vp), ("head1_b", vp), ("head2_w", vp), ("head2_b", vp)]


# name, restype, argtypes -- one entry per function declared in include/regtgcn.h
SIGNATURES = {
    "regt_abi_version": (C.c_int32, []),
    "regt_set_gemm_mode": (C.c_int32, [C.c_int32]),
    "regt_set_option": (C.c_int32, [C.c_char_p, C.c_int32]),
    "regt_cell_forward": (C.c_int32, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "regt_cell_backward": (C.c_int32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "regt_last_error": (C.c_char_p, []),
    "regt_graph_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "regt_gcn_csr": (C.c_int32, [vp, vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "regt_gcn_dis": (C.c_int32, [vp, vp, C.c_int64, C.c_int32, vp, vp, vp, C.c_size_t, vp]),
    "regt_cheb_edge_weights": (C.c_int32, [vp, vp, C.c_int64, C.c_int32, vp, vp, vp, C.c_size_t, vp]),
    "regt_raw_csr": (C.c_int32, [vp, vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "regt_graph_fingerprint": (C.c_int32, [vp, vp, C.c_int64, vp, vp]),
    "regt_spmm_csr": (C.c_int32, [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp]),
    "regt_spmm_dual": (C.c_int32, [vp, vp, vp, vp, vp, vp, vp, C.c_int32, C.c_int32, vp]),
    "regt_pack_x": (C.c_int32, [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp]),
    "regt_pack_x_bf16": (C.c_int32, [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp]),
    "regt_spmm_dual_bf16": (C.c_int32, [vp, vp, vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp]),
    "regt_forward_packed_bf16": (C.c_int32, [C.POINTER(Dims), C.POINTER(Graph), C.POINTER(Params), vp, C.c_int32, vp, vp, vp,
                                             C.c_size_t, vp]),
    "regt_linear": (C.c_int32, [vp, C.c_int64, C.c_int64, C.c_int32, vp, C.c_int64, C.c_int32, vp, C.c_int32,
                                C.c_float, vp, C.c_int64, vp]),
    "regt_wgrad_slab_floats": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "regt_wgrad": (C.c_int32, [vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp, C.c_int64, vp, vp, vp]),
    "regt_workspace_bytes": (C.c_size_t, [C.POINTER(Dims), C.c_int32, C.c_int32]),
    "regt_forward": (C.c_int32, [C.POINTER(Dims), C.POINTER(Graph), C.POINTER(Params), vp, vp, vp, vp, C.c_size_t, vp]),
    "regt_forward_packed": (C.c_int32, [C.POINTER(Dims), C.POINTER(Graph), C.POINTER(Params), vp, C.c_int32, vp, vp, vp,
                                        C.c_size_t, vp]),
    "regt_backward": (C.c_int32, [C.POINTER(Dims), C.POINTER(Graph), C.POINTER(Params), C.POINTER(Grads), vp, vp, vp,
                                  vp, vp, C.c_size_t, vp]),
    "regt_mean_csr": (C.c_int32, [vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "regt_cell0_workspace_bytes": (C.c_size_t, [C.POINTER(Dims), C.c_int32, C.c_int32]),
    "regt_cell0_forward": (C.c_int32, [C.POINTER(Dims), C.POINTER(Cell0Args), vp, vp, vp, C.c_size_t, vp]),
    "regt_cell0_backward": (C.c_int32, [C.POINTER(Dims), C.POINTER(Cell0Args), C.POINTER(Cell0Grads), vp, vp, vp, vp, C.c_size_t, vp]),
    "regt_gat_forward": (C.c_int32, [vp, vp, vp, vp, vp, C.c_float, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]),
    "regt_gat_backward": (C.c_int32, [vp, vp, vp, vp, vp, vp, C.c_float, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp]),
    "regt_graph_stats": (C.c_int32, [C.POINTER(C.c_int64)]),
    "regt_debug_trace": (C.c_int64, [C.POINTER(C.c_int64), C.c_int64]),
    "regt_profile_enable": (C.c_int32, [C.c_int32]),
    "regt_profile_collect": (C.c_int32, [C.c_char_p, C.c_size_t]),
    "regt_mse_loss_grad": (C.c_int32, [vp, vp, vp, vp, C.c_int64, C.c_int64, vp]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises RegtError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RegtError(
            f"{LIB_PATH} not found: the HIP library is not built. Run `python regt-gcn_amd/build.py` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing -> loud
        fn.restype = res
        fn.argtypes = args
    v = lib.regt_abi_version()
    if v != ABI_VERSION:
        raise RegtError(f"libregtgcn_hip.so ABI version {v} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().regt_last_error().decode("utf-8", "replace")
        raise RegtError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None) as a c_void_p value."""
    return None if t is None else C.c_void_p(t.data_ptr())
