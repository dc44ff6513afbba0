"""Host-side logic of round 4 that needs no GPU: graph replication for snapshot batching, the stacked window store, the
batched-graph cache, the CLI flag, the per-call arithmetic codes."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import regtgcn_amd as R                                   # noqa: E402
from regtgcn_amd import _lib                               # noqa: E402
from regtgcn_amd.graph import replicate_edges             # noqa: E402


def test_replicate_edges_builds_disjoint_copies():
    ei = torch.tensor([[0, 1, 2, 2], [1, 2, 0, 2]], dtype=torch.int64)
    w = torch.tensor([1.0, 2.0, 3.0, 4.0])
    same_i, same_w = replicate_edges(ei, w, 1, 3)
    assert same_i is ei and same_w is w
    ri, rw = replicate_edges(ei, w, 3, 3)
    assert tuple(ri.shape) == (2, 12) and tuple(rw.shape) == (12,)
    for b in range(3):                                     # copy b: the same edges shifted by b * N, in the original order
        assert torch.equal(ri[:, 4 * b:4 * b + 4], ei + 3 * b)
        assert torch.equal(rw[4 * b:4 * b + 4], w)
    assert int(ri.max()) == 8 and int(ri.min()) == 0
    assert (ri[0] // 3 == ri[1] // 3).all()                # no edge crosses a copy boundary
    none_i, none_w = replicate_edges(ei, None, 2, 3)
    assert none_w is None and tuple(none_i.shape) == (2, 8)


def test_window_store_batches_are_views_of_stacked_windows():
    node = torch.arange(5 * 3 * 20, dtype=torch.float32).reshape(5, 3, 20)
    xs, ys = R.data.snapshot_windows(node, 6, 2)
    store = R.train.WindowStore(xs, ys)
    assert len(store) == len(xs) == 20 - 8 + 1
    x, y = store.batch(4, 3)
    assert tuple(x.shape) == (15, 3, 6) and tuple(y.shape) == (15, 2)
    assert x.data_ptr() == store.X[4].data_ptr()           # a view: no copy per step
    for b in range(3):
        assert torch.equal(x[5 * b:5 * b + 5], xs[4 + b]) and torch.equal(y[5 * b:5 * b + 5], ys[4 + b])
    x_last, _ = store.batch(12, 3)                         # the epoch's last, shorter batch
    assert x_last.shape[0] == 5 * (len(store) - 12)


def test_batched_graphs_are_built_once_per_batch_size():
    built = []
    graphs = R.train.BatchedGraphs(lambda b: built.append(b) or ("graph", b))
    assert graphs.get(4) == ("graph", 4) and graphs.get(4) == ("graph", 4) and graphs.get(3) == ("graph", 3)
    assert built == [4, 3]


def test_cli_accepts_snap_batch_next_to_the_reference_flags():
    a = R.train.build_parser().parse_args("--model RegionalTemporalGCN --num_timesteps_in 6 --num_timesteps_out 1 --tr 0.2 --tf occrate "
                                          "--dataloading_type 2 --epochs 1 --decomp_type regional --snap_batch 64".split())
    assert a.snap_batch == 64 and a.model == "RegionalTemporalGCN"
    assert R.train.build_parser().parse_args([]).snap_batch == 1       # default: one launch sequence per snapshot, as run.py


def test_arithmetic_codes_of_regt_dims():
    assert [_lib.arith_code(v) for v in (None, "default", "fp32", "bf16x3", "bf16", 0, 3)] == [0, 0, 1, 2, 3, 0, 3]
    for bad in ("fp8", 4, -1, True, 1.5):
        with pytest.raises(ValueError):
            _lib.arith_code(bad)
    d = _lib.Dims(10, 6, 8, 256, 5, 1, 128, 1, 0.01)       # trailing ABI-v6 fields default to "process defaults"
    assert d.arith == 0 and d.flags == 0
    d = _lib.Dims(10, 6, 8, 256, 5, 1, 128, 1, 0.01, _lib.ARITH_BF16, _lib.DIMS_NO_SIDE_STREAM)
    assert d.arith == 3 and d.flags == 4
    lib = R.load_library()
    import ctypes
    assert lib.regt_workspace_bytes(ctypes.byref(d), 1, 0) > 0
    bad_dims = _lib.Dims(10, 6, 8, 256, 5, 1, 128, 1, 0.01, 7, 0)
    assert lib.regt_workspace_bytes(ctypes.byref(bad_dims), 1, 0) == 0 and b"arith" in lib.regt_last_error()
